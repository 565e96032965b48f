"""Independent sequences, one per GPU.

The EKF path shards by *sequence*: every frame is a serial dependency chain on
one covariance matrix, so a filter is never split across GPUs.  Rank r runs
sequence r on GPU r with no communication in the frame loop; when the run is
over the trajectories and maps are gathered ONCE with ``all_gather`` (RCCL over
xGMI on GPUs, gloo in the CPU tests).  Payloads are tens of KB per rank.
"""
from __future__ import annotations

import torch


def rank_seed(base_seed: int, rank: int) -> int:
    """Sequence r of a multi-GPU run uses seed base+r (seed r for base 0)."""
    return int(base_seed) + int(rank)


def gather_sequences(trajectory: torch.Tensor, landmark_map: torch.Tensor, dist=None):
    """Gather per-rank results.

    trajectory   [F_r, 7]  camera pose per frame (x y z qw qx qy qz), float64
    landmark_map [n_r, 6]  landmark xyz + variances, float64
    Ranks may hold different F_r / n_r: tensors are padded with NaN to the
    common maximum (two tiny all_reduce(MAX)), gathered, and returned together
    with the true lengths.  With ``dist`` None or world_size 1 this is a local
    reshape.  Returns ``(traj [W,Fmax,7], maps [W,nmax,6], frames [W], landmarks [W])``.
    """
    dev = trajectory.device
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return (trajectory[None], landmark_map[None],
                torch.tensor([trajectory.shape[0]], device=dev),
                torch.tensor([landmark_map.shape[0]], device=dev))
    world = dist.get_world_size()
    sizes = torch.tensor([trajectory.shape[0], landmark_map.shape[0]], dtype=torch.int64, device=dev)
    size_list = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(size_list, sizes)          # list form: same call on RCCL and gloo
    all_sizes = torch.stack(size_list)
    fmax, nmax = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    tpad = torch.full((fmax, 7), float("nan"), dtype=torch.float64, device=dev)
    mpad = torch.full((nmax, 6), float("nan"), dtype=torch.float64, device=dev)
    tpad[: trajectory.shape[0]] = trajectory.to(torch.float64)
    mpad[: landmark_map.shape[0]] = landmark_map.to(torch.float64)
    traj_list = [torch.empty_like(tpad) for _ in range(world)]
    map_list = [torch.empty_like(mpad) for _ in range(world)]
    dist.all_gather(traj_list, tpad)
    dist.all_gather(map_list, mpad)
    return (torch.stack(traj_list), torch.stack(map_list),
            all_sizes[:, 0].clone(), all_sizes[:, 1].clone())
