"""N > 1 host path on CPU: two ranks (gloo), one independent sequence each,
results gathered once at the end (aruco_slam_amd/sequences.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aruco_slam_amd.sequences import gather_sequences, rank_seed
    from aruco_slam_amd.synthetic import SyntheticStream, camera_truth
    # rank r: its own stream (seed r); ragged on purpose: different frame / landmark counts
    n, m, frames = 12 + 4 * rank, 4, 5 + 3 * rank
    stream = SyntheticStream(n, m, seed=rank_seed(0, rank))
    traj = torch.zeros((frames, 7), dtype=torch.float64)
    for f in range(frames):
        c, _ = camera_truth(f)
        traj[f, :3] = torch.from_numpy(c)
        traj[f, 3] = rank + 1.0
    lm_map = torch.cat([torch.from_numpy(stream.landmarks), torch.full((n, 3), 0.7 + rank)], dim=1)
    all_traj, all_map, nfr, nlm = gather_sequences(traj, lm_map, dist)
    torch.save({"traj": all_traj, "map": all_map, "nfr": nfr, "nlm": nlm,
                "own_traj": traj, "own_map": lm_map}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]
    for r in range(world):
        assert res[r]["traj"].shape == (2, 8, 7) and res[r]["map"].shape == (2, 16, 6)
        assert res[r]["nfr"].tolist() == [5, 8] and res[r]["nlm"].tolist() == [12, 16]
        # every rank sees every sequence, in rank order, padded with NaN
        for q in range(world):
            f, n = int(res[r]["nfr"][q]), int(res[r]["nlm"][q])
            assert torch.equal(res[r]["traj"][q, :f], res[q]["own_traj"])
            assert torch.equal(res[r]["map"][q, :n], res[q]["own_map"])
            assert torch.isnan(res[r]["traj"][q, f:]).all() and torch.isnan(res[r]["map"][q, n:]).all()
    # the two sequences really are different streams
    assert not torch.equal(res[0]["own_map"][:12, :3], res[1]["own_map"][:12, :3])


def test_single_process_gather_is_a_reshape():
    from aruco_slam_amd.sequences import gather_sequences
    t, m = torch.rand(4, 7, dtype=torch.float64), torch.rand(3, 6, dtype=torch.float64)
    at, am, nf, nl = gather_sequences(t, m, None)
    assert torch.equal(at[0], t) and torch.equal(am[0], m) and nf.tolist() == [4] and nl.tolist() == [3]
