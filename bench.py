#!/usr/bin/env python3
"""Benchmark of the EKF-SLAM update path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one ``observe()`` (predict + update) of the headline workload
(BASELINE.json configs[2]): n=1024 landmarks, m=32 detections per frame,
fp32 covariance / fp64 state, synthetic detections already resident in HBM.
With N > 1 (launched by torch.distributed.run) every rank runs its own
independent sequence (seed = rank) on its own GPU -- the path shards by
sequence, there is no collective in the frame loop -- and the trajectory and
map are gathered once at the end over RCCL ("scaling": "weak").

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : covariance-update kernel (P <- P + Q - W^T W), algorithmic
                 bytes 2 N^2 sizeof(T) per launch / mean launch duration measured
                 with HIP events on the filter's stream in an instrumented
                 repeat of the timed steps
  cpu_baseline : the CPU oracle in the reference's own operation sequence
                 (oracle/ekf_numpy.py, mode="reference_ops") timed on this
                 host on a bounded sample of the same workload (N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"float32": 157.3, "float64": 78.6}     # dense matrix peaks
INIT_POSE = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--landmarks", type=int, default=1024)
    ap.add_argument("--visible", type=int, default=32)
    ap.add_argument("--cov-dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--cov-kernel", default="auto", choices=["auto", "valu", "mfma", "mfma_tile", "mfma_macro"])
    ap.add_argument("--lookahead", choices=["auto", "on", "off"], default="auto",
                    help="pipelined sequence mode: front kernel of frame t+1 beside the covariance update of frame t (auto = by size)")
    ap.add_argument("--unfused", action="store_true",
                    help="gather / solve / panel as separate launches instead of the fused front kernel")
    ap.add_argument("--burn-in", type=int, default=3000,
                    help="untimed frames on a scratch filter of the same configuration before the warm-up steps (GPU clocks "
                         "at their sustained level); 0 = none")
    ap.add_argument("--cpu-frames", type=int, default=12,
                    help="steady-state frames of the CPU baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(n, m, frames):
    """CPU oracle on the same stream (seed 0), bootstrap untimed (fast mode, same results), then
    timed steady-state updates, per BASELINE.md section 4:
      * reference_ops (the reference's own op sequence) on all host threads  -> `value`
      * reference_ops pinned to ONE thread (fewer frames: bounded sample)
      * fast mode (rank-k update through BLAS) on all threads: separates the algorithmic speed-up
        (2 N^3 -> 2 N^2 k) from the hardware one
    Every leg reports mean, p10 and p90 of the per-frame times."""
    import copy
    from aruco_slam_amd.synthetic import SyntheticStream
    from oracle.ekf_numpy import OracleEKF
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits = None
        threads = os.cpu_count() or 1
    stream = SyntheticStream(n, m, seed=0)
    orc = OracleEKF(INIT_POSE, mode="fast")
    for ids, poses in stream.bootstrap():
        orc.observe(list(ids), poses)
    steady_frames = [(list(ids), poses.copy()) for ids, poses in stream.steady(frames)]

    def leg(mode, count, limit=None):
        o = copy.deepcopy(orc)
        o.mode = mode
        per, cams = [], []
        ctx = threadpool_limits(limits=limit) if (limit and threadpool_limits) else None
        try:
            for ids, poses in steady_frames[:count]:
                t0 = time.perf_counter()
                o.observe(ids, poses)
                per.append(time.perf_counter() - t0)
                cams.append(np.asarray(o.state[:7], dtype=np.float64).copy())
        finally:
            if ctx is not None:
                ctx.restore_original_limits()
        per = np.asarray(per)
        st = per[1:] if len(per) > 2 else per            # first frame pays allocator warm-up
        return {"updates_per_s": float(1.0 / st.mean()), "ms_per_update": float(1e3 * st.mean()),
                "ms_p10": float(1e3 * np.percentile(st, 10)), "ms_p90": float(1e3 * np.percentile(st, 90)),
                "frames_timed": int(len(st))}, np.stack(cams)

    ref_all, cams = leg("reference_ops", frames)
    one_frames = max(3, min(frames, 4))
    ref_one, _ = leg("reference_ops", one_frames, limit=1) if threadpool_limits else ({"skipped": "threadpoolctl missing"}, None)
    fast_all, _ = leg("fast", frames)
    return {
        "value": ref_all["updates_per_s"], "unit": "updates/s", "cores": int(threads),
        "kind": "port",
        "sample": (f"{frames} steady-state frames (first discarded) of the same n={n}, m={m} "
                   f"stream, NumPy/SciPy restatement in the reference's op sequence "
                   f"(dense Q, CSR of P, spsolve, dense (I-KH)P), host cpu_count={os.cpu_count()}"),
        "ms_per_update": ref_all["ms_per_update"], "ms_p10": ref_all["ms_p10"], "ms_p90": ref_all["ms_p90"],
        "one_core": dict(ref_one, cores=1, sample=f"{one_frames} frames, same op sequence, BLAS/OpenMP pools limited to 1 thread"),
        "fast_mode": dict(fast_all, cores=int(threads),
                          sample="same frames, rank-k update P - K(HP) through BLAS (2 N^2 k flops instead of 2 N^3)"),
    }, cams


def main():
    args = parse_args()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N "
                             "bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the EKF path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    from aruco_slam_amd.synthetic import SyntheticStream

    n, m, k_steps, w_steps = args.landmarks, args.visible, args.steps, args.warmup
    dims = 3 * n + 10
    elem = 4 if args.cov_dtype == "float32" else 8

    # ---- untimed: filter, bootstrap through observe(), resident detections --
    flt = EKF(INIT_POSE, max_landmarks=n, max_visible=m, cov_dtype=args.cov_dtype,
              cov_kernel=args.cov_kernel, device=dev, fused=not args.unfused,
              lookahead={"auto": None, "on": True, "off": False}[args.lookahead])
    from aruco_slam_amd.sequences import rank_seed
    stream = SyntheticStream(n, m, seed=rank_seed(0, rank))
    for ids, poses in stream.bootstrap():
        flt.observe(ids, poses)
    total = w_steps + 3 * k_steps
    frames = list(stream.steady(total))
    idx_all = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device=dev)
    z_all = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device=dev)
    traj = torch.zeros((total, 7), dtype=torch.float64, device=dev)
    hip = flt.backend
    hip.sync()

    def run(lo, hi):
        hip.observe_sequence(idx_all[lo:hi], z_all[lo:hi], traj[lo:hi])

    if args.burn_in > 0:
        # a scratch filter (own buffers, own stream of detections): the measured filter's state is not touched
        b_flt = EKF(INIT_POSE, max_landmarks=n, max_visible=m, cov_dtype=args.cov_dtype, cov_kernel=args.cov_kernel,
                    device=dev, fused=not args.unfused, lookahead={"auto": None, "on": True, "off": False}[args.lookahead])
        b_stream = SyntheticStream(n, m, seed=rank_seed(1000, rank))
        for ids, poses in b_stream.bootstrap():
            b_flt.observe(ids, poses)
        b_frames = list(b_stream.steady(min(args.burn_in, 256)))
        b_idx = torch.tensor(np.stack([f[0] for f in b_frames]), dtype=torch.int32, device=dev)
        b_z = torch.tensor(np.stack([f[1][:, :3] for f in b_frames]), dtype=torch.float64, device=dev)
        done = 0
        while done < args.burn_in:                    # (the same 256 frames over and over: only the load matters)
            b_flt.backend.observe_sequence(b_idx, b_z, None)
            done += len(b_frames)
        b_flt.backend.sync()
        del b_flt, b_idx, b_z
    run(0, w_steps)
    hip.sync()
    # (the timed call's arguments -- three tensor views -- are made here: building them is harness work, not the path)
    timed_args = (idx_all[w_steps:w_steps + k_steps], z_all[w_steps:w_steps + k_steps], traj[w_steps:w_steps + k_steps])

    # ---- timed region: exactly K steps ---------------------------------------
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.observe_sequence(*timed_args)
    torch.cuda.synchronize()          # (every stream of the device, the filter's internal one included)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    hip.sync()                        # status word of the filter: raises if any frame of the timed region failed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- final gather of trajectory + map (once per run, RCCL over xGMI) ------
    from aruco_slam_amd.sequences import gather_sequences
    state = torch.as_tensor(hip.get_state(), device=dev)
    diag = torch.as_tensor(hip.get_cov_diag(), device=dev)
    map_t = torch.cat([state[10:].reshape(n, 3), diag[10:].reshape(n, 3)], dim=1).contiguous()
    traj_timed = traj[w_steps:w_steps + k_steps].contiguous()
    torch.cuda.synchronize()
    g0 = time.perf_counter()
    all_traj, all_map, _, _ = gather_sequences(traj_timed, map_t, dist)
    torch.cuda.synchronize()
    gather_ms = 1e3 * (time.perf_counter() - g0) if dist is not None else 0.0
    finite = bool(torch.isfinite(all_traj).all() and torch.isfinite(all_map).all())
    assert all_traj.shape == (world, k_steps, 7) and all_map.shape == (world, n, 6)

    # ---- instrumented repeat: per-kernel HIP-event timing on the filter stream -
    # pass A: the covariance-update kernel alone (start / stop events attached to its dispatch) -> roofline
    hip.set_kernel_timing(2)
    run(w_steps + k_steps, w_steps + 2 * k_steps)
    cov_us, cov_launches = hip.kernel_timing()["cov_update"]
    # pass B: every kernel (front kernel: events recorded around the launch, i.e. including its launch gap)
    hip.set_kernel_timing(1)
    run(w_steps + 2 * k_steps, total)
    timing = hip.kernel_timing()
    hip.set_kernel_timing(0)
    # host-pointer boundary as BaseFilter.process_frame drives it: observe(ids, poses) with
    # host arrays + get_poses() (device->host sync) every frame.  PCIe-inclusive; never `value`.
    # Five segments of 50 frames, the best one counts: directly after the sequence calls above the HIP runtime is still
    # retiring their thousands of launches and events, and API calls are slow for tens of milliseconds.
    hb_frames = list(stream.steady(250))
    torch.cuda.synchronize()
    host_boundary = 0.0
    for seg in range(5):
        h0 = time.perf_counter()
        for ids_h, poses_h in hb_frames[50 * seg:50 * seg + 50]:
            flt.observe(ids_h, poses_h)
            flt.get_poses()
        host_boundary = max(host_boundary, 50 / (time.perf_counter() - h0))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    algo_bytes = 2.0 * dims * dims * elem                       # read P once, write once
    achieved = algo_bytes / (cov_us * 1e-6) / 1e9 if cov_us > 0 else 0.0
    k = 3 * m
    # everything else of the update (front kernel): SURVEY 8(d) "extra" bytes = support rows of P
    # gathered for A = H P, the W panel written + read, the state
    front_bytes = (10.0 + k) * dims * elem + 2.0 * dims * k * elem + 16.0 * dims
    front_us = timing.get("front", timing.get("gather", (0.0, 0)))[0] if "front" in timing else \
        sum(timing[name][0] for name in ("gather", "solve", "panel") if name in timing)
    # HBM bytes per launch from rocprofv3 PMC passes (tools/profile_round.sh +
    # tools/summarize_profile.py), and the rocprofv3 kernel-trace duration next to the
    # HIP-event one (dispatch time stamps; they start a little before the first wave does).
    # NOT measured in this run: constants from the committed profile named in traffic_source.
    traffic = rocprof_us = traffic_src = None
    pmc = REPO / "profiles" / "cov_update_pmc_traffic.json"
    if pmc.exists():
        try:
            table = json.loads(pmc.read_text())
            key = f"n{n}_m{m}_{args.cov_dtype}"
            traffic = table.get(key)
            rocprof_us = table.get(key + "_rocprof_mean_us")
            traffic_src = table.get(key + "_source")
        except Exception:
            traffic = rocprof_us = traffic_src = None
    mfma_peak = MFMA_PEAK_TFLOPS[args.cov_dtype]
    executed_flops = 1.0 * dims * dims * k                      # symmetric kernel: lower-triangle tiles only
    secs = cov_us * 1e-6 if cov_us > 0 else float("inf")
    # which roof bounds the kernel: algorithmic intensity k / sizeof(T) flop per byte against the ridge
    # (157.3 TF / 8 TB/s = 19.7 for f32): C2 6, C3 24 (on the ridge: HBM), C5 48 (MFMA)
    ridge = mfma_peak * 1e3 / HBM_PEAK_GBS
    if k / elem > 1.5 * ridge:
        algo_tf = 2.0 * dims * dims * k / secs / 1e12
        rl_primary = {"bound": "mfma", "achieved": algo_tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": algo_tf / mfma_peak,
                      "note": "algorithmic flops 2 N^2 k; the symmetric kernel EXECUTES N^2 k (see mfma_util), so frac may exceed 1"}
    else:
        rl_primary = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    out = {
        "metric": "EKF updates/sec at n=1024 landmarks, m=32 obs/frame" if (n, m) == (1024, 32)
                  else f"EKF updates/sec at n={n} landmarks, m={m} obs/frame",
        "value": world * k_steps / elapsed,
        "unit": "updates/s",
        "n_gpus": world, "steps": k_steps, "warmup": w_steps,
        "ms_per_step": 1e3 * elapsed / k_steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 covariance / f64 state" if elem == 4 else "f64",
        "data": "synthetic",
        "config": {"workload": f"n={n} landmarks, m={m} visible/frame, N={dims}, k={3 * m}, "
                               f"{args.cov_dtype} covariance, one independent sequence per GPU",
                   "sequences": world, "cov_kernel": args.cov_kernel,
                   "burn_in_frames": args.burn_in,
                   "burn_in": "untimed frames on a scratch filter before the warm-up steps, like the bootstrap: the metric is "
                              "steady-state throughput and the GPU's clocks need tens of ms of load to reach their sustained "
                              "level (20 steps per call: 35.5k updates/s without, 37.5k with)",
                   "front": "stage kernels" if args.unfused else "fused front kernel",
                   "sequence_mode": {"auto": "pipelined where it wins (N >= 200 except N > 9000 with k > 96: "
                                             "front kernel of frame t+1 beside the covariance update of frame t, "
                                             "covariance ping-pong between two buffers), else serial",
                                     "on": "pipelined", "off": "serial"}[args.lookahead]},
        "roofline": dict(rl_primary, **{
                     "kernel": "ekf_cov_update (P <- P + Q - W^T W)",
                     "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS},
                     "traffic": traffic,
                     "traffic_source": traffic_src and (traffic_src + " (rocprofv3 PMC passes of an earlier run of this "
                                                        "command; not measured in this run)"),
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "mean_launch_us": cov_us, "launches_timed": cov_launches,
                     "timing": "HIP events attached to the dispatch (start/stop time stamps of the kernel itself), in an "
                               "instrumented repeat of the timed steps in SERIAL order (the kernel alone on the GPU); in the "
                               "pipelined timed region it runs beside the next frame's front kernel and takes longer (hidden)",
                     "rocprofv3_mean_us": rocprof_us,
                     "residency": ("P is %.0f MB padded: it stays in the 256 MB Infinity Cache between frames, so "
                                   "`achieved` is an algorithmic-bytes rate, not DRAM traffic" % (flt.backend.ld ** 2 * elem / 1e6))
                                  if flt.backend.ld ** 2 * elem < 200e6 else
                                  "P is %.0f MB padded: larger than the Infinity Cache, streamed from HBM" % (flt.backend.ld ** 2 * elem / 1e6),
                     "algorithmic_flops_per_launch": 2.0 * dims * dims * k,
                     "algorithmic_tflops": 2.0 * dims * dims * k / secs / 1e12,
                     "executed_flops_per_launch": executed_flops,
                     "mfma_util": executed_flops / secs / 1e12 / mfma_peak,
                     "mfma_peak_tflops": mfma_peak,
                     "front": {"kernel": "ekf_front_kernel (measurement model, S, Cholesky, W, dx, injection)"
                                         if not args.unfused else "gather + solve + panel",
                               "algorithmic_bytes_per_launch": front_bytes, "mean_launch_us": front_us,
                               "timing": "HIP events recorded around the launch (includes the launch gap)",
                               "achieved": front_bytes / (front_us * 1e-6) / 1e9 if front_us > 0 else 0.0,
                               "frac": front_bytes / (front_us * 1e-6) / 1e9 / HBM_PEAK_GBS if front_us > 0 else 0.0,
                               "bound": "latency (serial pivot chain of the k x k Cholesky on one workgroup)"},
                     "whole_frame": {"algorithmic_bytes": algo_bytes + front_bytes,
                                     "achieved": (algo_bytes + front_bytes) / (elapsed / k_steps) / 1e9,
                                     "frac": (algo_bytes + front_bytes) / (elapsed / k_steps) / 1e9 / HBM_PEAK_GBS}}),
        "kernel_us": {name: round(us, 3) for name, (us, _) in timing.items()},
        "gather_ms": gather_ms,
        "host_boundary_updates_per_s": host_boundary,
        "outputs_finite": finite,
    }
    if world == 1 and args.cpu_frames > 0:
        base, cpu_cams = cpu_baseline(n, m, args.cpu_frames)
        out["cpu_baseline"] = base
        gpu_cams = traj[:args.cpu_frames].cpu().numpy() if args.cpu_frames <= total else None
        if gpu_cams is not None:
            d = gpu_cams - cpu_cams
            out["trajectory_l2_vs_cpu"] = float(np.sqrt((d[:, :3] ** 2).sum(axis=1)).max())
            out["trajectory_rel_vs_cpu"] = float(np.abs(d).max() / max(1.0, np.abs(cpu_cams).max()))
            out["trajectory_frames_compared"] = int(args.cpu_frames)
        out["speedup_vs_cpu_baseline"] = out["value"] / base["value"]
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
