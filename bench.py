#!/usr/bin/env python3
"""Benchmark of the EKF-SLAM update path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one ``observe()`` (predict + update) of the headline workload
(BASELINE.json configs[2]): n=1024 landmarks, m=32 detections per frame,
fp32 covariance / fp64 state, synthetic detections already resident in HBM.
With N > 1 every rank runs its own independent sequence (seed = rank) on its
own GPU -- the path shards by sequence, there is no collective in the frame
loop -- and the trajectory and map are gathered once at the end over RCCL
("scaling": "weak").  `python bench.py --gpus N` starts the N ranks itself
(`python -m torch.distributed.run` as a child process, before anything in
this process touches the GPU); under an outer torch.distributed.run it is a
rank.

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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--filter", default="ekf", choices=["ekf", "ekf_rotations"],
                    help="filter model (run_slam.py --filter); ekf_rotations defaults to n=400, m=27 (N=4010, k=189)")
    ap.add_argument("--landmarks", type=int, default=None)
    ap.add_argument("--visible", type=int, default=None)
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
    args = ap.parse_args(argv)
    if args.landmarks is None:
        args.landmarks = 400 if args.filter == "ekf_rotations" else 1024
    if args.visible is None:
        args.visible = 27 if args.filter == "ekf_rotations" else 32
    return args


def launcher_command(gpus, argv, port):
    """The command that runs this benchmark on `gpus` ranks of one node: what `python bench.py --gpus N` starts as a child
    process when it is not itself a rank (no WORLD_SIZE in the environment)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), str(Path(__file__).resolve()), *argv]


def relaunch_as_ranks(args, argv):
    """N > 1 without an outer launcher: start the ranks as a child process (a fresh process each: nothing here has touched
    the GPU yet), relay rank 0's JSON line and exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(launcher_command(args.gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(proc.stdout)
    return proc.returncode


def cpu_baseline(n, m, frames, model="ekf"):
    """CPU oracle on the same stream (seed 0), bootstrap untimed (fast mode, same results), then
    timed steady-state updates, per BASELINE.md section 4:
      * reference_ops (the reference's own op sequence) on all host threads  -> `value`
      * reference_ops pinned to ONE thread (fewer frames: bounded sample)
      * fast mode (rank-k update through BLAS) on all threads: separates the algorithmic speed-up
        (2 N^3 -> 2 N^2 k) from the hardware one
    Every leg reports mean, p10 and p90 of the per-frame times."""
    import copy
    from aruco_slam_amd.synthetic import SyntheticStream
    from oracle.ekf_numpy import OracleEKF, OracleEKFRotations
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits = None
        threads = os.cpu_count() or 1
    rot = model == "ekf_rotations"
    stream = SyntheticStream(n, m, seed=0, rvec_sigma=0.05 if rot else 0.0)
    orc = (OracleEKFRotations if rot else OracleEKF)(INIT_POSE, mode="fast")
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
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as the driver calls it: become the launcher (before torch / HIP are touched)
        raise SystemExit(relaunch_as_ranks(args, argv))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the EKF path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))
        world = dist.get_world_size()              # (what RCCL reports, not what the environment asked for)

    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    from aruco_slam_amd.filters.ekf_with_rotations import EKF_Rotations, euler_xyz_to_quat
    from aruco_slam_amd.synthetic import SyntheticStream

    rot = args.filter == "ekf_rotations"
    n, m, k_steps, w_steps = args.landmarks, args.visible, args.steps, args.warmup
    lmd, rd = (10, 7) if rot else (3, 3)
    dims = lmd * n + 10
    k = rd * m
    elem = 4 if args.cov_dtype == "float32" else 8
    lookahead = {"auto": None, "on": True, "off": False}[args.lookahead]

    def make_filter():
        if rot:
            return EKF_Rotations(INIT_POSE, max_landmarks=n, max_visible=m, cov_dtype=args.cov_dtype, cov_kernel=args.cov_kernel,
                                 device=dev, fused=not args.unfused, lookahead=lookahead)
        return EKF(INIT_POSE, max_landmarks=n, max_visible=m, cov_dtype=args.cov_dtype, cov_kernel=args.cov_kernel, device=dev,
                   fused=not args.unfused, lookahead=lookahead)

    def resident(frames):
        """Detections of `frames` as the device arrays observe_sequence takes: indices [F,m], z [F,m,rd]."""
        idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device=dev)
        if rot:       # z = [tvec | quaternion of from_euler("xyz", rvec)] (ekf_with_rotations.py:216-224)
            z_np = np.stack([np.hstack((f[1][:, :3], euler_xyz_to_quat(f[1][:, 3:6]))) for f in frames])
        else:
            z_np = np.stack([f[1][:, :3] for f in frames])
        return idx, torch.tensor(z_np, dtype=torch.float64, device=dev)

    # ---- untimed: filter, bootstrap through observe(), resident detections --
    flt = make_filter()
    from aruco_slam_amd.sequences import rank_seed
    stream = SyntheticStream(n, m, seed=rank_seed(0, rank), rvec_sigma=0.05 if rot else 0.0)
    for ids, poses in stream.bootstrap():
        flt.observe(ids, poses)
    total = w_steps + 3 * k_steps
    frames = list(stream.steady(total))
    idx_all, z_all = resident(frames)
    traj = torch.zeros((total, 7), dtype=torch.float64, device=dev)
    hip = flt.backend
    hip.sync()

    def run(lo, hi):
        hip.observe_sequence(idx_all[lo:hi], z_all[lo:hi], traj[lo:hi])

    if args.burn_in > 0:
        # a scratch filter (own buffers, own stream of detections): the measured filter's state is not touched
        b_flt = make_filter()
        b_stream = SyntheticStream(n, m, seed=rank_seed(1000, rank), rvec_sigma=0.05 if rot else 0.0)
        for ids, poses in b_stream.bootstrap():
            b_flt.observe(ids, poses)
        b_frames = list(b_stream.steady(min(args.burn_in, 256)))
        b_idx, b_z = resident(b_frames)
        done = 0
        while done < args.burn_in:                    # (the same 256 frames over and over: only the load matters)
            b_flt.backend.observe_sequence(b_idx, b_z, None)
            done += len(b_frames)
        b_flt.backend.sync()                          # (also hands the process-wide pipelining token back)
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
    own_elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    mode_ran = hip.last_sequence_mode()
    hip.sync()                        # status word of the filter: raises if any frame of the timed region failed
    per_rank = [k_steps / own_elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([k_steps / own_elapsed], dtype=torch.float64, device=dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(v.item()) for v in every]

    # ---- final gather of trajectory + map (once per run, RCCL over xGMI) ------
    from aruco_slam_amd.sequences import gather_sequences
    state = torch.as_tensor(hip.get_state(), device=dev)
    diag = torch.as_tensor(hip.get_cov_diag(), device=dev)
    map_t = torch.cat([state[10:].reshape(n, lmd)[:, :3], diag[10:].reshape(n, lmd)[:, :3]], dim=1).contiguous()
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
    cov_us_a, cov_launches_a = hip.kernel_timing()["cov_update"]
    # pass B: every kernel (front kernel: events recorded around the launch, i.e. including its launch gap)
    hip.set_kernel_timing(1)
    run(w_steps + 2 * k_steps, total)
    timing = hip.kernel_timing()
    hip.set_kernel_timing(0)
    # The roofline uses pass B's figure for the covariance update: the same pair of events attached to the dispatch, but with
    # an event record between the front kernel and the update.  Pass A (nothing between the two launches) reads 7 - 13 % more
    # than pass B, tools/cov_bench.py and rocprofv3's kernel trace of the same run, which agree with each other (C5: 339 vs
    # 295 / 297 / 296 us; C3: 16.8 vs 15.0 / 15.7); why pass A reads more in this process is not understood.  Kept in the
    # line as `mean_launch_us_back_to_back`.
    cov_us, cov_launches = timing["cov_update"]
    if not cov_us > 0:
        cov_us, cov_launches = cov_us_a, cov_launches_a
    # host-pointer boundary as BaseFilter.process_frame drives it: observe(ids, poses) with
    # host arrays + get_poses() (device->host sync) every frame.  PCIe-inclusive; never `value`.
    # Segments of 50 frames; median, minimum and maximum are reported (directly after the sequence calls above the HIP
    # runtime is still retiring their thousands of launches and events, and API calls are slow for tens of milliseconds:
    # two untimed segments first).
    hb_frames = list(stream.steady(450))
    torch.cuda.synchronize()
    hb_rates = []
    for seg in range(9):
        h0 = time.perf_counter()
        for ids_h, poses_h in hb_frames[50 * seg:50 * seg + 50]:
            flt.observe(ids_h, poses_h)
            flt.get_poses()
        if seg >= 2:
            hb_rates.append(50 / (time.perf_counter() - h0))
    hb_rates.sort()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    secs = cov_us * 1e-6 if cov_us > 0 else float("inf")
    # read P once, write P once, 2 N^2 k flops.  The symmetric kernel needs less: it reads the lower triangle
    # only (the upper one is its mirror image, bit for bit) and executes N^2 k flops -- `frac` is against what the kernel
    # needs, `frac_survey` against 8(d)'s figure (which exceeds 1 where the kernel is matrix-bound: half of those flops are
    # never executed).
    survey_bytes = 2.0 * dims * dims * elem
    need_bytes = 1.5 * dims * dims * elem
    survey_flops = 2.0 * dims * dims * k
    need_flops = 1.0 * dims * dims * k
    hbm = {"achieved": need_bytes / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": need_bytes / secs / 1e9 / HBM_PEAK_GBS,
           "bytes_per_launch": need_bytes, "bytes_definition": "1.5 N^2 sizeof(T): lower triangle read, whole matrix written",
           "achieved_survey": survey_bytes / secs / 1e9, "frac_survey": survey_bytes / secs / 1e9 / HBM_PEAK_GBS,
           "bytes_per_launch_survey": survey_bytes}
    mfma_peak = MFMA_PEAK_TFLOPS[args.cov_dtype]
    mfma = {"achieved": need_flops / secs / 1e12, "peak": mfma_peak, "unit": "TFLOP/s", "frac": need_flops / secs / 1e12 / mfma_peak,
            "flops_per_launch": need_flops, "flops_definition": "N^2 k: the symmetric update computes the lower triangle only",
            "achieved_survey": survey_flops / secs / 1e12, "frac_survey": survey_flops / secs / 1e12 / mfma_peak,
            "flops_per_launch_survey": survey_flops}
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
    if pmc.exists() and not rot:
        try:
            table = json.loads(pmc.read_text())
            key = f"n{n}_m{m}_{args.cov_dtype}"
            traffic = table.get(key)
            rocprof_us = table.get(key + "_rocprof_mean_us")
            traffic_src = table.get(key + "_source")
        except Exception:
            traffic = rocprof_us = traffic_src = None
    # which roof bounds the kernel: intensity k / (1.5 sizeof(T)) flop per byte against the ridge
    # (157.3 TF / 8 TB/s = 19.7 for f32): C2 4, C3 16 (HBM side of the ridge), C5 32 (MFMA)
    ridge = mfma_peak * 1e3 / HBM_PEAK_GBS
    primary = mfma if need_flops / need_bytes > ridge else hbm
    rl_primary = {"bound": "mfma" if primary is mfma else "hbm", "achieved": primary["achieved"], "peak": primary["peak"],
                  "unit": primary["unit"], "frac": primary["frac"], "frac_survey": primary["frac_survey"]}
    headline = (n, m, rot) == (1024, 32, False)
    out = {
        "metric": "EKF updates/sec at n=1024 landmarks, m=32 obs/frame" if headline
                  else f"{'EKF_Rotations' if rot else 'EKF'} updates/sec at n={n} landmarks, m={m} obs/frame",
        "value": world * k_steps / elapsed,
        "unit": "updates/s",
        "n_gpus": world, "steps": k_steps, "warmup": w_steps,
        "ms_per_step": 1e3 * elapsed / k_steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 covariance / f64 state" if elem == 4 else "f64",
        "data": "synthetic",
        "config": {"workload": f"{'EKF_Rotations' if rot else 'EKF'}: n={n} landmarks, m={m} visible/frame, N={dims}, k={k}, "
                               f"{args.cov_dtype} covariance, one independent sequence per GPU",
                   "sequences": world, "cov_kernel": args.cov_kernel,
                   "world_size_reported_by_rccl": world if dist is not None else None,
                   "updates_per_s_per_rank": per_rank,
                   "burn_in_frames": args.burn_in,
                   "burn_in": "untimed frames on a scratch filter before the warm-up steps, like the bootstrap: the metric is "
                              "steady-state throughput and the GPU's clocks need tens of ms of load to reach their sustained "
                              "level (20 steps per call: 35.5k updates/s without, 37.5k with)",
                   "front": "stage kernels" if args.unfused else "fused front kernel",
                   "sequence_mode_asked": args.lookahead,
                   "sequence_mode_ran": mode_ran},
        "roofline": dict(rl_primary, **{
                     "kernel": "ekf_cov_update (P <- P + Q - W^T W)",
                     "hbm": hbm, "mfma": mfma,
                     "traffic": traffic,
                     "traffic_source": traffic_src and (traffic_src + " (rocprofv3 PMC passes of an earlier run of this "
                                                        "command; not measured in this run)"),
                     "algorithmic_bytes_per_launch": survey_bytes,
                     "mean_launch_us": cov_us, "launches_timed": cov_launches,
                     "mean_launch_us_back_to_back": cov_us_a,
                     "timing": "HIP events attached to the dispatch (start/stop time stamps of the kernel itself), in an "
                               "instrumented repeat of the timed steps in SERIAL order (the kernel alone on the GPU); in the "
                               "pipelined timed region it runs beside the next frame's front kernel and takes longer (hidden)",
                     "rocprofv3_mean_us": rocprof_us,
                     "residency": ("P is %.0f MB padded: it stays in the 256 MB Infinity Cache between frames, so "
                                   "`achieved` is a rate of needed bytes, not DRAM traffic" % (flt.backend.ld ** 2 * elem / 1e6))
                                  if flt.backend.ld ** 2 * elem < 200e6 else
                                  "P is %.0f MB padded: larger than the Infinity Cache, streamed from HBM" % (flt.backend.ld ** 2 * elem / 1e6),
                     "mfma_util": mfma["frac"],
                     "front": {"kernel": "ekf_front_kernel (measurement model, S, Cholesky, W, dx, injection)"
                                         if not args.unfused else "gather + solve + panel",
                               "algorithmic_bytes_per_launch": front_bytes, "mean_launch_us": front_us,
                               "timing": "HIP events recorded around the launch (includes the launch gap)",
                               "achieved": front_bytes / (front_us * 1e-6) / 1e9 if front_us > 0 else 0.0,
                               "frac": front_bytes / (front_us * 1e-6) / 1e9 / HBM_PEAK_GBS if front_us > 0 else 0.0,
                               "bound": "latency (serial pivot chain of the k x k Cholesky on one workgroup)"},
                     "whole_frame": {"bytes": need_bytes + front_bytes,
                                     "achieved": (need_bytes + front_bytes) / (elapsed / k_steps) / 1e9,
                                     "frac": (need_bytes + front_bytes) / (elapsed / k_steps) / 1e9 / HBM_PEAK_GBS}}),
        "kernel_us": {name: round(us, 3) for name, (us, _) in timing.items()},
        "gather_ms": gather_ms,
        "host_boundary_updates_per_s": hb_rates[len(hb_rates) // 2],
        "host_boundary": {"median": hb_rates[len(hb_rates) // 2], "min": hb_rates[0], "max": hb_rates[-1],
                          "segments": len(hb_rates), "frames_per_segment": 50,
                          "what": "observe(ids, poses) with host arrays + get_poses() every frame, as BaseFilter.process_frame "
                                  "does (PCIe-inclusive)"},
        "outputs_finite": finite,
    }
    if world == 1 and args.cpu_frames > 0:
        base, cpu_cams = cpu_baseline(n, m, args.cpu_frames, args.filter)
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
