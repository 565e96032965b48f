"""Diagnostics: start-to-start times of the front kernels of ONE short pipelined call (device clock, no profiler):
python tools/frame_starts.py [K=16] [warmup=5]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, m = 1024, 32
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
s = SyntheticStream(n, m, seed=0)
flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32")
flt.backend.debug_enable_stamps(True)
for ids, poses in s.bootstrap(): flt.observe(ids, poses)
frames = list(s.steady(W + 3 * K))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
flt.backend.observe_sequence(idx[:W], z[:W], None); flt.backend.sync()
for rep in range(3):
    lo = W + rep * K
    torch.cuda.synchronize(); t0 = time.perf_counter()
    flt.backend.observe_sequence(idx[lo:lo + K], z[lo:lo + K], None)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e6
    st = flt.backend.debug_fetch("stamps", m)
    seq_last = 32 + W + (rep + 1) * K            # fused-frame number of the call's last frame (bootstrap: 32 frames)
    order = [(seq_last - K + 1 + i) & 15 for i in range(min(K, 16))] if K <= 16 else [(seq_last - 15 + i) & 15 for i in range(16)]
    starts = np.array([st[16 + o] for o in order]) / 100.0
    print("call %d: wall %.1f us = %.2f per frame; start-to-start of its %s frames: %s  (first to last start: %.1f us)" % (
        rep, wall, wall / K, "last 16" if K > 16 else "", np.round(np.diff(starts), 1).tolist(), starts[-1] - starts[0]), flush=True)
