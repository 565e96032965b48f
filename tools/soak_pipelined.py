#!/usr/bin/env python3
"""Soak test (diagnostics): the pipelined sequence mode (front kernel of frame t+1 beside the covariance update
of frame t, device-side gates between the two streams) against the serial order, thousands of frames back to
back in several calls, several sizes and both filters; state, covariance diagonal and trajectory must be
bitwise equal, no status bit.  Prints the wall time per frame of both."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.filters.ekf_with_rotations import EKF_Rotations, euler_xyz_to_quat
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
cases = [(EKF, 1024, 32, "float32", 6000, 3), (EKF, 256, 16, "float64", 3000, 2), (EKF, 4096, 64, "float32", 300, 2),
         (EKF, 2048, 48, "float32", 600, 3), (EKF_Rotations, 400, 27, "float32", 1000, 4), (EKF, 64, 3, "float32", 3000, 7)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for cls, n, m, dtype, nfr, ncalls in cases:
    s = SyntheticStream(n, m, seed=5, rvec_sigma=0.05)
    boot = list(s.bootstrap())
    frames = list(s.steady(nfr))
    idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
    if cls is EKF:
        z = np.stack([f[1][:, :3] for f in frames])
        kw = dict(quat_update="scalar_first")       # the as-written quaternion rule is chaotic over thousands of frames
    else:
        z = np.stack([np.hstack((f[1][:, :3], euler_xyz_to_quat(f[1][:, 3:6]))) for f in frames])
        kw = {}
    z = torch.tensor(z, dtype=torch.float64, device="cuda")
    outs, times = [], []
    for pipelined in (True, False):
        flt = cls(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, lookahead=pipelined, **kw)
        for ids, poses in boot:
            flt.observe(ids, poses)
        traj = torch.zeros((nfr, 7), dtype=torch.float64, device="cuda")
        flt.backend.sync()
        t0 = time.perf_counter()
        step = -(-nfr // ncalls)
        for lo in range(0, nfr, step):              # several calls: the join / restart of the pipeline is exercised too
            flt.backend.observe_sequence(idx[lo:lo + step], z[lo:lo + step], traj[lo:lo + step])
        flt.backend.sync()
        times.append((time.perf_counter() - t0) / nfr * 1e6)
        outs.append((flt.state, flt.backend.get_cov_diag(), traj.cpu().numpy()))
        del flt
    ok = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    print(cls.__name__, n, m, dtype, nfr, "frames in", ncalls, "calls: bitwise equal", ok, "finite", bool(np.isfinite(outs[0][0]).all()),
          " us/frame pipelined %.1f serial %.1f" % (times[0], times[1]), flush=True)
