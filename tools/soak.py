#!/usr/bin/env python3
"""Soak test (diagnostics): thousands of frames back to back through the sequence entry point, fused front
kernel vs separate launches, several sizes; final state and covariance diagonal must be bitwise equal."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.filters.ekf_with_rotations import EKF_Rotations, euler_xyz_to_quat
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
cases = [(EKF, 1024, 32, "float32", 3000), (EKF, 256, 16, "float64", 3000), (EKF, 4096, 64, "float32", 300),
         (EKF, 2048, 48, "float32", 600), (EKF_Rotations, 400, 27, "float32", 1000), (EKF, 64, 3, "float64", 3000)]
for cls, n, m, dtype, nfr in cases:
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
    outs = []
    for fused in (True, False):
        flt = cls(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, fused=fused, **kw)
        for ids, poses in boot:
            flt.observe(ids, poses)
        flt.backend.observe_sequence(idx, z, None)
        flt.backend.sync()
        outs.append((flt.state, flt.backend.get_cov_diag()))
        del flt
    ok = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    print(cls.__name__, n, m, dtype, nfr, "frames: bitwise equal", ok, "finite", bool(np.isfinite(outs[0][0]).all()), flush=True)
