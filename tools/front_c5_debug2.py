#!/usr/bin/env python3
"""Sequence entry point at n=4096, m=64: fused vs separate launches, with and without event timing."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
fs = [EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=f) for f in (True, False)]
s = SyntheticStream(n, m, seed=0)
for ids, poses in s.bootstrap():
    for f in fs:
        f.observe(ids, poses)
frames = list(s.steady(120))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
lo = 0
for name, timing, cnt in (("plain", 0, 30), ("timing=2", 2, 30), ("timing=1", 1, 30), ("plain again", 0, 30)):
    for f in fs:
        f.backend.set_kernel_timing(timing)
        f.backend.observe_sequence(idx[lo:lo + cnt], z[lo:lo + cnt], None)
    lo += cnt
    res = []
    for f in fs:
        try:
            f.backend.sync()
            res.append(f.state)
        except Exception as e:
            res.append(None)
            print(name, "error:", str(e)[:100])
    if res[0] is not None and res[1] is not None:
        print(name, "fused == separate:", np.array_equal(res[0], res[1]), "max diff", float(np.abs(res[0] - res[1]).max()), flush=True)
    else:
        break
