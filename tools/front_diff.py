#!/usr/bin/env python3
"""Where do the fused front kernel and the stage kernels first differ? (one frame, intermediates)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
res = []
for fused in (True, False):
    s = SyntheticStream(n, m, seed=1)
    f = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, fused=fused)
    f.backend.debug_enable_w()
    frames = list(s.bootstrap()) + list(s.steady(3))
    per = []
    for ids, poses in frames:
        f.observe(ids, poses)
        per.append({k: f.backend.debug_fetch(k, m) for k in ("jac", "resid", "A", "L", "W")} | {"state": f.state.copy()})
    res.append(per)
for t, (a, b) in enumerate(zip(*res)):
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    if bad:
        print("frame", t, "first differences in:", bad)
        for k in bad:
            d = np.abs(a[k] - b[k])
            idx = np.unravel_index(np.argmax(d), d.shape)
            print("  ", k, "max", d.max(), "at", idx, "count", int((d > 0).sum()))
        break
else:
    print("all frames identical")
