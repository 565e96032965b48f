#!/usr/bin/env python3
"""Find the first frame where the fused front kernel and the separate launches differ at n=4096, m=64."""
import sys
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
fs = [EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=f) for f in (True, False)]
s = SyntheticStream(n, m, seed=0)
frames = list(s.bootstrap()) + list(s.steady(20))
for t, (ids, poses) in enumerate(frames):
    for f in fs:
        f.observe(ids, poses)
    try:
        a, b = fs[0].state, fs[1].state
    except Exception as e:
        print("frame", t, "error", e)
        break
    d = np.abs(a - b)
    if not np.array_equal(a, b):
        bad = np.nonzero(a != b)[0]
        print("frame", t, "dims", len(a), "state differs at", len(bad), "entries; first", bad[:10], "max", d.max(), flush=True)
        pa, pb = fs[0].uncertainty, fs[1].uncertainty
        bc = np.nonzero((pa != pb).any(axis=0))[0]
        print("  cov columns differing:", len(bc), bc[:20], "...", bc[-5:])
        break
else:
    print("all", len(frames), "frames bitwise equal")
