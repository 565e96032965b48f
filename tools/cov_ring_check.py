#!/usr/bin/env python3
"""MFMA f32 covariance update vs the VALU reference kernel, bitwise, for every KB (k = 16 KB)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
for m in (2, 5, 8, 13, 16, 21, 27, 32, 38, 43, 48, 53, 59, 64):
    outs = []
    for kern in ("mfma", "valu"):
        s = SyntheticStream(128, m, seed=1)
        f = EKF(INIT, max_landmarks=128, max_visible=m, cov_dtype="float32", cov_kernel=kern)
        for ids, poses in list(s.bootstrap()) + list(s.steady(3)):
            f.observe(ids, poses)
        outs.append((f.state, f.uncertainty))
    print("m", m, "KB", -(-3 * m // 16), "equal", np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]),
          "finite", bool(np.isfinite(outs[0][1]).all()), flush=True)
