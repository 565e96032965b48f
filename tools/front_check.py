#!/usr/bin/env python3
"""Fused front kernel vs the three separate launches: bitwise comparison + timing (diagnostics)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF          # noqa: E402
from aruco_slam_amd.filters.ekf_with_rotations import EKF_Rotations    # noqa: E402
from aruco_slam_amd.synthetic import SyntheticStream                   # noqa: E402

INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])


def run(cls, n, m, dtype, fused, frames=6, rv=0.0):
    s = SyntheticStream(n, m, seed=1, rvec_sigma=rv)
    flt = cls(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, fused=fused)
    for ids, poses in s.bootstrap():
        flt.observe(ids, poses)
    for ids, poses in s.steady(frames):
        flt.observe(ids, poses)
    t0 = time.perf_counter()
    st = flt.state
    return st, flt.uncertainty, time.perf_counter() - t0


cases = [(EKF, 256, 16, "float32", 0.0), (EKF, 256, 8, "float32", 0.0), (EKF, 100, 16, "float32", 0.0), (EKF, 24, 8, "float64", 0.0), (EKF, 24, 8, "float32", 0.0), (EKF, 256, 16, "float64", 0.0),
         (EKF_Rotations, 20, 6, "float64", 0.05), (EKF, 1024, 32, "float32", 0.0), (EKF, 128, 64, "float32", 0.0),
         (EKF_Rotations, 40, 27, "float32", 0.05)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for cls, n, m, dtype, rv in cases:
    a = run(cls, n, m, dtype, True, rv=rv)
    b = run(cls, n, m, dtype, False, rv=rv)
    print(cls.__name__, n, m, dtype, "state equal", np.array_equal(a[0], b[0]), "cov equal", np.array_equal(a[1], b[1]),
          "max|dstate|", float(np.abs(a[0] - b[0]).max()), "finite", bool(np.isfinite(a[0]).all()), flush=True)
