"""Diagnostics: updates/s through the per-frame host boundary (observe(ids, poses) + get_poses() every frame), n=1024, m=32."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
n, m = 1024, 32
s = SyntheticStream(n, m, seed=0)
flt = EKF(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0]), max_landmarks=n, max_visible=m, cov_dtype="float32")
for ids, poses in s.bootstrap(): flt.observe(ids, poses)
frames = list(s.steady(400))
for ids, poses in frames[:50]: flt.observe(ids, poses); flt.get_poses()
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    for ids, poses in frames[50:]: flt.observe(ids, poses); flt.get_poses()
    best = max(best, 350 / (time.perf_counter() - t0))
print({k: v for k, v in os.environ.items() if k.startswith("EKF_")}, "host boundary: %.0f updates/s (%.1f us per frame)" % (best, 1e6 / best), flush=True)
