#!/usr/bin/env python3
"""Where the per-frame host boundary (observe + get_poses every frame, BaseFilter.process_frame) spends its time: the Python
observe() call, the get_poses() call (waits for the front kernel), and the same two steps straight through the C ABI."""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m = 1024, 32
s = SyntheticStream(n, m, seed=0)
flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32")
for ids, poses in s.bootstrap():
    flt.observe(ids, poses)
frames = list(s.steady(1200))
for ids, poses in frames[:200]:
    flt.observe(ids, poses); flt.get_poses()
t_obs, t_get = [], []
t0 = time.perf_counter()
for ids, poses in frames[200:700]:
    a = time.perf_counter(); flt.observe(ids, poses); b = time.perf_counter(); flt.get_poses(); c = time.perf_counter()
    t_obs.append(b - a); t_get.append(c - b)
tot = (time.perf_counter() - t0) / 500 * 1e6
print(f"python loop: {tot:.1f} us/frame  observe() {np.median(t_obs) * 1e6:.1f}  get_poses() {np.median(t_get) * 1e6:.1f} (medians)")
hip = flt.backend
lib = hip.lib
idx = [np.ascontiguousarray(f[0], dtype=np.int32) for f in frames[700:1200]]
zz = [np.ascontiguousarray(f[1][:, :3]) for f in frames[700:1200]]
ip = [a.ctypes.data_as(C.POINTER(C.c_int32)) for a in idx]
zp = [a.ctypes.data_as(C.POINTER(C.c_double)) for a in zz]
out = np.empty(hip.dims); op = out.ctypes.data_as(C.POINTER(C.c_double))
t_obs, t_get = [], []
t0 = time.perf_counter()
for i in range(500):
    a = time.perf_counter(); lib.ekf_observe(hip.h, ip[i], zp[i], m); b = time.perf_counter(); lib.ekf_get_state(hip.h, op, hip.dims); c = time.perf_counter()
    t_obs.append(b - a); t_get.append(c - b)
tot = (time.perf_counter() - t0) / 500 * 1e6
print(f"C ABI loop : {tot:.1f} us/frame  ekf_observe {np.median(t_obs) * 1e6:.1f}  ekf_get_state({hip.dims}) {np.median(t_get) * 1e6:.1f}")
t0 = time.perf_counter()
cam = np.empty(10); cp = cam.ctypes.data_as(C.POINTER(C.c_double))
for i in range(500):
    lib.ekf_observe(hip.h, ip[i], zp[i], m); lib.ekf_get_camera(hip.h, cp)
print(f"C ABI loop, camera only: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us/frame")
