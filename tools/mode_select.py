#!/usr/bin/env python3
"""Which order for ekf_observe_sequence_device: pipelined (front kernel of frame t+1 beside the covariance update of frame t)
or serial?  Both orders, 2000 frames per call after a warm-up call of the same length (the first pipelined call of a handle
also pays the one-time queue self-test), measured two ways in the same process:
  * device clock: start-to-start of the last 16 front kernels of the call (in-kernel stamps, s_memrealtime, no profiler);
  * host clock: wall time of the call / frames.
Output: one line per shape; the table is committed under profiles/ and the size rule in ekf_api.hip follows it."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
NFR = 2000
shapes = [(32, 3, "float32"), (64, 3, "float32"), (64, 8, "float32"), (64, 8, "float64"), (128, 16, "float32"), (128, 16, "float64"),
          (256, 16, "float64"), (256, 16, "float32"), (512, 16, "float32"), (512, 32, "float32"), (1024, 32, "float32")]
print("n m dtype | device start-to-start us/frame: pipelined serial | host wall us/frame: pipelined serial | mode that ran")
for n, m, dtype in shapes:
    s = SyntheticStream(n, m, seed=5)
    boot = list(s.bootstrap())
    frames = list(s.steady(NFR))
    idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
    z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
    dev, wall, ran = [], [], []
    for pipelined in (True, False):
        flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, lookahead=pipelined, quat_update="scalar_first")
        flt.backend.debug_enable_stamps(True)
        for ids, poses in boot:
            flt.observe(ids, poses)
        flt.backend.observe_sequence(idx, z, None)
        flt.backend.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        flt.backend.observe_sequence(idx, z, None)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / NFR * 1e6)
        ran.append(flt.backend.last_sequence_mode())
        st = flt.backend.debug_fetch("stamps", m)
        last = len(boot) + 2 * NFR                       # fused-frame number of the call's last frame
        starts = np.array([st[16 + ((last - 15 + i) & 15)] for i in range(16)]) / 100.0
        dev.append(float(np.diff(starts).mean()))
        del flt
    print(f"{n} {m} {dtype} | {dev[0]:.2f} {dev[1]:.2f} | {wall[0]:.2f} {wall[1]:.2f} | {ran[0]} / {ran[1]}", flush=True)
