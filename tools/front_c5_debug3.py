#!/usr/bin/env python3
"""Single filter, frames back to back through the sequence entry point: fused vs separate launches."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
s = SyntheticStream(n, m, seed=0)
boot = list(s.bootstrap())
frames = list(s.steady(60))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")


def run(fused, nfr, sync_every=0):
    f = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=fused)
    for ids, poses in boot:
        f.observe(ids, poses)
    f.backend.sync()
    if sync_every:
        for t in range(0, nfr, sync_every):
            f.backend.observe_sequence(idx[t:t + sync_every], z[t:t + sync_every], None)
            f.backend.sync()
    else:
        f.backend.observe_sequence(idx[:nfr], z[:nfr], None)
    try:
        f.backend.sync()
        return f.state
    except Exception as e:
        return str(e)[:80]


for nfr in (1, 2, 3, 5, 10, 30, 60):
    ref = run(False, nfr)
    for rep in range(2):
        a = run(True, nfr)
        ok = isinstance(a, np.ndarray) and isinstance(ref, np.ndarray) and np.array_equal(a, ref)
        info = a if isinstance(a, str) else ("max diff %.3e, entries %d" % (np.abs(a - ref).max(), int((a != ref).sum())))
        print("frames", nfr, "rep", rep, "fused == separate:", ok, info, flush=True)
b = run(True, 60, sync_every=1)
print("60 frames, sync after every frame:", isinstance(b, np.ndarray) and np.array_equal(b, run(False, 60)))
