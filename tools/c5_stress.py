#!/usr/bin/env python3
"""Stress (diagnostics): n=4096, m=64, fused front kernel, chunks of CH frames back to back; compare the state after
every chunk with the separate-launch reference computed beforehand; report the first mismatching chunk."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[4]) if len(sys.argv) > 4 else 64
CH, NCH = int(sys.argv[5]) if len(sys.argv) > 5 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = SyntheticStream(n, m, seed=0)
boot = list(s.bootstrap())
frames = list(s.steady(CH * NCH))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")


def run(fused, ref=None):
    f = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=bool(fused), quat_update="scalar_first")
    for ids, poses in boot:
        f.observe(ids, poses)
    out = []
    for c in range(NCH):
        f.backend.observe_sequence(idx[c * CH:(c + 1) * CH], z[c * CH:(c + 1) * CH], None)
        try:
            f.backend.sync()
        except Exception as e:
            print("  chunk", c, "error:", str(e)[60:260])
            return out
        st = f.backend.get_state()
        if ref is not None and not np.array_equal(st, ref[c]):
            d = np.nonzero(st != ref[c])[0]
            print("  chunk", c, "MISMATCH:", len(d), "of", len(st), "state entries differ; first", d[:8], " max |d|", float(np.abs(st - ref[c]).max()))
            return out
        out.append(st)
    return out


ref = run(False)
print("reference chunks:", len(ref), flush=True)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    got = run(True, ref)
    print("rep", rep, "fused chunks equal to reference:", len(got), "/", NCH, flush=True)
