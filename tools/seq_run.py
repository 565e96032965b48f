#!/usr/bin/env python3
"""One sequence call pattern and nothing else (for kernel traces):  seq_run.py n m frames lookahead(on|off) [repeats]
prints µs per frame of each repeat and the sequence mode that ran."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    from aruco_slam_amd.synthetic import SyntheticStream
    n, m, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    look = sys.argv[4] == "on"
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    init = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    s = SyntheticStream(n, m, seed=0)
    flt = EKF(init, max_landmarks=n, max_visible=m, cov_dtype="float32", lookahead=look)
    for ids, poses in s.bootstrap():
        flt.observe(ids, poses)
    fr = list(s.steady(frames))
    idx = torch.tensor(np.stack([f[0] for f in fr]), dtype=torch.int32, device="cuda:0")
    z = torch.tensor(np.stack([f[1][:, :3] for f in fr]), dtype=torch.float64, device="cuda:0")
    hip = flt.backend
    for r in range(reps):
        hip.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.observe_sequence(idx, z)
        hip.sync()
        torch.cuda.synchronize()
        print(f"n={n} m={m} repeat {r}: {(time.perf_counter() - t0) / frames * 1e6:8.2f} us per frame, mode {hip.last_sequence_mode()}", flush=True)


if __name__ == "__main__":
    main()
