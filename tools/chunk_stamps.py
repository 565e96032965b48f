#!/usr/bin/env python3
"""In-kernel timeline of column chunk 0 of the front kernel (role-level stamps):  chunk_stamps.py n m on|off [frames]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    from aruco_slam_amd.synthetic import SyntheticStream
    n, m, look = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "on"
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    s = SyntheticStream(n, m, seed=0)
    f = EKF(np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0]), max_landmarks=n, max_visible=m, cov_dtype="float32", lookahead=look)
    f.backend.debug_enable_stamps(True)
    for ids, p in s.bootstrap():
        f.observe(ids, p)
    fr = list(s.steady(frames))
    idx = torch.tensor(np.stack([x[0] for x in fr]), dtype=torch.int32, device="cuda")
    z = torch.tensor(np.stack([x[1][:, :3] for x in fr]), dtype=torch.float64, device="cuda")
    f.backend.observe_sequence(idx, z, None)
    f.backend.sync()
    st = f.backend.debug_fetch("stamps", m)
    nb = (3 * m + 15) // 16
    t0 = st[32]
    us = lambda i: (st[i] - t0) / 100.0
    print(f"n={n} m={m} mode {f.backend.last_sequence_mode()}: chunk 0 of the last frame, us since its start (100 MHz clock)")
    if "diag" in sys.argv:
        D = 41
        print("  indices / masks in LDS", us(D + 0))
        for r in range(2):
            for h in range(2):
                print(f"  round {r} stage {h}: operands staged {us(D + 1 + 4 * r + 2 * h):.2f}  MFMAs done {us(D + 2 + 4 * r + 2 * h):.2f}")
        print("  support rows complete", us(D + 10), " Jacobian in LDS", us(D + 11))
    if "svb" in sys.argv:
        t0 = st[24]
        print("  factorisation (us since the first chain started): column: chain starts / X in LDS / publisher's stores issued / complete")
        for b in range(nb):
            print(f"    {b:2d}: {us(24 + b):7.2f} {us(36 + b):7.2f} {us(12 + b):7.2f} {us(b):7.2f}")
        t0 = st[48]
        print("  one worker during one block column, us since its barrier: [A] done %.2f  [B] done %.2f  S blocks of column b+2 %.2f  pivot row staged %.2f  history terms %.2f  last term %.2f ; previous column's [C] ended %.2f" % tuple(us(48 + i) for i in (1, 2, 3, 4, 5, 6, 7)))
        print("    batched form: column b-1 ready %.2f  pivot share staged %.2f  first pivot block there %.2f" % (us(56), us(57), us(58)))
        t0 = st[32]
    print("  A chunk in LDS", us(33))
    print("  substitution steps done:", " ".join(f"{us(34 + q):.2f}" for q in range(nb)))
    print("  W / dx stored", us(34 + nb))


if __name__ == "__main__":
    main()
