#!/usr/bin/env python3
"""Covariance-update kernel alone: mean launch time (HIP events attached to the dispatch) and whole-frame time in serial
order, per covariance kernel.   usage: cov_bench.py "n m kernel[,kernel...] [frames]" ...
   e.g. tools/cov_bench.py "4096 64 mfma_tile,mfma_macro 40" "2048 32 mfma_tile,mfma_macro"
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from aruco_slam_amd.filters.extended_kalman_filter import EKF
    from aruco_slam_amd.synthetic import SyntheticStream
    init = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    for spec in sys.argv[1:]:
        parts = spec.split()
        n, m, kernels = int(parts[0]), int(parts[1]), parts[2].split(",")
        frames = int(parts[3]) if len(parts) > 3 else 60
        dims = 3 * n + 10
        for kern in kernels:
            s = SyntheticStream(n, m, seed=0)
            flt = EKF(init, max_landmarks=n, max_visible=m, cov_dtype="float32", cov_kernel=kern, lookahead=False)
            for ids, poses in s.bootstrap():
                flt.observe(ids, poses)
            fr = list(s.steady(2 * frames))
            idx = torch.tensor(np.stack([f[0] for f in fr]), dtype=torch.int32, device="cuda:0")
            z = torch.tensor(np.stack([f[1][:, :3] for f in fr]), dtype=torch.float64, device="cuda:0")
            hip = flt.backend
            hip.observe_sequence(idx[:frames], z[:frames])
            hip.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hip.observe_sequence(idx[:frames], z[:frames])
            torch.cuda.synchronize()
            per_frame = (time.perf_counter() - t0) / frames * 1e6
            hip.set_kernel_timing(2)
            hip.observe_sequence(idx[frames:], z[frames:])
            us, cnt = hip.kernel_timing()["cov_update"]
            hip.set_kernel_timing(0)
            hip.sync()
            import os
            if os.environ.get("COV_STATS"):      # (diagnostic builds of the macro-tile kernel: -DCM_STAMPS)
                hip.debug_enable_stamps(light=True)
                hip.debug_fetch("cov_stats", m)
                nf = 10
                hip.observe_sequence(idx[:nf], z[:nf])
                hip.sync()
                a = hip.debug_fetch("A", m)      # per-workgroup timeline of the last launch (diagnostic build)
                nwg = int(os.environ.get("COV_GRID", "4800"))
                rows = [a[b // 1500, 8 * (b % 1500): 8 * (b % 1500) + 8] for b in range(nwg)]
                tl = np.array([r for r in rows if r[1] > 0])
                if len(tl):
                    np.save(os.environ.get("COV_TL", "gpurun_out/cov_timeline.npy"), tl)
                    dur = (tl[:, 1] - tl[:, 0]) / 100
                    print(f"   timeline: {len(tl)} workgroups, mean {dur.mean():.2f} us each; cycles start->chunk0 {tl[:, 4].mean():.0f}  loop {tl[:, 5].mean():.0f}"
                          f"  epilogue {tl[:, 6].mean():.0f}")
            k = 3 * m
            tf = dims * dims * k / (us * 1e-6) / 1e12
            print(f"n={n} m={m} {kern:11s} cov_update {us:8.2f} us ({cnt} launches)  executed {tf:6.1f} TF = {tf / 157.3:.3f} of peak"
                  f"  serial frame {per_frame:8.2f} us", flush=True)
            del flt


if __name__ == "__main__":
    main()
