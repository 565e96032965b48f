#!/usr/bin/env python3
"""Bootstrap at n=4096, m=64 through host-pointer observe(), frames back to back: where does the fused path first differ?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m = 4096, 64
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 8
boot = list(SyntheticStream(n, m, seed=0).bootstrap())
ref = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=False)
refs = []
for t, (ids, poses) in enumerate(boot):
    ref.observe(ids, poses)
    if (t + 1) % chunk == 0:
        refs.append(ref.state.copy())
for rep in range(4):
    f = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", fused=True)
    bad = None
    for t, (ids, poses) in enumerate(boot):
        f.observe(ids, poses)
        if (t + 1) % chunk == 0:
            try:
                st = f.state
            except Exception as e:
                bad = (t, "error " + str(e)[:60])
                break
            r = refs[(t + 1) // chunk - 1]
            if not np.array_equal(st, r):
                d = np.nonzero(st != r)[0]
                bad = (t, "dims %d, %d entries differ, first %s, max %.3e" % (len(st), len(d), d[:6], np.abs(st - r).max()))
                break
    print("rep", rep, "first bad chunk ends at frame:", bad, flush=True)
