import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
n, m, nfr = 1024, 32, 1000
s = SyntheticStream(n, m, seed=5)
boot = list(s.bootstrap()); frames = list(s.steady(nfr))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
for mode in (None, False):
    flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype="float32", lookahead=mode)
    for ids, poses in boot: flt.observe(ids, poses)
    flt.backend.observe_sequence(idx[:50], z[:50], None); flt.backend.sync()
    t0 = time.perf_counter(); flt.backend.observe_sequence(idx, z, None); t1 = time.perf_counter(); flt.backend.sync(); t2 = time.perf_counter()
    print("mode", mode, "host enqueue %.1f us/frame, total %.1f us/frame" % ((t1 - t0) / nfr * 1e6, (t2 - t0) / nfr * 1e6))
