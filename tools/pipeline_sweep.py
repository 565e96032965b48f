import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
for n, m, dtype, nfr in [(64, 8, "float32", 3000), (128, 8, "float32", 3000), (128, 16, "float32", 3000), (192, 16, "float32", 3000), (256, 16, "float32", 2000), (512, 16, "float32", 2000), (512, 32, "float32", 2000), (1024, 32, "float32", 2000), (1024, 64, "float32", 1000),
                         (2048, 32, "float32", 600), (2048, 64, "float32", 600), (4096, 64, "float32", 200), (4096, 32, "float32", 200), (1024, 32, "float64", 500), (256, 16, "float64", 2000), (128, 16, "float64", 3000), (64, 8, "float64", 3000)]:
    s = SyntheticStream(n, m, seed=5)
    boot = list(s.bootstrap()); frames = list(s.steady(nfr))
    idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
    z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
    res = []
    for pipelined in (True, False):
        flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, lookahead=pipelined, quat_update="scalar_first")
        for ids, poses in boot: flt.observe(ids, poses)
        flt.backend.observe_sequence(idx[:50], z[:50], None); flt.backend.sync()
        t0 = time.perf_counter(); flt.backend.observe_sequence(idx, z, None); flt.backend.sync()
        res.append((time.perf_counter() - t0) / nfr * 1e6); del flt
    print("LDS_KB=%s n=%d m=%d %s: pipelined %.1f serial %.1f us/frame (x%.2f)" % (os.environ.get("EKF_LA_LDS_KB", "148"), n, m, dtype, res[0], res[1], res[1] / res[0]), flush=True)
