"""Diagnostics: us per frame of the pipelined sequence mode at one shape, for the environment it is started in
(EKF_LA_LDS_KB is read once per process):  python tools/pipe_probe.py [n m frames [dtype]]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
n, m, nfr = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 32, 2000)
dtype = sys.argv[4] if len(sys.argv) > 4 else "float32"
opts = sys.argv[5:]            # "traj": write the trajectory rows; "aswritten": the reference's quaternion rule; "seed0"; "once": no best-of-3
INIT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
s = SyntheticStream(n, m, seed=0 if "seed0" in opts else 5)
boot = list(s.bootstrap()); frames = list(s.steady(nfr))
idx = torch.tensor(np.stack([f[0] for f in frames]), dtype=torch.int32, device="cuda")
z = torch.tensor(np.stack([f[1][:, :3] for f in frames]), dtype=torch.float64, device="cuda")
res = []
for pipelined in (True, False):
    flt = EKF(INIT, max_landmarks=n, max_visible=m, cov_dtype=dtype, lookahead=pipelined, quat_update="as_written" if "aswritten" in opts else "scalar_first")
    for ids, poses in boot: flt.observe(ids, poses)
    flt.backend.observe_sequence(idx[:50], z[:50], None); flt.backend.sync()
    best = 1e9
    traj = torch.zeros((nfr, 7), dtype=torch.float64, device="cuda") if "traj" in opts else None
    for rep in range(1 if "once" in opts else 3):
        t0 = time.perf_counter(); flt.backend.observe_sequence(idx, z, traj); flt.backend.sync()
        best = min(best, (time.perf_counter() - t0) / nfr * 1e6)
    res.append(best); del flt
env = {k: v for k, v in os.environ.items() if k.startswith("EKF_")}
print("n=%d m=%d %s %s %s: pipelined %.2f serial %.2f us/frame" % (n, m, dtype, env, opts, res[0], res[1]), flush=True)
