import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
n,m=1024,32
s=SyntheticStream(n,m,seed=0)
f=EKF(np.array([0,0,0,1,0,0,0,0,0,0]),max_landmarks=n,max_visible=m,cov_dtype="float32")
light = 'light' in sys.argv
if len(sys.argv) > 1 and sys.argv[1] == 'debug': f.backend.debug_enable_w()
else: f.backend.debug_enable_stamps(light)
for ids,p in s.bootstrap(): f.observe(ids,p)
if len(sys.argv) > 1 and sys.argv[1] == 'seq':      # the last frame of a pipelined sequence (beside the previous frame's covariance update)
    import torch
    frames = list(s.steady(40))
    idx = torch.tensor(np.stack([x[0] for x in frames]), dtype=torch.int32, device="cuda")
    z = torch.tensor(np.stack([x[1][:, :3] for x in frames]), dtype=torch.float64, device="cuda")
    f.backend.observe_sequence(idx, z, None); f.backend.sync()
else:
    for ids,p in s.steady(5): f.observe(ids,p)
st=f.backend.debug_fetch("stamps",m)
nb=6
t0=min(st[62],st[32],st[60])
us=lambda i:(st[i]-t0)/100.0
print("fused front kernel timeline, us since the earliest stamp (100 MHz wall clock)")
print("S-block wg0 start",us(60),"  measure wg published",us(55),"  S-block wg nS-1 stored",us(61))
print("factor: start",us(62)," end",us(63))
if not light: print("  cycles: blocks into registers",st[1]-st[0], " per wave:", [int(st[24+w]-st[0]) for w in range(8)])
if light: print('  X_b in LDS, seen by the publishing wave (cycles since the start of the factorisation):', [int(st[2+2*b]-st[0]) for b in range(nb)])
for b in range(0 if light else nb):
    prev = st[1] if b == 0 else st[3+2*(b-1)]
    print("  col",b," chain phase + barrier",st[2+2*b]-prev," panel + barrier + urgent update",st[3+2*b]-st[2+2*b],
          "  (chain alone %d; from the barrier: next owner's panel posted +%d, next chain starts +%d)" % (
              (st[48+2*b]-st[47+2*b]) if b < 4 else (st[49+2*b]-st[48+2*b]), st[16+b]-st[2+2*b],
              ((st[47+2*(b+1)] if b+1 < 4 else st[48+2*(b+1)]) - st[2+2*b]) if b+1 < nb else 0))
if light: print("chunk0: start",us(32)," A in LDS",us(33), "  (prologue detail: tools/chunk_stamps.py ... diag with a -DFR_CHUNK_DIAG build)")
else: print("chunk0: start",us(32)," indices in LDS",us(41)," operands staged",us(45)," MFMAs done",us(46)," tile done (wave 0)",us(42)," all tiles",us(43)," Jacobian in LDS",us(44)," picked",us(14)," prefetch issued",us(15)," A in LDS",us(33))
for q in range(nb): print("  step",q,"done",us(34+q))
print("  W/dx stored",us(34+nb))
