import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from aruco_slam_amd.filters.extended_kalman_filter import EKF
from aruco_slam_amd.synthetic import SyntheticStream
n,m=1024,32
s=SyntheticStream(n,m,seed=0)
f=EKF(np.array([0,0,0,1,0,0,0,0,0,0]),max_landmarks=n,max_visible=m,cov_dtype="float32", fused=False)
f.backend.debug_enable_w()
for ids,p in s.bootstrap(): f.observe(ids,p)
for ids,p in s.steady(5): f.observe(ids,p)
st=f.backend.debug_fetch("stamps",m)
nb=6
d=np.diff(st[:4+2*nb])
print("solve kernel phase stamps (shader cycles)")
print("prologue: fetch S blocks", d[0], " put to LDS + barrier", d[1])
for b in range(nb): print("b",b,"(A) last term",d[2+2*b],"(B) pivot chain",d[3+2*b])
print("total", st[3+2*nb]-st[0])
