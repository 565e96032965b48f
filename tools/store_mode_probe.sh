#!/bin/bash
# GPU box: covariance-update store flavour (plain / nontemporal / write-through): kernel time, frame time serial and pipelined.
for mode in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DEKF_COV_STORE_MODE=$mode -c aruco_slam_amd/csrc/ekf_cov_update.hip -o /tmp/cov_$mode.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o aruco_slam_amd/lib/libekf_slam_hip.so aruco_slam_amd/build/ekf_api.o aruco_slam_amd/build/ekf_small_kernels.o aruco_slam_amd/build/ekf_front.o aruco_slam_amd/build/ekf_front_f64.o /tmp/cov_$mode.o
  echo "== store mode $mode"
  python tools/enqueue_cost.py 2>/dev/null | grep mode
  python bench.py --cpu-frames 0 --lookahead off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial bench', round(d['value']), d['kernel_us'])"
done
