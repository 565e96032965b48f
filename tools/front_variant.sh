#!/bin/bash
# GPU box or here: a variant of the library with extra flags for the f32 front kernel:  tools/front_variant.sh <out.so> <flags...>
set -o pipefail
out=$1; shift
B=aruco_slam_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c aruco_slam_amd/csrc/ekf_front.hip -o /tmp/front_variant.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $B/ekf_api.o $B/ekf_small_kernels.o /tmp/front_variant.o $B/ekf_front_f64.o $B/ekf_cov_update.o $B/ekf_cov_macro.o $B/ekf_pose_ippe.o
