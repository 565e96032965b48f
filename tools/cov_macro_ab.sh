#!/bin/bash
# GPU box: timing variants of the macro-tile covariance update (compile-time switches of ekf_cov_macro.hip / ekf_api.hip).
#   tools/cov_macro_ab.sh <tag> "<name> <macro-flags> [| <api-flags>]" ...      shape from $SHAPE (default "4096 64 mfma_macro 30")
set -o pipefail
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
SHAPE=${SHAPE:-"4096 64 mfma_macro 30"}
KB=${KB:-12}
B=aruco_slam_amd/build
cp aruco_slam_amd/lib/libekf_slam_hip.so /tmp/orig.so
for spec in "$@"; do
  name=${spec%% *}; rest=${spec#* }; mflags=${rest%%|*}; aflags=""
  [[ "$rest" == *"|"* ]] && aflags=${rest#*|}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCM_ONLY_KB=$KB $mflags -c aruco_slam_amd/csrc/ekf_cov_macro.hip -o /tmp/cm_$name.o 2> $out/build_$name.log || { cat $out/build_$name.log; exit 1; }
  api=$B/ekf_api.o
  if [ -n "$aflags" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $aflags -c aruco_slam_amd/csrc/ekf_api.hip -o /tmp/api_$name.o 2>> $out/build_$name.log || exit 1; api=/tmp/api_$name.o; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o aruco_slam_amd/lib/libekf_slam_hip.so $api $B/ekf_small_kernels.o $B/ekf_front.o $B/ekf_front_f64.o $B/ekf_cov_update.o /tmp/cm_$name.o $B/ekf_pose_ippe.o || exit 1
  echo "== $name ($mflags |$aflags)" | tee -a $out/ab.log
  timeout -k 10 120 python tools/cov_bench.py "$SHAPE" 2>$out/err_$name.log | tee -a $out/ab.log
done
cp /tmp/orig.so aruco_slam_amd/lib/libekf_slam_hip.so
