#!/bin/bash
# A/B of prebuilt library variants (aruco_slam_amd/lib/variants/<name>.so) on the GPU box:  tools/quick_ab.sh <tag> <name> ...
set -o pipefail
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
if [ -f tools/probes/mfma_order_probe.hip ]; then
  hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_order_probe tools/probes/mfma_order_probe.hip > $out/probe_build.log 2>&1 && timeout -k 10 120 /tmp/mfma_order_probe > $out/mfma_order.log 2>&1; cat $out/mfma_order.log
fi
for v in "$@"; do
  cp aruco_slam_amd/lib/variants/$v.so aruco_slam_amd/lib/libekf_slam_hip.so || exit 1
  timeout -k 10 120 python tools/front_stamps.py seq light > $out/${v}_stamps_light.log 2>&1 || { tail -5 $out/${v}_stamps_light.log; exit 1; }
  timeout -k 10 200 python tools/pipe_probe.py 1024 32 2000 > $out/${v}_pipe.log 2>&1 || exit 1
  timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bitwise or fused or pipelined or intermediates or k192 or g2_teacher" > $out/${v}_pytest.log 2>&1
  echo "== $v"; tail -2 $out/${v}_pytest.log; grep -E "factor:|X_b in LDS|S-block|step 5|W/dx" $out/${v}_stamps_light.log; grep "us/frame" $out/${v}_pipe.log
done
