#!/bin/bash
# A/B of prebuilt library variants, us per frame only (pipelined and serial order), alternating:  tools/ab_serial.sh <name> <name> ...
for v in "$@"; do
  cp aruco_slam_amd/lib/variants/$v.so aruco_slam_amd/lib/libekf_slam_hip.so || exit 1
  echo "== $v $(timeout -k 10 200 python tools/pipe_probe.py 1024 32 1500 2>/dev/null | grep us/frame)"
done
