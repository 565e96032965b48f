#!/bin/bash
# GPU box: the same sequence runs with two or more prebuilt libraries (aruco_slam_amd/lib/variants/<name>.so; "new" = the
# library in place):  tools/lib_ab.sh <tag> <name> ...     shapes from $SHAPES ("n m frames on|off;...")
set -o pipefail
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
SHAPES=${SHAPES:-"4096 64 30 off;2048 64 60 on;1024 64 100 on;1024 32 200 on"}
cp aruco_slam_amd/lib/libekf_slam_hip.so /tmp/new.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = new ]; then cp /tmp/new.so aruco_slam_amd/lib/libekf_slam_hip.so; else cp aruco_slam_amd/lib/variants/$v.so aruco_slam_amd/lib/libekf_slam_hip.so || exit 1; fi
  echo "== $v (pass $rep)" | tee -a $out/ab.log
  while read -r n m fr look; do
    [ -z "$n" ] && continue
    timeout -k 10 200 python3 tools/seq_run.py $n $m $fr $look 4 2>/dev/null | tail -1 | tee -a $out/ab.log || exit 1
  done <<< "$(echo "$SHAPES" | tr ';' '\n')"
  if [ -n "$ROT" ]; then timeout -k 10 200 python3 bench.py --filter ekf_rotations --cpu-frames 0 --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotations', d['value'], d['config'].get('sequence_mode_ran'))" | tee -a $out/ab.log; fi
done
done
cp /tmp/new.so aruco_slam_amd/lib/libekf_slam_hip.so
