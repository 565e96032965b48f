#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run via gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench.py arguments, default: --steps 100 --warmup 10]
#   e.g. tools/profile_round.sh r02_c5 --landmarks 4096 --visible 64 --steps 50 --warmup 5
# 1) --kernel-trace --stats  2) --pmc FETCH_SIZE  3) --pmc WRITE_SIZE   (separate passes; PMC is
# never combined with trace domains other than the kernel trace).  Output: gpurun_out/<tag>/
set -o pipefail
TAG=${1:-prof}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-frames 0 --burn-in 0 ${*:---steps 100 --warmup 10}"
echo "$ARGS" > "$OUT/bench_args.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.json" 2> "$OUT/trace.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 1
ls -R "$OUT" | head -40
