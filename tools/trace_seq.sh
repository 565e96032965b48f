#!/bin/bash
# rocprofv3 kernel trace of tools/seq_run.py:  tools/trace_seq.sh <tag> n m frames on|off
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$ROOT/tools/seq_run.py" "$@" > "$OUT/plain.log" 2>&1 || { cat "$OUT/plain.log"; exit 1; }
cat "$OUT/plain.log"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/seq_run.py" "$@" > "$OUT/run.log" 2> "$OUT/trace.err" || { tail -20 "$OUT/trace.err"; exit 1; }
cat "$OUT/run.log"
python3 "$ROOT/tools/trace_gaps.py" "$(ls $OUT/trace/*/*kernel_trace.csv | head -1)" 120 40 > "$OUT/gaps.log" 2>&1
cat "$OUT/gaps.log"
