#!/bin/bash
# rocprofv3 kernel trace of one short bench call + the timeline of its timed call:  tools/trace_call.sh [K [W]]
set -o pipefail
K=${1:-20}; W=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/trace_call; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps $K --warmup $W --cpu-frames 0 --burn-in 0 > "$OUT/bench.json" 2> "$OUT/trace.err" || exit 1
python3 "$ROOT/tools/call_timeline.py" "$(ls $OUT/trace/*/*kernel_trace.csv | head -1)" $K $W > "$OUT/timeline.log" 2>&1
cat "$OUT/timeline.log"
