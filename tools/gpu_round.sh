#!/bin/bash
# One GPU-box session: the -m gpu tests, the default bench line, then optional extra commands.
# A step that is killed by its timeout ends the session (no further GPU step after a hang).
#   tools/gpu_round.sh <tag> [extra command ...]
set -o pipefail
tag=${1:-run}; shift
out=gpurun_out/$tag; mkdir -p $out
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $log 2>&1; local rc=$?
    echo "[$(date +%H:%M:%S)] rc=$rc  $*" | tee -a $out/steps.log
    if [ $rc -ge 124 ]; then echo "step timed out / was killed: stopping" | tee -a $out/steps.log; exit $rc; fi
    return 0
}
step 1000 $out/pytest.log python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider
tail -n 15 $out/pytest.log
step 300 $out/bench.json python bench.py
tail -c 3000 $out/bench.json
for cmd in "$@"; do
    name=$(echo "$cmd" | tr -c 'A-Za-z0-9' '_' | cut -c1-60)
    step 600 $out/extra_$name.log bash -c "$cmd"
    tail -n 12 $out/extra_$name.log
done
