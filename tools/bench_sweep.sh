#!/bin/bash
# bench.py at several "steps warmup burn-in" triples:  tools/bench_sweep.sh "20 5 0" "20 5 3000" ...
for a in "$@"; do
  set -- $a
  python bench.py --steps $1 --warmup $2 --burn-in $3 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps', d['steps'], 'warmup', d['warmup'], 'burn-in', $3, 'updates/s', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2))"
done
