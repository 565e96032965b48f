#!/bin/bash
# GPU box: the pipelined sequence mode -- bitwise tests, bench with / without it, kernel timeline.
set -o pipefail
out=gpurun_out/${1:-pipe}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sequence_entry" -p no:cacheprovider > $out/pytest.log 2>&1; rc=$?
tail -n 5 $out/pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out: stopping"; exit $rc; fi
timeout -k 10 200 python bench.py --cpu-frames 0 --lookahead on > $out/bench_la.json 2> $out/bench_la.err; rc=$?
if [ $rc -ge 124 ]; then echo "bench timed out: stopping"; exit $rc; fi
timeout -k 10 200 python bench.py --cpu-frames 0 > $out/bench_serial.json 2> $out/bench_serial.err
python - <<PY
import json
for f in ("bench_la", "bench_serial"):
    try:
        d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["ms_per_step"], d["kernel_us"], d["outputs_finite"])
    except Exception as e:
        print(f, "no result", e); print(open("$out/%s.err" % f).read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$out/trace -- python3 $ROOT/bench.py --cpu-frames 0 --steps 100 --warmup 10 --lookahead on > $ROOT/$out/trace.json 2> $ROOT/$out/trace.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$ROOT/$out/trace/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "ekf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cov_rows" in r["Kernel_Name"]]
dur = collections.defaultdict(list)
for r in rows: dur[r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items(): print("%-32s n=%4d mean %7.2f min %7.2f" % (k, len(v), sum(v) / len(v), min(v)))
if len(idx) > 60:
    i0 = idx[50]; t0 = int(rows[i0 - 4]["Start_Timestamp"])
    for r in rows[i0 - 4:i0 + 14]:
        print("%-30s %8.2f %8.2f  q%s" % (r["Kernel_Name"][:30], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id")))
PY
