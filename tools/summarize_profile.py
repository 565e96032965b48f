#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs collected by tools/profile_round.sh into the small
tracked summaries under profiles/<tag>/ and refresh profiles/cov_update_pmc_traffic.json
(which bench.py reads for roofline.traffic).

    python tools/summarize_profile.py gpurun_out/r01_v6 r01_v6 [--key n1024_m32_float32]

HBM bytes per launch of the covariance-update kernel, per MI355X_MICROARCH.md (HBM /
rocprofv3 section): separate --pmc passes; FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so reads are doubled:
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
The doubling is cross-checked on this kernel's own access pattern: WRITE_SIZE matches
the bytes the kernel stores exactly, and 2*FETCH_SIZE matches the lower-triangle P
bytes plus one W-panel fetch per XCD L2.
"""
import argparse
import collections
import csv
import glob
import json
import shutil
from pathlib import Path

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("tag")
ap.add_argument("--key", default="n1024_m32_float32")
ap.add_argument("--steady", type=int, default=100, help="launches to average: the instrumented serial-order repeat of bench.py")
ap.add_argument("--skip-last", type=int, default=450,
                help="launches at the very end that are left out (bench.py ends with 450 per-frame observes through the host boundary, "
                     "whose front kernel also reads the detections from / mirrors the state to host memory)")
ap.add_argument("--parity", default=None, help="parity_metrics.jsonl of the same session to keep next to the profile")
args = ap.parse_args()
src, repo = Path(args.src), Path(__file__).resolve().parent.parent
dst = repo / "profiles" / args.tag
dst.mkdir(parents=True, exist_ok=True)

stats = glob.glob(str(src / "trace" / "*" / "*kernel_stats.csv"))[0]
shutil.copy(stats, dst / "kernel_stats.csv")
trace = glob.glob(str(src / "trace" / "*" / "*kernel_trace.csv"))[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    if "ekf_" in r["Kernel_Name"]:
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
def window(v):
    w = v[-(args.steady + args.skip_last):-args.skip_last] if args.skip_last and len(v) > args.steady + args.skip_last else v[-args.steady:]
    return w or v
steady = {k: sum(window(v)) / len(window(v)) / 1e3 for k, v in dur.items()}

# the covariance-update kernel of the steady state: the one of the window's launches (bootstrap frames of a large
# configuration run smaller instantiations / the other kernel)
cov_name = max((k for k in steady if "cov_update" in k), key=lambda k: len(dur[k]) * steady[k])
pmc = {}
for kind, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(str(src / kind / "*" / "*counter_collection.csv"))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Kernel_Name"] == cov_name and r["Counter_Name"] == name]
    pmc[name] = sum(window(vals)) / len(window(vals))
traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
bench_args = (src / "bench_args.txt").read_text().strip() if (src / "bench_args.txt").exists() else "--cpu-frames 0 --steps 100 --warmup 10"
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py " + bench_args,
    "steady_state_mean_us": {k: round(v, 3) for k, v in steady.items()},
    "cov_update": {"kernel": cov_name, "mean_us": round(steady[cov_name], 3),
                   "FETCH_SIZE_KiB": pmc["FETCH_SIZE"], "WRITE_SIZE_KiB": pmc["WRITE_SIZE"],
                   "hbm_bytes_per_launch": traffic,
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B reads as 64 B)"},
}
bench_line = (src / "trace.json").read_text().strip().splitlines()[-1]
summary["bench_line_under_rocprof"] = json.loads(bench_line)
(dst / "summary.json").write_text(json.dumps(summary, indent=1))
if args.parity and Path(args.parity).exists():
    shutil.copy(args.parity, dst / "parity_metrics.jsonl")
tfile = repo / "profiles" / "cov_update_pmc_traffic.json"
table = json.loads(tfile.read_text()) if tfile.exists() else {}
table[args.key] = traffic
table[args.key + "_source"] = f"profiles/{args.tag}/summary.json"
table[args.key + "_rocprof_mean_us"] = round(steady[cov_name], 3)
tfile.write_text(json.dumps(table, indent=1))
print(json.dumps(summary["steady_state_mean_us"], indent=1))
print("traffic MB", traffic / 1e6)
