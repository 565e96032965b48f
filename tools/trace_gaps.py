#!/usr/bin/env python3
"""Per-kernel durations and idle gaps between consecutive kernels from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ekf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -400:]
short = lambda n: n.split("<")[0].replace("void ", "")
dur = collections.defaultdict(list)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    print(f"{k:40s} n={len(v):4d} mean {sum(v)/len(v):7.2f} us")
# timeline of the last 12 kernels
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 16
t0 = int(rows[-NT]["Start_Timestamp"])
for r in rows[-NT:]:
    print(f"{short(r['Kernel_Name']):32s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.2f} end {(int(r['End_Timestamp'])-t0)/1e3:8.2f} queue {r.get('Queue_Id','?')}")
