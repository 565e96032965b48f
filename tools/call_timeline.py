#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --steps K --warmup W` (pipelined mode): the K front kernels of the TIMED
call -- start of each relative to the first, duration, gap to the previous one -- and the span from the first front
kernel's start to the end of the last kernel of the call.    python tools/call_timeline.py <kernel_trace.csv> K W"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ekf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
K, W = int(sys.argv[2]), int(sys.argv[3])
fronts = [i for i, r in enumerate(rows) if "ekf_front_kernel" in r["Kernel_Name"]]
# front kernels in launch order: bootstrap (one per frame), warm-up call (W), timed call (K), then the instrumented repeats
boot = len(fronts) - (W + 3 * K) - 250         # (bench.py: 250 host-boundary frames at the end)
first = fronts[boot + W]
last_front = fronts[boot + W + K - 1]
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for n, i in enumerate(fronts[boot + W: boot + W + K]):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"F({n:2d}) start {(s - t0) / 1e3:8.2f}  dur {(e - s) / 1e3:6.2f}  gap {((s - prev_end) / 1e3) if prev_end else 0:5.2f}")
    prev_end = e
end = max(int(r["End_Timestamp"]) for r in rows[first: fronts[boot + W + K]])
print(f"first front start -> last kernel of the call ends: {(end - t0) / 1e3:.2f} us = {(end - t0) / 1e3 / K:.2f} us per step")
prev = rows[first - 1]
print(f"idle before the call: {(t0 - int(prev['End_Timestamp'])) / 1e3:.2f} us (previous kernel: {prev['Kernel_Name'][:40]})")
for r in rows[last_front: fronts[boot + W + K]]:
    print(f"   {r['Kernel_Name'][:48]:48s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.2f} end {(int(r['End_Timestamp']) - t0) / 1e3:8.2f}")
