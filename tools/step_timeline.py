#!/usr/bin/env python3
"""One step of a single-stream run as a timeline: kernels in launch order between two Adam updates, with start offset,
duration and the idle gap in front of each (rocprofv3 --kernel-trace CSV).
usage: python tools/step_timeline.py <kernel_trace.csv> [step_index, negative = from the end]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
if k < 0:  # counted from the end of the run
    k += len(idx)
a, b = idx[k] + 1, idx[k + 1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0.0
groups = {}
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")[:60]
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r["Grid_Size_Y"]), 1) * max(int(r["Grid_Size_Z"]), 1)
    print("%9.1f us  +%6.1f gap  %7.1f us  %5d wg  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, wg, name))
    busy += (en - st) / 1e3
    key = name.split("<")[0]
    groups[key] = groups.get(key, 0.0) + (en - st) / 1e3
    prev_end = max(prev_end, en)
span = (prev_end - t0) / 1e3
print("step span %.1f us, kernel time %.1f us, idle %.1f us, %d launches" % (span, busy, span - busy, b - a))
for key, v in sorted(groups.items(), key=lambda kv: -kv[1])[:25]:
    print("   %-40s %8.1f us" % (key, v))
