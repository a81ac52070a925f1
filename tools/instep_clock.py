#!/usr/bin/env python3
"""Per-dispatch durations and counters of the top-layer GEMM launches by context (tools/instep_clock_run.py under
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace):
usage: python tools/instep_clock.py <counter_collection.csv> <kernel_trace.csv>
Context of a dispatch: `isolated` = inside a run of >= 8 consecutive dispatches of the same kernel; `after 80 tiny` / `after
an element-wise pass` = phases L / W of the workload; `in step` = everything else (preceded by another kernel of the step).
Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import csv
import re
import statistics as st
import sys

cnt = {}
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = int(r["Dispatch_Id"])
    cnt.setdefault(d, {})[r["Counter_Name"]] = cnt.get(d, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[d] = r["Kernel_Name"]
tr = {}
for r in csv.DictReader(open(sys.argv[2])):
    tr[int(r["Dispatch_Id"])] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                                 int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r["Grid_Size_Y"]), 1) * max(int(r["Grid_Size_Z"]), 1))
ids = sorted(tr)
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
names = [short(tr[i][2]) for i in ids]
# run lengths of identical consecutive kernels
run_len = [1] * len(ids)
i = 0
while i < len(ids):
    j = i
    while j + 1 < len(ids) and names[j + 1] == names[i] and tr[ids[j + 1]][3] == tr[ids[i]][3]:
        j += 1
    for k in range(i, j + 1):
        run_len[k] = j - i + 1
    i = j + 1
rows = {}
for k, d in enumerate(ids):
    nm = names[k]
    if "gemm_f64_kernel" not in nm or tr[d][3] < 400:      # top-layer launches only (1024 / 496 workgroups)
        continue
    if run_len[k] >= 8:
        ctx = "isolated (run of the same launch)"
    else:
        prev = names[max(0, k - 40):k]
        tiny = sum(1 for p in prev[-30:] if "elementwise" in p or "vectorized" in p)
        if tiny >= 25:
            ctx = "after 80 one-workgroup launches"
        elif prev and ("elementwise" in prev[-1] or "vectorized" in prev[-1]) and run_len[k - 1] == 1 and tr[ids[k - 1]][3] > 1000:
            ctx = "after a 268 MB element-wise pass"
        else:
            ctx = "in step"
    s, e = tr[d][0], tr[d][1]
    c = cnt.get(d, {})
    rows.setdefault((nm, ctx), []).append(((e - s) / 1e3, c.get("GRBM_GUI_ACTIVE", float("nan")), c.get("SQ_BUSY_CYCLES", float("nan")),
                                           c.get("SQ_WAVES", float("nan"))))
print("| kernel | context | launches | median us | GRBM_GUI_ACTIVE / 8 | effective clock GHz | SQ_BUSY_CYCLES | SQ_WAVES |")
print("|---|---|---|---|---|---|---|---|")
for (nm, ctx), v in sorted(rows.items()):
    med = lambda i: st.median(x[i] for x in v)
    dur, gui = med(0), med(1)
    clk = st.median(x[1] / 8.0 / (x[0] * 1e3) for x in v)
    print("| %s | %s | %d | %.1f | %.0f | %.3f | %.3e | %.0f |" % (nm, ctx, len(v), dur, gui / 8.0, clk, med(2), med(3)))
