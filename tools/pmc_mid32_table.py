#!/usr/bin/env python3
"""usage: python tools/pmc_mid32_table.py <dir with pass*/...counter_collection.csv> -> per-launch mean of every counter for the
gemm_mid32 launches, by grid size."""
import csv
import glob
import sys

acc = {}
for path in glob.glob(sys.argv[1] + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "gemm_mid32" not in r["Kernel_Name"]:
            continue
        key = "%s workgroups" % (int(r["Grid_Size"]) // 256)
        acc.setdefault(key, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
names = sorted(acc, key=lambda s: int(s.split()[0]))
ctrs = sorted({c for n in names for c in acc[n]})
print("| counter (mean per launch, summed over the chip) | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for c in ctrs:
    row = []
    for n in names:
        v = list(acc[n].get(c, {}).values())
        row.append("%.4g" % (sum(v) / len(v)) if v else "-")
    print("| %s | %s |" % (c, " | ".join(row)))
