#!/usr/bin/env python3
"""usage: python tools/pmc_table.py <dir with pass*/.../counter_collection.csv>  -> per-kernel mean of every counter."""
import csv
import glob
import re
import sys

acc = {}
for path in glob.glob(sys.argv[1] + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "gemm_f64_kernel" not in name:
            continue
        acc.setdefault(name, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
names = sorted(acc)
ctrs = sorted({c for n in names for c in acc[n]})
print("| counter | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for c in ctrs:
    row = []
    for n in names:
        v = list(acc[n].get(c, {}).values())
        row.append("%.4g" % (sum(v) / len(v)) if v else "-")
    print("| %s | %s |" % (c, " | ".join(row)))
