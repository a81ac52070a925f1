#!/usr/bin/env python3
"""HBM-side traffic of ONE ELBO step of one C3 surrogate, kernel by kernel: sums rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
(separate passes over `bench.py --surrogates 1 --eager`) over one step (from one ELBO tail to the next).
usage: python tools/pmc_step.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import csv
import re
import sys


def step_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "elbo_combine_fwd" in r["Kernel_Name"]]      # once per step
    return rows[idx[-2]:idx[-1]]


tot = {}
for path, counter, scale in ((sys.argv[1], "FETCH_SIZE", 2048.0), (sys.argv[2], "WRITE_SIZE", 1024.0)):
    for r in step_rows(path, counter):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48]
        t = tot.setdefault(name, [0.0, 0.0, 0])
        t[0 if counter == "FETCH_SIZE" else 1] += float(r["Counter_Value"]) * scale
        if counter == "FETCH_SIZE":
            t[2] += 1
rd = sum(v[0] for v in tot.values())
wr = sum(v[1] for v in tot.values())
print("one ELBO step of one C3 surrogate (d=8, M=512, N=8192, S=8): %.2f GB read + %.2f GB written = %.2f GB" % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
print("(FETCH_SIZE x 2 KiB -- the gfx950 correction for wide reads, an upper bound for narrow ones -- and WRITE_SIZE x 1 KiB; the counters sit")
print(" on the L2's fabric side and include Infinity-Cache hits)")
print("%-50s %8s %10s %10s" % ("kernel", "launches", "read MB", "written MB"))
for k, v in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:14]:
    print("%-50s %8d %10.1f %10.1f" % (k, v[2], v[0] / 1e6, v[1] / 1e6))
