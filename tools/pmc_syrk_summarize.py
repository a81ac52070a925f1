#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) over tools/pmc_syrk.py M N -> profiles/rNN_pmc_syrk.json: HBM-side
traffic per launch of the weighted syrk's two kernels (k-sliced main loop into slabs, slab reduction).
usage: python tools/pmc_syrk_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> M N"""
import csv
import json
import sys


def per_launch(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


f = per_launch(sys.argv[1], "FETCH_SIZE")
w = per_launch(sys.argv[2], "WRITE_SIZE")
M, N = int(sys.argv[4]), int(sys.argv[5])
out = {"shape": "H = A diag(w) A^T, A %d x %d (mobocmf_syrk_weighted_f64 as the layer backward launches it)" % (M, N),
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/pmc_syrk.py %d %d  (two separate passes)" % (M, N),
       "corrections": "FETCH_SIZE x 1024 B x 2 (MI355X_MICROARCH.md HBM section: gfx950 reports half the bytes of wide coalesced "
                      "reads); WRITE_SIZE x 1024 B",
       "algorithmic_bytes": {"main loop": "A read once = %d, w read = %d, slabs written" % (M * N * 8, N * 8),
                             "reduction": "slabs read, H written = %d" % (M * M * 8)},
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    if "gemm_f64_kernel" not in k and "reduce_slabs" not in k:
        continue
    fr, nf = f.get(k, (0.0, 0))
    wr, nw = w.get(k, (0.0, 0))
    out["kernels"][k] = {"launches": [nf, nw], "read_bytes_per_launch": fr * 1024 * 2, "write_bytes_per_launch": wr * 1024,
                         "raw_KiB": {"FETCH_SIZE": fr, "WRITE_SIZE": wr}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
