#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_gemm.py -> profiles/rNN_pmc_gemm.json.
usage: python tools/pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [M N']"""
import csv
import hashlib
import json
import os
import sys


def per_launch(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if "gemm_f64_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


f, nf = per_launch(sys.argv[1], "FETCH_SIZE")
w, nw = per_launch(sys.argv[2], "WRITE_SIZE")
M = int(sys.argv[4]) if len(sys.argv) > 4 else 512
N = int(sys.argv[5]) if len(sys.argv) > 5 else 16384
read_b, write_b = f * 1024 * 2, w * 1024
alg = (M * N * 2 + M * M // 2) * 8 + 2 * M * N // M * 8 * 0   # K_mn read + A written (+ the L2-resident L^-1 once)
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mobocmf_amd", "csrc", "gemm_f64.hip")
out = {
    "kernel": "gemm_f64_kernel<false, true, 1>  A = L^-1 K_mn with the column-statistics epilogue, %d x %d x %d "
              "lower-triangular (as mobocmf_layer_forward launches it)" % (M, N, M),
    "shape": [M, N, M],
    "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/pmc_gemm.py  (two separate passes)",
    "raw": {"FETCH_SIZE": {"per_launch_KiB": f, "launches": nf}, "WRITE_SIZE": {"per_launch_KiB": w, "launches": nw}},
    "corrections": "FETCH_SIZE x 1024 B x 2 (MI355X_MICROARCH.md HBM section: gfx950 reports half the bytes of wide "
                   "coalesced reads; the tiles are read by 16 B/lane LDS-DMA); WRITE_SIZE x 1024 B (8 B/lane stores: "
                   "width uncalibrated)",
    "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_launch": alg,
    "note": "reads are %.2fx the K_mn bytes: the two workgroup pairs of a 128-column block walk k together (long row "
            "block first, same direction) so the slab one pulls into the XCD's L2 serves the other; the second parts "
            "re-read part of the rows.  The counter sits on the L2's fabric side and also counts Infinity-Cache hits, "
            "so it bounds HBM traffic from above." % (read_b / (M * N * 8)),
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
