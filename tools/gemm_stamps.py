#!/usr/bin/env python3
"""Where a workgroup of the triangular GEMM spends its time (diagnostic build -DGEMM_STAMPS, 100 MHz wall-clock stamps):
    bash tools/build_variant.sh stamps -DGEMM_STAMPS          (build container)
    MOBOCMF_HIP_LIB=$PWD/abtest/libstamps.so python tools/gemm_stamps.py      (GPU box)
Prints per-phase durations (median over workgroups), how the phases of all workgroups line up on the wall clock (the
number of workgroups inside a main loop in each 5 us bin), and which workgroups share a CU."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import _lib  # noqa: E402
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
avec = torch.randn(M, dtype=torch.float64, device=dev)
p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
p2 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
nwg = 4096
stamps = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
A2 = torch.randn(M, N, dtype=torch.float64, device=dev)
gmu, cgv, gv = (torch.randn(N, dtype=torch.float64, device=dev) for _ in range(3))
rdp = torch.empty(2 * (N // 128), M, dtype=torch.float64, device=dev)
Hs = torch.empty(M, M, dtype=torch.float64, device=dev)
for name, tri, epi in (("lower colstats", 1, 1), ("lower store", 1, 0), ("lower dA", 1, 2), ("weighted syrk", 0, 9), ("dense store", 0, 0)):
    if epi == 9:
        fn = lambda: F.syrk_weighted(A2, gv, Hs)
    elif epi == 2:
        fn = lambda: F.gemm_f64_epilogue(A, B, C, tri, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp)
    else:
        fn = lambda: F.gemm_f64_epilogue(A, B, C, tri, epi, colsq_part=p1, coldot_part=p2, avec=avec)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    raw.mobocmf_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    stamps.zero_()
    fn()
    torch.cuda.synchronize()
    raw.mobocmf_debug_set_stamps(ctypes.c_void_p(0))
    s = stamps.cpu().numpy().reshape(nwg, 32)
    used = s[:, 0] > 0
    s = s[used]
    t0 = s[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    print("== %s: %d workgroups, kernel span %.1f us" % (name, len(s), us(s[:, [9, 10]].max())))
    start = us(s[:, 0])
    print("   start: first %.1f, median %.1f, p90 %.1f, last %.1f us" %
          (start.min(), np.median(start), np.percentile(start, 90), start.max()))
    for part in (0, 1):
        sel = s[:, 1 + 4 * part] > 0
        if not sel.any():
            continue
        prev = s[sel, 0] if part == 0 else s[sel, 9]
        pro = (s[sel, 1 + 4 * part] - prev) / 100.0
        loop = (s[sel, 4 + 4 * part] - s[sel, 1 + 4 * part]) / 100.0
        epi_t = (s[sel, 9 + part] - s[sel, 4 + 4 * part]) / 100.0
        print("   part %d: prologue %.2f | main loop %.2f (min %.2f max %.2f) | epilogue issue %.2f  [us, median]" %
              (part, np.median(pro), np.median(loop), loop.min(), loop.max(), np.median(epi_t)))
        if part == 0 and (s[sel, 13] > 0).any():
            base = s[sel, 4]
            f = lambda i: np.median((s[sel, i] - base) / 100.0) if (s[sel, i] > 0).any() else float("nan")
            print("           epilogue of part 0 from the end of the main loop: sums %.2f | partials %.2f | C stores issued %.2f us" %
                  (f(11), f(12), f(13)))
        if (s[sel, 2 + 4 * part] > 0).any():
            d0 = (s[sel, 2 + 4 * part] - s[sel, 1 + 4 * part]) / 100.0
            d1 = (s[sel, 3 + 4 * part] - s[sel, 2 + 4 * part]) / 100.0
            d2 = (s[sel, 4 + 4 * part] - s[sel, 3 + 4 * part]) / 100.0
            print("           diag-first %.2f | dense %.2f | diag-last %.2f" % (np.median(d0), np.median(d1), np.median(d2)))
    for lbl, b in (("K step nk-2 of part 0 (a light diagonal-block step for a triangular A)", 16), ("K step 4 of part 0 (dense)", 21)):
        sel = s[:, b] > 0
        if sel.any():
            dd = lambda i, j: np.median((s[sel, j] - s[sel, i]) / 100.0)
            print("   %s: fragment reads + DMA issue %.2f | MFMA phase %.2f | wait for DMA / LDS %.2f | barrier %.2f  = %.2f us [median, wavefront 0]" %
                  (lbl, dd(b, b + 1), dd(b + 1, b + 2), dd(b + 2, b + 3), dd(b + 3, b + 4), dd(b, b + 4)))
    end = us(np.maximum(s[:, 9], s[:, 10]))
    print("   end: first %.1f, median %.1f, last %.1f us" % (end.min(), np.median(end), end.max()))
    span = end.max()
    bins = np.arange(0, span + 5, 5.0)
    inloop = np.zeros(len(bins))
    for part in (0, 1):
        sel = s[:, 1 + 4 * part] > 0
        a, b = us(s[sel, 1 + 4 * part]), us(s[sel, 4 + 4 * part])
        for i, tb in enumerate(bins):
            inloop[i] += ((a <= tb) & (b > tb)).sum()
    print("   workgroups inside a main loop at t = 0, 5, 10 ... us:", " ".join("%d" % v for v in inloop))
    # who is slow?  total busy time (first stamp -> last stamp) per workgroup, grouped by XCD and by dispatch round
    hw = s[:, 15]
    xcc = (hw >> 32) & 0xf
    dur = (np.maximum(s[:, 9], s[:, 10]) - s[:, 0]) / 100.0
    rnd = (start > np.median(start)).astype(int)
    print("   workgroup duration by XCD (median us, round 1 | round 2):",
          " ".join("%d:%.0f|%.0f" % (x, np.median(dur[(xcc == x) & (rnd == 0)]) if ((xcc == x) & (rnd == 0)).any() else 0,
                                     np.median(dur[(xcc == x) & (rnd == 1)]) if ((xcc == x) & (rnd == 1)).any() else 0)
                   for x in range(8)))
    print("   duration percentiles (us): p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f; XCD of block 0..7: %s" %
          (tuple(np.percentile(dur, [5, 25, 50, 75, 95])) + (xcc[:8].tolist(),)))
    lo = hw & 0xffffffff
    cu_key = (hw >> 32) * 100000 + ((lo >> 8) & 0xf) * 64 + ((lo >> 12) & 0x1) * 16 + ((lo >> 13) & 0x7)   # xcc, cu, sh, se
    order = np.nonzero(used)[0]
    groups = {}
    for b, k, st in zip(order, cu_key, start):
        groups.setdefault(int(k), []).append((int(b), round(float(st), 1)))
    ex = list(groups.items())[:4]
    print("   distinct (xcc, cu, sh, se) keys: %d; examples (block id, start us) per key: %s" % (len(groups), [v[:6] for _, v in ex]))
