#!/usr/bin/env python3
"""Round-3 experiments at the pruned panel shapes: (a) pair stagger of the triangular products, (b) workgroup budget of the
k-sliced weighted syrk (+ its slab reduction), (c) Gram forward / backward of a layer call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(1)
rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)


def timeit(fn, iters=40):
    for _ in range(8):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


warm = torch.tril(rnd(512, 512)), rnd(512, 65536), torch.empty(512, 65536, dtype=torch.float64, device=dev)
for _ in range(300):
    F.gemm_f64_epilogue(warm[0], warm[1], warm[2], 1, 0)
torch.cuda.synchronize()
del warm
shapes = [(512, 8192), (512, 16384), (1024, 8192), (1024, 16384), (512, 65536)]
print("# (a) triangular products: us per launch, colstats(lower) / dA(lower) / store(upper); stagger 0..3")
for M, N in shapes:
    Lw, Up = torch.tril(rnd(M, M)), torch.triu(rnd(M, M))
    B, A2 = rnd(M, N), rnd(M, N)
    C = torch.empty(M, N, dtype=torch.float64, device=dev)
    avec, gmu, cgv, gv = rnd(M), rnd(N), rnd(N), rnd(N)
    p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
    p2 = torch.empty_like(p1)
    rdp = torch.empty(2 * max(N // 128, N // 16), M, dtype=torch.float64, device=dev)
    row = []
    for st in range(4):
        F.set_tile_rows(0, 4 * st)
        t1 = timeit(lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=avec))
        t2 = timeit(lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp))
        t3 = timeit(lambda: F.gemm_f64_epilogue(Up, B, C, 2, 0))
        row.append("s%d %.1f/%.1f/%.1f" % (st, t1, t2, t3))
    F.set_tile_rows(0, 0)
    print("M=%4d N'=%6d | %s" % (M, N, " | ".join(row)), flush=True)
print("# (b) weighted syrk + slab reduction: us per call by workgroup budget")
for M, N in shapes:
    A = rnd(M, N)
    w = rnd(N)
    H = torch.empty(M, M, dtype=torch.float64, device=dev)
    row = []
    for wg in (128, 192, 256, 320, 384, 512, 768):
        F.set_syrk_workgroups(wg)
        row.append("%d: %.1f" % (wg, timeit(lambda: F.syrk_weighted(A, w, H))))
    F.set_syrk_workgroups(0)
    print("M=%4d N'=%6d | %s" % (M, N, " | ".join(row)), flush=True)
