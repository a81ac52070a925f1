#!/usr/bin/env python3
"""Tile height x row-block pairing sweep of the triangular panel product (A = L^-1 K with column statistics, and the dA
launch) over the panel shapes of the configs: picks the policy of api.hip panel_tile_rows / gemm_f64.hip launch_gemm."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(1)
rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


shapes = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or \
    [(512, 8192), (512, 16384), (512, 32768), (512, 65536), (1024, 8192), (1024, 65536), (256, 8192), (256, 65536), (640, 8192), (768, 16384)]
warm = torch.tril(rnd(512, 512)), rnd(512, 65536), torch.empty(512, 65536, dtype=torch.float64, device=dev)
for _ in range(300):
    F.gemm_f64_epilogue(warm[0], warm[1], warm[2], 1, 0)
torch.cuda.synchronize()
for M, N in shapes:
    Lw = torch.tril(rnd(M, M))
    B, A2 = rnd(M, N), rnd(M, N)
    C = torch.empty(M, N, dtype=torch.float64, device=dev)
    avec, gmu, cgv, gv = rnd(M), rnd(N), rnd(N), rnd(N)
    p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
    p2 = torch.empty_like(p1)
    rdp = torch.empty(2 * max(N // 128, N // 16), M, dtype=torch.float64, device=dev)
    row = []
    for rows in (128, 64):
        for pm in (1, 2):
            F.set_tile_rows(rows, pm)
            t1 = timeit(lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=avec))
            t2 = timeit(lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp))
            row.append("%d%s %.3f/%.3f" % (rows, "p" if pm == 2 else "u", t1, t2))
    F.set_tile_rows(0, 0)
    fl = float(M) * M * N
    best = min(float(r.split()[1].split("/")[0]) for r in row)
    print("M=%4d N'=%6d | %s | best colstats %.3f ms = %.3f of 78.6" % (M, N, " | ".join(row), best, fl / best / 1e9 / 78.6), flush=True)
