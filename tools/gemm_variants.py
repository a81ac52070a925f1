#!/usr/bin/env python3
"""Warm-clock timing matrix of the triangular GEMM at the headline shape: lower / upper x store / colstats / dA x plain /
non-temporal stores, three interleaved rounds (the chip needs ~1 s of load before its clock settles: a fresh process
measures the first variants slow)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
F.set_tile_rows(int(os.environ.get("TILE_ROWS", "0")), int(os.environ.get("PAIR_MODE", "0")))      # rows: 0 automatic | 64 | 128; pairing: 0 auto | 1 never | 2 always
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
g = torch.Generator(device=dev)
g.manual_seed(1)
rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)
Lw, Up = torch.tril(rnd(M, M)), torch.triu(rnd(M, M))
B, A2 = rnd(M, N), rnd(M, N)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
avec, gmu, cgv, gv = rnd(M), rnd(N), rnd(N), rnd(N)
p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
p2 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
rdp = torch.empty(2 * (N // 128), M, dtype=torch.float64, device=dev)


def run(tri, epi, so):
    T = Lw if tri == 1 else Up
    if epi == 3:      # column statistics without the a^T C partials (the C = U^T A launch)
        F.gemm_f64_epilogue(T, B, C, tri, 1, stream_out=so, colsq_part=p1, avec=avec)
    elif epi == 0:
        F.gemm_f64_epilogue(T, B, C, tri, 0, stream_out=so)
    elif epi == 1:
        F.gemm_f64_epilogue(T, B, C, tri, 1, stream_out=so, colsq_part=p1, coldot_part=p2, avec=avec)
    else:
        F.gemm_f64_epilogue(T, B, C, tri, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp)


def timeit(fn, iters=20):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


for _ in range(400):      # ~0.3 s of load: let the clock settle
    run(1, 0, False)
torch.cuda.synchronize()
combos = [(t, e, s) for t in (1, 2) for e in (0, 1, 3, 2) for s in ((False, True) if e != 2 else (False,))]
res = {c: [] for c in combos}
for rnd_i in range(3):
    for c in combos:
        res[c].append(timeit(lambda: run(*c)))
fl = float(M) * M * N
for (t, e, s), v in res.items():
    best = min(v)
    print("%s %-8s %-3s: %s ms  -> best %.3f ms = %.1f TFLOP/s = %.3f of 78.6" %
          ("lower" if t == 1 else "upper", ("store", "colstats", "dA", "colsq")[e], "nt" if s else "", " ".join("%.3f" % x for x in v),
           best, fl / best / 1e9, fl / best / 1e9 / 78.6))
H = torch.empty(M, M, dtype=torch.float64, device=dev)
ts = [timeit(lambda: F.syrk_weighted(A2, gv, H)) for _ in range(3)]
print("weighted syrk (incl. slab reduction): %s ms  -> best %.3f ms = %.1f TFLOP/s = %.3f of 78.6" %
      (" ".join("%.3f" % x for x in ts), min(ts), fl / min(ts) / 1e9, fl / min(ts) / 1e9 / 78.6))
D = rnd(M, M)
for _ in range(3):
    t_d = timeit(lambda: F.gemm_f64(D, B, C))
    t_r = timeit(lambda: torch.matmul(D, B, out=C))
    print("dense: ours %.3f ms (%.1f TF/s)   rocBLAS %.3f ms (%.1f TF/s)" % (t_d, 2 * fl / t_d / 1e9, t_r, 2 * fl / t_r / 1e9))
