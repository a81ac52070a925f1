#!/usr/bin/env python3
"""Why the top-layer GEMM takes ~0.335 ms inside a step and ~0.307 ms launched alone in a loop: the same launch (A = L^-1 K
with column statistics, 512 x 65536 x 512) timed (a) on ONE pair of operand / result panels, as bench.py's roofline leg and
tools/gemm_variants.py do, (b) cycling through six pairs (3.2 GB: nothing of a panel survives in the 256 MB Infinity Cache
until it is read again), (c) as (b) with a Gram-sized element-wise pass over another panel between the launches, (d) as (a) with
~0.4 ms of one-workgroup kernels (the M x M chain's launch pattern) in front of every timed launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
M, N = 512, 65536
g = torch.Generator(device=dev)
g.manual_seed(1)
rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)
Lw = torch.tril(rnd(M, M))
Bs = [rnd(M, N) for _ in range(6)]
Cs = [torch.empty(M, N, dtype=torch.float64, device=dev) for _ in range(6)]
avec = rnd(M)
p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
p2 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
run = lambda i: F.gemm_f64_epilogue(Lw, Bs[i], Cs[i], 1, 1, colsq_part=p1, coldot_part=p2, avec=avec)


def timeit(fn, iters=24):
    st = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    en = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        fn(i, st[i], en[i])
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(st, en))
    return ts[len(ts) // 2]


def one(i, s, e):
    s.record(); run(0); e.record()


def cyc(i, s, e):
    s.record(); run(i % 6); e.record()


def cyc_ew(i, s, e):
    Bs[(i + 3) % 6].mul_(1.0000001)      # a 268 MB read + write on another panel in between (not timed)
    s.record(); run(i % 6); e.record()


tiny = torch.zeros(64, dtype=torch.float64, device=dev)


def light(i, s, e):
    for _ in range(80):      # ~80 x 5 us of latency-bound launches on a nearly idle chip
        tiny.add_(1.0)
    s.record(); run(0); e.record()


for _ in range(400):
    run(0)
torch.cuda.synchronize()
for rnd_i in range(3):
    print("same panels %.3f ms | six panel pairs in turn %.3f ms | + an element-wise pass in between %.3f ms | after 80 one-workgroup launches %.3f ms" %
          (timeit(one), timeit(cyc), timeit(cyc_ew), timeit(light)))
