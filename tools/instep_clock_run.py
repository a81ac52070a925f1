#!/usr/bin/env python3
"""Workload for the in-step vs isolated GEMM comparison by PER-DISPATCH counters, one process so that every
phase shares the box, the clocks and the allocator state:

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d <dir> -- python3 tools/instep_clock_run.py

  phase I1  0.3 s of warm-up launches, then 24 consecutive launches of each top-layer variant (isolated, settled)
  phase S   8 eager ELBO steps of ONE C3 surrogate on one stream (the launch sequence the captured graph replays)
  phase I2  the isolated blocks again, right after the steps
  phase L   24 x [80 one-workgroup launches, then ONE A = L^-1 K launch]   (the chain's launch pattern in front of a GEMM)
  phase W   24 x [a 268 MB element-wise pass, then ONE A = L^-1 K launch]  (dirty lines / cache state in front of a GEMM)
tools/instep_clock.py turns the two CSVs into one table (duration, GRBM_GUI_ACTIVE -> effective clock, SQ_BUSY_CYCLES)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402
from mobocmf_amd.mlls import VariationalELBOMF  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.graphed_step import GraphedELBOStep  # noqa: E402

dev = torch.device("cuda")
M, N = 512, 65536
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
H = torch.empty(M, M, dtype=torch.float64, device=dev)
variants = [
    lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=avec),
    lambda: F.gemm_f64_epilogue(Up, B, C, 2, 1, stream_out=True, colsq_part=p1, avec=avec),
    lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp),
    lambda: F.gemm_f64_epilogue(Up, B, C, 2, 0),
    lambda: F.syrk_weighted(A2, gv, H),
]


def isolated():
    for fn in variants:
        for _ in range(24):
            fn()
    torch.cuda.synchronize()


import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(20):
        variants[0]()
    torch.cuda.synchronize()
isolated()                                                   # I1
cfg = synthetic.CONFIGS["C3"]
prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], output=0, seed=0)
model = synthetic.model_from_problem(prob, device=dev)
elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
step = GraphedELBOStep(model, elbo, t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], lr=1e-3, use_graph=False,
                       stream=torch.cuda.current_stream(dev))
for _ in range(8):                                           # S
    step.step()
torch.cuda.synchronize()
isolated()                                                   # I2
tiny = torch.zeros(64, dtype=torch.float64, device=dev)
for _ in range(24):                                          # L
    for _ in range(80):
        tiny.add_(1.0)
    variants[0]()
torch.cuda.synchronize()
for _ in range(24):                                          # W
    A2.mul_(1.0000001)
    variants[0]()
torch.cuda.synchronize()
print("done")
