#!/usr/bin/env python3
"""One-launch step (mobocmf_tiny_elbo_step) vs the layer path (HIP-graph replay) for ONE surrogate over the sizes the
one-launch step accepts: where it stops paying (util/tiny_step.py MAX_COLUMNS).  Usage: python tools/tiny_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.mlls import VariationalELBOMF  # noqa: E402
from mobocmf_amd.util import synthetic, tiny_step  # noqa: E402
from mobocmf_amd.util.graphed_step import GraphedELBOStep  # noqa: E402

tiny_step.MAX_COLUMNS = 1 << 20
dev = torch.device("cuda")
CASES = [(1, 2, 16, 16, 1), (1, 2, 16, 16, 4), (2, 2, 16, 64, 4), (2, 2, 16, 256, 2), (4, 2, 24, 24, 4), (4, 2, 32, 32, 1), (4, 2, 32, 64, 4), (4, 2, 32, 128, 4),
         (4, 2, 32, 256, 4), (4, 2, 32, 512, 4), (8, 3, 32, 256, 8), (8, 3, 32, 1024, 8)]


def timed(step, sync, n=300):
    for _ in range(20):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


for d, L, M, N, S in CASES:
    prob = synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, output=0, seed=0)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    perm = torch.roll(torch.arange(N, device=dev), 1)
    x, y, fid = t(prob["x"])[perm].contiguous(), t(prob["y"])[perm].contiguous(), t(prob["fid"])[perm].contiguous()
    ma = synthetic.model_from_problem(prob, num_samples_for_training=S, device=dev)
    mb = synthetic.model_from_problem(prob, num_samples_for_training=S, device=dev)
    g = GraphedELBOStep(ma, VariationalELBOMF(ma, N, L), x, y[:, None], fid[:, None], lr=1e-3)
    tl = timed(g.step, g.stream.synchronize)
    ts = tiny_step.TinyELBOStep([mb], [N], [x], [y], [fid], lr=1e-3, force=True)
    tt = timed(ts.step, ts.stream.synchronize)
    ts.check()
    cols = [int((fid >= l).sum()) * (S if l else 1) for l in range(L)]
    print("d=%d L=%d M=%2d N=%4d S=%d columns %-18s one launch %7.1f us (rule's estimate %6.0f) | layer path %7.1f us | x%.2f" %
          (d, L, M, N, S, cols, tt, tiny_step.estimated_us(M, cols), tl, tl / tt), flush=True)
