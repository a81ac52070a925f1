#!/usr/bin/env python3
"""ELBO steps/s of ONE surrogate (HIP-graph replay) over problem sizes in the reference's own regime (M = N, S = 1, two
fidelities): shows where the small-problem kernels hand over to the tiled ones.  Usage: python tools/size_sweep.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.mlls import VariationalELBOMF  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.graphed_step import GraphedELBOStep  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [16, 64, 128, 200, 256, 300, 384, 512]
dev = torch.device("cuda")
for n in sizes:
    prob = synthetic.make_problem(d=4, L=2, M=n, N=n, S=1, output=0, seed=0)
    model = synthetic.model_from_problem(prob, device=dev)
    elbo = VariationalELBOMF(model, n, 2)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    # Z = x (M = N): the rows are shuffled once, as the reference's loader does -- the general branch, not the shortcut
    perm = torch.roll(torch.arange(n, device=dev), 1) if n > 1 else torch.arange(n, device=dev)
    g = GraphedELBOStep(model, elbo, t(prob["x"])[perm].contiguous(), t(prob["y"])[:, None][perm].contiguous(),
                        t(prob["fid"])[:, None][perm].contiguous(), lr=1e-3)
    for _ in range(20):
        g.step()
    g.stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.step()
    g.stream.synchronize()
    dt = (time.perf_counter() - t0) / 200
    g.check()
    print("M = N = %4d (padded %4d): %.3f ms per step, %6.0f steps/s" % (n, (n + 127) // 128 * 128, dt * 1e3, 1.0 / dt), flush=True)
