#!/usr/bin/env python3
"""Time of the blocked Cholesky + triangular inverse of one n x n matrix (mobocmf_exact_gp_factor: pad, factorise, invert, a
matrix-vector product and the likelihood) for the one-launch factorisation (0) against the launch pair per 64 columns (4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(0)
for n in (192, 256, 384, 512, 768, 1024):
    x = torch.rand(n, 4, dtype=torch.float64, device=dev, generator=g)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    K = torch.exp(-0.5 * d2) + 1e-4 * torch.eye(n, dtype=torch.float64, device=dev)
    y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    line = []
    ref = None
    for cols in (0, 4):
        F.set_potrf_cols(cols)
        st = F.exact_gp_factor(K, y)
        assert F.check_info(st.info) == 0
        if ref is None:
            ref = float(st.mll)
        for _ in range(5):
            F.exact_gp_factor(K, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            F.exact_gp_factor(K, y)
        e1.record()
        torch.cuda.synchronize()
        line.append("cols=%d %.1f us (mll %.10e)" % (cols, e0.elapsed_time(e1) * 1e3 / 50, float(st.mll)))
    F.set_potrf_cols(0)
    print("n=%4d  %s" % (n, "  |  ".join(line)), flush=True)
