#!/usr/bin/env python3
"""Backward error of the blocked Cholesky + triangular inverse (through mobocmf_exact_gp_factor) for its three forms
(0 = all steps in one launch; 4 / 1 = a launch pair per 64 columns, 4 columns / 1 column per hand-over) and torch.linalg.cholesky (rocSOLVER), on Gram matrices of growing condition number."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(0)
for n, d, ls in ((80, 3, 0.6), (512, 8, 1.4), (1000, 8, 1.4), (700, 2, 0.5), (1024, 3, 0.8)):
    x = torch.rand(n, d, dtype=torch.float64, device=dev, generator=g)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    K = torch.exp(-0.5 * d2 / ls ** 2) + 1e-6 * torch.eye(n, dtype=torch.float64, device=dev)
    y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    cond = float(torch.linalg.cond(K))
    npad = (n + 127) // 128 * 128
    out = []
    for cols in (0, 4, 1):
        F.set_potrf_cols(cols)
        st = F.exact_gp_factor(K, y)
        assert F.check_info(st.info) == 0
        buf = st.state.view(torch.float64)
        L = buf[:npad * npad].view(npad, npad)[:n, :n]
        off = (npad * npad * 8 + 255) // 256 * 256 // 8
        Li = buf[off:off + npad * npad].view(npad, npad)[:n, :n]
        r1 = float(torch.linalg.norm(L @ L.T - K) / torch.linalg.norm(K))
        r2 = float(torch.linalg.norm(Li @ L - torch.eye(n, dtype=torch.float64, device=dev)))
        out.append("cols=%d: |LL^T-K|/|K| %.2e  |L^-1 L - I| %.2e  mll %.12e" % (cols, r1, r2, float(st.mll)))
    F.set_potrf_cols(0)
    Lt = torch.linalg.cholesky(K)
    r1 = float(torch.linalg.norm(Lt @ Lt.T - K) / torch.linalg.norm(K))
    print("n=%d d=%d cond %.1e | %s | %s | %s | torch: %.2e" % (n, d, cond, out[0], out[1], out[2], r1), flush=True)
