#!/usr/bin/env python3
"""Time of one M x M x M product on the mid-size kernel (gemm_mid32_kernel, the chain's products at 384 < M <= 1024): plain,
lower-triangular A, and the z-batched pair of a two-layer surrogate; events around 200 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


for M in (512, 768, 1024):
    A = torch.randn(M, M, dtype=torch.float64, device=dev)
    B = torch.randn(M, M, dtype=torch.float64, device=dev)
    C = torch.empty(M, M, dtype=torch.float64, device=dev)
    ref = A @ B
    F.gemm_f64(A, B, C)
    err = float((C - ref).abs().max() / ref.abs().max())
    Al = torch.tril(A)
    F.gemm_f64(Al, B, C, tri=1)
    err_l = float((C - Al @ B).abs().max() / ref.abs().max())
    print("M = %4d: plain %.1f us | lower-triangular A %.1f us | A B^T %.1f us   (max rel err %.1e / %.1e)"
          % (M, timeit(lambda: F.gemm_f64(A, B, C)), timeit(lambda: F.gemm_f64(Al, B, C, tri=1)),
             timeit(lambda: F.gemm_f64(A, B, C, trans_b=True)), err, err_l), flush=True)
