#!/usr/bin/env python3
"""A few launches of the mid-size product kernel alone (gemm_mid32_kernel: 512^3 and 1024^3, lower-triangular A, two layers
z-batched as the chain issues them) for counter passes: rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/pmc_mid32.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402

dev = torch.device("cuda")
for M in (512, 1024):
    A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
    B = torch.randn(M, M, dtype=torch.float64, device=dev)
    C = torch.empty(M, M, dtype=torch.float64, device=dev)
    for _ in range(6):
        F.gemm_f64(A, B, C, tri=1)
    torch.cuda.synchronize()
