#!/usr/bin/env python3
"""Does the leading dimension of the M x N' panels matter?  N' = 65536 doubles makes the rows of a panel exactly 512 KB apart:
the 128-byte row segments an A-type operand tile is made of then sit at the same offset of 128 consecutive 512 KB rows.
Times the weighted syrk (both operands of that type) and the triangular NN product for padded leading dimensions."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import _lib  # noqa: E402
from mobocmf_amd import functional as F  # noqa: E402

lib = _lib.require_device()
dev = torch.device("cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=30):
    for _ in range(8):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


warm = torch.tril(torch.randn(512, 512, dtype=torch.float64, device=dev)), torch.randn(512, 65536, dtype=torch.float64, device=dev), torch.empty(512, 65536, dtype=torch.float64, device=dev)
for _ in range(600):
    F.gemm_f64_epilogue(warm[0], warm[1], warm[2], 1, 0)
torch.cuda.synchronize()
for M in (512, 1024):
    N = 65536
    for pad in (0, 16, 32, 64, 144, 272, 1040):
        ld = N + pad
        A = torch.randn(M, ld, dtype=torch.float64, device=dev)
        C = torch.empty(M, ld, dtype=torch.float64, device=dev)
        w = torch.randn(N, dtype=torch.float64, device=dev)
        H = torch.empty(M, M, dtype=torch.float64, device=dev)
        L = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
        nb = _lib._SZ()
        _lib.check(lib.mobocmf_syrk_workspace_bytes(M, N, None, ctypes.byref(nb)), "ws")
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        t_syrk = timeit(lambda: lib.mobocmf_syrk_weighted_f64(M, N, P(A), ld, P(w), P(H), P(ws), nb.value, None, None, st))
        t_gemm = timeit(lambda: lib.mobocmf_gemm_f64_epilogue(1, 0, M, N, M, P(L), M, P(A), ld, P(C), ld, 1.0, 0, None, None, None, None, None, None, None, None, None, None, st))
        if pad == 0:
            ref = (A[:, :N] * w[None, :]) @ A[:, :N].T
        print("M=%d ld=N'+%-5d syrk %.4f ms   lower-tri NN product (plain store) %.4f ms" % (M, pad, t_syrk, t_gemm), flush=True)
        del A, C
