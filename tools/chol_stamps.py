#!/usr/bin/env python3
"""Where a step of the one-launch Cholesky goes: wall-clock stamps of its panel workgroup (diagnostic build:
bash tools/build_variant.sh pcstamps -DPC_STAMPS; MOBOCMF_HIP_LIB=abtest/libpcstamps.so python tools/chol_stamps.py [n])."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import _lib, functional as F  # noqa: E402

NAMES = {1: "(loop top)", 2: "four pivot tiles, panel, updates, inverse rows 1-2", 3: "last inverse row + fetch of the next block row",
         4: "L_jj, L_jj^-1 out", 5: "L[jb+1, jb]", 6: "publish", 7: "next pivot tile updated (wavefront 0; the other nine tiles by wavefronts 1-7)"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(0)
x = torch.rand(n, 4, dtype=torch.float64, device=dev, generator=g)
K = torch.exp(-0.5 * ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)) + 1e-4 * torch.eye(n, dtype=torch.float64, device=dev)
y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
F.set_potrf_cols(0)
for _ in range(3):
    st = F.exact_gp_factor(K, y)
torch.cuda.synchronize()
lib = _lib.require_device()
buf = (ctypes.c_double * 1024)()
lib.mobocmf_debug_potrf_stamps.argtypes = [ctypes.c_void_p]
assert lib.mobocmf_debug_potrf_stamps(buf) == 0
ns = int(buf[0])
ids = [int(buf[2 + 2 * i]) for i in range(ns)]
tk = [buf[3 + 2 * i] for i in range(ns)]
print("n = %d: %d stamps, %.1f us from the first to the last (100 MHz clock)" % (n, ns, (tk[-1] - tk[0]) / 100.0))
it = 0
for i in range(1, ns):
    if ids[i - 1] == 1:
        print(" step %d" % it)
        it += 1
    print("   %6.2f us  -> %s" % ((tk[i] - tk[i - 1]) / 100.0, NAMES.get(ids[i], str(ids[i]))))
