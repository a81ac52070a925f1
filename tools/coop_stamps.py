#!/usr/bin/env python3
"""Where a cooperative one-launch step (csrc/coop_step.hip) spends its time: the kernel built with -DCOOP_STAMPS lets the first
workgroup of a surrogate write the 100 MHz wall clock at every phase boundary (before and after every in-launch barrier).
    bash tools/build_variant.sh cstamps -DCOOP_STAMPS
    MOBOCMF_HIP_LIB=$PWD/abtest/libcstamps.so python tools/coop_stamps.py d L M N S [n_surrogates] [wgs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.coop_step import CoopELBOStep  # noqa: E402

NAMES = {0: "start", 1: "P0 hyper-parameters", 2: "K_mm", 3: "Cholesky", 4: "inverse", 5: "L, L^-1 out; L_S in", 6: "U, a, KL", 7: "Z~, a of every layer staged", 8: "K_mm by all workgroups",
         10: "forward layer 0", 11: "forward layer 1", 12: "forward layer 2", 15: "coupling", 20: "backward columns layer 0",
         21: "backward columns layer 1", 22: "backward columns layer 2", 25: "+ syrk of the layer above (0)", 26: "+ syrk (1)",
         27: "+ syrk (2)", 30: "syrk layer 0", 31: "CB1 G1 = U^T H", 32: "CB2+3 Y, g_LS", 33: "CB4-6 dL, P, T4", 34: "CB7 T5",
         35: "CB7+8 T5 = T4 L^-1, Gram backward of K_mm", 40: "gradients + Adam", 50: "    . layer staged", 51: "    . rows staged",
         52: "    . K block", 53: "    . A = L^-1 K", 54: "    . C = U^T A", 55: "    . sums + A, C out", 60: "    . layer staged",
         61: "    . A, C in + upstream gradients", 62: "    . dA", 63: "    . dK", 64: "    . Gram backward", 99: "  (barrier wait)"}
d, L, M, N, S = [int(v) for v in sys.argv[1:6]]
n_sur = int(sys.argv[6]) if len(sys.argv) > 6 else 1
wgs = int(sys.argv[7]) if len(sys.argv) > 7 else 0
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device="cuda")
models, xs, ys, fs = [], [], [], []
for o in range(n_sur):
    prob = synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, seed=o, output=o)
    models.append(synthetic.model_from_problem(prob, num_samples_for_training=S, device="cuda"))
    perm = torch.as_tensor(np.random.default_rng(5).permutation(N), device="cuda")
    xs.append(t(prob["x"])[perm].contiguous()), ys.append(t(prob["y"])[perm].contiguous()), fs.append(t(prob["fid"])[perm].contiguous())
step = CoopELBOStep(models, [N] * n_sur, xs, ys, fs, lr=1e-3, force=True)
step.wgs_per_model = wgs
for _ in range(20):
    step.step()
step.check()
st = step._work[0][-256:].cpu().numpy().reshape(-1, 2)
n = int(np.max(np.nonzero(st[:, 1])[0])) + 1
ids, clk = st[:n, 0].astype(int), st[:n, 1]
dt = np.diff(clk) * 0.01
print("d=%d L=%d M=%d N=%d S=%d, %d surrogate(s), %d workgroups each: %.1f us between the first and the last stamp of workgroup 0" %
      (d, L, M, N, S, n_sur, step.wgs_used, dt.sum()))
wait = 0.0
for i in range(1, n):
    print("%8.1f us  %s" % (dt[i - 1], NAMES.get(ids[i], str(ids[i]))))
    if ids[i] == 99:
        wait += dt[i - 1]
print("barrier waits (incl. waiting for the slowest workgroup): %.1f us of %.1f" % (wait, dt.sum()))
