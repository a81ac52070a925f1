#!/usr/bin/env python3
"""Randomised campaign (GPU): layer backward with zero-gradient column blocks skipped vs the dense backward, and a pruned model
forward vs the reference layout, over random shapes / replica counts / zero patterns.  Prints the worst relative differences."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from tests.test_hip_layer import _mk  # noqa: E402
from tests.test_hip_sparse_backward import _layer_grads, _elbo_and_grads  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, worst = time.time(), 0, 0.0
while time.time() - t0 < budget * 0.7:
    kind = int(rng.integers(0, 2))
    d = int(rng.choice([1, 2, 3, 5, 8, 12]))
    M = int(rng.choice([7, 16, 40, 100, 130, 200, 384, 400, 512, 520]))
    xdiv = 1 if kind == 0 else int(rng.choice([1, 2, 3, 4, 8, 16, 25]))
    nbase = int(rng.integers(3, 1 + 6000 // xdiv))
    branch = int(rng.integers(0, 2))
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=int(rng.integers(1 << 30)))
    on = np.zeros(nbase, dtype=bool)
    mode = int(rng.integers(0, 5))
    if mode == 0:
        on[:max(1, nbase // int(rng.integers(2, 9)))] = True
    elif mode == 1:
        on[rng.permutation(nbase)[:max(1, nbase // 40)]] = True
    elif mode == 2:
        on[int(rng.integers(0, nbase)):] = True
    elif mode == 3:
        on[:] = True
    cols = np.repeat(on, xdiv)
    wm = torch.tensor(rng.standard_normal(nbase * xdiv) * cols)
    wv = torch.tensor(rng.standard_normal(nbase * xdiv) * cols * (rng.random() < 0.8))
    sp = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, True)
    de = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, False)
    for k in de:
        assert bool(torch.isfinite(sp[k]).all()), (k, kind, d, M, nbase, xdiv, mode)
        err = float((sp[k] - de[k]).abs().max()) / max(float(de[k].abs().max()), 1e-300)
        worst = max(worst, err)
        assert err < 1e-8, (k, err, kind, d, M, nbase, xdiv, branch, mode)
    n += 1
print("layer backward, skipping vs dense: %d random cases, worst relative difference %.2e" % (n, worst))
n2, worst2 = 0, 0.0
while time.time() - t0 < budget:
    L = int(rng.choice([2, 3]))
    S = int(rng.choice([1, 2, 4, 8]))
    N = int(rng.integers(40, 1500))
    M = int(min(N, rng.choice([16, 48, 96, 160])))
    prob = synthetic.make_problem(d=int(rng.choice([1, 2, 4])), L=L, M=M, N=N, S=S, seed=int(rng.integers(1 << 30)))
    fid = np.asarray(prob["fid"])
    rows = [int((fid >= l).sum()) for l in range(L)]
    if rows[-1] < 1:
        continue
    e0, k0, g0, _ = _elbo_and_grads(prob, L, N, S, None, sparse=False)
    e1, k1, g1, _ = _elbo_and_grads(prob, L, N, S, rows)
    assert abs(e1 - e0) <= 1e-9 * abs(e0), (e0, e1, L, N, S, M)
    for (nm, a), (_, b) in zip(g1, g0):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)
        worst2 = max(worst2, err)
        assert err < 1e-4, (nm, err, L, N, S, M)      # the north-star tolerance: d = 1 problems carry cond(K_mm) ~ 1e12
    n2 += 1
print("model step, dead rows pruned vs reference layout: %d random problems, worst relative gradient difference %.2e" % (n2, worst2))
