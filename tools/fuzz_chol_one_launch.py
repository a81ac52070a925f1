#!/usr/bin/env python3
"""Randomised campaign for the one-launch Cholesky + inverse (potrf_coop_kernel, mobocmf_tuning.potrf_cols = 0) against the launch
pair per 64 columns (potrf_cols = 4): random orders n in 129..1024 (every residue of 64 and 128 gets hit), random kernels and
jitters, through mobocmf_exact_gp_factor (one layer) and through the z-batched layer chain of 2-3 layer models (several layers per
launch, M in 130..520).  usage: python tools/fuzz_chol_one_launch.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd import functional as F  # noqa: E402
from mobocmf_amd.mlls import VariationalELBOMF  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
dev = torch.device("cuda")
worst = {"L": 0.0, "Linv": 0.0, "mll": 0.0, "res": 0.0, "elbo": 0.0, "grad": 0.0}
for case in range(n_cases):
    if case % 3 != 2:      # one matrix through mobocmf_exact_gp_factor
        n = int(rng.integers(129, 1025))
        d = int(rng.integers(2, 9))
        ls = float(rng.uniform(0.4, 1.5))
        jit = float(10.0 ** rng.uniform(-6, -2))
        g = torch.Generator(device=dev)
        g.manual_seed(int(rng.integers(1 << 30)))
        x = torch.rand(n, d, dtype=torch.float64, device=dev, generator=g)
        K = torch.exp(-0.5 * ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1) / ls ** 2) + jit * torch.eye(n, dtype=torch.float64, device=dev)
        y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        out = {}
        for cols in (0, 4):
            with F.tuning(potrf_cols=cols):
                st = F.exact_gp_factor(K, y)
            assert F.check_info(st.info) == 0, (n, d, ls, jit, cols)
            L, Li, _, npad = F._exact_state_views(st)
            out[cols] = (L.clone(), Li.clone(), float(st.mll))
        L0, Li0, m0 = out[0]
        L4, Li4, m4 = out[4]
        errs = {"L": float((L0 - L4).abs().max() / L4.abs().max()), "Linv": float((Li0 - Li4).abs().max() / Li4.abs().max()),
                "mll": abs(m0 - m4) / abs(m4), "res": float(torch.linalg.norm(L0[:n, :n] @ L0[:n, :n].T - K) / torch.linalg.norm(K))}
        assert torch.equal(torch.triu(L0, 1), torch.zeros_like(L0)) and torch.equal(torch.triu(Li0, 1), torch.zeros_like(Li0))
    else:                  # a 2-3 layer model: the layers' chains z-batched in one launch
        Lr = int(rng.integers(2, 4)); M = int(rng.integers(130, 521)); d = int(rng.integers(2, 7)); S = int(rng.choice([1, 2]))
        N = int(rng.integers(M, M + 200))
        prob = synthetic.make_problem(d=d, L=Lr, M=M, N=N, S=S, seed=int(rng.integers(1 << 30)))
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
        res = {}
        for cols in (0, 4):
            with F.tuning(potrf_cols=cols):
                model = synthetic.model_from_problem(prob, num_samples_for_training=S, device="cuda")
                model.set_check_pd(False)
                e, _ = VariationalELBOMF(model, N, Lr)(model(t(prob["x"]), eps=[None] + [t(v) for v in prob["eps"][1:]]),
                                                      t(prob["y"])[None, :], t(prob["fid"])[:, None])
                (-e).backward()
            res[cols] = (float(e.detach()), [p.grad.detach().clone() for p in model.parameters() if p.grad is not None])
        errs = {"elbo": abs(res[0][0] - res[4][0]) / abs(res[4][0]),
                "grad": max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300) for a, b in zip(res[0][1], res[4][1]))}
    for k, v in errs.items():
        assert np.isfinite(v), (case, k)
        worst[k] = max(worst[k], v)
    if (case + 1) % 20 == 0:
        print("%d cases: worst  " % (case + 1) + "  ".join("%s %.2e" % kv for kv in worst.items()), flush=True)
# both forms are backward stable; what separates them is cond * eps (Gram matrices with jitter down to 1e-6: cond up to ~1e9)
assert worst["res"] < 1e-14 and worst["L"] < 1e-6 and worst["Linv"] < 1e-4 and worst["mll"] < 1e-7 and worst["elbo"] < 1e-6 and worst["grad"] < 1e-3, worst
print("campaign passed: %d cases" % n_cases)
