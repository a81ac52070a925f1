#!/usr/bin/env python3
"""Randomised campaign for CoopPredictGroup (modes 2 and 3 of mobocmf_coop_elbo_step: the acquisition moments of several models
at T candidates and their gradient w.r.t. the candidates, with and without the chains kept across launches) against
MFDGP.predict_for_acquisition through the layer entry points.  usage: python tools/fuzz_coop_predict.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.coop_step import CoopPredictGroup, fits_predict  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
DEV = "cuda"
rel = lambda a, b: float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-300))
worst, worst_case = {}, {}
done = 0
while done < n_cases:
    L = int(rng.integers(1, 4)); d = int(rng.integers(1, 9)); M = int(rng.integers(1, 129)); S = int(rng.choice([1, 2, 3, 5, 10, 25]))
    fidelity = int(rng.integers(0, L)); T = int(rng.integers(1, 120)); nmod = int(rng.integers(1, 5))
    if T * S > 1500:
        continue
    N = int(rng.integers(max(M, 4), M + 60))
    models = [synthetic.model_from_problem(synthetic.make_problem(d=d, L=L, M=M, N=N, S=1, seed=int(rng.integers(1 << 30))),
                                           num_samples_for_training=1, num_samples_for_acquisition=S, device=DEV) for _ in range(nmod)]
    for m in models:      # (prob["samples"] holds S = 1 values: give every layer its S fixed samples)
        for l in range(1, L):
            lay = getattr(m, "hidden_layer_%d" % l)
            lay.samples = torch.randn(S, 1, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(done + l))
    if not all(fits_predict(m, fidelity, T, d) for m in models):
        continue
    g = torch.Generator().manual_seed(done)
    X = torch.rand(T, d, dtype=torch.float64, generator=g).to(DEV)
    wm = torch.randn(nmod, T, dtype=torch.float64, generator=g).to(DEV)
    wv = torch.randn(nmod, T, dtype=torch.float64, generator=g).to(DEV)
    Xa = X.clone().requires_grad_(True)
    rm, rv = [], []
    for m in models:
        m.eval()
        mu, v = m.predict_for_acquisition(Xa, fidelity)
        m.train()
        rm.append(mu), rv.append(v)
    rm, rv = torch.stack(rm), torch.stack(rv)
    ((rm * wm).sum() + (rv * wv).sum()).backward()
    grp = CoopPredictGroup(models, fidelity, T, d)
    grp.wgs_per_model = int(rng.choice([0, 0, 1, 2, 3, 7, 16]))
    errs = {}
    for frozen in (False, True, True):
        if frozen:
            grp.freeze()
        Xb = X.clone().requires_grad_(True)
        mu, v = grp.acquisition_moments(Xb)
        ((mu * wm).sum() + (v * wv).sum()).backward()
        for k, e in (("mean", rel(mu, rm)), ("var", rel(v, rv)), ("dX", rel(Xb.grad, Xa.grad))):
            errs[k] = max(errs.get(k, 0.0), e)
    grp.thaw()
    hard = d <= 2 and M > 32      # many inducing points in one or two dimensions: cond(K_mm + 1e-6 I) ~ 1e9 .. 1e13
    for k, e in errs.items():
        kk = k + "_illcond" if hard else k
        if not np.isfinite(e) or e > worst.get(kk, 0.0):
            worst[kk], worst_case[kk] = e, (L, d, M, S, fidelity, T, nmod)
    done += 1
    if done % 20 == 0:
        print("%d cases: worst relative differences  " % done + "  ".join("%s %.2e" % kv for kv in sorted(worst.items())), flush=True)
print("worst cases (L, d, M, S, fidelity, T, models):", worst_case)
assert worst.get("mean", 0) < 1e-6 and worst.get("var", 0) < 1e-5 and worst.get("dX", 0) < 1e-4, worst
assert all(v < 5e-2 for k, v in worst.items() if k.endswith("_illcond")), worst
print("campaign passed: %d cases" % done)
