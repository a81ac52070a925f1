#!/usr/bin/env python3
"""Wall time of the acquisition search (optimize_acqf_multistart over JES) at C3-sized surrogates, with
the chains recomputed at every evaluation vs MFDGP.frozen_chains().  Usage: python tools/acq_bench.py [maxiter]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import _JES_MFDGP, optimize_acqf_multistart  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402

cfg = synthetic.CONFIGS[os.environ.get("CONFIG", "C3")]
maxiter = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
models = []
for seed in (0, 1):
    prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], seed=seed)
    models.append(synthetic.model_from_problem(prob, device=dev))      # S = the config's 8 fixed samples
jes = _JES_MFDGP(cfg["L"] - 1, models[0], models[1])
bounds = torch.stack([torch.zeros(cfg["d"], dtype=torch.float64, device=dev), torch.ones(cfg["d"], dtype=torch.float64, device=dev)])
for frozen in (False, True):
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if frozen:
        with jes.frozen():
            cand, val = optimize_acqf_multistart(jes, bounds, maxiter=maxiter, generator=gen)
    else:
        cand, val = optimize_acqf_multistart(jes, bounds, maxiter=maxiter, generator=gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("frozen_chains=%s: %d iterations in %.2f s (%.1f ms per iteration), value %.6e at %s" %
          (frozen, maxiter, dt, dt / maxiter * 1e3, float(val), [round(float(v), 4) for v in cand[0]]))
