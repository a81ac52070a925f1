#!/usr/bin/env python3
"""Time of one acquisition evaluation WITH its candidate gradient (JESMOC_MFDGP.py:137-184) at the reference's later loop sizes:
moments of n models at T candidates + d/dX, through CoopPredictGroup (two cooperative launches) and through the layer path
(MFDGP.predict_for_acquisition per model, autograd).  usage: python tools/acq_mid_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.coop_step import CoopPredictGroup  # noqa: E402

dev = "cuda"
for M, T, S, nmod in ((48, 50, 10, 6), (64, 50, 10, 6), (75, 200, 10, 6), (128, 100, 10, 6)):
    models = [synthetic.model_from_problem(synthetic.make_problem(d=2, L=2, M=M, N=M, S=1, seed=s), num_samples_for_training=1,
                                           num_samples_for_acquisition=S, device=dev) for s in range(nmod)]
    X = torch.rand(T, 2, dtype=torch.float64, device=dev)
    grp = CoopPredictGroup(models, 1, T, 2)

    def one_launch():
        Xb = X.clone().requires_grad_(True)
        mu, v = grp.acquisition_moments(Xb)
        (mu.sum() + v.sum()).backward()
        return Xb.grad

    def layer_path():
        Xa = X.clone().requires_grad_(True)
        tot = 0.0
        for m in models:
            m.eval()
            mu, v = m.predict_for_acquisition(Xa, 1)
            m.train()
            tot = tot + mu.sum() + v.sum()
        tot.backward()
        return Xa.grad

    import contextlib

    def layer_path_frozen():      # what JESMOC_MFDGP._optimize does on the layer path: the chains computed once per search
        return layer_path()

    out = []
    def one_launch_frozen():
        return one_launch()

    for name, fn, reps in (("cooperative launches", one_launch, 200), ("the same, chains kept (freeze())", one_launch_frozen, 200),
                           ("layer path", layer_path, 20),
                           ("layer path, frozen chains", layer_path_frozen, 50)):
        stack = contextlib.ExitStack()
        if fn is layer_path_frozen:
            for m in models:
                stack.enter_context(m.frozen_chains())
        if fn is one_launch_frozen:
            grp.freeze()
            stack.callback(grp.thaw)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append("%s %.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
        stack.close()
    print("M = N = %3d, %d models, T = %d candidates, S = %d samples, fidelity 1: %s  (%d workgroups per model)"
          % (M, nmod, T, S, " | ".join(out), grp.wgs_used), flush=True)
