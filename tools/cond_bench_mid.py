#!/usr/bin/env python3
"""Time per joint iteration of the conditioned training (blackbox_mfdgp_fitter.py:245-354) at the sizes the
reference's BO loop reaches later on: n black-boxes (2 objectives + the rest constraints), M = N inducing / training points in
d = 2, 50 Pareto points, 10 x~ points -- through the cooperative one-launch step (mode 4, and its three-launch form), and
through the layer path (HIP-graph replay).  Usage: python tools/cond_bench_mid.py [M=64] [black-boxes=4] [iters=300]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.mlls import VariationalELBOMF  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter, MFDGPHandler  # noqa: E402
from mobocmf_amd.util.coop_step import CoopConditionedStep  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nbb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
DEV = "cuda"


def build():
    from torch.utils.data import TensorDataset
    fitter = BlackBoxMFDGPFitter(2, M, device=DEV)
    fitter.verbose = False
    n_obj = 2
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=DEV)
    for o in range(nbb):
        prob = synthetic.make_problem(d=2, L=2, M=M, N=M, S=1, output=o, seed=o)
        prob["noise"] = [np.array(1e-2), np.array(2e-2)]
        model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
        h = MFDGPHandler.__new__(MFDGPHandler)
        h.mfdgp, h.num_data, h.num_fidelities, h.batch_size = model, M, 2, M
        h.elbo = VariationalELBOMF(model, M, 2)
        perm = torch.as_tensor(np.random.default_rng(3 + o).permutation(M), device=DEV)
        h.train_dataset = TensorDataset(t(prob["x"])[perm], t(prob["y"])[perm][:, None], t(prob["fid"])[perm][:, None])
        h.iter_train_loader = None
        (fitter.mfdgp_handlers_objs if o < n_obj else fitter.mfdgp_handlers_cons)["bb%d" % o] = h
    fitter.num_obj, fitter.num_con = n_obj, nbb - n_obj
    fitter.thresholds_cons = torch.tensor([0.1] * (nbb - n_obj), dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    fitter.set_pareto_solution(torch.rand(50, 2, dtype=torch.float64, generator=g), torch.randn(50, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in fitter._handlers():
        h.mfdgp.fix_variational_hypers_cond(True)
    return fitter


print("conditioned training, %d black-boxes, M = N = %d, d = 2, 50 Pareto points, 10 x~ points" % (nbb, M))
for label, one in (("ONE launch per iteration (mobocmf_coop_elbo_step mode 4, HIP-graph replay)", True),
                   ("forward-only launch + factor launches + step launch (HIP-graph replay)", False)):
    fitter = build()
    step = CoopConditionedStep(fitter, lr=1e-3)
    step.one_launch = one
    for _ in range(10):
        step.step()
    step.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step.step()
    step.stream.synchronize()
    dt = time.perf_counter() - t0
    step.check()
    print("  %-82s %.3f ms per iteration (%d workgroups per surrogate)" % (label, dt / iters * 1e3, step.wgs_used))
fitter = build()
fitter.use_tiny_step = False
torch.cuda.synchronize()
fitter.train_conditioned_mfdgps(num_iters=20, use_graphs=True)      # (capture + warm-up)
torch.cuda.synchronize()
t0 = time.perf_counter()
fitter.train_conditioned_mfdgps(num_iters=iters, use_graphs=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("  %-82s %.3f ms per iteration (incl. one graph capture: %d iterations)" % ("layer path (HIP-graph replay)", dt / iters * 1e3, iters))
