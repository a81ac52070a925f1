#!/usr/bin/env python3
"""One whole BO iteration of the reference's 2-D toy loop (examples/toy_synthetic_2D_JESMOCMF/...py:305-331) at a LATER iteration's
size -- N = M = 64 points (44 low + 20 high fidelity), two objectives + one constraint -- through the mirrored caller surface:
unconditioned fit, Pareto sample, conditioned fit, JES acquisition search; once with the one-launch kernels (cooperative launch:
training step, conditioned iteration, acquisition moments + candidate gradients), once on the layer entry points.
usage: python tools/bo_iteration_mid.py [epochs per phase] [conditioned iterations] [search iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import bo_iteration_toy2d as B  # noqa: E402
from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP  # noqa: E402
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
A = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for one_launch in (True, False, True):
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    np.random.seed(0)
    x = rng.uniform(size=(64, 2))
    fid = np.concatenate([np.zeros(44), np.ones(20)])
    fitter = BlackBoxMFDGPFitter(2, 64, num_epochs_1=E, num_epochs_2=E, pareto_set_size=10, opt_grid_size=100,
                                 type_lengthscale=TL.MEDIAN, device="cuda")
    fitter.verbose = False
    fitter.use_tiny_step = one_launch
    for name, (lo, hi, is_con) in B.blackboxes().items():
        y = np.where(fid == 0, lo(x), hi(x))
        fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name, is_constraint=is_con)
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    fitter.train_mfdgps()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    fitter.sample_and_store_pareto_solution()
    t.append(time.perf_counter())
    fitter.num_epochs_2 = C
    acq = JESMOC_MFDGP(model=fitter, num_fidelities=2,
                       standard_bounds=torch.tensor([[0.0, 0.0], [1.0, 1.0]], dtype=torch.float64, device="cuda"))
    acq.use_tiny_step = one_launch
    torch.cuda.synchronize(); t.append(time.perf_counter())
    for f in range(2):
        for name, (_, _, is_con) in B.blackboxes().items():
            acq.add_blackbox(f, name, cost_evaluation=1.0 if f == 0 else 10.0, is_constraint=is_con)
    cand, fidelity = acq.get_nextpoint_coupled(iteration=0, verbose=False, maxiter=A)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    d = [b - a for a, b in zip(t[:-1], t[1:])]
    print("%s: fit (2 x %d epochs x 3 surrogates) %.2f s | Pareto sample %.2f s | conditioned fit (%d iterations) %.2f s | acquisition "
          "search (2 fidelities x %d iterations, 5 restarts) %.2f s | total %.2f s | next point %s at fidelity %d"
          % ("one-launch kernels" if one_launch else "layer entry points", E, d[0], d[1], C, d[2], A, d[3], sum(d),
             np.round(cand.cpu().numpy(), 4), fidelity), flush=True)
