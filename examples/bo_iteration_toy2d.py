#!/usr/bin/env python3
"""One full iteration of the reference's BO loop on a toy 2-D problem (2 objectives, 1 constraint, 2 fidelities), every
step on the MI355X path -- the flow of examples/example_acquisition_mfdgp_toy_2d (reference) with short schedules:

  fit the unconditioned MFDGPs  ->  sample a Pareto solution (RFF posterior samples + MOOP)  ->  fit the conditioned
  MFDGPs (theta / omega factors)  ->  maximise the cost-weighted JES acquisition per fidelity  ->  next (x, fidelity).

    python examples/bo_iteration_toy2d.py [--epochs 300] [--seed 0]
"""
import argparse
import faulthandler
import os
import sys
import time

import numpy as np

faulthandler.enable()      # a native crash leaves the Python stack on stderr
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP  # noqa: E402
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402


def blackboxes():
    """name -> (low-fidelity f, high-fidelity f, is_constraint); inputs in [0, 1]^2, objectives are minimised."""
    o1 = lambda x: np.sin(3.0 * x[:, 0]) + x[:, 1] ** 2
    o2 = lambda x: np.cos(2.0 * x[:, 0] + 1.0) * (1.0 - x[:, 1])
    c1 = lambda x: 0.9 - x[:, 0] * x[:, 1] - 0.5 * x[:, 0]
    hi = lambda f, s: (lambda x: f(x) * (1.0 + 0.3 * np.cos(4.0 * x[:, 0] + s)) + 0.05 * x[:, 1])
    return {"obj1": (o1, hi(o1, 0.0), False), "obj2": (o2, hi(o2, 1.0), False), "con1": (c1, hi(c1, 2.0), True)}


def run(epochs=300, cond_iters=200, acq_iters=50, n_low=14, n_high=6, grid=100, seed=0, device="cuda", verbose=True,
        data=None):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    if data is None:
        x = rng.uniform(size=(n_low + n_high, 2))
        fid = np.concatenate([np.zeros(n_low), np.ones(n_high)])
    else:
        x, fid = data
    fitter = BlackBoxMFDGPFitter(2, x.shape[0], num_epochs_1=epochs, num_epochs_2=epochs, pareto_set_size=10,
                                 opt_grid_size=grid, type_lengthscale=TL.MEDIAN, device=device)
    fitter.verbose = False
    for name, (lo, hi, is_con) in blackboxes().items():
        y = np.where(fid == 0, lo(x), hi(x))
        fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name,
                                is_constraint=is_con)
    t = [time.perf_counter()]
    fitter.train_mfdgps()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    fitter.sample_and_store_pareto_solution()
    t.append(time.perf_counter())
    fitter.num_epochs_2 = cond_iters
    acq = JESMOC_MFDGP(model=fitter, num_fidelities=2,
                       standard_bounds=torch.tensor([[0.0, 0.0], [1.0, 1.0]], dtype=torch.float64, device=device))
    torch.cuda.synchronize(); t.append(time.perf_counter())
    for f in range(2):
        for name, (_, _, is_con) in blackboxes().items():
            acq.add_blackbox(f, name, cost_evaluation=1.0 if f == 0 else 10.0, is_constraint=is_con)
    cand, fidelity = acq.get_nextpoint_coupled(iteration=0, verbose=verbose, maxiter=acq_iters)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if verbose:
        print("pareto set %s, front %s" % (tuple(fitter.pareto_set.shape), tuple(fitter.pareto_front.shape)))
        print("seconds: fit %.2f | pareto sample %.2f | conditioned fit %.2f | acquisition search %.2f" %
              tuple(b - a for a, b in zip(t[:-1], t[1:])))
        print("next point", cand.cpu().numpy(), "at fidelity", fidelity)
    return fitter, acq, cand, fidelity


def loop(iters=3, seed=0, verbose=True, **kw):
    """``iters`` BO iterations as the reference's driver script runs them (toy_synthetic_2D_JESMOCMF.py:305-470): a fresh
    fitter on the grown data set each time, the chosen point evaluated at the chosen fidelity for every black-box."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(20, 2))
    fid = np.concatenate([np.zeros(14), np.ones(6)])
    history = []
    for it in range(iters):
        t0 = time.perf_counter()
        _, _, cand, fidelity = run(seed=seed + it, data=(x, fid), verbose=False, **kw)
        x = np.vstack([x, cand.detach().cpu().numpy()[None, :]])
        fid = np.concatenate([fid, [float(fidelity)]])
        history.append((cand.detach().cpu().numpy(), fidelity, time.perf_counter() - t0))
        if verbose:
            print("BO iteration %d: evaluate x = %s at fidelity %d  (%.1f s, %d points now)" %
                  (it, np.round(history[-1][0], 4), fidelity, history[-1][2], x.shape[0]))
    return x, fid, history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--iters", type=int, default=1, help="number of BO iterations (1 = a single, verbose iteration)")
    a = ap.parse_args()
    if a.iters > 1:
        loop(iters=a.iters, seed=a.seed, epochs=a.epochs)
    else:
        run(epochs=a.epochs, seed=a.seed)
