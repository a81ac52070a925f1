#!/usr/bin/env python3
"""The reference's Forrester walk-through (examples/example_acquisition_mfdgp_forrester/...py) on the MI355X path, with
the reference's own schedule by default: 2 objectives (+-Forrester) and 1 constraint (sin / cos), 12 low- and 4
high-fidelity points, 5000 + 15000 unconditioned epochs per surrogate, a Pareto solution from RFF posterior samples,
15000 conditioned iterations, then the coupled and decoupled JES acquisitions on a grid for both fidelities.

The fitter and the acquisition object go through the same dill round-trips as in the reference; plots are written only if
matplotlib is importable.  This is configuration C1 of SURVEY 8 (the reference's own CPU-runnable case).

    python examples/example_acquisition_mfdgp_forrester.py                 # the reference's schedule
    python examples/example_acquisition_mfdgp_forrester.py --scale 0.02    # 2 % of every schedule (smoke run)
"""
import argparse
import faulthandler
import os
import sys
import tempfile
import time

import numpy as np

faulthandler.enable()      # a native crash leaves the Python stack on stderr
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP  # noqa: E402
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402
from mobocmf_amd.util.util import read_pickle, save_pickle  # noqa: E402

COST_LOWER_FIDELITY, COST_HIGHER_FIDELITY = 1.0, 10.0


def main(scale=1.0, seed=0, device="cuda", out_dir=None, verbose=True):
    np.random.seed(seed)
    torch.manual_seed(seed)
    ep1, ep2, cond = max(int(5000 * scale), 20), max(int(15000 * scale), 20), max(int(15000 * scale), 20)
    out_dir = out_dir or tempfile.mkdtemp(prefix="mobocmf_forrester_")
    t = [time.perf_counter()]

    fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=ep1, num_epochs_2=ep2, type_lengthscale=TL.MEDIAN, device=device)
    fitter.verbose = False
    for o, (name, is_con) in enumerate((("obj1", False), ("obj2", False), ("con1", True))):
        x, y, fid = synthetic.forrester_problem(o)      # the example's data and pooled standardisation (:51-104)
        kw = {"threshold_constraint": 0.0} if is_con else {}
        fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name,
                                is_constraint=is_con, **kw)
    fitter.train_mfdgps()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    save_pickle(out_dir, "fitter_uncond.dat", fitter)
    fitter = read_pickle(out_dir, "fitter_uncond.dat")

    fitter.num_epochs_1, fitter.num_epochs_2 = 0, cond           # one phase in the conditioned training
    acq = JESMOC_MFDGP(model=fitter, num_fidelities=2,
                       standard_bounds=torch.tensor([[0.0], [1.0]], dtype=torch.float64, device=device))
    torch.cuda.synchronize(); t.append(time.perf_counter())
    for name, is_con in (("obj1", False), ("obj2", False), ("con1", True)):
        acq.add_blackbox(0, name, cost_evaluation=COST_LOWER_FIDELITY, is_constraint=is_con)
        acq.add_blackbox(1, name, cost_evaluation=COST_HIGHER_FIDELITY, is_constraint=is_con)
    save_pickle(out_dir, "jesmoc_mfdgp.dat", acq)
    acq = read_pickle(out_dir, "jesmoc_mfdgp.dat")

    grid = torch.linspace(0.0, 1.0, 200, dtype=torch.float64, device=device)[:, None]
    with torch.no_grad():
        coupled = {f: acq.coupled_acq(grid, fidelity=f).cpu().numpy() for f in (0, 1)}
        decoupled = {(f, n): acq.decoupled_acq(grid, f, n, is_constraint=(n == "con1")).cpu().numpy()
                     for f in (0, 1) for n in ("obj1", "obj2", "con1")}
        preds = {(f, n): [v.cpu().numpy() for v in fitter.get_model(n, is_constraint=(n == "con1")).predict(grid, f)]
                 for f in (0, 1) for n in ("obj1", "obj2", "con1")}
    cand, fidelity = acq.get_nextpoint_coupled(iteration=0, verbose=False)
    torch.cuda.synchronize(); t.append(time.perf_counter())

    # the fit interpolates the high-fidelity data (noise-free black-boxes)
    xs, ys, fids = synthetic.forrester_problem(0)
    hi = torch.from_numpy(xs[fids == 1]).to(device)
    mu_hi = acq.blackbox_mfdgp_fitter_uncond.get_model("obj1").predict(hi, 1)[0].detach().cpu().numpy().reshape(-1)
    fit_err = float(np.abs(mu_hi - ys[fids == 1]).max())
    if verbose:
        print("schedule: %d + %d unconditioned epochs x 3 surrogates, %d conditioned iterations" % (ep1, ep2, cond))
        print("seconds : unconditioned fit %.1f | Pareto sample + conditioned fit %.1f | acquisition grids + search %.1f"
              % tuple(b - a for a, b in zip(t[:-1], t[1:])))
        print("Pareto set: %d points; max |mean - y| at the high-fidelity data: %.2e" % (fitter.pareto_set.shape[0], fit_err))
        for f in (0, 1):
            print("fidelity %d: coupled JES max %.4f at x = %.3f (cost-weighted %.4f)" %
                  (f, coupled[f].max(), float(grid[int(coupled[f].argmax()), 0]),
                   coupled[f].max() / (3 * (COST_LOWER_FIDELITY if f == 0 else COST_HIGHER_FIDELITY))))
        print("next evaluation: x = %.4f at fidelity %d" % (float(cand[0]), fidelity))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(2, 1, figsize=(10, 8))
        g = grid.cpu().numpy()[:, 0]
        for f, c in ((0, "m"), (1, "g")):
            m, v = preds[(f, "obj1")]
            ax[0].plot(g, m.reshape(-1), c + "-", label="obj1 fidelity %d" % f)
            ax[0].fill_between(g, m.reshape(-1) - np.sqrt(v.reshape(-1)), m.reshape(-1) + np.sqrt(v.reshape(-1)), color=c, alpha=0.3)
            ax[1].plot(g, coupled[f], c + "-", label="coupled JES fidelity %d" % f)
        ax[0].legend(); ax[1].legend()
        fig.savefig(os.path.join(out_dir, "forrester_acquisition.png"))
        if verbose:
            print("figure:", os.path.join(out_dir, "forrester_acquisition.png"))
    except ImportError:
        pass
    return {"coupled": coupled, "decoupled": decoupled, "next": (cand, fidelity), "fit_err": fit_err,
            "times": [b - a for a, b in zip(t[:-1], t[1:])], "fitter": fitter}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the reference's 5000/15000/15000 schedule")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    main(scale=a.scale, seed=a.seed)
