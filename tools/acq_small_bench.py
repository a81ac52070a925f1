#!/usr/bin/env python3
"""The acquisition phase of a BO iteration at the reference's own sizes (Forrester: 3 black-boxes, M = N = 16, 25 fixed
samples): grids and the multi-start search of JESMOC_MFDGP.get_nextpoint_coupled, through the one-launch kernel
(TinyPredictGroup) and through the layer entry points.  Usage: python tools/acq_small_bench.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mobocmf_amd.models.mfdgp import TL
from mobocmf_amd.util import synthetic
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter
from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP
torch.manual_seed(0); np.random.seed(0)
fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=500, num_epochs_2=1500, type_lengthscale=TL.MEDIAN, device="cuda")
fitter.verbose = False
for o, (name, is_con) in enumerate((("obj1", False), ("obj2", False), ("con1", True))):
    x, y, fid = synthetic.forrester_problem(o)
    fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name, is_constraint=is_con)
fitter.train_mfdgps()
fitter.num_epochs_1, fitter.num_epochs_2 = 0, 1500
def T(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print("%-40s %.3f s" % (label, time.perf_counter() - t0), flush=True); return r
acq = T("JESMOC ctor (pareto sample + cond fit)", lambda: JESMOC_MFDGP(model=fitter, num_fidelities=2, standard_bounds=torch.tensor([[0.0], [1.0]], dtype=torch.float64, device="cuda")))
for name, is_con in (("obj1", False), ("obj2", False), ("con1", True)):
    acq.add_blackbox(0, name, cost_evaluation=1.0, is_constraint=is_con)
    acq.add_blackbox(1, name, cost_evaluation=10.0, is_constraint=is_con)
grid = torch.linspace(0.0, 1.0, 200, dtype=torch.float64, device="cuda")[:, None]
with torch.no_grad():
    T("coupled grids (2 fidelities)", lambda: {f: acq.coupled_acq(grid, fidelity=f).cpu().numpy() for f in (0, 1)})
    T("decoupled grids (6)", lambda: {(f, n): acq.decoupled_acq(grid, f, n, is_constraint=(n == "con1")).cpu().numpy() for f in (0, 1) for n in ("obj1", "obj2", "con1")})
    T("predict grids (6)", lambda: {(f, n): [v.cpu().numpy() for v in fitter.get_model(n, is_constraint=(n == "con1")).predict(grid, f)] for f in (0, 1) for n in ("obj1", "obj2", "con1")})
T("search fidelity 0 (200 it)", lambda: acq._optimize(0))
T("search fidelity 1 (200 it)", lambda: acq._optimize(1))
T("search fidelity 0 again (warm)", lambda: acq._optimize(0))
T("search fidelity 1 again (warm)", lambda: acq._optimize(1))
acq.use_tiny_step = False
T("search fidelity 0, layer path (warm)", lambda: acq._optimize(0))
T("search fidelity 1, layer path (warm)", lambda: acq._optimize(1))
