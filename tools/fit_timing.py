#!/usr/bin/env python3
"""Unconditioned fit of the three Forrester surrogates at the reference's schedule (5000 + 15000 epochs each), timed per
phase; repeated to look for run-to-run trouble."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
t = lambda a: torch.as_tensor(a, dtype=torch.float64)
for rep in range(reps):
    fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=int(5000 * scale), num_epochs_2=int(15000 * scale),
                                 type_lengthscale=TL.MEDIAN, device="cuda")
    fitter.verbose = False
    for o, (name, con) in enumerate([("obj1", False), ("obj2", False), ("con1", True)]):
        x, y, fid = synthetic.forrester_problem(o)
        fitter.initialize_mfdgp(t(x), t(y)[:, None], t(fid)[:, None], name, is_constraint=con)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fitter._train_mfdgp_graphed(True, fitter.num_epochs_1, fitter.lr_1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fitter._train_mfdgp_graphed(False, fitter.num_epochs_2, fitter.lr_2)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: phase 1 %.2f s (%.0f surrogate-steps/s), phase 2 %.2f s (%.0f surrogate-steps/s)" %
          (rep, t1 - t0, 3 * fitter.num_epochs_1 / (t1 - t0), t2 - t1, 3 * fitter.num_epochs_2 / (t2 - t1)), flush=True)
