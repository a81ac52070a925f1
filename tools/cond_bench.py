#!/usr/bin/env python3
"""Time per iteration of the conditioned training on the Forrester problem (C1 sizes: 3 surrogates,
M = N = 16, 50 Pareto points): HIP-graph replay vs eager issue.  Usage: python tools/cond_bench.py [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
np.random.seed(0)
torch.manual_seed(0)
ep = int(os.environ.get("EPOCHS", "200"))
fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=ep, num_epochs_2=ep, type_lengthscale=TL.MEDIAN)
fitter.verbose = False
for o, (name, is_con) in enumerate((("obj1", False), ("obj2", False), ("con1", True))):
    x, y, fid = synthetic.forrester_problem(o)
    fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name,
                            is_constraint=is_con)
fitter.train_mfdgps()
g = torch.Generator().manual_seed(0)
fitter.set_pareto_solution(torch.rand(50, 1, dtype=torch.float64, generator=g),
                           torch.randn(50, 2, dtype=torch.float64, generator=g) * 0.3)
for label, tiny, use_graphs in (("ONE launch per iteration (mobocmf_tiny_elbo_step mode 4; default at these sizes)", True, True),
                                ("layer path, HIP-graph replay", False, True), ("layer path, eager", False, False)):
    fitter.use_tiny_step = tiny
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fitter.train_conditioned_mfdgps(num_iters=iters, use_graphs=use_graphs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("conditioned training, %s: %.3f ms per iteration (%d iterations incl. set-up)" % (label, dt / iters * 1e3, iters))
