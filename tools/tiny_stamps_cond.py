#!/usr/bin/env python3
"""Phase stamps (library built with -DTINY_STAMPS) of the two launches of a conditioned iteration at Forrester sizes
(3 surrogates, 50 Pareto points, 10 x~, 16 batch rows: 76 columns per layer).  usage: python tools/tiny_stamps_cond.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.models.mfdgp import TL  # noqa: E402
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter  # noqa: E402
from mobocmf_amd.util.tiny_step import TinyConditionedStep  # noqa: E402

np.random.seed(0)
torch.manual_seed(0)
fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=50, num_epochs_2=50, type_lengthscale=TL.MEDIAN)
fitter.verbose = False
for o, (name, is_con) in enumerate((("obj1", False), ("obj2", False), ("con1", True))):
    x, y, fid = synthetic.forrester_problem(o)
    fitter.initialize_mfdgp(torch.from_numpy(x), torch.from_numpy(y)[:, None], torch.from_numpy(fid)[:, None], name,
                            is_constraint=is_con)
fitter.train_mfdgps()
g = torch.Generator().manual_seed(0)
fitter.set_pareto_solution(torch.rand(50, 1, dtype=torch.float64, generator=g), torch.randn(50, 2, dtype=torch.float64, generator=g) * 0.3)
for _, _, h in fitter._handlers():
    h.mfdgp.fix_variational_hypers_cond(True)
step = TinyConditionedStep(fitter, lr=1e-3)
step.use_graph = False
for mode in (2, 1):
    for _ in range(5):
        step._launch(2)
        step._factors()
        step._launch(1)
    step._launch(mode)
    step.check()
    st = step._work[0][-128:].cpu().numpy()
    n = int(np.max(np.nonzero(st)[0])) + 1
    d = np.diff(st[:n]) * 0.01
    print("mode %d: %d phases, total %.1f us" % (mode, n - 1, d.sum()))
    print(" ".join("%.1f" % v for v in d))
    step._work[0][-128:].zero_()
