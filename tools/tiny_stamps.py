#!/usr/bin/env python3
"""Where a one-launch step (csrc/tiny_step.hip) spends its time: the kernel built with -DTINY_STAMPS writes the 100 MHz
wall clock at every phase boundary; this prints the differences for one C1-sized surrogate.
usage (library built with EXTRA_HIPCC_FLAGS=-DTINY_STAMPS): python tools/tiny_stamps.py [config]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.util import synthetic  # noqa: E402
from mobocmf_amd.util.tiny_step import TinyELBOStep  # noqa: E402

cfg = dict(synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C1"])
prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], seed=0)
model = synthetic.model_from_problem(prob, device="cuda")
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device="cuda")
step = TinyELBOStep([model], [cfg["N"]], [t(prob["x"])], [t(prob["y"])], [t(prob["fid"])], lr=1e-3)
for _ in range(20):
    step.step()
step.check()
st = step._work[0][-128:].cpu().numpy()
n = int(np.max(np.nonzero(st)[0])) + 1
d = np.diff(st[:n]) * 0.01      # 100 MHz -> us
print("config", cfg, "phases", n - 1, "total %.1f us" % d.sum())
print(" ".join("%.1f" % v for v in d))
