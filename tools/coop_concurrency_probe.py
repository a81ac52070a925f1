"""Several cooperative one-launch steps on several streams at once: does an in-launch barrier ever get abandoned?
    python tools/coop_concurrency_probe.py <dummy streams created first> <step objects> <iterations> [sync]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.util import synthetic      # noqa: E402
from mobocmf_amd.util.coop_step import CoopELBOStep      # noqa: E402

ndummy, nobj, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sync = len(sys.argv) > 4 and sys.argv[4] == "sync"
dev = torch.device("cuda", 0)
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1.0)
torch.cuda.synchronize()
steps = []
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
for o in range(nobj):
    prob = synthetic.make_problem(d=2, L=2, M=48, N=56, S=1, seed=o, output=o)
    model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=dev)
    steps.append(CoopELBOStep([model], [56], [t(prob["x"])], [t(prob["y"])], [t(prob["fid"])], lr=1e-3, force=True))
    steps[-1].wgs_per_model = int(os.environ.get("WGS", "0"))
print("streams:", [hex(s.stream.cuda_stream) for s in steps])
t0 = time.perf_counter()
bad = 0
for it in range(iters):
    for s in steps:
        s.step()
        if sync:
            s.stream.synchronize()
    if it % int(os.environ.get("CHECK", "50")) == int(os.environ.get("CHECK", "50")) - 1 or it == iters - 1:
        torch.cuda.synchronize()
        inf = [int(s.infos.min()) for s in steps]
        if min(inf) < 0:
            print("iteration", it, "infos", inf, "wgs", [s.wgs_used for s in steps], "detail (info words)", [s.infos.cpu().tolist() for s in steps],
                  "sync words", [s._sync_words()[::16].cpu().tolist() for s in steps])
            bad += 1
            for s in steps:
                s.infos.zero_()
                s._sync_words().zero_()
print("dummy streams %d, %d step objects, %d iterations%s: %d bad checks, %.2f s" % (ndummy, nobj, iters, " (serialised)" if sync else "", bad, time.perf_counter() - t0))
