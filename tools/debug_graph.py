import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobocmf_amd.util import synthetic
from mobocmf_amd.util.graphed_step import GraphedELBOStep
from mobocmf_amd.mlls import VariationalELBOMF
from tests.helpers import to_t
mode = sys.argv[1]
dev = "cuda"
prob = synthetic.make_problem(d=2, L=2, M=16, N=16 if mode != "big" else 64, S=1, seed=0)
model = synthetic.model_from_problem(prob, device=dev)
elbo = VariationalELBOMF(model, prob["N"], 2)
t = lambda a: to_t(a).to(dev)
x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
print("capture 1", flush=True)
model.fix_variational_hypers(True)
g1 = GraphedELBOStep(model, elbo, x, y, fid, lr=1e-3)
for _ in range(5): g1.step()
g1.check(); print("ok 1", float(g1.loss), flush=True)
if mode == "del":
    del g1; gc.collect(); torch.cuda.synchronize()
model.fix_variational_hypers(False)
print("capture 2", flush=True)
g2 = GraphedELBOStep(model, elbo, x, y, fid, lr=1e-3)
for _ in range(5): g2.step()
g2.check(); print("ok 2", float(g2.loss), flush=True)
