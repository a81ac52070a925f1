import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
from mobocmf_amd.util import synthetic
from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter
from mobocmf_amd.util.graphed_step import GraphedELBOStep
from mobocmf_amd import functional as F
from tests.helpers import to_t
x, y, fid = synthetic.forrester_problem(0)
fitter = BlackBoxMFDGPFitter(2, 16, num_epochs_1=150, num_epochs_2=150, device="cuda")
fitter.verbose = False
fitter.initialize_mfdgp(to_t(x), to_t(y)[:, None], to_t(fid)[:, None], "obj1")
h = fitter.mfdgp_handlers_objs["obj1"]
fitter._train_mfdgp_graphed(True, 150, 3e-3)
print("phase 1 done")
h.mfdgp.fix_variational_hypers(False)
xb, yb, fb = h.train_dataset.tensors
g = GraphedELBOStep(h.mfdgp, h.elbo, xb, yb, fb, lr=1e-3, use_graph=(len(sys.argv) > 1))
for i in range(30):
    g.step(); g.stream.synchronize()
    infos = [F.check_info(l._info) for l in h.mfdgp._layers()]
    L1 = h.mfdgp.hidden_layer_1
    print(i, float(g.loss), infos, float(L1.variational_strategy.zf.abs().max()),
          [float(v) for v in __import__("mobocmf_amd.gp", fromlist=["x"]).pack_hypers(L1.covar_module, 1)][:6])
    if any(infos): break
