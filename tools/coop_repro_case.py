import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mobocmf_amd.mlls import VariationalELBOMF
from mobocmf_amd.util import synthetic
from mobocmf_amd.util.coop_step import CoopELBOStep
DEV = "cuda"
cases = [tuple(int(v) for v in sys.argv[i:i + 6]) for i in range(1, len(sys.argv), 6)]
for (d, L, M, N, S, seed) in cases:
  for top, wgs in ((0.25, 0),):
    rng = np.random.default_rng(1)
    prob = synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, seed=seed, top_fraction=top)
    perm = rng.permutation(N)
    tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    x, y, fid = tc(prob["x"])[perm], tc(prob["y"])[perm], tc(prob["fid"])[perm]
    eps = [None] + [tc(e).reshape(N, S)[perm].reshape(-1) for e in prob["eps"][1:]]
    ma = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    mb = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    ma.set_check_pd(False)
    rows = [int((fid >= l).sum()) for l in range(L)]
    order = torch.argsort(fid, descending=True, stable=True)
    xo, yo, fo = x[order].to(DEV), y[order][:, None].to(DEV), fid[order][:, None].to(DEV)
    eo = [None if e is None else e.reshape(N, S)[order][:rows[l]].reshape(-1).contiguous().to(DEV) for l, e in enumerate(eps)]
    e_ref, skl_ref = VariationalELBOMF(ma, N, L)(ma(xo, eps=eo, rows=rows), yo.T, fo)
    (-e_ref).backward()
    step = CoopELBOStep([mb], [N], [x.to(DEV)], [y.to(DEV)], [fid.to(DEV)], lr=1e-3,
                        fixed_eps=[[None if e is None else e.to(DEV) for e in eps]], want_grad=True, force=True)
    step.wgs_per_model = wgs
    grads = step.gradients()[0]
    step.check()
    worst = []
    for (na, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        if pa.grad is None: continue
        ga = torch.tril(pa.grad) if (pa.dim() == 2 and pa.shape[0] == pa.shape[1] == M) else pa.grad
        sc = float(ga.abs().max())
        if sc > 0:
            worst.append((float((grads[pb] - ga).abs().max()) / sc, na))
    worst.sort(reverse=True)
    print("d%d L%d M%d N%d S%d" % (d, L, M, N, S), "top %.2f wgs %2d (used %d) rows %s: elbo rel %.2e | worst grads %s" % (top, wgs, step.wgs_used, rows, abs(float(step.losses[0][0]) - float(e_ref)) / abs(float(e_ref)), [(("%.1e" % v), n.split(".")[-1] + "@" + n.split(".")[0][-1]) for v, n in worst[:3]]), flush=True)
