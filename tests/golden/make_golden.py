"""Generates tests/golden/*.npz.  Run in the BUILD container only (needs /root/reference for the
helper pins; the GPU box only ever reads the committed .npz files).

Three kinds of vectors:
  ref_moop.npz      outputs of the reference's own MOOP (mobocmf/util/moop.py: numpy/scipy only, importable here) on
                    seeded inputs -- Pareto masks, front summaries, feasible grids, constrained optima, whole
                    extractions: these pin mobocmf_amd/util/moop.py against the REAL reference.
  ref_helpers.npz   outputs of the pieces of the reference that ARE importable here
                    (mobocmf.util.util.compute_dist / triu_indices, the Forrester test functions,
                    the nearest-same-fidelity init loop of mfdgp.py:290-317 executed with the
                    reference's own compute_dist) -- these pin the oracle's init heuristics.
  oracle_*.npz      inputs + outputs of oracle/mfdgp_oracle.py (the reference's MFDGP itself cannot
                    run: gpytorch is absent) -- these freeze the oracle so the HIP parity tests on
                    the GPU box and any later oracle edit are checked against the same numbers.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mobocmf_amd.util import synthetic  # noqa: E402
from oracle import mfdgp_oracle as O  # noqa: E402
from tests.helpers import oracle_state, state_leaves, to_t  # noqa: E402


def ref_helpers():
    sys.path.insert(0, "/root/reference")
    from mobocmf.test_functions.forrester import forrester_mf0, forrester_mf1
    from mobocmf.util.util import compute_dist, triu_indices
    out = {}
    for seed, (n, d) in enumerate([(6, 2), (9, 1), (12, 3)]):
        torch.manual_seed(seed)
        x = torch.rand(n, d, dtype=torch.float64)
        D = compute_dist(x)
        out[f"med_x_{seed}"] = x.numpy()
        out[f"med_dist_{seed}"] = D.numpy()
        out[f"med_ls_{seed}"] = torch.sqrt(torch.median(D[triu_indices(n, 1)])).numpy()
    x0 = np.linspace(0, 1.0, 12).reshape(12, 1)
    x1 = np.array([0.1, 0.3, 0.5, 0.7]).reshape(4, 1)
    out["forr_lo"] = forrester_mf0(x0)
    out["forr_hi"] = forrester_mf1(x1)
    # nearest same-fidelity init values, the reference's loop (mfdgp.py:304-307) with its compute_dist
    xs, ys, fid = synthetic.forrester_problem(0)
    x_t, y_t, f_t = torch.from_numpy(xs), torch.from_numpy(ys)[:, None], torch.from_numpy(fid)[:, None]
    for layer in (0, 1):
        vals = torch.zeros(x_t.shape[0], dtype=torch.float64)
        for i in range(x_t.shape[0]):
            tmp = torch.cat((x_t[f_t[:, 0] == layer, :], x_t[i:i + 1, :]), 0)
            sel = torch.argmin(compute_dist(tmp)[0:tmp.shape[0] - 1, tmp.shape[0] - 1])
            vals[i] = y_t[f_t[:, 0] == layer, :][sel]
        out[f"init_vals_{layer}"] = vals.numpy()
    np.savez(os.path.join(HERE, "ref_helpers.npz"), **out)


def ref_moop():
    """Outputs of the reference's own MOOP (numpy/scipy only, importable here) on seeded inputs: pins
    mobocmf_amd/util/moop.py."""
    sys.path.insert(0, "/root/reference")
    from mobocmf.util.moop import MOOP as RefMOOP
    from tests.helpers import moop_callable
    out = {}
    rng = np.random.default_rng(123)
    for c, (n, k) in enumerate([(1, 2), (40, 1), (300, 2), (300, 3), (2000, 4), (5000, 2)]):
        pts = rng.standard_normal((n, k))
        if n > 10:
            pts[5] = pts[3]                                   # exact duplicate
            pts[7] = pts[3] + np.eye(k)[-1]                   # weakly dominated (ties in k-1 coordinates)
            pts[::7] = np.round(pts[::7], 1)                  # ties in single coordinates
        out[f"front_pts_{c}"] = pts
        out[f"front_mask_{c}"] = RefMOOP.compute_pareto_front(pts.copy())
        out[f"front_mask_sorted_{c}"] = RefMOOP([], [], k).obtain_indices_pareto(pts.copy())
    for c, (n, k, size) in enumerate([(120, 2, 10), (400, 3, 50), (30, 2, 50), (200, 2, 2)]):
        ps, pf = rng.uniform(size=(n, 4)), rng.standard_normal((n, k))
        a, b = RefMOOP([], [], 4).compute_pareto_front_and_set_summary_y_space(ps, pf, size)
        out[f"sum_set_{c}"], out[f"sum_front_{c}"], out[f"sum_size_{c}"] = ps, pf, np.array(size)
        out[f"sum_out_set_{c}"], out[f"sum_out_front_{c}"] = a, b
    # feasibility + constrained optimum + the whole extraction on analytic "samples" (d = 2)
    cons = [moop_callable("lin", [-0.3, 1.0, 0.0]), moop_callable("lin", [0.8, 0.0, -1.0])]      # x0 >= 0.3, x1 <= 0.8
    objs = [moop_callable("quad", [0.1, 0.2]), moop_callable("quad", [0.9, 0.9]), ]
    grid = rng.uniform(size=(500, 2))
    m = RefMOOP(objs, cons, 2, feasible_values=np.zeros(2))
    out["feas_grid"] = grid
    out["feas_out"] = m.find_feasible_grid(cons, grid, feasible_values=np.zeros(2))
    hard = [moop_callable("lin", [-1.5, 1.0, 0.0]), moop_callable("lin", [-0.2, 0.0, -1.0])]     # infeasible on [0,1]^2
    out["feas_none"] = np.array(m.find_feasible_grid(hard, grid, feasible_values=np.zeros(2)) is None)
    out["feas_neg_out"] = m.find_feasible_grid(hard, grid, feasible_values=np.zeros(2), allow_negative_constraints=True)
    fg = out["feas_out"]
    for j, obj in enumerate(objs):
        out[f"opt_x_{j}"] = m.optimize_obj_globally(obj, cons, obj(fg), fg)
    for c, (objs_c, size) in enumerate([(objs, None), (objs, 7),
                                        ([moop_callable("wave", [0.3, 1.1]), moop_callable("wave", [2.0, 0.4])], 12)]):
        np.random.seed(77 + c)
        inputs = rng.uniform(size=(9, 2))
        res = RefMOOP(objs_c, cons, 2, grid_size=300, pareto_set_size=size,
                      feasible_values=np.zeros(2)).compute_pareto_solution_from_samples(inputs)
        out[f"sol_inputs_{c}"] = inputs
        out[f"sol_set_{c}"], out[f"sol_front_{c}"] = res[0].numpy(), res[1].numpy()
    np.savez(os.path.join(HERE, "ref_moop.npz"), **out)


def permuted_problem(prob, seed=5):
    """The same problem with its DATA rows shuffled (Z_x keeps the original order): what one batch of the reference's
    DataLoader(shuffle=True) looks like at M = N (blackbox_mfdgp_fitter.py:35) -- the general branch, not torch.equal's."""
    perm = np.random.default_rng(seed).permutation(prob["x"].shape[0])
    if (perm == np.arange(perm.size)).all():
        perm = np.roll(perm, 1)
    out = dict(prob)
    out.update(x=prob["x"][perm], y=prob["y"][perm], fid=prob["fid"][perm], perm=perm)
    return out


def oracle_case(name, prob, S, T=6, shortcut=True):
    st = oracle_state(prob, requires_grad=True)
    x, y, fid = to_t(prob["x"]), to_t(prob["y"]), to_t(prob["fid"])
    eps = [None] + [to_t(e) for e in prob["eps"][1:]]
    e, skl = O.elbo(st, x, y, fid, eps=eps, S=S, shortcut=shortcut)
    leaves = state_leaves(st)
    grads = torch.autograd.grad(e, leaves)
    out = {"elbo": e.detach().numpy(), "scaled_kl": skl.detach().numpy()}
    for i, g in enumerate(grads):
        out[f"grad_{i}"] = g.numpy()
    with torch.no_grad():
        outs = O.model_forward(st, x, eps=eps, S=S, shortcut=shortcut)
        for l, (mu, v) in enumerate(outs):
            out[f"mean_{l}"], out[f"var_{l}"] = mu.numpy(), v.numpy()
        X = to_t(np.random.default_rng(123).random((T, prob["d"])))
        out["acq_X"] = X.numpy()
        for f in range(prob["L"]):
            for flag in (True, False):
                mus, vs = O.predict_for_acquisition(st, X, f, S, training=flag)
                tag = "train" if flag else "eval"
                out[f"acq_mu_{f}_{tag}"], out[f"acq_var_{f}_{tag}"] = mus.numpy(), vs.numpy()
    np.savez(os.path.join(HERE, f"oracle_{name}.npz"), **out)


def forrester_state_problem(output):
    """C1 as a synthetic-style problem dict: Forrester data + the reference's init heuristics."""
    x, y, fid = synthetic.forrester_problem(output)
    prob = synthetic.make_problem(d=1, L=2, M=16, N=16, S=4, seed=output)
    prob.update(x=x, y=y, fid=fid, Zx=x.copy())
    xt, yt, ft = to_t(x), to_t(y), to_t(fid)
    ls_lo = float(O.median_lengthscale(xt[ft == 0]))
    ls_hi = float(O.median_lengthscale(xt[ft == 1]))
    y_high_std = float(np.std(y[fid == 1]))
    m0 = O.nearest_same_fidelity_values(xt, yt, ft, xt, 0).numpy()
    m1 = O.nearest_same_fidelity_values(xt, yt, ft, xt, 1).numpy()
    l0, l1 = prob["layers"]
    l0["hyp"] = {"ls": np.array([ls_lo]), "alpha": np.array(1.0)}
    l0["m"], l0["L_S"] = m0, 1e-4 * np.eye(16)
    l1["hyp"] = {"ls1": np.array([10 * ls_hi]), "a1": np.array(1.0), "lsf": np.array(1.0), "af": np.array(1.0),
                 "nu": np.array(1.0), "ls2": np.array([ls_hi]), "a2": np.array(0.01)}
    Zraw = np.concatenate([x, m1[:, None]], 1)
    Kinit = O.gram({k: to_t(v) for k, v in l1["hyp"].items()}, to_t(Zraw), to_t(Zraw)).numpy()
    S1 = Kinit * (1e-2 * y_high_std ** 2) ** 2 + 1e-12 * np.eye(16)
    l1["m"], l1["L_S"] = m1, np.linalg.cholesky(S1)
    prob["noise"] = [np.array(1e-6), np.array(1e-2 * y_high_std)]
    return prob


if __name__ == "__main__":
    if os.path.isdir("/root/reference"):
        ref_helpers()
        ref_moop()
    for o in range(3):
        oracle_case(f"C1_forrester_out{o}", forrester_state_problem(o), S=4)
        # the branch the reference's training executes: shuffled batch rows, no equal-inputs shortcut
        oracle_case(f"C1_forrester_out{o}_general", permuted_problem(forrester_state_problem(o)), S=4, shortcut=False)
    for seed in range(3):
        oracle_case(f"small2d_seed{seed}", synthetic.make_problem(d=2, L=2, M=8, N=12, S=3, seed=seed), S=3)
    oracle_case("small3layer", synthetic.make_problem(d=3, L=3, M=10, N=16, S=2, seed=7), S=2)
    oracle_case("C2_seed0", synthetic.make_problem(**{k: v for k, v in synthetic.CONFIGS["C2"].items() if k != "outputs"},
                                                  seed=0), S=8, T=16)
    print("golden written")
