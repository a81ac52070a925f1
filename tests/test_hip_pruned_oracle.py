"""The configuration bench.py times -- GraphedELBOStep(prune_rows=True) on HIP-graph replay, zero-gradient column blocks
skipped in every layer backward -- against the ORACLE (which evaluates every layer at every row and masks afterwards, as the
reference does: variational_elbo_mf.py:33-38, mfdgp.py:174-196), not against the package's own dense step.

* a fixed-seed slice of the randomised campaign tools/fuzz_sparse_backward.py: 40 layer cases (random shapes, replica counts,
  zero patterns of the upstream gradients) and 6 whole problems (2-3 fidelities, S = 1...8, ragged sizes);
* the captured, pruned step's Adam trajectory vs ``oracle.elbo_step``.
Full-size (C3 seeds 0-2 x {objective, constraint}, C5) pruned cases live in test_hip_fullsize.py."""
import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.test_hip_layer import _close, _mk, _oracle, _pack
from tests.test_hip_model import _model_param_for, _raw_from_model, rel
from tests.test_hip_sparse_backward import _layer_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fuzz_layer_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        kind = int(rng.integers(0, 2))
        d = int(rng.choice([1, 2, 3, 5, 8, 12]))
        M = int(rng.choice([7, 16, 40, 100, 130, 200, 384, 400]))
        xdiv = 1 if kind == 0 else int(rng.choice([1, 2, 3, 4, 8, 16, 25]))
        nbase = int(rng.integers(3, 1 + 3000 // xdiv))
        cases.append((kind, d, M, nbase, xdiv, int(rng.integers(0, 2)), int(rng.integers(0, 5)), int(rng.integers(1 << 30))))
    return cases


LAYER_FUZZ = _fuzz_layer_cases(40, seed=2026)


@pytest.mark.parametrize("kind,d,M,nbase,xdiv,branch,mode,seed", LAYER_FUZZ,
                         ids=["k%d_d%d_M%d_n%d_x%d_b%d_z%d" % c[:7] for c in LAYER_FUZZ])
def test_fuzz_layer_backward_with_block_skipping_matches_oracle(kind, d, M, nbase, xdiv, branch, mode, seed):
    """Layer backward with the zero-gradient column blocks skipped vs the oracle's autograd through the dense layer over the
    same upstream gradients.  Zero patterns: a leading share of the base rows (the dead-row shape), 1/40 scattered, a tail,
    all rows, none (only the KL feeds the parameters).  Gate 1e-6 relative to the largest entry (d = 1 cases: cond(K_mm) is
    large, either implementation carries ~cond * eps); the north star's tolerance is 1e-4."""
    from mobocmf_amd import functional as F
    rng = np.random.default_rng(seed)
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=seed)
    on = np.zeros(nbase, dtype=bool)
    if mode == 0:
        on[:max(1, nbase // int(rng.integers(2, 9)))] = True
    elif mode == 1:
        on[rng.permutation(nbase)[:max(1, nbase // 40)]] = True
    elif mode == 2:
        on[int(rng.integers(0, nbase)):] = True
    elif mode == 3:
        on[:] = True
    cols = np.repeat(on, xdiv)
    wm = torch.tensor(rng.standard_normal(nbase * xdiv) * cols)
    wv = torch.tensor(rng.standard_normal(nbase * xdiv) * cols * (rng.random() < 0.8))
    sp = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, True)
    _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, [wm, wv, torch.tensor(0.37)])
    tol = 1e-6
    _close(sp["g_m"], m.grad, tol, "g_m")
    _close(sp["g_LS"], torch.tril(L_S.grad), tol, "g_LS")
    _close(sp["g_hyp"], _pack(kind, {k: v.grad for k, v in hyp.items()}), tol, "g_hyp")
    _close(sp["g_x"], x.grad, tol, "g_x")
    if kind == 1:
        _close(sp["g_f"], f.grad, tol, "g_f")
        _close(sp["g_zf"], zf.grad, tol, "g_zf")


def _fuzz_problems(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        L = int(rng.choice([2, 3]))
        S = int(rng.choice([1, 2, 4, 8]))
        N = int(rng.integers(40, 1500))
        M = int(min(N, rng.choice([16, 48, 96, 160])))
        d = int(rng.choice([2, 4, 8]))
        fid = np.asarray(synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, seed=0)["fid"])
        if int((fid >= L - 1).sum()) >= 1:
            out.append((L, S, N, M, d, int(rng.integers(1 << 30))))
    return out


MODEL_FUZZ = _fuzz_problems(6, seed=4)


def _pruned_hip(model, prob, S, shuffle_seed=None):
    """ELBO of the model on the fidelity-ordered prefix layout, built the way GraphedELBOStep.__init__ builds it."""
    from mobocmf_amd.mlls import VariationalELBOMF
    L, N = prob["L"], prob["x"].shape[0]
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=DEV)
    x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    fidv = fid.reshape(-1)
    rows = [int((fidv >= l).sum()) for l in range(L)]
    order = torch.argsort(fidv, descending=True, stable=True)
    x, y, fid = x[order].contiguous(), y[order].contiguous(), fid[order].contiguous()
    eps = [None if e is None else e.reshape(N, S)[order][:rows[l]].reshape(-1).contiguous() for l, e in enumerate(eps)]
    out = model(x, eps=eps, rows=rows)
    return VariationalELBOMF(model, N, L)(out, y.T, fid), out, rows


@pytest.mark.parametrize("L,S,N,M,d,seed", MODEL_FUZZ, ids=["L%d_S%d_N%d_M%d_d%d" % c[:5] for c in MODEL_FUZZ])
def test_fuzz_pruned_model_matches_oracle(L, S, N, M, d, seed):
    """Whole problems: the pruned forward / backward (dead rows cut, blocks skipped) vs the oracle's dense evaluation: ELBO,
    scaled KL, each layer's moments on its prefix, every raw-parameter gradient."""
    prob = synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, seed=seed)
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    model.set_check_pd(False)
    raw = _raw_from_model(model, L)
    tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    x, y, fid = tc(prob["x"]), tc(prob["y"]), tc(prob["fid"])
    eps = [None] + [tc(e) for e in prob["eps"][1:]]
    st = O.state_from_raw(raw)
    e_o, skl_o = O.elbo(st, x, y, fid, eps=eps, S=S)
    (-e_o).backward()
    with torch.no_grad():
        outs_o = O.model_forward(st, x, eps=eps, S=S)
    (e, skl), out, rows = _pruned_hip(model, prob, S)
    (-e).backward()
    assert rows[-1] < N
    assert rel(e, e_o) < 1e-8 and rel(skl, skl_o) < 1e-8
    for l in range(L):
        n = out[l].mean.numel()
        assert n == rows[l] * (1 if l == 0 else S)
        assert rel(out[l].mean.reshape(-1), outs_o[l][0].reshape(-1)[:n]) < 1e-7
        assert rel(out[l].variance.reshape(-1), outs_o[l][1].reshape(-1)[:n]) < 1e-6
    for l in range(L):
        for key, tt in raw["layers"][l].items():
            p = _model_param_for(model, l, key)
            gref = tt.grad if key != "L_S" else torch.tril(tt.grad)
            assert rel(p.grad.reshape(gref.shape), gref) < 1e-5, (l, key, rel(p.grad.reshape(gref.shape), gref))
        assert rel(getattr(model, f"hidden_layer_likelihood_{l}").raw_noise.grad.reshape(()), raw["raw_noise"][l].grad) < 1e-5


TRAJ = [dict(d=2, L=2, M=128, N=512, S=8, seed=0),          # C2
        dict(d=8, L=2, M=256, N=2048, S=4, seed=1),         # C3's shape, scaled to what the oracle steps through on the CPU
        dict(d=3, L=3, M=48, N=400, S=2, seed=5)]


@pytest.mark.parametrize("cfg", TRAJ, ids=["C2", "d8_M256_N2048", "three_fidelities"])
def test_graphed_pruned_step_trajectory_matches_oracle(cfg):
    """GraphedELBOStep(prune_rows=True, use_graph=True) -- zero_grad + pruned forward + fused ELBO + backward + the fused
    multi-tensor Adam, replayed from ONE HIP graph on a SHUFFLED batch -- vs the oracle's step
    (blackbox_mfdgp_fitter.py:161-171 with torch.optim.Adam) on the unordered dense batch: the loss of each of 3 steps and
    every parameter afterwards.  C2's gate is the north star's (cond(K_mm + 1e-6 I) ~ 1e9)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    prob = synthetic.make_problem(**cfg)
    S, L, N = cfg["S"], cfg["L"], cfg["N"]
    ill = cfg["M"] == 128 and cfg["d"] == 2
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    raw = _raw_from_model(model, L)
    perm = np.random.default_rng(3).permutation(N)
    tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    xo, yo, fo = tc(prob["x"])[perm], tc(prob["y"])[perm], tc(prob["fid"])[perm]
    epso = [None] + [tc(e).reshape(N, S)[perm].reshape(-1) for e in prob["eps"][1:]]
    step = GraphedELBOStep(model, VariationalELBOMF(model, N, L), xo.to(DEV), yo[:, None].to(DEV), fo[:, None].to(DEV),
                           lr=1e-2, use_graph=True, fixed_eps=[None if e is None else e.to(DEV) for e in epso],
                           prune_rows=True)
    assert step.graph is not None and step.layer_rows == [int((fo >= l).sum()) for l in range(L)]
    opt = torch.optim.Adam(O.flatten_raw(raw), lr=1e-2)
    for k in range(3):
        lo, klo = O.elbo_step(raw, opt, xo, yo, fo, epso, S, ref_equiv=False)
        loss, kl = step.step()
        step.stream.synchronize()
        assert rel(loss, lo) < (1e-4 if ill else 1e-7), (k, rel(loss, lo))
        assert rel(kl, klo) < (1e-4 if ill else 1e-7), k
    step.check()
    for l in range(L):
        for key, tt in raw["layers"][l].items():
            p = _model_param_for(model, l, key)
            assert rel(p.reshape(tt.shape), tt.detach()) < (1e-3 if ill else 1e-6), (l, key, rel(p.reshape(tt.shape), tt.detach()))
        lk = getattr(model, f"hidden_layer_likelihood_{l}")
        assert rel(lk.raw_noise.reshape(()), raw["raw_noise"][l].detach()) < (1e-3 if ill else 1e-6)
    step.retire()
