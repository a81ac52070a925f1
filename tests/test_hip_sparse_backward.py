"""Zero-gradient column blocks of the layer backward (mobocmf_tuning.sparse_backward, include/mobocmf_hip.h).

The reference's ELBO scores every row at its own fidelity only (variational_elbo_mf.py:33-38), so autograd hands the top
layer exact zeros for the rows of every other fidelity; the HIP backward finds the 128-column blocks whose upstream
gradients are all zero on the device and leaves them out of dA, the weighted syrk, da, dK and the Gram backward.  Checked
here: the building blocks directly (the col_activity / k_activity arguments of mobocmf_gemm_f64_epilogue / mobocmf_syrk_weighted_f64), a layer with several zero patterns against its own dense
backward and against the oracle, and a whole model step."""
import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.test_hip_layer import _mk, _oracle, _pack, _close

pytestmark = pytest.mark.gpu
DEV = "cuda"
SENTINEL = 777.0


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _activity(nblk, pattern, seed=0):
    act = torch.zeros(nblk, dtype=torch.int32)
    if pattern == "all":
        act[:] = 1
    elif pattern == "prefix":
        act[:max(1, nblk // 4)] = 1
    elif pattern == "scattered":
        rng = np.random.default_rng(seed)
        act[torch.as_tensor(rng.permutation(nblk)[:max(1, nblk // 3)])] = 1
    elif pattern == "last":
        act[-1] = 1
    else:
        assert pattern == "none"
    return act


@pytest.mark.parametrize("rows", [64, 128], ids=["tiles64x128", "tiles128x128"])
@pytest.mark.parametrize("pattern", ["all", "prefix", "scattered", "last", "none"])
@pytest.mark.parametrize("Mp,Np", [(128, 1024), (512, 8192), (384, 32768), (1024, 16384)])
def test_gemm_skips_inactive_column_blocks(Mp, Np, pattern, rows):
    """A B form: the tiles of an inactive column block are not computed (C keeps what it held), an EPI_DA tile still zeroes
    its row-dot partials; active blocks are bit-identical to the dense launch.  Small-panel kernel, tiled kernel with and
    without row-block pairing, both tile heights."""
    from mobocmf_amd import functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(Mp + Np)
    rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=DEV, generator=g)
    act = _activity(Np // 128, pattern, seed=Np).to(DEV)
    colmask = act.bool().repeat_interleave(128)
    Lw, Up = torch.tril(rnd(Mp, Mp)), torch.triu(rnd(Mp, Mp))
    B, Aaux = rnd(Mp, Np), rnd(Mp, Np)
    avec = rnd(Mp)
    gmu, cgv, gv = (rnd(Np) * colmask for _ in range(3))
    nparts = 2 * max(Np // 128, Np // 16)
    F.set_tile_rows(rows)
    try:
        for tri, T in ((1, Lw), (2, Up)):
            dense = torch.empty(Mp, Np, dtype=torch.float64, device=DEV)
            rd_dense = torch.zeros(nparts, Mp, dtype=torch.float64, device=DEV)
            F.gemm_f64_epilogue(T, B, dense, tri, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=Aaux,
                                rowdot_part=rd_dense)
            plain_dense = torch.empty(Mp, Np, dtype=torch.float64, device=DEV)
            F.gemm_f64_epilogue(T, B, plain_dense, tri, 0)
            F.set_block_activity(act)
            try:
                C = torch.full((Mp, Np), SENTINEL, dtype=torch.float64, device=DEV)
                # NaN in the partial rows the launch writes (one per 16 columns on the small-panel kernel, two per 128 on
                # the tiled one: an inactive tile must zero its own), zeros beyond
                small = Mp <= 512 and (Np // 16) * (Mp // 128) <= 512
                rd = torch.zeros(nparts, Mp, dtype=torch.float64, device=DEV)
                rd[:(Np // 16 if small else Np // 64)] = float("nan")
                F.gemm_f64_epilogue(T, B, C, tri, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=Aaux, rowdot_part=rd)
                P = torch.full((Mp, Np), SENTINEL, dtype=torch.float64, device=DEV)
                F.gemm_f64_epilogue(T, B, P, tri, 0)
            finally:
                F.set_block_activity(None)
            torch.cuda.synchronize()
            assert torch.equal(C[:, colmask], dense[:, colmask])
            assert torch.equal(P[:, colmask], plain_dense[:, colmask])
            assert bool((C[:, ~colmask] == SENTINEL).all()) and bool((P[:, ~colmask] == SENTINEL).all())
            assert bool(torch.isfinite(rd).all())
            assert rel(rd.sum(0), rd_dense.sum(0)) < 1e-14 or float(rd_dense.abs().max()) == 0.0
            assert rel(rd.sum(0), Aaux @ gmu) < 1e-11 or pattern == "none"
            if pattern == "none":
                assert float(rd.abs().max()) == 0.0
    finally:
        F.set_tile_rows(0)


@pytest.mark.parametrize("pattern", ["all", "prefix", "scattered", "last", "none"])
@pytest.mark.parametrize("Mp,Np", [(512, 65536), (640, 8192), (1024, 4096), (128, 2048), (384, 16384)])
def test_syrk_contracts_over_active_blocks_only(Mp, Np, pattern):
    """k-sliced A diag(w) A^T with w zero throughout the inactive 128-column blocks: the slices split the ACTIVE K steps;
    the inactive columns of A are never read (they hold NaN here), the result equals the dense product."""
    from mobocmf_amd import functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(Mp * 3 + Np)
    act = _activity(Np // 128, pattern, seed=Mp).to(DEV)
    colmask = act.bool().repeat_interleave(128)
    A = torch.randn(Mp, Np, dtype=torch.float64, device=DEV, generator=g)
    w = torch.randn(Np, dtype=torch.float64, device=DEV, generator=g) * colmask
    Hd = torch.full((Mp, Mp), float("nan"), dtype=torch.float64, device=DEV)
    F.syrk_weighted(A, w, Hd)
    ref = (A * w[None, :]) @ A.T
    Ap = A.clone()
    Ap[:, ~colmask] = float("nan")
    H = torch.full((Mp, Mp), float("nan"), dtype=torch.float64, device=DEV)
    F.set_block_activity(act)
    try:
        F.syrk_weighted(Ap, w, H)
    finally:
        F.set_block_activity(None)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(H).all())
    assert torch.equal(H, H.T)
    if pattern == "none":
        assert float(H.abs().max()) == 0.0
    else:
        assert rel(H, ref) < 1e-12
        assert rel(H, Hd) < 1e-12


LAYER_CASES = [
    # kind, d, M, nbase, xdiv, branch
    (1, 5, 130, 400, 4, 0),          # 13 column blocks, generic replica path
    (1, 3, 64, 50, 25, 0),           # replica runs of 25 straddle block boundaries (per-column masks)
    (0, 4, 200, 1500, 1, 0),         # first layer, want_dx
    (1, 8, 400, 1200, 8, 0),         # 8 replicas in registers; k-sliced DUAL syrk launch (M > 384)
    (1, 2, 16, 64, 16, 1),           # 16 replicas, test branch
    (1, 8, 512, 1100, 8, 0),         # 69 blocks: tiled kernels on 64-row tiles
]


def _weights(nbase, xdiv, pattern, rng):
    """Upstream gradients of (mean, var) per column; zero for the base rows the pattern leaves out."""
    on = np.zeros(nbase, dtype=bool)
    if pattern == "all":
        on[:] = True
    elif pattern == "top_quarter":
        on[:max(1, nbase // 4)] = True
    elif pattern == "scattered":
        on[rng.permutation(nbase)[:max(1, nbase // 50)]] = True
    elif pattern == "mean_only_tail":
        on[-3:] = True
    cols = np.repeat(on, xdiv)
    wm = rng.standard_normal(nbase * xdiv) * cols
    wv = rng.standard_normal(nbase * xdiv) * cols
    if pattern == "mean_only_tail":
        wv[:] = 0.0
    return torch.tensor(wm), torch.tensor(wv)


def _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, wkl, sparse):
    dev = torch.device(DEV)
    g = lambda t, rg=True: None if t is None else t.detach().to(dev).requires_grad_(rg)
    xg, fg, Zg, zfg, mg, LSg = g(x), g(f), g(Zx, False), g(zf), g(m), g(L_S)
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(dev).requires_grad_(True)
    F.set_sparse_backward(sparse)
    try:
        mean, var, kl = F.layer_forward(xg, fg, Zg, zfg, hg, mg, LSg, kind, xdiv=xdiv, branch=branch, want_dx=True)
        ((wm.to(dev) * mean).sum() + (wv.to(dev) * var).sum() + wkl * kl).backward()
        torch.cuda.synchronize()
    finally:
        F.set_sparse_backward(True)
    out = {"g_m": mg.grad, "g_LS": LSg.grad, "g_hyp": hg.grad, "g_x": xg.grad}
    if kind == 1:
        out["g_f"], out["g_zf"] = fg.grad, zfg.grad
    return out


@pytest.mark.parametrize("pattern", ["all", "top_quarter", "scattered", "mean_only_tail", "none"])
@pytest.mark.parametrize("kind,d,M,nbase,xdiv,branch", LAYER_CASES)
def test_layer_backward_sparse_equals_dense(kind, d, M, nbase, xdiv, branch, pattern):
    """Every gradient of a layer with the block skipping on equals the dense backward's -- also when NO column has upstream
    gradient and only the KL term feeds the parameters.  Same kernels, but the syrk sums its active K steps in other slices:
    H differs in the last bit, and the hyper-parameter gradients (sums with cancellation) by ~1e-12 of their largest entry;
    the gate is 1e-10 (the oracle comparison of the dense backward runs at 1e-7)."""
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=kind * 100 + M + xdiv)
    rng = np.random.default_rng(11)
    wm, wv = _weights(nbase, xdiv, pattern, rng)
    sp = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, True)
    de = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, False)
    for k in de:
        assert bool(torch.isfinite(sp[k]).all()), k
        scale = max(float(de[k].abs().max()), 1e-300)
        err = float((sp[k] - de[k]).abs().max()) / scale
        assert err < 1e-10, (k, err)
    if pattern in ("none", "mean_only_tail"):
        # the skipped rows get exact zeros (not "whatever the workspace held")
        on = np.repeat((wm != 0).numpy().reshape(nbase, xdiv).any(1), 1)
        assert float(sp["g_x"][torch.as_tensor(~on)].abs().max()) == 0.0
        if kind == 1:
            off_cols = torch.as_tensor(np.repeat(~on, xdiv))
            assert float(sp["g_f"][off_cols].abs().max()) == 0.0


@pytest.mark.parametrize("kind,d,M,nbase,xdiv,branch", [LAYER_CASES[0], LAYER_CASES[1], LAYER_CASES[2]])
def test_layer_backward_sparse_matches_oracle(kind, d, M, nbase, xdiv, branch):
    """... and the oracle's autograd over the same zero-padded upstream gradients."""
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=kind * 100 + M + xdiv)
    rng = np.random.default_rng(13)
    wm, wv = _weights(nbase, xdiv, "top_quarter", rng)
    sp = _layer_grads(F, kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, wm, wv, 0.37, True)
    _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, [wm, wv, torch.tensor(0.37)])
    _close(sp["g_m"], m.grad, 1e-7, "g_m")
    _close(sp["g_LS"], torch.tril(L_S.grad), 1e-7, "g_LS")
    _close(sp["g_hyp"], _pack(kind, {k: v.grad for k, v in hyp.items()}), 1e-7, "g_hyp")
    _close(sp["g_x"], x.grad, 1e-7, "g_x")
    if kind == 1:
        _close(sp["g_f"], f.grad, 1e-7, "g_f")
        _close(sp["g_zf"], zf.grad, 1e-7, "g_zf")


@pytest.mark.parametrize("L,N,S", [(2, 2000, 4), (3, 1536, 8)])
def test_model_step_sparse_equals_dense(L, N, S):
    """A whole ELBO step (synthetic problem: the first N/4 rows are the top fidelity): every parameter gradient with the
    skipping on equals the dense step's -- up to the last-bit difference of H (see above) carried through the chain backward,
    i.e. ~cond(K_mm) * eps (96 inducing points in 3-D: measured 3e-9; tests/test_hip_model.py discusses the same factor)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd import functional as F
    prob = synthetic.make_problem(d=3, L=L, M=96, N=N, S=S, seed=5)
    grads = {}
    for sparse in (True, False):
        model = synthetic.model_from_problem(prob, device=DEV)
        elbo = VariationalELBOMF(model, N, L)
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=DEV)
        eps = [None] + [t(e) for e in prob["eps"][1:]]
        F.set_sparse_backward(sparse)
        try:
            out = model(t(prob["x"]), eps=eps)
            res = elbo(out, t(prob["y"])[None, :], t(prob["fid"])[:, None])
            (-res[0]).backward()
            torch.cuda.synchronize()
        finally:
            F.set_sparse_backward(True)
        model.clear_kl_cache()
        grads[sparse] = [(n, p.grad.clone()) for n, p in model.named_parameters() if p.grad is not None]
    assert len(grads[True]) == len(grads[False]) > 0
    for (n, a), (_, b) in zip(grads[True], grads[False]):
        scale = max(float(b.abs().max()), 1e-300)
        assert float((a - b).abs().max()) / scale < 1e-6, n


# ---------------------------------------------------------------------------------------------------------------------
# Dead rows: layer l evaluated on the rows of fidelity >= l only (MFDGP.forward(rows=...), GraphedELBOStep(prune_rows=True))

def _elbo_and_grads(prob, L, N, S, rows, sparse=True, **model_kwargs):
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd import functional as F
    model = synthetic.model_from_problem(prob, device=DEV, **model_kwargs)
    elbo = VariationalELBOMF(model, N, L)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=DEV)
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    if rows is not None:
        eps = [None] + [e.reshape(N, S)[:rows[l + 1]].reshape(-1).contiguous() for l, e in enumerate(eps[1:])]
    F.set_sparse_backward(sparse)
    try:
        out = model(t(prob["x"]), eps=eps, rows=rows)
        res = elbo(out, t(prob["y"])[None, :], t(prob["fid"])[:, None])
        (-res[0]).backward()
        torch.cuda.synchronize()
    finally:
        F.set_sparse_backward(True)
    model.clear_kl_cache()
    return float(res[0]), float(res[1]), [(n, p.grad.clone()) for n, p in model.named_parameters() if p.grad is not None], out


@pytest.mark.parametrize("L,N,S", [(2, 2000, 4), (3, 1536, 8), (2, 64, 10)])
def test_pruned_forward_gives_the_same_elbo_and_gradients(L, N, S):
    """MFDGP.forward(rows=[#rows of fidelity >= l]) on a fidelity-ordered batch: the layers above 0 run on a prefix of the
    rows, the ELBO and every parameter gradient equal those of the reference layout (every layer at every row, dense
    backward), and the pruned layers' moments equal the full ones on the rows they cover."""
    prob = synthetic.make_problem(d=3, L=L, M=min(96, N), N=N, S=S, seed=5)
    fid = np.asarray(prob["fid"])
    assert bool((np.diff(fid) <= 0).all())                 # synthetic batches are ordered by descending fidelity
    rows = [int((fid >= l).sum()) for l in range(L)]
    assert rows[0] == N and rows[-1] < N
    e0, k0, g0, out0 = _elbo_and_grads(prob, L, N, S, None, sparse=False)
    e1, k1, g1, out1 = _elbo_and_grads(prob, L, N, S, rows)
    assert abs(e1 - e0) <= 1e-10 * abs(e0) and abs(k1 - k0) <= 1e-13 * abs(k0)
    for l in range(L):
        n = rows[l] * (1 if l == 0 else S)
        assert out1[l].mean.numel() == n and out1[l].batch_rows == rows[l]
        # another N' may take another kernel / tile height (another summation order) through L^-1: ~cond(K_mm) * eps
        assert rel(out1[l].mean.reshape(-1), out0[l].mean.reshape(-1)[:n]) < 1e-9
        assert rel(out1[l].variance.reshape(-1), out0[l].variance.reshape(-1)[:n]) < 1e-9
    assert len(g0) == len(g1) > 0
    for (n, a), (_, b) in zip(g1, g0):
        scale = max(float(b.abs().max()), 1e-300)
        assert float((a - b).abs().max()) / scale < 1e-6, n      # ~cond(K_mm) * eps, as above


def test_pruned_forward_with_the_only_highest_fidelity_ablation():
    """use_only_highest_fidelity (mfdgp.py:189-190: the previous layer's output enters as zeros): the zeros are cut to the
    layer's rows as well; same ELBO and gradients as the reference layout."""
    L, N, S = 2, 1200, 4
    prob = synthetic.make_problem(d=3, L=L, M=64, N=N, S=S, seed=8)
    fid = np.asarray(prob["fid"])
    rows = [int((fid >= l).sum()) for l in range(L)]
    e0, k0, g0, _ = _elbo_and_grads(prob, L, N, S, None, sparse=False, use_only_highest_fidelity=True)
    e1, k1, g1, out1 = _elbo_and_grads(prob, L, N, S, rows, use_only_highest_fidelity=True)
    assert out1[1].mean.numel() == rows[1] * S
    assert abs(e1 - e0) <= 1e-10 * abs(e0) and abs(k1 - k0) <= 1e-13 * abs(k0)
    for (n, a), (_, b) in zip(g1, g0):
        assert float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300) < 1e-6, n


def test_pruned_forward_rejects_bad_row_counts():
    prob = synthetic.make_problem(d=2, L=2, M=16, N=64, S=2, seed=1)
    model = synthetic.model_from_problem(prob, device=DEV)
    x = torch.as_tensor(prob["x"], dtype=torch.float64, device=DEV)
    for bad in ([64], [32, 64], [65, 10], [64, 0]):
        with pytest.raises(ValueError):
            model(x, rows=bad)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("L,N,S", [(2, 600, 4), (3, 512, 4)])
def test_graphed_step_orders_the_batch_and_prunes(L, N, S, use_graph):
    """GraphedELBOStep on a SHUFFLED batch: with prune_rows it orders the rows by descending fidelity once (explicit eps follow
    their rows) and prunes; parameters after three Adam steps and the reported loss equal those of the unpruned step."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    prob = synthetic.make_problem(d=3, L=L, M=48, N=N, S=S, seed=9)
    perm = torch.as_tensor(np.random.default_rng(3).permutation(N), device=DEV)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=DEV)
    x, y, fid = t(prob["x"])[perm].contiguous(), t(prob["y"])[perm, None].contiguous(), t(prob["fid"])[perm, None].contiguous()
    eps = [None] + [t(e).reshape(N, S)[perm].reshape(-1).contiguous() for e in prob["eps"][1:]]
    finals, losses = [], []
    for prune in (False, True):
        model = synthetic.model_from_problem(prob, device=DEV)
        step = GraphedELBOStep(model, VariationalELBOMF(model, N, L), x, y, fid, lr=1e-2, use_graph=use_graph,
                               fixed_eps=eps, prune_rows=prune)
        assert (step.layer_rows is not None) == prune
        if prune:
            assert bool((step.fid.reshape(-1)[:-1] >= step.fid.reshape(-1)[1:]).all())
            assert step.layer_rows == [int((fid >= l).sum()) for l in range(L)]
        ls = []
        for _ in range(3):
            loss, _ = step.step()
            step.stream.synchronize()
            ls.append(float(loss))
        step.check()
        finals.append([p.detach().clone() for p in model.parameters()])
        losses.append(ls)
        step.retire()
    for a, b in zip(losses[0], losses[1]):
        assert abs(a - b) <= 1e-9 * abs(a)
    for a, b in zip(finals[0], finals[1]):
        assert float((a - b).abs().max()) <= 1e-7 * max(float(a.abs().max()), 1e-30)
