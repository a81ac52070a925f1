"""The split layer call (MOBOCMF_PHASE_CHAIN / _PANEL, chain halves on a side stream) must reproduce the fused call:
same ELBO, same gradients, same trajectories -- eagerly, over several iterations (cross-iteration stream hazards), under
HIP-graph capture (fork/join of the side stream inside the capture), for a KL-only backward (CHAIN_ONLY) and for the
acquisition gradient w.r.t. X."""
import pytest
import torch

from mobocmf_amd.util import synthetic
from tests.helpers import to_t
from tests.test_hip_model import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grads(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("cfg", [dict(d=3, L=2, M=20, N=60, S=2, seed=9), dict(d=4, L=3, M=150, N=300, S=3, seed=2)],
                         ids=["L2", "L3_M150"])
def test_split_call_equals_fused_call(cfg):
    from mobocmf_amd.mlls import VariationalELBOMF
    prob = synthetic.make_problem(**cfg)
    t = lambda a: to_t(a).to(DEV)
    x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    out = {}
    for overlap in (False, True):
        model = build_model(prob, S_train=cfg["S"])
        model.set_check_pd(False)
        model.overlap_chains = overlap
        elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        vals = []
        for it in range(3):                       # no synchronisation between iterations
            opt.zero_grad(set_to_none=True)
            res = elbo(model(x, eps=eps), y.T, fid)
            (-res[0]).backward()
            vals.append((res[0].detach().clone(), res[1].detach().clone(), _grads(model).clone()))
            opt.step()
        torch.cuda.synchronize()
        for layer in model._layers():
            assert int(layer._info.item()) == 0
        out[overlap] = vals
    # iteration 0: identical inputs -> identical kernels, only the g_zf / g_hyp partial sums are added in another order;
    # later iterations inherit that rounding through Adam (ill-conditioned toy problems: looser)
    for it, ((e0, k0, g0), (e1, k1, g1)) in enumerate(zip(out[False], out[True])):
        tol = 1e-13 if it == 0 else 1e-8
        assert abs(float(e0 - e1)) <= tol * abs(float(e0)) and abs(float(k0 - k1)) <= tol * abs(float(k0))
        assert _rel(g1, g0) < (1e-12 if it == 0 else 1e-7)


def test_split_call_kl_only_and_data_only_backward():
    from mobocmf_amd.mlls import VariationalELBOMF
    cfg = dict(d=2, L=2, M=12, N=30, S=2, seed=4)
    prob = synthetic.make_problem(**cfg)
    t = lambda a: to_t(a).to(DEV)
    x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
    eps = [None, t(prob["eps"][1])]
    res = {}
    for overlap in (False, True):
        model = build_model(prob, S_train=2)
        model.set_check_pd(False)
        model.overlap_chains = overlap
        elbo = VariationalELBOMF(model, cfg["N"], 2)
        r = elbo(model(x, eps=eps), y.T, fid)
        r[1].backward()                              # KL only: the PANEL halves take no part
        g_kl = _grads(model).clone()
        model.zero_grad(set_to_none=True)
        data = elbo(model(x, eps=eps), y.T, fid, include_kl_term=False)
        data.backward()                              # data term only: no KL gradient reaches the CHAIN halves
        res[overlap] = (g_kl, _grads(model).clone())
    assert _rel(res[True][0], res[False][0]) < 1e-12
    assert _rel(res[True][1], res[False][1]) < 1e-11


def test_graphed_step_with_overlap_equals_serial_eager():
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    cfg = dict(d=3, L=2, M=40, N=130, S=2, seed=5)
    prob = synthetic.make_problem(**cfg)
    t = lambda a: to_t(a).to(DEV)
    traj = []
    for use_graph, overlap in ((False, False), (True, True)):
        model = build_model(prob, S_train=2)
        model.overlap_chains = overlap
        elbo = VariationalELBOMF(model, cfg["N"], 2)
        g = GraphedELBOStep(model, elbo, t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], lr=1e-2,
                            use_graph=use_graph, fixed_eps=[None, t(prob["eps"][1])])
        ls = []
        for _ in range(8):
            l, _ = g.step()
            g.stream.synchronize()
            ls.append(float(l))
        g.check()
        traj.append(ls)
    for a, b in zip(*traj):
        assert abs(a - b) <= 1e-9 * abs(a)
    assert traj[1][-1] < traj[1][0]


def test_acquisition_gradient_with_overlap():
    cfg = dict(d=3, L=2, M=16, N=40, S=4, seed=6)
    prob = synthetic.make_problem(**cfg)
    X = to_t(synthetic.make_problem(d=3, L=2, M=4, N=9, S=1, seed=77)["x"]).to(DEV)
    out = {}
    for overlap in (False, True):
        model = build_model(prob, S_train=1, S_acq=4)
        model.set_check_pd(False)
        model.overlap_chains = overlap
        model.eval()
        Xr = X.clone().requires_grad_(True)
        mus, vs = model.predict_for_acquisition(Xr, 1)
        (mus.sum() + 3.0 * vs.sum()).backward()
        out[overlap] = (mus.detach().clone(), vs.detach().clone(), Xr.grad.clone())
    for a, b in zip(out[True], out[False]):
        assert _rel(a, b) < 1e-12


def test_frozen_chains_reproduce_acquisition_and_its_gradient():
    """MFDGP.frozen_chains(): CHAIN half computed once, PANEL half per call (PHASE_PANEL_INPUTS backward) -- same moments,
    same JES value and the same dX as recomputing everything at every call, for varying numbers of test points."""
    from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import _JES_MFDGP
    cfg = dict(d=3, L=2, M=33, N=50, S=5, seed=8)
    prob = synthetic.make_problem(**cfg)
    prob2 = synthetic.make_problem(**dict(cfg, seed=9))
    mu, mc = build_model(prob, S_train=1, S_acq=5), build_model(prob2, S_train=1, S_acq=5)
    jes = _JES_MFDGP(1, mu, mc)
    for T in (7, 200, 1):
        X = to_t(synthetic.make_problem(d=3, L=2, M=4, N=T, S=1, seed=70 + T)["x"]).to(DEV)
        Xa = X.clone().requires_grad_(True)
        va = jes(Xa)
        va.sum().backward()
        with jes.frozen():
            for _ in range(2):                         # second pass reuses the cached chains
                Xb = X.clone().requires_grad_(True)
                vb = jes(Xb)
                vb.sum().backward()
                assert _rel(vb.detach(), va.detach()) < 1e-13 or float(va.abs().max()) == 0.0
                assert float((Xb.grad - Xa.grad).abs().max()) <= 1e-12 * max(float(Xa.grad.abs().max()), 1e-30)
            m0, v0 = mu.predict_for_acquisition(X, 0)
        m1, v1 = mu.predict_for_acquisition(X, 0)
        assert torch.equal(m0, m1) and torch.equal(v0, v1)
    assert mu._frozen is None and mc._frozen is None


@pytest.mark.parametrize("cfg", [dict(d=3, L=2, M=20, N=60, S=2, seed=9), dict(d=2, L=3, M=130, N=200, S=2, seed=3),
                                 dict(d=4, L=2, M=300, N=700, S=1, seed=5)],
                         ids=["2layers_small", "3layers_M130", "2layers_M300"])
def test_batched_chains_match_the_layer_by_layer_path(cfg):
    """All layers' CHAIN halves in one z-batched sequence of launches (mobocmf_layers_chain_*, the fast path of the graphed
    step) vs one fused call per layer: ELBO, KL and every parameter gradient agree to rounding (the batched M x M products
    are not k-sliced, so the summation order differs)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.models import MFDGP
    prob = synthetic.make_problem(**cfg)
    t = lambda a: to_t(a).to(DEV)
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    res = []
    for batched in (False, True):
        model = build_model(prob, S_train=cfg["S"])
        model.set_check_pd(False)
        keep, MFDGP.batch_chains = MFDGP.batch_chains, batched
        try:
            out = model(t(prob["x"]), eps=eps)
            e, skl = VariationalELBOMF(model, cfg["N"], cfg["L"])(out, t(prob["y"])[None, :], t(prob["fid"])[:, None])
            (-e).backward()
        finally:
            MFDGP.batch_chains = keep
        res.append((float(e), float(skl), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (e0, k0, g0), (e1, k1, g1) = res
    assert abs(e0 - e1) < 1e-11 * abs(e0) and abs(k0 - k1) < 1e-11 * abs(k0)
    assert g0.keys() == g1.keys() and len(g0) >= 4 * cfg["L"]
    for n in g0:
        err = float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-300))
        assert err < 1e-7, (n, err)


def test_batched_chains_kl_only_and_truncated_forward():
    """(i) only the KL is differentiated (no panel backward ran: the chain backward must treat H, da as zero);
    (ii) a forward truncated at max_fidelity batches the layers it runs."""
    from mobocmf_amd.models import MFDGP
    prob = synthetic.make_problem(d=2, L=3, M=12, N=30, S=1, seed=2)
    t = lambda a: to_t(a).to(DEV)
    grads = []
    for batched in (False, True):
        model = build_model(prob, S_train=1)
        model.set_check_pd(False)
        keep, MFDGP.batch_chains = MFDGP.batch_chains, batched
        try:
            out = model(t(prob["x"]), max_fidelity=1, eps=[None, t(prob["eps"][1]), None])
            assert len(out) == 2
            kl = model.variational_strategy.kl_terms()
            (kl[0] + 2.0 * kl[1] + out[1].mean.sum()).backward()
        finally:
            MFDGP.batch_chains = keep
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        err = float((grads[0][n] - grads[1][n]).abs().max() / grads[0][n].abs().max().clamp_min(1e-300))
        assert err < 1e-7, (n, err)


def test_chain_entry_points_refuse_a_short_block_buffer():
    """mobocmf_layers_chain_forward / _backward with n = 3 layers and a buffer of two blocks must return
    MOBOCMF_WORKSPACE_TOO_SMALL before anything is enqueued (layer 2 would write past the allocation)."""
    import ctypes

    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F
    lib = _lib.require_device()
    dev = torch.device("cuda")
    M, d, n = 24, 3, 3
    descs = [F.make_desc(0 if z == 0 else 1, d, M, 1, 1, 0, False, F.JITTER, F.MIN_VARIANCE, F.PHASE_CHAIN) for z in range(n)]
    bb, st = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(lib.mobocmf_chain_block_bytes(ctypes.byref(descs[0]), ctypes.byref(bb), ctypes.byref(st)), "chain_block_bytes")
    stride = (bb.value + 255) // 256 * 256
    blocks = torch.zeros(2 * stride, dtype=torch.uint8, device=dev)            # one block short
    t = lambda *s: torch.rand(*s, dtype=torch.float64, device=dev)
    Zx, zf, m, LS = t(M, d), t(M), t(M), torch.tril(t(M, M)) + torch.eye(M, dtype=torch.float64, device=dev)
    hyps = [torch.ones(F.hyp_len(descs[z].kind, d), dtype=torch.float64, device=dev) for z in range(n)]
    kls = [torch.zeros((), dtype=torch.float64, device=dev) for _ in range(n)]
    infos = [torch.zeros((), dtype=torch.int32, device=dev) for _ in range(n)]
    T = lambda ts: F._table([x.data_ptr() for x in ts])
    rc = lib.mobocmf_layers_chain_forward(n, F._desc_table(descs), T([Zx] * n), T([zf] * n), T(hyps), T([m] * n), T([LS] * n),
                                          T(kls), T(infos), F._ptr(blocks), stride, blocks.numel(), F._stream())
    assert rc == _lib.WORKSPACE_TOO_SMALL
    g = [t(M) for _ in range(n)]
    gh = [torch.zeros_like(h) for h in hyps]
    gL = [t(M, M) for _ in range(n)]
    rc = lib.mobocmf_layers_chain_backward(n, F._desc_table(descs), T([Zx] * n), T([zf] * n), T(hyps), T(kls),
                                           (ctypes.c_int32 * n)(0, 0, 0), T(g), T(gh), T(g), T(gL), F._ptr(blocks), stride,
                                           blocks.numel(), F._stream())
    assert rc == _lib.WORKSPACE_TOO_SMALL
    torch.cuda.synchronize()
    assert float(blocks.sum()) == 0.0                                          # nothing was written
