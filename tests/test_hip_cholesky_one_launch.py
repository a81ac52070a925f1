"""The blocked Cholesky + triangular inverse as ONE launch (chol.hip potrf_coop_kernel, mobocmf_tuning.potrf_cols = 0, the
default for 128 < M <= 1024) against the launch pair per 64 columns (potrf_cols = 4) and against torch.linalg on the same
matrices: factors, inverse, failed pivots, the layer chain (z-batched layers) and graph replay.
Reference behaviour: gpytorch psd_safe_cholesky(K_mm) inside the variational strategy."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _gram(n, d, ls, jitter, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    x = torch.rand(n, d, dtype=torch.float64, device=DEV, generator=g)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    K = torch.exp(-0.5 * d2 / ls ** 2) + jitter * torch.eye(n, dtype=torch.float64, device=DEV)
    return K, torch.randn(n, dtype=torch.float64, device=DEV, generator=g)


def _factors(st, n):
    npad = (n + 127) // 128 * 128
    buf = st.state.view(torch.float64)
    L = buf[:npad * npad].view(npad, npad)
    off = (npad * npad * 8 + 255) // 256 * 256 // 8
    Li = buf[off:off + npad * npad].view(npad, npad)
    return L, Li, npad


@pytest.mark.parametrize("n,d,ls", [(129, 3, 0.7), (192, 4, 1.0), (200, 2, 0.6), (320, 5, 1.2), (512, 8, 1.4), (700, 2, 0.5),
                                     (1000, 8, 1.4), (1024, 3, 0.8)])
def test_one_launch_factor_and_inverse_match_the_launch_pair_form_and_torch(n, d, ls):
    from mobocmf_amd import functional as F
    K, y = _gram(n, d, ls, 1e-6, n)
    out = {}
    for cols in (0, 4):
        with F.tuning(potrf_cols=cols):
            st = F.exact_gp_factor(K, y)
        assert F.check_info(st.info) == 0
        L, Li, npad = _factors(st, n)
        out[cols] = (L.clone(), Li.clone(), float(st.mll))
        # the padding is the identity, the strict upper triangles are zero
        assert torch.equal(torch.triu(L, 1), torch.zeros_like(L)) and torch.equal(torch.triu(Li, 1), torch.zeros_like(Li))
        eye = torch.eye(npad - n, dtype=torch.float64, device=DEV)
        assert torch.equal(L[n:, n:], eye) and torch.equal(Li[n:, n:], eye)
        assert not L[n:, :n].any() and not Li[n:, :n].any()
    L0, Li0, mll0 = out[0]
    L4, Li4, mll4 = out[4]
    nk = float(torch.linalg.norm(K))
    # backward error of the factor at the level of rocSOLVER's (a few ulp), the inverse a left inverse to what the
    # conditioning allows, both forms the same numbers up to rounding
    Lt = torch.linalg.cholesky(K)
    ref = float(torch.linalg.norm(Lt @ Lt.T - K)) / nk
    assert float(torch.linalg.norm(L0[:n, :n] @ L0[:n, :n].T - K)) / nk < max(4.0 * ref, 4e-15)
    e0 = float(torch.linalg.norm(Li0[:n, :n] @ L0[:n, :n] - torch.eye(n, dtype=torch.float64, device=DEV)))
    e4 = float(torch.linalg.norm(Li4[:n, :n] @ L4[:n, :n] - torch.eye(n, dtype=torch.float64, device=DEV)))
    assert e0 < max(2.0 * e4, 1e-12)
    assert float((L0 - L4).abs().max()) < 1e-9 * float(L4.abs().max())
    assert abs(mll0 - mll4) < 1e-9 * abs(mll4)


def test_one_launch_factor_reports_the_failed_pivot():
    from mobocmf_amd import functional as F
    n = 400
    K, y = _gram(n, 3, 0.9, 1e-4, 1)
    for bad in (5, 64, 130, 333, 399):
        Kb = K.clone()
        Kb[bad, bad] = -3.0
        got = {}
        for cols in (0, 4):
            with F.tuning(potrf_cols=cols):
                got[cols] = F.check_info(F.exact_gp_factor(Kb, y).info)
        assert got[0] == got[4] == bad + 1
    with F.tuning(potrf_cols=0):
        assert F.check_info(F.exact_gp_factor(K, y).info) == 0      # (the status word is rewritten by every call)


def test_layer_chain_is_the_same_through_both_forms_and_replays_from_a_graph():
    """z-batched layers (both layers of a surrogate in one chain launch sequence, M = 200 -> four 64-blocks, the last partly
    padding): ELBO and every gradient through the one-launch factorisation equal those through the launch pairs up to
    rounding; a captured step replays the eager trajectory bit for bit."""
    from mobocmf_amd import functional as F
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util import synthetic
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    from tests.helpers import to_t
    from tests.test_hip_model import build_model, hip_elbo
    prob = synthetic.make_problem(d=3, L=2, M=200, N=320, S=2, seed=4)
    res = {}
    for cols in (0, 4):
        with F.tuning(potrf_cols=cols):
            model = build_model(prob, S_train=2)
            (loss, _), _ = hip_elbo(model, prob, 2)
            (-loss).backward()
        res[cols] = (float(loss.detach()), [p.grad.detach().clone() for p in model.parameters() if p.grad is not None])
    # (200 inducing points in 3-D: cond(K_mm + 1e-6 I) ~ 1e9, either form carries ~cond * eps)
    assert abs(res[0][0] - res[4][0]) <= 1e-8 * abs(res[4][0])
    assert len(res[0][1]) == len(res[4][1]) > 0
    for a, b in zip(res[0][1], res[4][1]):
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-3, float(b.abs().max()))
    t = lambda a: to_t(a).to(DEV)
    losses = []
    for use_graph in (False, True):
        model = build_model(prob, S_train=2)
        elbo = VariationalELBOMF(model, 320, 2)
        g = GraphedELBOStep(model, elbo, t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], lr=1e-3,
                            use_graph=use_graph, fixed_eps=[None, t(prob["eps"][1])])
        ls = []
        for _ in range(5):
            l, _ = g.step()
            g.stream.synchronize()
            ls.append(float(l))
        g.check()
        losses.append(ls)
    assert losses[0] == losses[1]


def test_more_concurrent_one_launch_factorisations_than_the_chip_holds_are_still_right_or_say_so():
    """Six streams, each with the one-launch factorisation of a 1024 x 1024 matrix in flight (49 workgroups each that must be
    resident together: 294 on 256 CUs).  The launches are ordinary ones: the hardware may hold some of them only partly until
    others finish.  Every result must be EITHER bit-identical to the factorisation done alone OR carry info = -1 (a bounded
    in-launch wait was abandoned: functional.InLaunchWaitAbandoned) -- never a wrong factor, never a hang."""
    from mobocmf_amd import functional as F
    n, ns = 1024, 6
    mats = [_gram(n, 4, 1.0, 1e-4, 10 + i) for i in range(ns)]
    alone = []
    for K, y in mats:
        st = F.exact_gp_factor(K, y)
        assert F.check_info(st.info) == 0
        alone.append((_factors(st, n)[0].clone(), _factors(st, n)[1].clone(), float(st.mll)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(ns)]
    abandoned = 0
    for rnd in range(3):
        states = []
        for (K, y), s in zip(mats, streams):
            with torch.cuda.stream(s):
                states.append(F.exact_gp_factor(K, y))      # (scratch is per (device, stream): functional.scratch_buffer)
        torch.cuda.synchronize()
        for i, st in enumerate(states):
            piv = F.check_info(st.info)
            if piv == -1:
                abandoned += 1
                continue
            assert piv == 0
            L, Li, _ = _factors(st, n)
            assert torch.equal(L, alone[i][0]) and torch.equal(Li, alone[i][1]) and float(st.mll) == alone[i][2]
    print("abandoned waits: %d of %d concurrent factorisations" % (abandoned, 3 * ns))
