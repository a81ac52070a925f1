"""C-ABI entry points added in round 2: the GEMM with the layer's epilogues (what bench.py's per-variant roofline times),
the dense prior Gram behind MFDGPHiddenLayer.forward, the tuning thresholds."""
import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.helpers import oracle_state, to_t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("rows", [64, 128], ids=["tiles64x128", "tiles128x128"])
@pytest.mark.parametrize("Mp,Np", [(128, 256), (256, 2048), (512, 8192), (640, 1024), (512, 32768), (384, 65536), (1024, 16384)])
def test_gemm_epilogue_variants_match_torch(Mp, Np, rows):
    """The four launches of a layer's panel work through mobocmf_gemm_f64_epilogue vs float64 torch ops: small-panel
    kernel (Mp <= 512, few workgroups), tiled kernel without and with row-block pairing, on 128 x 128 tiles and on 64 x 128
    tiles (three workgroups per CU, triangular operand resolved in 64-row blocks; mobocmf_tuning.tile_rows)."""
    from mobocmf_amd import functional as F
    F.set_tile_rows(rows)
    try:
        _gemm_epilogue_variants(F, Mp, Np)
    finally:
        F.set_tile_rows(0)


def _gemm_epilogue_variants(F, Mp, Np):
    g = torch.Generator(device=DEV)
    g.manual_seed(Mp + Np)
    rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=DEV, generator=g)
    Lw, Up = torch.tril(rnd(Mp, Mp)), torch.triu(rnd(Mp, Mp))
    B, Aaux = rnd(Mp, Np), rnd(Mp, Np)
    avec, gmu, cgv, gv = rnd(Mp), rnd(Np), rnd(Np), rnd(Np)
    nrb = Mp // 128
    for tri, T in ((1, Lw), (2, Up)):
        ref = T @ B
        # plain store
        C = torch.full((Mp, Np), float("nan"), dtype=torch.float64, device=DEV)
        F.gemm_f64_epilogue(T, B, C, tri, 0)
        assert rel(C, ref) < 1e-12
        # column statistics (+ non-temporal stores)
        for so in (False, True):
            C.fill_(float("nan"))
            npart = F.gemm_colstat_rows(Mp, Np, Mp, tri)                      # two partial rows per row block of the tile height
            assert npart in (2 * nrb, 4 * nrb)
            p1 = torch.full((npart, Np), float("nan"), dtype=torch.float64, device=DEV)
            p2 = torch.full((npart, Np), float("nan"), dtype=torch.float64, device=DEV)
            F.gemm_f64_epilogue(T, B, C, tri, 1, stream_out=so, colsq_part=p1, coldot_part=p2, avec=avec)
            assert rel(C, ref) < 1e-12
            assert rel(p1.sum(0), (ref * ref).sum(0)) < 1e-12
            assert rel(p2.sum(0), avec @ ref) < 1e-11
        # dA epilogue + row dots
        C.fill_(float("nan"))
        rdp = torch.zeros(2 * max(Np // 128, Np // 16), Mp, dtype=torch.float64, device=DEV)
        F.gemm_f64_epilogue(T, B, C, tri, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=Aaux, rowdot_part=rdp)
        want = 2.0 * ref * gv[None, :] + avec[:, None] * gmu[None, :] - 2.0 * Aaux * cgv[None, :]
        assert rel(C, want) < 1e-12
        assert rel(rdp.sum(0), Aaux @ gmu) < 1e-11


def test_layer_forward_is_the_dense_prior():
    """MFDGPHiddenLayer.forward (mfdgp_hidden_layer.py:232-243): N(0, k(x, x)) of the layer's kernel, both kinds."""
    prob = synthetic.make_problem(d=3, L=2, M=10, N=40, S=1, seed=3)
    model = synthetic.model_from_problem(prob, device=DEV)
    st = oracle_state(prob)
    x = to_t(prob["x"])
    f = torch.randn(40, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    p0 = model.hidden_layer_0.forward(x.to(DEV))
    assert rel(p0.covariance_matrix, O.gram(st["layers"][0]["hyp"], x, x)) < 1e-13
    assert float(p0.mean.abs().max()) == 0.0 and p0.variance.shape == (40,)
    X1 = torch.cat([x, f], 1)
    p1 = model.hidden_layer_1.forward(X1.to(DEV))
    assert rel(p1.covariance_matrix, O.gram(st["layers"][1]["hyp"], X1, X1)) < 1e-13


def test_set_tuning_switches_kernels_not_results():
    """mobocmf_tuning.small_gemm_max / small_panel_max move the hand-over between the small-operand kernels and the tiled MFMA pipeline: results agree
    to rounding on either side; bad values are refused."""
    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F
    from mobocmf_amd.mlls import VariationalELBOMF
    prob = synthetic.make_problem(d=2, L=2, M=100, N=300, S=2, seed=1)
    t = lambda a: to_t(a).to(DEV)
    vals = []
    try:
        for lim in ((16, 16), (384, 512)):
            F.set_tuning(*lim)
            model = synthetic.model_from_problem(prob, num_samples_for_training=2, device=DEV)
            out = model(t(prob["x"]), eps=[None, t(prob["eps"][1])])
            e, _ = VariationalELBOMF(model, 300, 2)(out, t(prob["y"])[None, :], t(prob["fid"])[:, None])
            (-e).backward()
            vals.append((float(e), model.hidden_layer_1.variational_strategy._variational_distribution
                         .variational_mean.grad.clone()))
    finally:
        F.set_tuning(384, 512)
    assert abs(vals[0][0] - vals[1][0]) < 1e-9 * abs(vals[1][0])
    assert rel(vals[0][1], vals[1][1]) < 1e-6
    with pytest.raises(_lib.MobocmfError):
        F.set_tuning(4096, 0)


def test_limits_are_reported_with_a_message():
    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F
    with pytest.raises(_lib.MobocmfError, match="input dimensions"):
        F.make_desc(0, 33, 8, 16)
    with pytest.raises(_lib.MobocmfError, match="samples per input row"):
        F.make_desc(1, 4, 8, 49 * 2, xdiv=49)


@pytest.mark.parametrize("d,L,nF", [(2, 2, 150), (8, 2, 500), (3, 3, 64), (32, 2, 97)])
def test_rff_grid_evaluation_kernel_matches_oracle(d, L, nF):
    """mobocmf_rff_eval (one function sample of every layer on a Pareto-sized grid, layer recursion, no
    F x n feature matrix) against the numpy restatement of the reference's feature maps, oracle/rff_oracle.py
    ``layer0_features`` / ``layer1_features`` (mfdgp_hidden_layer.py:288-292, :319-321, :384-399) times theta, with explicit
    W, b, theta and hyper-parameters: nothing of mobocmf_amd.layers.rff takes part."""
    from mobocmf_amd import functional as F
    from oracle import rff_oracle as R
    rng = np.random.default_rng(100 * d + L)
    n = 4200 + 37                                       # >= the 1000 d^2 + N rows of moop.py:232 at d = 2; ragged last block
    X = rng.random((n, d))
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)
    f_ref, f_dev = None, None
    for l in range(L):
        if l == 0:
            alpha, ls = 1.3, 0.3 + rng.random(d)
            W, b, theta = rng.standard_normal((nF, d)) / ls, 2 * np.pi * rng.random((nF, 1)), rng.standard_normal(nF)
            ref = theta @ R.layer0_features(X, W, b, alpha)
            out = F.rff_eval(0, dev(X), None, dev(W), dev(b[:, 0]), None, None, None, dev(theta), np.sqrt(2.0 * alpha / nF))
        else:
            a1, af, a2, nu = 0.8 + rng.random(), 0.5 + rng.random(), 0.01 + 0.1 * rng.random(), 0.4 + rng.random()
            ls1, lsf, ls2 = 2.0 + rng.random(d), 0.7 + rng.random(), 0.3 + rng.random(d)
            W1, Wf, W2 = rng.standard_normal((nF, d)) / ls1, rng.standard_normal(nF) / lsf, rng.standard_normal((nF, d)) / ls2
            b1, b2 = 2 * np.pi * rng.random((nF, 1)), 2 * np.pi * rng.random((nF, 1))
            theta = rng.standard_normal(3 * nF)
            ref = theta @ R.layer1_features(X, f_ref, W1, Wf, W2, b1, b2, a1, af, a2, nu)
            out = F.rff_eval(1, dev(X), f_dev, dev(W1), dev(b1[:, 0]), dev(Wf), dev(W2), dev(b2[:, 0]), dev(theta),
                             np.sqrt(2.0 * a1 * nu / nF), np.sqrt(2.0 * a1 * af / nF), np.sqrt(2.0 * a2 / nF))
        got = out.cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), (l, np.abs(got - ref).max())
        f_ref, f_dev = ref, dev(ref)                    # the next layer of both sides sees the ORACLE's sample


@pytest.mark.parametrize("d,L", [(2, 2), (8, 2), (3, 3)])
def test_rff_sample_callables_use_the_kernel_for_grids(d, L):
    """Host plumbing of the samplers only (the kernel itself is checked against the oracle above): the callables of
    ``sample_function_from_each_layer`` send batches of >= GRID_ROWS_ON_DEVICE rows through mobocmf_rff_eval and smaller
    ones through the host evaluation of the SAME sample, and both agree."""
    from mobocmf_amd.layers import rff
    prob = synthetic.make_problem(d=d, L=L, M=12, N=40, S=1, seed=d)
    model = synthetic.model_from_problem(prob, device=DEV)
    g = torch.Generator().manual_seed(5)
    fs = model.sample_function_from_each_layer(nFeatures=150, generator=g)
    n = max(rff.GRID_ROWS_ON_DEVICE, 4200)
    X = np.random.default_rng(0).random((n, d))
    small = X[:7]
    for l, f in enumerate(fs):
        dev_vals = f(X)                                   # >= GRID_ROWS_ON_DEVICE rows: the kernel
        host_vals = f._torch(torch.as_tensor(X)).detach().numpy()
        assert np.abs(dev_vals - host_vals).max() < 1e-10 * max(1.0, np.abs(host_vals).max()), l
        assert np.abs(f(small) - host_vals[:7]).max() < 1e-12 * max(1.0, np.abs(host_vals).max())      # host path
    pr = model.sample_function_from_prior_each_layer(nFeatures=64, generator=g)
    assert np.abs(pr[-1](X) - pr[-1]._torch(torch.as_tensor(X)).numpy()).max() < 1e-10


@pytest.mark.parametrize("Mp,Np", [(128, 2048), (256, 16384), (384, 4096), (512, 8192), (640, 8192), (1024, 4096), (512, 65536)])
def test_weighted_syrk_matches_torch(Mp, Np):
    """H = A diag(w) A^T through mobocmf_syrk_weighted_f64 (small-operand kernel; k-sliced tiled kernel with per-class slice
    counts -- one, two, ... five row blocks -- and the skipped 16 x 16 blocks above the diagonal of diagonal tiles mirrored
    by the slab reduction) vs float64 torch; the result is exactly symmetric."""
    from mobocmf_amd import functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(Mp * 7 + Np)
    A = torch.randn(Mp, Np, dtype=torch.float64, device=DEV, generator=g)
    w = torch.randn(Np, dtype=torch.float64, device=DEV, generator=g)
    H = torch.full((Mp, Mp), float("nan"), dtype=torch.float64, device=DEV)
    F.syrk_weighted(A, w, H)
    ref = (A * w[None, :]) @ A.T
    assert rel(H, ref) < 1e-12
    assert torch.equal(H, H.T)


def test_softplus_pack_matches_torch():
    """mobocmf_softplus_pack / _backward (the constrained hyper-parameter vector of a layer in one launch) vs
    softplus(cat(raws)) and its autograd gradient, including 0-d tensors and values beyond softplus' threshold."""
    from mobocmf_amd import functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    shapes = [(), (), (1, 1), (), (1, 1), (1, 8), (1, 8)]
    raws = [(5.0 * torch.randn(s, dtype=torch.float64, device=DEV, generator=g)).requires_grad_(True) for s in shapes]
    with torch.no_grad():
        raws[1].fill_(25.0)      # beyond the threshold: identity branch
    ref_in = [r.detach().clone().requires_grad_(True) for r in raws]
    w = torch.randn(sum(r.numel() for r in raws), dtype=torch.float64, device=DEV, generator=g)
    out = F.softplus_pack(raws)
    (out * w).sum().backward()
    ref = torch.nn.functional.softplus(torch.cat([r.reshape(-1) for r in ref_in]))
    (ref * w).sum().backward()
    assert rel(out, ref) < 1e-15
    for a, b in zip(raws, ref_in):
        assert a.grad.shape == b.grad.shape and rel(a.grad.reshape(-1), b.grad.reshape(-1)) < 1e-14



def test_propagate_rng_draws_standard_normals_and_advances():
    """mobocmf_propagate_rng_forward (eps of mfdgp_hidden_layer.py:272-274 drawn inside the propagation launch, Philox4x32-10 +
    Box-Muller): f = mean + sqrt(var) eps holds exactly for the returned eps; eps has the moments of N(0, 1) and no serial
    correlation; a call advances the counter (fresh draw), the same (seed, calls) reproduces the draw; gradients equal
    those of the explicit-eps entry point."""
    from mobocmf_amd import functional as F
    nb, div = 250_000, 8
    n = nb * div
    g = torch.Generator(device=DEV).manual_seed(1)
    mean = torch.randn(nb, dtype=torch.float64, device=DEV, generator=g).requires_grad_(True)
    var = (torch.rand(nb, dtype=torch.float64, device=DEV, generator=g) + 0.1).requires_grad_(True)
    st = torch.tensor([123456789, 0, 0], dtype=torch.int64, device=DEV)
    f1, e1 = F.propagate_rng(mean, var, st, n, div)
    assert st.tolist() == [123456789, 1, 0]
    ref = mean.detach().repeat_interleave(div) + var.detach().sqrt().repeat_interleave(div) * e1
    assert float((f1.detach() - ref).abs().max()) <= 1e-15 * float(ref.abs().max())      # (the kernel fuses the multiply-add)
    e = e1.double()
    m1, m2 = float(e.mean()), float(e.var())
    skew, kurt = float((e ** 3).mean()), float((e ** 4).mean())
    assert abs(m1) < 5.0 / n ** 0.5 and abs(m2 - 1.0) < 5.0 * (2.0 / n) ** 0.5
    assert abs(skew) < 5.0 * (15.0 / n) ** 0.5 and abs(kurt - 3.0) < 5.0 * (96.0 / n) ** 0.5
    assert abs(float((e[1:] * e[:-1]).mean())) < 5.0 / n ** 0.5            # neighbouring rows
    assert abs(float((e[div:] * e[:-div]).mean())) < 5.0 / n ** 0.5        # neighbouring base rows
    assert float(e.abs().max()) < 7.0 and float((e.abs() > 3.0).double().mean()) == pytest.approx(0.0027, abs=2e-4)
    f2, e2 = F.propagate_rng(mean, var, st, n, div)                            # the next call: another draw
    assert st.tolist() == [123456789, 2, 0] and abs(float((e1 * e2).mean())) < 5.0 / n ** 0.5 and not torch.equal(e1, e2)
    st.copy_(torch.tensor([123456789, 0, 0]))                                  # same (seed, calls): the same draw
    f3, e3 = F.propagate_rng(mean, var, st, n, div)
    assert torch.equal(e3, e1) and torch.equal(f3, f1)
    st2 = torch.tensor([987654321, 0, 0], dtype=torch.int64, device=DEV)       # another seed: another stream
    _, e4 = F.propagate_rng(mean, var, st2, n, div)
    assert abs(float((e1 * e4).mean())) < 5.0 / n ** 0.5
    w = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    (f3 * w).sum().backward()
    gm, gv = mean.grad.clone(), var.grad.clone()
    mean.grad = var.grad = None
    (F.propagate(mean, var, e1, div) * w).sum().backward()
    assert torch.equal(mean.grad, gm) and torch.equal(var.grad, gv)


@pytest.mark.parametrize("waves", [4, 8, 32])
@pytest.mark.parametrize("Mr,Nc,Kd", [(512, 512, 512), (640, 640, 640), (1024, 1024, 1024), (128, 384, 384), (128, 896, 896),
                                      (768, 512, 1024)])
def test_mid_gemm_matches_torch_and_the_tiled_kernel(Mr, Nc, Kd, waves):
    """Plain products with every dimension in (384, 1024] -- the M x M chain of C3 / C5 -- run on the mid-size kernel (64 x 64
    tiles, whole contraction per workgroup, one launch; mobocmf_tuning.mid_gemm_max): all triangular-operand flags the chain uses,
    A B and A B^T, alpha / accumulate, vs float64 torch and vs the 128 x 128 pipeline (knob off)."""
    from mobocmf_amd import functional as F
    F.set_mid_gemm_max(1024)
    F.set_mid_gemm_waves(waves)   # the three forms of the kernel
    try:
        _mid_gemm_cases(F, Mr, Nc, Kd)
    finally:
        F.set_mid_gemm_max(1024)
        F.set_mid_gemm_waves(32)


def _mid_gemm_cases(F, Mr, Nc, Kd):
    g = torch.Generator(device=DEV)
    g.manual_seed(Mr + 3 * Nc + 7 * Kd)
    rnd = lambda *s: torch.randn(*s, dtype=torch.float64, device=DEV, generator=g)
    A0, B0, Bt0 = rnd(Mr, Kd), rnd(Kd, Nc), rnd(Nc, Kd)
    LOWER_A, UPPER_A, LOWER_B, UPPER_B = 1, 2, 4, 8
    cases = [(0, False), (LOWER_A, False), (UPPER_A, False), (LOWER_B, False), (LOWER_A | LOWER_B, False),
             (UPPER_A | LOWER_B, False), (LOWER_A | UPPER_B, False), (0, True), (LOWER_A | UPPER_B, True)]
    for tri, tb in cases:
        # the unused triangle holds zeros (as every chain operand does); square blocks decide what "triangle" means
        A = torch.tril(A0) if tri & LOWER_A else torch.triu(A0) if tri & UPPER_A else A0
        Bl = Bt0.T if tb else B0                       # logical B [Kd x Nc]
        Bl = torch.tril(Bl) if tri & LOWER_B else torch.triu(Bl) if tri & UPPER_B else Bl
        Bop = Bl.T.contiguous() if tb else Bl.contiguous()
        ref = A @ Bl
        for alpha, acc in ((1.0, False), (-0.5, True)):
            C0 = rnd(Mr, Nc)
            want = alpha * ref + (C0 if acc else 0.0)
            C = C0.clone() if acc else torch.full((Mr, Nc), float("nan"), dtype=torch.float64, device=DEV)
            F.gemm_f64(A.contiguous(), Bop, C, tri=tri, trans_b=tb, alpha=alpha, accumulate=acc)
            assert rel(C, want) < 1e-12, (tri, tb, alpha)
            F.set_mid_gemm_max(0)
            try:
                C2 = C0.clone() if acc else torch.full((Mr, Nc), float("nan"), dtype=torch.float64, device=DEV)
                F.gemm_f64(A.contiguous(), Bop, C2, tri=tri, trans_b=tb, alpha=alpha, accumulate=acc)
            finally:
                F.set_mid_gemm_max(1024)
            assert rel(C2, want) < 1e-12 and rel(C, C2) < 1e-12, (tri, tb, alpha)


@pytest.mark.parametrize("n_obj,n_con,P,T", [(2, 1, 50, 10), (1, 0, 7, 3), (3, 2, 33, 300), (8, 8, 5, 17)])
def test_conditioned_factor_losses_match_oracle(n_obj, n_con, P, T):
    """mobocmf_cond_factors_forward (omega factors, blackbox_mfdgp_fitter.py:235-243; theta factors :227-233) and the segment
    glue of the conditioned loss vs the oracle's torch statement: value and the gradient w.r.t. every mean / variance entry."""
    from mobocmf_amd import functional as F
    from oracle import mfdgp_oracle as O
    rng = np.random.default_rng(n_obj * 100 + n_con * 10 + T)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    fm, fv = t(rng.standard_normal((n_obj, T))), t(0.2 + rng.random((n_obj, T)))
    cm, cv = t(rng.standard_normal((n_con, T))), t(0.2 + rng.random((n_con, T)))
    front, thr = t(rng.standard_normal((P, n_obj))), t(0.3 * rng.standard_normal(n_con))
    eps = 1e-8
    leaves_o = [x.clone().requires_grad_(True) for x in (fm, fv, cm, cv)]
    lo = O.loss_omega_factors(leaves_o[0], leaves_o[1], leaves_o[2], leaves_o[3], front, thr, eps)
    lo.backward()
    leaves = [x.clone().to(DEV).requires_grad_(True) for x in (fm, fv, cm, cv)]
    rows = lambda x: list(x.unbind(0))
    lh = F.cond_factors(rows(leaves[0]), rows(leaves[1]), rows(leaves[2]), rows(leaves[3]), front.to(DEV), thr.to(DEV),
                        float(np.log(eps)), float(np.log(1 - eps)))
    (2.5 * lh).backward()
    assert rel(lh, lo) < 1e-12
    for a, b in zip(leaves, leaves_o):
        if b.numel():
            assert rel(a.grad, 2.5 * b.grad) < 1e-11
    if n_con:      # theta factors of the first constraint (no objective rows, P = 1)
        mu, var = cm[0].clone().requires_grad_(True), cv[0].clone().requires_grad_(True)
        lt_o = O.loss_theta_factors(mu, var, thr[0], eps)
        lt_o.backward()
        mu_h, var_h = cm[0].clone().to(DEV).requires_grad_(True), cv[0].clone().to(DEV).requires_grad_(True)
        lt = F.cond_factors([], [], [mu_h], [var_h], None, thr[:1].to(DEV), float(np.log(1 - eps)), float(np.log(eps)))
        lt.backward()
        assert rel(lt, lt_o) < 1e-12 and rel(mu_h.grad, mu.grad) < 1e-11 and rel(var_h.grad, var.grad) < 1e-11


def test_split_rows_and_scalar_combine_are_exact():
    """The glue nodes of the conditioned loss: row ranges as views with a one-launch backward (zeros where a range got no
    gradient), and a signed sum of scalar terms."""
    from mobocmf_amd import functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    m = torch.randn(37, dtype=torch.float64, device=DEV, generator=g, requires_grad=True)
    v = torch.rand(37, dtype=torch.float64, device=DEV, generator=g).requires_grad_(True)
    (m0, m1, m2), (v0, v1, v2) = F.split_rows(m, v, [5, 20, 12])
    assert torch.equal(m1, m.detach()[5:25]) and torch.equal(v2, v.detach()[25:])
    terms = [(m0 * m0).sum(), (m2 * v2).sum(), v1.sum()]            # m1 and v0 get no gradient at all
    loss = F.scalar_combine(terms, [-2.0, 0.5, 3.0])
    loss.backward()
    md, vd = m.detach(), v.detach()
    want = -2.0 * (md[:5] ** 2).sum() + 0.5 * (md[25:] * vd[25:]).sum() + 3.0 * vd[5:25].sum()
    assert abs(float(loss) - float(want)) < 1e-12 * max(1.0, abs(float(want)))
    gm = torch.cat([-4.0 * md[:5], torch.zeros(20, dtype=torch.float64, device=DEV), 0.5 * vd[25:]])
    gv = torch.cat([torch.zeros(5, dtype=torch.float64, device=DEV), 3.0 * torch.ones(20, dtype=torch.float64, device=DEV), 0.5 * md[25:]])
    assert torch.equal(m.grad, gm) and torch.equal(v.grad, gv)
    # more terms than one launch takes (a conditioned loss over 16+ black-boxes on a rank: 2 * handlers + 1 terms): chunked
    w = torch.randn(71, dtype=torch.float64, device=DEV, generator=g, requires_grad=True)
    cs = [float(c) for c in torch.linspace(-1.5, 2.0, 71)]
    F.scalar_combine([w[i] * w[i] for i in range(71)], cs).backward()
    assert rel(w.grad, 2.0 * torch.tensor(cs, dtype=torch.float64, device=DEV) * w.detach()) < 1e-14
