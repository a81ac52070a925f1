"""Known-answer tests D1-D7: pin the oracle from GP theory, since the
reference has no tests of its own (parity unpinned by the reference)."""
import numpy as np
import torch

from oracle import mfdgp_oracle as O
from tests.helpers import oracle_state, small_problem, state_leaves, to_t

import pytest


@pytest.fixture(autouse=True, scope="module")
def _float64_default():
    """torch.zeros / torch.eye / torch.rand below are float64 -- for THIS module only (a global default would leak into
    every other test module of the session, and the mirrored model deliberately builds its constants at torch's default
    dtype like the reference)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def _layer(prob, l):
    st = oracle_state(prob)
    return st, st["layers"][l]


def test_D1_prior_recovery():
    prob = small_problem(M=10, N=30)
    st, lay = _layer(prob, 0)
    Z = st["Zx"]
    K = O.gram(lay["hyp"], Z, Z) + O.JITTER * torch.eye(Z.shape[0])
    Lp = torch.linalg.cholesky(K)
    X = to_t(prob["x"])
    mean, var, _ = O.layer_moments(lay["hyp"], X, Z, torch.zeros(Z.shape[0]), Lp)
    assert mean.abs().max() < 1e-14
    assert (var - O.gram_diag(lay["hyp"], X)).abs().max() < 1e-12
    assert abs(O.kl_layer(lay["hyp"], Z, torch.zeros(Z.shape[0]), Lp)) < 1e-10


def _optimal_q(hyp, Z, X, y, tau):
    """Titsias' optimal q(u) for inducing inputs Z with the jittered prior K~ = K_zz + eps I."""
    M = Z.shape[0]
    Kt = O.gram(hyp, Z, Z) + O.JITTER * torch.eye(M)
    Kzx = O.gram(hyp, Z, X)
    Sig = torch.linalg.inv(Kt + Kzx @ Kzx.T / tau)
    m = Kt @ Sig @ Kzx @ y / tau
    S = Kt @ Sig @ Kt
    S = 0.5 * (S + S.T)
    return m, torch.linalg.cholesky(S), Kt, Kzx


def _titsias_bound(hyp, Z, X, y, tau):
    m, LS, Kt, Kzx = _optimal_q(hyp, Z, X, y, tau)
    n = X.shape[0]
    Q = Kzx.T @ torch.linalg.solve(Kt, Kzx)
    bound = torch.distributions.MultivariateNormal(torch.zeros(n), Q + tau * torch.eye(n)).log_prob(y) \
        - (O.gram_diag(hyp, X) - torch.diagonal(Q)).sum() / (2 * tau)
    return m, LS, bound


def test_D2_optimal_q_elbo_equals_collapsed_bound():
    prob = small_problem(d=2, M=10, N=10)
    st, lay = _layer(prob, 0)
    X = to_t(prob["x"])
    y = to_t(prob["y"])
    tau = torch.tensor(0.05)
    perm = torch.randperm(10, generator=torch.Generator().manual_seed(1))
    for Z in (X[perm], X[perm][:6] + 0.01):       # Z = X permuted (shortcut does not fire); M < N
        m, LS, bound = _titsias_bound(lay["hyp"], Z, X, y, tau)
        mean, var, _ = O.layer_moments(lay["hyp"], X, Z, m, LS)
        elbo = O.expected_log_prob(y, mean, var, tau).sum() - O.kl_layer(lay["hyp"], Z, m, LS)
        assert abs(elbo - bound) / abs(bound) < 1e-7
    # Z = X: collapsed bound = exact log marginal up to the 1e-6 jitter
    Ky = O.gram(lay["hyp"], X, X) + tau * torch.eye(10)
    lml = torch.distributions.MultivariateNormal(torch.zeros(10), Ky).log_prob(y)
    m, LS, bound = _titsias_bound(lay["hyp"], X[perm], X, y, tau)
    assert abs(bound - lml) / abs(lml) < 1e-4


def test_D3_optimal_q_predictive_is_exact_gp():
    prob = small_problem(d=2, M=10, N=10)
    st, lay = _layer(prob, 0)
    X = to_t(prob["x"])
    y = to_t(prob["y"])
    tau = torch.tensor(0.05)
    m, LS, _, _ = _optimal_q(lay["hyp"], X, X, y, tau)
    Xs = torch.rand(7, 2, generator=torch.Generator().manual_seed(3))
    mean, var, _ = O.layer_moments(lay["hyp"], Xs, X, m, LS)
    Ky = O.gram(lay["hyp"], X, X) + tau * torch.eye(10)
    Ks = O.gram(lay["hyp"], X, Xs)
    mu_ex = Ks.T @ torch.linalg.solve(Ky, y)
    var_ex = O.gram_diag(lay["hyp"], Xs) - (Ks * torch.linalg.solve(Ky, Ks)).sum(0)
    assert (mean - mu_ex).abs().max() < 1e-4      # exact GP up to the 1e-6 jitter
    assert (var - var_ex).abs().max() < 1e-4
    # exact identity: Titsias' sparse predictive with the jittered prior
    _, _, Kt, Kzx = _optimal_q(lay["hyp"], X, X, y, tau)
    Sig = torch.linalg.inv(Kt + Kzx @ Kzx.T / tau)
    mu_sp = Ks.T @ Sig @ Kzx @ y / tau
    var_sp = O.gram_diag(lay["hyp"], Xs) - (Ks * torch.linalg.solve(Kt, Ks)).sum(0) + (Ks * (Sig @ Ks)).sum(0)
    assert (mean - mu_sp).abs().max() < 1e-8
    assert (var - var_sp).abs().max() < 1e-8


def test_D4_shortcut():
    prob = small_problem(M=9, N=9)
    st, lay = _layer(prob, 0)
    Z = st["Zx"]
    mean, var, ex = O.layer_moments(lay["hyp"], Z.clone(), Z, lay["m"], lay["L_S"])
    assert ex["shortcut"]
    assert torch.equal(mean, lay["m"])
    assert torch.allclose(var, (torch.tril(lay["L_S"]) ** 2).sum(1))
    mean2, var2, ex2 = O.layer_moments(lay["hyp"], Z.clone(), Z, lay["m"], lay["L_S"], shortcut=False)
    assert not ex2["shortcut"]
    assert (mean2 - mean).abs().max() < 1e-3 and (var2 - var).abs().max() < 1e-3  # O(jitter * |K^-1 m|)


def test_D5_train_vs_eval_branch_and_full_cov():
    prob = small_problem(M=8, N=20)
    st = oracle_state(prob)
    lay = st["layers"][1]
    Zt = O.inducing_inputs(st, 1)
    X = torch.cat([to_t(prob["x"]), to_t(prob["eps"][1][:20])[:, None]], 1)
    mt, vt, _ = O.layer_moments(lay["hyp"], X, Zt, lay["m"], lay["L_S"], training=True)
    me, ve, ex = O.layer_moments(lay["hyp"], X, Zt, lay["m"], lay["L_S"], training=False, full_cov=True)
    assert torch.allclose(mt, me) and torch.allclose(vt, ve, rtol=1e-12, atol=1e-14)
    cov = ex["cov"]
    assert (cov - cov.T).abs().max() < 1e-12
    assert torch.allclose(torch.diagonal(cov), ve, rtol=1e-10, atol=1e-12)
    assert torch.linalg.eigvalsh(0.5 * (cov + cov.T)).min() > -1e-9


def test_D6_only_hf_reduction():
    prob = small_problem(M=8, N=15)
    st = oracle_state(prob)
    lay = st["layers"][1]
    hyp = dict(lay["hyp"])
    hyp["a1"] = torch.tensor(0.0)
    Zt = O.inducing_inputs(st, 1)
    x = to_t(prob["x"][:15])
    f1 = torch.randn(15, generator=torch.Generator().manual_seed(0))
    f2 = torch.randn(15, generator=torch.Generator().manual_seed(1))
    o1 = O.layer_moments(hyp, torch.cat([x, f1[:, None]], 1), Zt, lay["m"], lay["L_S"])
    o2 = O.layer_moments(hyp, torch.cat([x, f2[:, None]], 1), Zt, lay["m"], lay["L_S"])
    assert torch.allclose(o1[0], o2[0]) and torch.allclose(o1[1], o2[1])


def test_ref_equiv_sequence_matches_whitened():
    prob = small_problem(M=12, N=25, S=2)
    st = oracle_state(prob)
    x, y, fid = to_t(prob["x"]), to_t(prob["y"]), to_t(prob["fid"])
    eps = [None, to_t(prob["eps"][1])]
    e1 = O.elbo(st, x, y, fid, eps=eps, S=2, ref_equiv=False)
    e2 = O.elbo(st, x, y, fid, eps=eps, S=2, ref_equiv=True)
    assert abs(e1[0] - e2[0]) / abs(e1[0]) < 1e-9
    assert abs(e1[1] - e2[1]) / abs(e1[1]) < 1e-9    # expanded-norm distances round differently


def test_D7_finite_difference_gradients():
    """Every gradient path incl. m0 via Z~_1 and via f~."""
    prob = small_problem(d=2, M=6, N=10, S=2)
    x, y, fid = to_t(prob["x"]), to_t(prob["y"]), to_t(prob["fid"])
    eps = [None, to_t(prob["eps"][1])]

    def f(*leaves):
        st = oracle_state(prob)
        it = iter(leaves)
        for lay in st["layers"]:
            for k in sorted(lay["hyp"]):
                lay["hyp"][k] = next(it)
            lay["m"] = next(it)
            lay["L_S"] = next(it)
        st["noise"] = [next(it) for _ in st["noise"]]
        return O.elbo(st, x, y, fid, eps=eps, S=2)[0]

    leaves = [t.clone().requires_grad_(True) for t in state_leaves(oracle_state(prob))]
    # noise 1e-6 makes FD ill-scaled: use a benign value for the check
    leaves[-2] = torch.tensor(1e-2, requires_grad=True)
    assert torch.autograd.gradcheck(f, leaves, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_S1_reduces_to_reference_semantics():
    """With S=1 the extension is the reference's path: one eps per row, no replication."""
    prob = small_problem(M=7, N=11, S=1)
    st = oracle_state(prob)
    x, y, fid = to_t(prob["x"]), to_t(prob["y"]), to_t(prob["fid"])
    eps = [None, to_t(prob["eps"][1])]
    outs = O.model_forward(st, x, eps=eps, S=1)
    assert outs[0][0].shape == (11,) and outs[1][0].shape == (11,)
    e, skl = O.elbo(st, x, y, fid, eps=eps, S=1)
    # manual
    m0, v0 = outs[0]
    m1, v1 = outs[1]
    d0 = O.expected_log_prob(y, m0, v0, st["noise"][0])[fid == 0].sum()
    d1 = O.expected_log_prob(y, m1, v1, st["noise"][1])[fid == 1].sum()
    assert torch.allclose(e + skl, d0 + d1)


def test_predict_for_acquisition_shapes_and_moments():
    prob = small_problem(M=8, N=12, S=4)
    st = oracle_state(prob)
    X = torch.rand(5, 1, 2, generator=torch.Generator().manual_seed(0))
    mus, vs = O.predict_for_acquisition(st, X, 1, S=4)
    assert mus.shape == (5,) and vs.shape == (5,)
    assert (vs > 0).all()
    mus0, vs0 = O.predict_for_acquisition(st, X, 0, S=4)
    # layer 0 has no sampling: variance = sigma^2 + tau exactly
    m0, v0 = O.predict(st, X[:, 0, :], 0, training=False)
    assert torch.allclose(mus0, m0) and torch.allclose(vs0, v0, rtol=1e-9, atol=1e-12)


def test_rff_posterior_weight_samplers_have_the_same_moments():
    """the product draws the posterior RFF weights by Matheron's rule in the M-dimensional function space
    (O(F M^2)); the reference factorises F x F matrices (mfdgp_hidden_layer.py:296-307, restated in oracle/rff_oracle.py).
    Both are samplers of ONE Gaussian: their closed-form means and covariances agree to rounding, the product's draws have
    those moments, and so have draws of the restated reference sampler."""
    import torch

    from mobocmf_amd.layers import rff
    from oracle import rff_oracle as R
    rng = np.random.default_rng(3)
    Fn, M, d, s2 = 40, 7, 2, 1e-3
    x = rng.random((M, d))
    W, b = rng.standard_normal((Fn, d)) / 0.4, rng.uniform(0, 2 * np.pi, (Fn, 1))
    Phi = R.layer0_features(x, W, b, 1.3)
    m = rng.standard_normal(M)
    Ls = np.tril(rng.standard_normal((M, M))) * 0.3 + 0.5 * np.eye(M)
    S = Ls @ Ls.T
    mean_ref, cov_ref = R.posterior_moments(m, S, Phi, s2)
    mean_mat, cov_mat = R.matheron_moments(m, S, Phi, s2)
    assert np.abs(mean_ref - mean_mat).max() < 1e-8 * np.abs(mean_ref).max()
    assert np.abs(cov_ref - cov_mat).max() < 1e-8 * np.abs(cov_ref).max()
    # the product's sampler: empirical moments of 20000 draws
    g = torch.Generator().manual_seed(0)
    Pt, mt, Lt = torch.as_tensor(Phi), torch.as_tensor(m), torch.as_tensor(Ls)
    draws = torch.stack([rff._posterior_weights(Pt, mt, Lt, s2, g) for _ in range(20000)]).numpy()
    se = np.sqrt(np.diag(cov_ref) / draws.shape[0])
    assert (np.abs(draws.mean(0) - mean_ref) < 5 * se + 1e-12).all()
    emp = np.cov(draws.T)
    assert np.abs(emp - cov_ref).max() < 0.06 * np.abs(cov_ref).max()
    # the restated reference sampler, same check
    ref = np.stack([R.posterior_weights_reference(m, S, Phi, s2, rng) for _ in range(20000)])
    assert (np.abs(ref.mean(0) - mean_ref) < 5 * se + 1e-12).all()
    assert np.abs(np.cov(ref.T) - cov_ref).max() < 0.06 * np.abs(cov_ref).max()
    # layer >= 1 features: the product's feature map equals the restated one
    f = rng.standard_normal(M)
    W2, b2, Wf = rng.standard_normal((Fn, d)) / 0.7, rng.uniform(0, 2 * np.pi, (Fn, 1)), rng.standard_normal(Fn) / 0.9
    want = R.layer1_features(x, f, W, Wf, W2, b, b2, 1.1, 0.8, 0.05, 0.6)
    T = torch.as_tensor
    got = torch.cat([rff._phi(T(x), T(W), T(b), 1.1) * T(f) * np.sqrt(0.6),
                     rff._phi(torch.cat([T(x), T(f)[:, None]], 1), torch.cat([T(W), T(Wf)[:, None]], 1), T(b), 1.1 * 0.8),
                     rff._phi(T(x), T(W2), T(b2), 0.05)], 0).numpy()
    assert np.abs(got - want).max() < 1e-13
