"""Independent numpy/scipy textbook-formula implementation of the 2-layer MFDGP ELBO and
predictive moments (dense inverses, slogdet, explicit kernel loops).  Shares no code with
oracle/mfdgp_oracle.py; pins the oracle's algebra."""
import numpy as np
import torch

from oracle import mfdgp_oracle as O
from tests.helpers import oracle_state, small_problem, to_t

JIT = 1e-6


def k0(h, a, b):
    return float(h["alpha"]) * np.exp(-0.5 * np.sum(((a - b) / h["ls"]) ** 2))


def k1(h, a, b):
    xa, fa, xb, fb = a[:-1], a[-1], b[:-1], b[-1]
    e1 = np.exp(-0.5 * np.sum(((xa - xb) / h["ls1"]) ** 2))
    e2 = np.exp(-0.5 * np.sum(((xa - xb) / h["ls2"]) ** 2))
    ef = np.exp(-0.5 * (fa - fb) ** 2 / float(h["lsf"]) ** 2)
    return float(h["a1"]) * e1 * (float(h["nu"]) * fa * fb + float(h["af"]) * ef) + float(h["a2"]) * e2


def gram_np(k, h, A, B):
    return np.array([[k(h, a, b) for b in B] for a in A])


def layer_np(k, h, X, Z, m, LS):
    Kt = gram_np(k, h, Z, Z) + JIT * np.eye(len(Z))
    Ki = np.linalg.inv(Kt)
    Kzx = gram_np(k, h, Z, X)
    S = np.tril(LS) @ np.tril(LS).T
    mu = Kzx.T @ Ki @ m
    knn = np.array([k(h, x, x) for x in X])
    q = np.einsum("in,ij,jn->n", Kzx, Ki, Kzx)
    r = np.einsum("in,ij,jn->n", Kzx, Ki @ S @ Ki, Kzx)
    var = np.maximum(knn - q, 0.0) + r
    kl = 0.5 * (np.trace(Ki @ S) + m @ Ki @ m - len(Z) + np.linalg.slogdet(Kt)[1] - np.linalg.slogdet(S)[1])
    return mu, np.maximum(var, 1e-10), kl


def elp_np(y, mu, var, tau):
    return -0.5 * (((y - mu) ** 2 + var) / tau + np.log(tau) + np.log(2 * np.pi))


def elbo_np(prob, S):
    x, y, fid = prob["x"], prob["y"], prob["fid"]
    l0, l1 = prob["layers"]
    Z0 = prob["Zx"]
    mu0, v0, kl0 = layer_np(k0, l0["hyp"], x, Z0, l0["m"], l0["L_S"])
    f = np.repeat(mu0, S) + np.sqrt(np.repeat(v0, S)) * prob["eps"][1]
    X1 = np.concatenate([np.repeat(x, S, 0), f[:, None]], 1)
    Z1 = np.concatenate([Z0, l0["m"][:, None]], 1)
    mu1, v1, kl1 = layer_np(k1, l1["hyp"], X1, Z1, l1["m"], l1["L_S"])
    t0, t1 = float(prob["noise"][0]), float(prob["noise"][1])
    d0 = elp_np(y, mu0, v0, t0)[fid == 0].sum()
    d1 = elp_np(np.repeat(y, S), mu1, v1, t1)[np.repeat(fid, S) == 1].sum() / S
    return d0 + d1 - (kl0 + kl1), kl0 + kl1, (mu0, v0, mu1, v1)


def test_elbo_and_moments_match_textbook_numpy():
    for seed in (0, 1, 2):
        prob = small_problem(d=2, M=8, N=12, S=3, seed=seed)
        st = oracle_state(prob)
        x, y, fid = to_t(prob["x"]), to_t(prob["y"]), to_t(prob["fid"])
        e, skl = O.elbo(st, x, y, fid, eps=[None, to_t(prob["eps"][1])], S=3)
        e_np, kl_np, (mu0, v0, mu1, v1) = elbo_np(prob, 3)
        assert abs(float(e) - e_np) / abs(e_np) < 1e-7
        assert abs(float(skl) - kl_np) / abs(kl_np) < 1e-7
        outs = O.model_forward(st, x, eps=[None, to_t(prob["eps"][1])], S=3)
        np.testing.assert_allclose(outs[0][0].numpy(), mu0, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(outs[0][1].numpy(), v0, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(outs[1][0].numpy(), mu1, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(outs[1][1].numpy(), v1, rtol=1e-4, atol=1e-6)


def test_predict_for_acquisition_matches_numpy():
    prob = small_problem(d=2, M=8, N=12, S=4, seed=5)
    st = oracle_state(prob)
    T, S = 5, 4
    X = np.random.default_rng(0).random((T, 2))
    l0, l1 = prob["layers"]
    Xt = np.repeat(X, S, 0)
    mu0, v0, _ = layer_np(k0, l0["hyp"], Xt, prob["Zx"], l0["m"], l0["L_S"])
    f = mu0 + np.sqrt(v0) * np.tile(prob["samples"][1], T)
    X1 = np.concatenate([Xt, f[:, None]], 1)
    Z1 = np.concatenate([prob["Zx"], l0["m"][:, None]], 1)
    mu1, v1, _ = layer_np(k1, l1["hyp"], X1, Z1, l1["m"], l1["L_S"])
    v1 = v1 + float(prob["noise"][1])
    mus = mu1.reshape(T, S).mean(1)
    vs = (v1 + mu1 ** 2).reshape(T, S).mean(1) - mus ** 2
    # train-branch (clamped) variant so the numpy max(.,0) matches
    om, ov = O.predict_for_acquisition(st, to_t(X), 1, S, training=True)
    np.testing.assert_allclose(om.numpy(), mus, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ov.numpy(), vs, rtol=1e-4, atol=1e-6)
