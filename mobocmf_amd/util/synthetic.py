"""Synthetic MFDGP problems of SURVEY.md section 8(d): inputs, targets,
fidelity labels, inducing inputs, variational parameters, hyper-parameters and explicit eps.

Pure numpy (float64, ``numpy.random.default_rng(seed)``) so the same bytes are produced on the
build container and on the GPU box.  Configs C1..C5 are the BASELINE.json ``configs``.
"""
import numpy as np

CONFIGS = {
    # id: d, layers, M, N, S, outputs
    "C1": dict(d=1, L=2, M=16, N=16, S=4, outputs=3),
    "C2": dict(d=2, L=2, M=128, N=512, S=8, outputs=4),
    "C3": dict(d=8, L=2, M=512, N=8192, S=8, outputs=3),
    "C4": dict(d=32, L=3, M=1024, N=65536, S=16, outputs=1),
    "C5": dict(d=8, L=2, M=1024, N=8192, S=8, outputs=8),
}


def target(x, o):
    """g_o(x) = d^-1/2 sum_k sin(2 pi x_k + k + o); low = g, high = g (1 + .5 cos(2 pi x_0)) + .1 mean(x)."""
    d = x.shape[1]
    k = np.arange(d)[None, :]
    g = np.sin(2.0 * np.pi * x + k + o).sum(1) / np.sqrt(d)
    y_low = g
    y_high = g * (1.0 + 0.5 * np.cos(2.0 * np.pi * x[:, 0])) + 0.1 * x.mean(1)
    return y_low, y_high


def make_problem(d, L, M, N, S, output=0, seed=0, num_fidelities=None, top_fraction=0.25):
    """Returns a dict of numpy float64 arrays (see SURVEY 8(d) 'Synthetic inputs').  ``top_fraction``: share of the rows at
    every fidelity above the lowest (SURVEY / BASELINE.md: a quarter; other values only for sensitivity sweeps)."""
    L = L if num_fidelities is None else num_fidelities
    rng = np.random.default_rng(seed)
    x = rng.random((N, d))
    fid = np.zeros(N)
    # first N/4 rows highest fidelity; for L=3 the next N/4 rows are the middle fidelity
    q = N // 4 if top_fraction == 0.25 else min(int(N * top_fraction), N // max(L - 1, 1))
    for l in range(L - 1, 0, -1):
        lo = (L - 1 - l) * q
        fid[lo:lo + q] = float(l)
    y_low, y_high = target(x, output)
    y = np.where(fid == L - 1, y_high, y_low)
    if L == 3:
        y = np.where(fid == 1, 0.5 * (y_low + y_high), y)
    Zx = x[:M].copy()
    layers = []
    for l in range(L):
        m = 0.1 * rng.standard_normal(M)
        L_S = 0.1 * np.eye(M) + 0.01 * np.tril(rng.standard_normal((M, M)))
        ls = np.full(d, np.sqrt(d) / 2.0)
        if l == 0:
            hyp = {"ls": ls, "alpha": np.array(1.0)}
        else:
            hyp = {"ls1": ls.copy(), "a1": np.array(1.0), "lsf": np.array(1.0), "af": np.array(1.0),
                   "nu": np.array(1.0), "ls2": ls.copy(), "a2": np.array(0.01)}
        layers.append({"hyp": hyp, "m": m, "L_S": L_S})
    noise = [np.array(1e-6)] * (L - 1) + [np.array(1e-2)]
    eps = [None] + [rng.standard_normal(N * S) for _ in range(1, L)]
    samples = [None] + [rng.standard_normal(S) for _ in range(1, L)]
    return {"x": x, "y": y, "fid": fid, "Zx": Zx, "layers": layers, "noise": noise, "eps": eps,
            "samples": samples, "d": d, "L": L, "M": M, "N": N, "S": S}


def forrester_problem(output=0):
    """C1: the exact Forrester data of examples/example_acquisition_mfdgp_forrester/...py:51-62,97-104
    (RNG-free): 12 low-fidelity points on linspace(0,1,12), 4 high at [.1,.3,.5,.7]; rows ordered
    high first; outputs 0/1 = +-Forrester, 2 = sin/cos constraint; standardised with the pooled mean/std."""
    x0 = np.linspace(0, 1.0, 12).reshape(12, 1)
    x1 = np.array([0.1, 0.3, 0.5, 0.7]).reshape(4, 1)

    def f_hi(x):
        return ((6 * x - 2) ** 2) * np.sin(12 * x - 4)

    def f_lo(x):
        return 0.5 * f_hi(x) + 10 * (x - 0.5) + 5

    if output == 0:
        y0, y1 = f_lo(x0), f_hi(x1)
    elif output == 1:
        y0, y1 = -f_lo(x0), -f_hi(x1)
    else:
        y0, y1 = np.sin(x0 * np.pi * 2.5), np.cos(x1 * np.pi * 2.5)
    mu, sd = np.mean(np.vstack((y1, y0))), np.std(np.vstack((y1, y0)))
    y0, y1 = (y0 - mu) / sd, (y1 - mu) / sd
    x = np.vstack((x1, x0))
    y = np.vstack((y1, y0))[:, 0]
    fid = np.concatenate((np.ones(4), np.zeros(12)))
    return x, y, fid


def model_from_problem(prob, num_samples_for_training=None, num_samples_for_acquisition=None, device="cuda",
                       **model_kwargs):
    """``mobocmf_amd.models.MFDGP`` carrying exactly the parameters of a problem dict (section 8(d): parameters are
    explicit inputs so that parity / throughput runs do not depend on the init heuristics)."""
    import torch

    from .. import gp
    from ..models import MFDGP, TL

    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    S = prob["S"]
    model = MFDGP(t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], num_fidelities=prob["L"],
                  type_lengthscale=TL.ONES, inducing_points=t(prob["Zx"]),
                  num_samples_for_acquisition=num_samples_for_acquisition or S,
                  num_samples_for_training=num_samples_for_training or S, **model_kwargs)
    model.double()
    with torch.no_grad():
        for l, lay in enumerate(prob["layers"]):
            layer = getattr(model, f"hidden_layer_{l}")
            h, cm = lay["hyp"], layer.covar_module
            if l == 0:
                cm.base_kernel.lengthscale = t(h["ls"])
                cm.outputscale = t(h["alpha"])
            else:
                k1, kf = cm.kernels[0].kernels[0], cm.kernels[0].kernels[1].kernels[1]
                kl, k2 = cm.kernels[0].kernels[1].kernels[0], cm.kernels[1]
                k1.base_kernel.lengthscale, k1.outputscale = t(h["ls1"]), t(h["a1"])
                kf.base_kernel.lengthscale, kf.outputscale = t(h["lsf"]), t(h["af"])
                k2.base_kernel.lengthscale, k2.outputscale = t(h["ls2"]), t(h["a2"])
                kl.variance = t(h["nu"])
                layer.samples.copy_(t(prob["samples"][l]).reshape(-1, 1))
            vd = layer.variational_strategy._variational_distribution
            vd.variational_mean.copy_(t(lay["m"]))
            vd.chol_variational_covar.copy_(t(lay["L_S"]))
            lik = getattr(model, f"hidden_layer_likelihood_{l}")
            lik.raw_noise_constraint = gp.Interval(1e-8, 1.0)
            lik.noise = t(prob["noise"][l])
    return model.to(device)
