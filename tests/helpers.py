"""Shared test helpers: synthetic problems -> oracle state (torch float64, CPU)."""
import numpy as np
import torch

from mobocmf_amd.util import synthetic


def to_t(a, requires_grad=False):
    t = torch.as_tensor(np.asarray(a), dtype=torch.float64).clone()
    return t.requires_grad_(requires_grad)


def oracle_state(prob, requires_grad=False):
    layers = []
    for lay in prob["layers"]:
        hyp = {k: to_t(v, requires_grad) for k, v in lay["hyp"].items()}
        layers.append({"hyp": hyp, "m": to_t(lay["m"], requires_grad), "L_S": to_t(lay["L_S"], requires_grad)})
    return {"Zx": to_t(prob["Zx"]), "layers": layers,
            "noise": [to_t(v, requires_grad) for v in prob["noise"]],
            "samples": [None if s is None else to_t(s) for s in prob["samples"]]}


def state_leaves(state):
    leaves = []
    for lay in state["layers"]:
        leaves += [lay["hyp"][k] for k in sorted(lay["hyp"])] + [lay["m"], lay["L_S"]]
    return leaves + list(state["noise"])


def small_problem(d=2, L=2, M=8, N=12, S=3, output=0, seed=0):
    return synthetic.make_problem(d=d, L=L, M=M, N=N, S=S, output=output, seed=seed)


# ---- analytic "function samples" for the MOOP fixtures (same callables on the reference side and in the tests)
def moop_callable(kind, a):
    """f(x, gradient=False) with the reference's calling convention (values (n,), gradient (d,) for one point)."""
    a = np.asarray(a, dtype=np.float64)

    def f(x, gradient=False):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        if kind == "quad":                      # sum (x - a)^2
            return 2.0 * (x[0] - a) if gradient else ((x - a) ** 2).sum(1)
        if kind == "lin":                       # a[0] + a[1:] . x
            return a[1:].copy() if gradient else a[0] + x @ a[1:]
        if kind == "wave":                      # sum sin(3 x + a)
            return 3.0 * np.cos(3.0 * x[0] + a) if gradient else np.sin(3.0 * x + a).sum(1)
        raise ValueError(kind)
    return f
