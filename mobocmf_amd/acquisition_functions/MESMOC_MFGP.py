"""Max-value entropy search on the exact-GP baselines -- host mirror of mobocmf/acquisition_functions/MESMOC_MFGP.py
(``_MES_MFGP.forward`` :38-71, ``MESMOC_MFGP`` :75-157).  SURVEY row N4: a comparison baseline, plain float64 torch;
botorch's ``optimize_acqf`` is replaced by the same batched multi-start projected Adam as in JESMOC_MFDGP."""
import math

import torch

from .JESMOC_MFDGP import optimize_acqf_multistart

CLAMP_LB = torch.finfo(torch.float32).eps


def _ncdf(z):
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))


class _MES_MFGP:

    def __init__(self, fidelity, model, best_value, is_constraint):
        self.fidelity, self.model, self.best_value, self.is_constraint = fidelity, model, best_value, is_constraint

    def forward(self, X):
        """:38-71 -- moments of the truncated Gaussian below the best value, then the noise; constraints: P(feasible)."""
        pred = self.model.predict(X, self.fidelity)
        mean, var = pred.mean, pred.variance
        stdv = var.sqrt()
        z = (self.best_value - mean) / stdv
        if self.is_constraint:
            return 1.0 - _ncdf(z)
        cdf = _ncdf(z).clamp_max(1 - CLAMP_LB)
        pdf = torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
        ratio = pdf / (1.0 - cdf)
        noise = self.model.likelihood.noise.reshape(())
        var_trunc = var * (1 + (z - ratio) * ratio).clamp_min(CLAMP_LB) + noise
        return torch.clamp(0.5 * torch.log(var + noise) - 0.5 * torch.log(var_trunc), min=0.0)

    __call__ = forward


class MESMOC_MFGP:

    def __init__(self, objectives, constraints, input_dim, num_fidelities, best_objective_values, constraint_thresholds,
                 standard_bounds=None):
        self.standard_bounds = standard_bounds
        self.num_fidelities, self.input_dim = num_fidelities, input_dim
        self.objectives, self.constraints = objectives, constraints
        self.best_objective_values, self.constraint_thresholds = best_objective_values, constraint_thresholds
        self.costs_blackboxes, self.acquisition_objs, self.acquisition_cons = {}, {}, {}
        for n_f in range(num_fidelities):
            self.acquisition_objs[n_f], self.acquisition_cons[n_f] = {}, {}
            self.costs_blackboxes[n_f] = {"total": 0.0}

    def add_blackbox(self, fidelity, blackbox_name, cost_evaluation=1.0, is_constraint=False):
        if not is_constraint:
            acq = _MES_MFGP(fidelity, self.objectives[blackbox_name], self.best_objective_values[blackbox_name], False)
            self.acquisition_objs[fidelity][blackbox_name] = acq
            self.costs_blackboxes[fidelity]["total"] += cost_evaluation
            self.costs_blackboxes[fidelity][blackbox_name] = cost_evaluation
        else:
            acq = _MES_MFGP(fidelity, self.constraints[blackbox_name], self.constraint_thresholds[blackbox_name], True)
            self.acquisition_cons[fidelity][blackbox_name] = acq
        return acq

    def coupled_acq(self, X, fidelity):
        """:120-132: sum of the objectives' MES at ``fidelity`` times the probability of feasibility at the HIGHEST one."""
        X = X.double()
        acq = torch.zeros(X.shape[0], dtype=X.dtype, device=X.device)
        for a in self.acquisition_objs[fidelity].values():
            acq = acq + a(X)
        feas = torch.ones(X.shape[0], dtype=X.dtype, device=X.device)
        for a in self.acquisition_cons[self.num_fidelities - 1].values():
            feas = feas * a(X)
        return acq * feas

    def get_nextpoint_coupled(self, iteration=None, verbose=False, maxiter=200):
        """:134-157: best cost-weighted candidate over the fidelities."""
        best = None
        for fidelity in range(self.num_fidelities):
            cand, val = optimize_acqf_multistart(lambda x: self.coupled_acq(x, fidelity=fidelity), self.standard_bounds,
                                                 num_restarts=5, raw_samples=200, maxiter=maxiter)
            w = val / self.costs_blackboxes[fidelity]["total"]
            if best is None or best[0] < w:
                best = (w, cand, fidelity)
        w, cand, fidelity = best
        if verbose:
            print("Iter:", iteration, "Acquisition:", float(w * self.costs_blackboxes[fidelity]["total"]),
                  " Evaluating fidelity", fidelity, "at", cand[0].cpu().numpy())
        return cand[0, :], fidelity
