"""ELBO of the multi-fidelity DGP -- host mirror of mobocmf/mlls/variational_elbo_mf.py:15-51."""
import math

import torch
from torch import nn

from .. import functional as F
from .. import gp


class VariationalELBOMF(nn.Module):
    """``elbo(l_dists, target (1,B), fidelities (B,1), include_kl_term=True)`` -> ``(elbo, scaled_kl)`` or the
    data term alone.  The per-fidelity masked expected log-likelihood is one fused HIP reduction; with S
    training samples per row (layer rows = B*S) it is averaged over the S samples."""

    def __init__(self, model, num_data, num_fidelities):
        super().__init__()
        object.__setattr__(self, "model", model)
        self.num_data = num_data
        self.num_fidelities = num_fidelities

    def forward(self, l_approximate_dist_f, target, fidelities, include_kl_term=True):
        assert target.shape[0] <= target.shape[1]        # (1, B), as the reference checks
        num_batch = target.shape[1]
        y = target.reshape(-1)
        fid = fidelities.reshape(-1).to(y.dtype)
        data_terms = []
        for i in range(min(self.num_fidelities, len(l_approximate_dist_f))):
            dist = l_approximate_dist_f[i]
            if dist is None:
                continue
            likelihood = getattr(self.model, self.model.name_hidden_layer_likelihood + str(i))
            mean, var = dist.mean.reshape(-1), dist.variance.reshape(-1)
            # rows with fid != i contribute nothing (an empty mask gives 0, as the reference's skip at :33)
            c = likelihood.raw_noise_constraint
            if type(c) is gp.Interval and math.isfinite(c.upper_bound) and c.upper_bound > c.lower_bound:
                data_terms.append(F.elbo_data(mean, var, y, fid, likelihood.raw_noise, float(i),
                                              div=mean.numel() // num_batch, interval=(c.lower_bound, c.upper_bound)))
            else:
                data_terms.append(F.elbo_data(mean, var, y, fid, likelihood.noise, float(i), div=mean.numel() // num_batch))
        # the tail -- sum of the data terms, sum of the layer KLs, batch/num_data scaling, the difference -- is one launch
        if not include_kl_term:
            return F.elbo_combine(data_terms, [], 0.0)[0] if data_terms else 0.0
        kls = self.model.variational_strategy.kl_terms()
        return F.elbo_combine(data_terms, kls, num_batch / self.num_data)
