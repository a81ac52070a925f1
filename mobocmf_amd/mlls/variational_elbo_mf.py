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

    last_neg_elbo = None      # -elbo of the last fused call (no gradient): the loss value a training step reports

    def forward(self, l_approximate_dist_f, target, fidelities, include_kl_term=True):
        assert target.shape[0] <= target.shape[1]        # (1, B), as the reference checks
        num_batch = target.shape[1]
        y = target.reshape(-1)
        fid = fidelities.reshape(-1).to(y.dtype)
        n_lev = min(self.num_fidelities, len(l_approximate_dist_f))
        layers = []
        for i in range(n_lev):
            dist = l_approximate_dist_f[i]
            if dist is None:
                # rows with fid != i contribute nothing (an empty mask gives 0, as the reference's skip at :33)
                layers.append(None)
                continue
            likelihood = getattr(self.model, self.model.name_hidden_layer_likelihood + str(i))
            mean, var = dist.mean.reshape(-1), dist.variance.reshape(-1)
            # MFDGP.forward(..., rows=...) evaluates layer i on the first rows[i] rows of the batch only (those that can reach
            # the loss) and says so on the distribution; otherwise every layer holds the whole batch
            rows = getattr(dist, "batch_rows", None)
            rows = num_batch if rows is None else int(rows)
            div = mean.numel() // max(rows, 1)
            c = likelihood.raw_noise_constraint
            if type(c) is gp.Interval and math.isfinite(c.upper_bound) and c.upper_bound > c.lower_bound:
                layers.append((mean, var, likelihood.raw_noise, div, c.lower_bound, c.upper_bound, rows))
            else:
                layers.append((mean, var, likelihood.noise, div, 0.0, 0.0, rows))
        if all(lay is None for lay in layers):
            if not include_kl_term:
                return 0.0
            return F.elbo_combine([], self.model.variational_strategy.kl_terms(), num_batch / self.num_data)
        kls = self.model.variational_strategy.kl_terms() if include_kl_term else []
        if n_lev <= F.ELBO_MAX_LAYERS and len(kls) <= F.ELBO_MAX_LAYERS:
            # data terms of every fidelity + the KL tail: one reduction launch + a one-block tail (the same in backward)
            elbo, skl, neg = F.elbo_fused(layers, y, fid, kls, num_batch / self.num_data if include_kl_term else 0.0)
            self.last_neg_elbo = neg
            return (elbo, skl) if include_kl_term else elbo
        self.last_neg_elbo = None
        if any(lay is not None and lay[6] != num_batch for lay in layers):
            raise ValueError("row-pruned layer outputs need the fused ELBO (at most %d fidelities)" % F.ELBO_MAX_LAYERS)
        data_terms = [F.elbo_data(lay[0], lay[1], y, fid, lay[2], float(i), div=lay[3],
                                  interval=(lay[4], lay[5]) if lay[5] > lay[4] else None)
                      for i, lay in enumerate(layers) if lay is not None]
        # the tail -- sum of the data terms, sum of the layer KLs, batch/num_data scaling, the difference -- is one launch
        if not include_kl_term:
            return F.elbo_combine(data_terms, [], 0.0)[0]
        return F.elbo_combine(data_terms, kls, num_batch / self.num_data)
