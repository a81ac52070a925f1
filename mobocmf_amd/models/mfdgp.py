"""Multi-fidelity deep GP -- host mirror of mobocmf/models/mfdgp.py (same constructor, attributes,
``forward`` / ``predict`` / ``predict_for_acquisition`` / ``fix_variational_hypers*`` surface).

Extensions (documented in DESIGN.md; defaults reproduce the reference):
  * ``num_inducing`` / ``inducing_points``: decouple M from N (reference: Z = all training inputs, F6).
  * ``num_samples_for_training`` (S): S-fold sample replication through layers >= 1 during training
    (reference: exactly one sample, F7); the ELBO then averages the S samples of each row.
  * more than two fidelities: Z~_l = [Z_x, m_{l-1}] for every l >= 1 (reference works for 2 only, F8).
"""
from enum import Enum

import contextlib

import numpy as np
import torch
from torch import nn

from .. import functional as F
from .. import gp
from ..layers.mfdgp_hidden_layer import MFDGPHiddenLayer
from ..util.util import compute_dist, triu_indices


class TL(Enum):  # type of lengthscale (mfdgp.py:15-18)
    ONES = 1
    MEDIAN = 2
    CENTESIMAL = 3


class _DeepGPVariationalStrategy:
    def __init__(self, model):
        self.model = model

    def kl_terms(self):
        """The per-layer KLs (every ApproximateGP sub-module counted once)."""
        return [getattr(self.model, self.model.name_hidden_layer + str(i)).variational_strategy.kl_divergence()
                for i in range(self.model.num_hidden_layers)]

    def kl_divergence(self):
        return sum(self.kl_terms())


class MFDGP(nn.Module):

    def __init__(self, x_train, y_train, fidelities, num_fidelities, type_lengthscale=TL.MEDIAN,
                 num_samples_for_acquisition=25, previously_trained_model=None, ini_inducing_using_layer_0=False,
                 use_only_highest_fidelity=False, init_params_to_prior_and_fix_them=False,
                 num_inducing=None, inducing_points=None, num_samples_for_training=1, median_mode="reference"):
        super().__init__()
        self.init_params_to_prior_and_fix_them = init_params_to_prior_and_fix_them
        self._eval_mode = False
        self.num_samples_for_acquisition = num_samples_for_acquisition
        self.num_samples_for_training = num_samples_for_training
        self.use_only_highest_fidelity = use_only_highest_fidelity
        self.ini_inducing_using_layer_0 = ini_inducing_using_layer_0
        self.median_mode = median_mode
        self.input_dims = x_train.shape[-1]
        x_train, y_train, fidelities = x_train.detach().cpu(), y_train.detach().cpu(), fidelities.detach().cpu()
        y_high_std = np.std(y_train[(fidelities == num_fidelities - 1).flatten()].numpy())

        if inducing_points is None and num_inducing is not None:
            inducing_points = x_train[:num_inducing]
        self._Zx_user = inducing_points

        hidden_layers = []
        for i in range(num_fidelities):
            Z, u = self.find_good_initial_inducing_points_and_values(x_train, y_train, fidelities, i)
            init_ls = self.get_init_lengthscale(type_lengthscale, inputs=x_train[(fidelities == i).flatten(), :])
            prev = None if previously_trained_model is None else \
                getattr(previously_trained_model, previously_trained_model.name_hidden_layer + str(i))
            hidden_layers.append(MFDGPHiddenLayer(
                input_dims=self.input_dims + (0 if i == 0 else 1), num_layer=i, inducing_points=Z, inducing_values=u,
                num_fidelities=num_fidelities, init_lengthscale=init_ls, y_high_std=y_high_std,
                num_samples_for_acquisition=num_samples_for_acquisition, previously_trained_layer=prev,
                init_params_to_prior_and_fix_them=init_params_to_prior_and_fix_them,
                previous_layer_in_hierarchy=hidden_layers[-1] if i > 0 else None,
                only_hf=use_only_highest_fidelity and i > 0))

        self.name_hidden_layer = "hidden_layer_"
        self.name_hidden_layer_likelihood = "hidden_layer_likelihood_"
        self.name_hidden_layer_likelihood_noiseless = "hidden_layer_likelihood_noiseless_"
        self.num_hidden_layers = num_fidelities
        self.num_fidelities = num_fidelities

        for i, hidden_layer in enumerate(hidden_layers):
            y_std = np.std(y_train[(fidelities == i).flatten()].numpy())
            setattr(self, self.name_hidden_layer + str(i), hidden_layer)
            likelihood = gp.GaussianLikelihood(noise_constraint=gp.Interval(lower_bound=1e-8, upper_bound=0.1 * y_std))
            likelihood.noise = 1e-2 * y_high_std if i == self.num_fidelities - 1 else 1e-6
            setattr(self, self.name_hidden_layer_likelihood + str(i), likelihood)
        self.variational_strategy = _DeepGPVariationalStrategy(self)

    # ------------------------------------------------------------------ init heuristics
    def clip_inducing_values(self, x_0, x_1, y_1):
        """Value of the nearest x_1 point for every x_0 point (mfdgp.py:125-135; unused by the reference's own callers)."""
        d2 = (x_0 * x_0).sum(1, keepdim=True) - 2.0 * x_0 @ x_1.T + (x_1 * x_1).sum(1)[None, :]
        return y_1[torch.argmin(d2, dim=1)]

    def get_init_lengthscale(self, type_lengthscale, inputs=None):
        if type_lengthscale == TL.ONES:
            return torch.ones(self.input_dims)
        if type_lengthscale == TL.MEDIAN:
            n = inputs.shape[0]
            if n > 20000:
                raise ValueError("TL.MEDIAN needs the full n x n distance matrix; pass type_lengthscale=TL.ONES (or "
                                 "set the lengthscales explicitly) for n > 20000")
            dists = compute_dist(inputs)
            # as written at mfdgp.py:143-144 the 2 x K index tensor selects ROWS: (2, K, n) elements, i.e.
            # O(n^3) memory -- reproduced only while that fits in ~1 GiB, otherwise the intended upper-triangle median
            if self.median_mode == "reference" and n * (n - 1) * n * 8 < 2 ** 30:
                return torch.sqrt(torch.median(dists[triu_indices(inputs.shape[0], 1)]))
            rows, cols = torch.triu_indices(inputs.shape[0], inputs.shape[0], offset=1)
            return torch.sqrt(torch.median(dists[rows, cols]))
        if type_lengthscale == TL.CENTESIMAL:
            return 0.01 * torch.ones(self.input_dims)
        raise ValueError("Wrong type of lengthscale.")

    def find_good_initial_inducing_points_and_values(self, x_train, y_train, fidelities, layer):
        """mfdgp.py:290-317 with the O(M N^2) loop replaced by one argmin over distances."""
        if self._Zx_user is not None:
            inducing_points = self._Zx_user
        elif self.use_only_highest_fidelity:
            inducing_points = x_train[fidelities[:, 0] == layer, :]
        else:
            inducing_points = x_train
        sel = fidelities[:, 0] == layer
        xs, ys = x_train[sel, :], y_train[sel, :]
        d2 = (xs ** 2).sum(1)[None, :] - 2.0 * inducing_points @ xs.T + (inducing_points ** 2).sum(1)[:, None]
        inducing_values = torch.zeros(inducing_points.shape[0])                 # float32, as the reference (B.4)
        inducing_values[:] = ys[torch.argmin(d2, 1), 0]
        if layer != 0:
            inducing_points = torch.cat((inducing_points, inducing_values[:, None].to(inducing_points.dtype)), 1)
        return inducing_points, inducing_values

    # ------------------------------------------------------------------ modes
    def _layers(self):
        return [getattr(self, self.name_hidden_layer + str(i)) for i in range(self.num_hidden_layers)]

    def train_mode(self):
        for layer in self._layers():
            layer.train_mode()
        self._eval_mode = False

    def eval_mode(self):
        for layer in self._layers():
            layer.eval_mode()
        self._eval_mode = True

    def clear_kl_cache(self):
        """Forget the KL memoised by the last forward (it keeps that iteration's autograd graph alive)."""
        for layer in self._layers():
            layer.variational_strategy._kl_cache = None

    def set_check_pd(self, value):
        """False: no host sync after the Cholesky on the training fast path (NaNs then surface in the loss)."""
        for layer in self._layers():
            layer.check_pd = bool(value)

    # ------------------------------------------------------------------ forward
    def forward(self, inputs, max_fidelity=None, eps=None, want_dx=False, _xdiv=None, rows=None):
        """List of per-layer predictive distributions (mfdgp.py:174-196).

        train mode: layer 0 sees the N rows; layers >= 1 see N * num_samples_for_training rows (row n*S+s is
        sample s of input row n).  ``eps``: optional list, eps[l] (N*S values) for layer l >= 1.
        eval_mode: the caller tiled the inputs (mfdgp.py:248), eps = the layer's fixed ``samples``.

        ``rows`` (training fast path, util/graphed_step.py): non-increasing counts, layer l is evaluated on the FIRST
        rows[l] input rows only and its distribution says so (``batch_rows``).  The reference evaluates every layer at
        every row and the ELBO then keeps, per layer, the rows of that layer's fidelity (variational_elbo_mf.py:33-38): a
        row of fidelity f reaches the loss through layers 0..f only, so with the batch ordered by descending fidelity layer
        l needs the rows of fidelity >= l -- a prefix -- and the rest of its work is dead.  ``eps[l]`` then holds
        rows[l] * S values.  Same ELBO, same gradients.
        """
        num_layers = self.num_hidden_layers if max_fidelity is None else max_fidelity + 1
        if rows is not None:
            rows = [int(r) for r in rows[:num_layers]]
            if len(rows) != num_layers or any(a < b for a, b in zip(rows, rows[1:])) or rows[0] > inputs.shape[0] or \
                    rows[-1] < 1 or want_dx or self._eval_mode:
                raise ValueError("rows: one non-increasing count per layer, 1 <= rows[l] <= N (training forward only)")
        if self.training:
            self.clear_kl_cache()      # drop the previous iteration's graph (and its AccumulateGrad nodes) up front
        S = 1 if self._eval_mode else self.num_samples_for_training
        if _xdiv is not None:
            S = _xdiv
        l_outputs = [None] * num_layers
        output_layer = None
        if self._frozen is not None:
            chains = self._frozen_chains(num_layers)
        else:
            chains = self._batched_chains(inputs, num_layers) or self._launch_chains(inputs, num_layers, S, want_dx, rows)
        for i in range(num_layers):
            hidden_layer = getattr(self, self.name_hidden_layer + str(i))
            x_i = inputs if rows is None else inputs[:rows[i]]
            if i == 0:
                output_layer = hidden_layer(x_i, want_dx=want_dx, chain=chains[i])
            else:
                if self.use_only_highest_fidelity:
                    output_layer = output_layer.mean * 0.0
                    if rows is not None:      # the previous layer's entries of this layer's rows
                        output_layer = output_layer.reshape(-1)[:rows[i] * (1 if i == 1 else S)]
                output_layer = hidden_layer(x_i, output_layer, eps=None if eps is None else eps[i], xdiv=S,
                                            want_dx=want_dx, chain=chains[i])
            if rows is not None:
                output_layer.batch_rows = rows[i]
            l_outputs[i] = output_layer
        return l_outputs

    _warned_off = False
    _frozen = None

    @contextlib.contextmanager
    def frozen_chains(self):
        """Inside this context the parameters are constants (acquisition optimisation against a fitted model,
        JESMOC_MFDGP.py:137-184): the CHAIN half of every layer -- K_mm, its Cholesky and inverse, U, a: ~150 launches
        of latency-bound M x M work -- is computed once per (layer, train/eval branch) and reused by every call, and
        backward yields input gradients only.  Results are identical to recomputing the chain at every call."""
        self._frozen = {}
        try:
            yield self
        finally:
            self._frozen = None

    def _frozen_chains(self, num_layers):
        out = []
        for i in range(num_layers):
            layer = getattr(self, self.name_hidden_layer + str(i))
            key = (i, layer.training)
            if key not in self._frozen:
                self._frozen[key] = layer.freeze_chain()
            out.append(self._frozen[key])
        return out
    batch_chains = True                   # all layers' CHAIN halves as one z-batched sequence of launches (fast path)

    def _batched_chains(self, inputs, num_layers):
        """The parameter-only (CHAIN) halves of all layers in ONE batch of launches (functional.layers_chain): the layers'
        chains are independent -- Z~_l = [Z_x, m_{l-1}] is made of parameters -- and each is a serial string of
        latency-bound M x M kernels, so batching divides that string's length by the number of layers.  Only without
        per-call host checks (``set_check_pd(False)``: graphed step / bench / fitter fast path) and with one M for all
        layers; otherwise None (the layers then run one by one)."""
        layers = [getattr(self, self.name_hidden_layer + str(i)) for i in range(num_layers)]
        if not (self.batch_chains and not self.overlap_chains and inputs.is_cuda and 2 <= num_layers <= 4
                and not any(l.check_pd for l in layers)):
            return None
        params, kinds, jit, infos, hyps = [], [], [], [], []
        # the constrained hyper-parameters of ALL layers from one launch (and one backward launch for every raw gradient)
        packed = gp.pack_hypers_many([(layer.covar_module, layer.kind) for layer in layers])
        for li, layer in enumerate(layers):
            vs = layer.variational_strategy
            vd = vs._variational_distribution
            if vs.Zx.shape[0] != layers[0].variational_strategy.Zx.shape[0] or (layer.kind == 1 and vs.zf is None):
                return None
            if layer._info is None or layer._info.device != inputs.device:
                layer._info = torch.zeros((), dtype=torch.int32, device=inputs.device)
            hyp = packed[li] if packed is not None else gp.pack_hypers(layer.covar_module, layer.kind)
            hyps.append(hyp)
            params.append((vs.Zx, vs.zf if layer.kind == 1 else None, hyp, vd.variational_mean, vd.chol_variational_covar))
            kinds.append(layer.kind)
            jit.append(vs.jitter_val)
            infos.append(layer._info)
        CB = F.layers_chain(params, kinds, 0 if self.training else 1, jit, infos)
        return [(CB, z, hyps[z]) for z in range(num_layers)]

    overlap_chains = False                # opt-in (HIP-graph replay of forked captures is slower on ROCm 7.2)
    OVERLAP_MAX_ROWS = 1 << 21            # above this the private backward scratch per layer is not worth its memory

    def _launch_chains(self, inputs, num_layers, S, want_dx, rows=None):
        """The parameter-only (CHAIN) half of every layer, issued up front on a side stream so the latency-bound M x M
        work of layer l runs under the grid-filling panel work of the other layers -- forward here, and backward through
        autograd, which replays each half on the stream its forward used.  Only without per-call host checks
        (``set_check_pd(False)``: graphed / bench / fitter fast path); otherwise the layers run serially."""
        layers = [getattr(self, self.name_hidden_layer + str(i)) for i in range(num_layers)]
        if not (self.overlap_chains and inputs.is_cuda and num_layers > 1 and not any(l.check_pd for l in layers)
                and inputs.shape[0] * S <= self.OVERLAP_MAX_ROWS):      # (rows: an upper bound)
            return [None] * num_layers
        main = torch.cuda.current_stream(inputs.device)
        side = F.side_stream_for(main)
        if not MFDGP._warned_off and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # parameters feed both halves, i.e. both streams, on purpose; the engine orders the accumulation
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            MFDGP._warned_off = True
        chains = []
        with torch.cuda.stream(side):
            for i, layer in enumerate(layers):
                nb = inputs.shape[0] if rows is None else rows[i]
                chains.append(layer.launch_chain(nb * (1 if i == 0 else S), 1 if i == 0 else S, want_dx, main, side))
        return chains

    def fix_variational_hypers(self, value):
        for i in range(self.num_hidden_layers):
            getattr(self, self.name_hidden_layer_likelihood + str(i)).raw_noise.requires_grad = not value
        for layer in self._layers():
            layer.variational_strategy._variational_distribution.chol_variational_covar.requires_grad = not value

    def fix_variational_hypers_cond(self, value):
        for i in range(self.num_hidden_layers):
            getattr(self, self.name_hidden_layer_likelihood + str(i)).raw_noise.requires_grad = not value
        for layer in self._layers():
            for _, param in layer.covar_module.named_parameters():
                param.requires_grad = not value

    # ------------------------------------------------------------------ prediction
    def predict(self, test_x, fidelity_layer=0, want_dx=False, _xdiv=None):
        assert 0 <= fidelity_layer < self.num_fidelities
        likelihood = getattr(self, self.name_hidden_layer_likelihood + str(fidelity_layer))
        preds = likelihood(self(test_x, max_fidelity=fidelity_layer, want_dx=want_dx, _xdiv=_xdiv)[fidelity_layer])
        return preds.mean, preds.variance

    def predict_for_acquisition(self, test_x, fidelity_layer=0):
        """Moments over the S = num_samples_for_acquisition fixed samples (mfdgp.py:237-262).  Layer 0 is
        evaluated once per test point (its S tiled copies are identical); layers >= 1 see the S-fold rows."""
        if len(test_x.shape) > 2:
            assert test_x.shape[1] == 1
            test_x = test_x[:, 0, :]
        S = self.num_samples_for_acquisition
        self.eval_mode()
        try:
            mus_tilde, vars_tilde = self.predict(test_x, fidelity_layer=fidelity_layer,
                                                 want_dx=test_x.requires_grad, _xdiv=S)
        finally:
            self.train_mode()
        if fidelity_layer == 0:          # no sampling below layer 0: the S copies coincide
            return mus_tilde.reshape(-1), vars_tilde.reshape(-1)
        return F.acq_moments(mus_tilde, vars_tilde, S)

    # ------------------------------------------------------------------ function sampling (RFF): SURVEY row N2
    def sample_function_from_each_layer(self, nFeatures=500, generator=None):
        """One posterior function sample per layer (mfdgp.py:264-275): list of callables f(x: ndarray) -> (n,)."""
        from ..layers import rff
        result, prev = [], None
        for layer in self._layers():
            prev = rff.sample_from_posterior(layer, self.input_dims, prev, nFeatures=nFeatures, generator=generator,
                                             device=self._sample_device())
            result.append(prev)
        return result

    def _sample_device(self):
        dev = next(self.parameters()).device
        return dev if dev.type == "cuda" else None

    def sample_function_from_prior_each_layer(self, nFeatures=500, generator=None):
        """Prior function samples used to build synthetic problems (mfdgp.py:277-288)."""
        from ..layers import rff
        result, prev = [], None
        for layer in self._layers():
            prev = rff.sample_from_prior(layer, self.input_dims, prev, nFeatures=nFeatures, generator=generator,
                                         device=self._sample_device())
            result.append(prev)
        return result
