"""One variational GP layer of the MFDGP -- host mirror of mobocmf/layers/mfdgp_hidden_layer.py.

Same constructor arguments, attributes and call convention as the reference class (``:26-188``,
``:232-286``, ``:520-559``); the arithmetic that the reference reaches through GPyTorch's
``UnwhitenedVariationalStrategy`` runs in the HIP library (mobocmf_amd.functional.layer_forward).
"""
import warnings

import torch
from torch import nn

from .. import functional as F
from .. import gp


class NotPSDError(RuntimeError):
    """K_mm + jitter not positive definite after the jitter retries (gpytorch NotPSDError)."""


class UnwhitenedVariationalStrategy(nn.Module):
    """Parameter holder for q(u) and the (fixed) inducing inputs of layer 0."""

    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=False,
                 jitter_val=None):
        super().__init__()
        object.__setattr__(self, "model", model)       # not registered: avoids a module cycle (as GPyTorch does)
        if learn_inducing_locations:
            self.register_parameter("_inducing_points", nn.Parameter(inducing_points.clone()))
        else:
            self.register_buffer("_inducing_points", inducing_points.clone())
        self._variational_distribution = variational_distribution
        self.jitter_val = F.JITTER if jitter_val is None else jitter_val
        self.register_buffer("variational_params_initialized", torch.tensor(1))
        self._kl_cache = None

    def __getstate__(self):
        # the memoised KL is a graph tensor: never copied / pickled (deepcopy + dill of the fitter must work)
        state = self.__dict__.copy()
        state["_kl_cache"] = None
        state.pop("_Zx_contig", None)
        return state

    @property
    def inducing_points(self):
        return self._inducing_points

    @property
    def variational_distribution(self):
        return self._variational_distribution()

    # pieces the HIP call consumes
    @property
    def Zx(self):
        return self._inducing_points

    @property
    def zf(self):
        return None

    def kl_divergence(self):
        """KL(q(u) || p(u)); reuses the value the last training forward produced (GPyTorch memoises the
        prior the same way), otherwise evaluates it with a one-row layer call."""
        if self._kl_cache is not None:
            return self._kl_cache
        return self.model._moments(self.Zx[:1], None if self.zf is None else self.zf[:1].detach(), 1, False)[2]


class MFDGUnwhitenedVariationalStrategy(UnwhitenedVariationalStrategy):
    """Layers >= 1: Z~ = [Z_x, mean of the previous layer at Z_x] recomputed on every access
    (mfdgp_hidden_layer.py:542-559).  With Z_x shared by all layers GPyTorch's ``torch.equal`` shortcut makes
    that mean exactly the previous layer's variational mean m_{l-1}; gradient flows into it.
    For l >= 2 the same rule is the documented extension."""

    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True,
                 jitter_val=None, previous_layer=None):
        super().__init__(model, inducing_points, variational_distribution, learn_inducing_locations, jitter_val)
        self.previous_layer = previous_layer            # registered sub-module, as in the reference

    @property
    def original_inducing_points(self):
        return self._inducing_points

    @property
    def Zx(self):
        """The x columns of Z~ as a CONTIGUOUS tensor (the C-ABI takes dense rows): the column slice of the stored
        (M, d + 1) matrix is copied once and reused while that matrix is unchanged (same storage, same version counter) --
        not twice per forward."""
        if self.previous_layer is None:
            return self._inducing_points
        Z = self._inducing_points
        if Z.requires_grad:
            # a TRAINABLE inducing matrix may be written through its raw pointer (the fused Adam kernel, any other C-ABI
            # caller): such writes do not bump the version counter, so a cached copy could go stale unnoticed -- and a step
            # captured on a cache hit would replay on it for ever.  No cache then: one copy launch per access.
            return Z.detach()[:, :-1].contiguous()
        tag = (Z.data_ptr(), Z._version, Z.device)
        c = self.__dict__.get("_Zx_contig")
        if c is None or c[0] != tag:
            c = (tag, Z.detach()[:, :-1].contiguous())
            self.__dict__["_Zx_contig"] = c
        return c[1]

    @property
    def zf(self):
        if self.previous_layer is None:
            return self._inducing_points[:, -1]
        prev_m = self.previous_layer.variational_strategy._variational_distribution.variational_mean
        if prev_m.shape[0] != self._inducing_points.shape[0]:
            # only-highest-fidelity ablation: the layers have different inducing inputs, and the kernel of this
            # layer ignores the f column (a_x1 = a_f = nu = 0, mfdgp_hidden_layer_only_hf.py:85-89): any value does
            return self._inducing_points[:, -1]
        return prev_m

    @property
    def inducing_points(self):
        if self.previous_layer is None:
            return self._inducing_points
        return torch.cat((self.Zx, self.zf[:, None]), 1)


class MFDGPHiddenLayer(nn.Module):

    def __init__(self, num_layer, input_dims, inducing_points, inducing_values, num_fidelities, init_lengthscale,
                 y_high_std=1.0, num_samples_for_acquisition=25, previously_trained_layer=None,
                 init_params_to_prior_and_fix_them=False, previous_layer_in_hierarchy=None, only_hf=False):
        super().__init__()
        self.init_params_to_prior_and_fix_them = init_params_to_prior_and_fix_them
        self.num_layer = num_layer
        self.input_dims = input_dims
        num_inducing = inducing_points.shape[0]
        self.num_inducing = num_inducing
        self.kind = 0 if num_layer == 0 else 1
        self.check_pd = True          # sync + jitter retry after every Cholesky (psd_safe_cholesky semantics)

        if num_layer == 0:
            covar_module = gp.ScaleKernel(gp.RBFKernel(ard_num_dims=input_dims, active_dims=list(range(input_dims))))
            covar_module.base_kernel.initialize(lengthscale=init_lengthscale)
            covar_module.initialize(outputscale=1.0)
            if init_params_to_prior_and_fix_them:
                covar_module.base_kernel.initialize(lengthscale=0.25 * input_dims)
                covar_module.initialize(outputscale=1.0)
        else:
            D_range = list(range(input_dims))
            k_x_1 = gp.ScaleKernel(gp.RBFKernel(ard_num_dims=input_dims - 1, active_dims=D_range[0:input_dims - 1]))
            k_f = gp.ScaleKernel(gp.RBFKernel(ard_num_dims=1, active_dims=D_range[input_dims - 1:input_dims]))
            k_x_2 = gp.ScaleKernel(gp.RBFKernel(ard_num_dims=input_dims - 1, active_dims=D_range[0:input_dims - 1]))
            k_lin = gp.LinearKernel(active_dims=D_range[input_dims - 1:input_dims])
            k_x_1.base_kernel.initialize(lengthscale=init_lengthscale * 10.0)
            k_f.base_kernel.initialize(lengthscale=1.0)
            k_x_2.base_kernel.initialize(lengthscale=init_lengthscale)
            if only_hf:   # mfdgp_hidden_layer_only_hf.py:85-89
                k_lin.initialize(variance=torch.zeros(1))
                k_x_1.initialize(outputscale=0.0)
                k_f.initialize(outputscale=0.0)
                k_x_2.initialize(outputscale=1.0)
            else:
                k_lin.initialize(variance=torch.ones(1))
                k_x_1.initialize(outputscale=1.0)
                k_f.initialize(outputscale=1.0)
                k_x_2.initialize(outputscale=0.01)
            if init_params_to_prior_and_fix_them:
                k_x_1.base_kernel.initialize(lengthscale=10 * 0.25 * (input_dims - 1))
                k_f.base_kernel.initialize(lengthscale=1.0)
                k_x_2.base_kernel.initialize(lengthscale=0.25 * (input_dims - 1))
                k_lin.initialize(variance=torch.ones(1))
                k_x_1.initialize(outputscale=1.0)
                k_f.initialize(outputscale=1.0)
                k_x_2.initialize(outputscale=0.01)
            covar_module = k_x_1 * (k_lin + k_f) + k_x_2

        if previously_trained_layer is not None:
            covar_module.load_state_dict(previously_trained_layer.covar_module.state_dict())

        variational_distribution = gp.CholeskyVariationalDistribution(num_inducing_points=num_inducing,
                                                                      mean_init_std=0.0)
        if num_layer == num_fidelities - 1:
            cov = gp.gram_cpu_init(covar_module, self.kind, inducing_points) * (1e-2 * y_high_std ** 2) ** 2
            init_dist = gp.MultivariateNormal(inducing_values, covariance_matrix=cov)
        else:
            init_dist = gp.MultivariateNormal(inducing_values, covariance_matrix=torch.eye(num_inducing) * 1e-8)
        variational_distribution.initialize_variational_distribution(init_dist)

        if num_layer == 0:
            variational_strategy = UnwhitenedVariationalStrategy(self, inducing_points, variational_distribution,
                                                                 learn_inducing_locations=False)
        else:
            variational_strategy = MFDGUnwhitenedVariationalStrategy(
                self, inducing_points, variational_distribution, learn_inducing_locations=False,
                previous_layer=previous_layer_in_hierarchy)
        self.variational_strategy = variational_strategy
        self.covar_module = covar_module

        if previously_trained_layer is not None:
            samples = previously_trained_layer.samples.detach().clone().reshape(-1, 1)   # (S,1) as intended
        else:
            samples = torch.normal(mean=torch.zeros([num_samples_for_acquisition]),
                                   std=torch.ones([num_samples_for_acquisition]))[:, None]
        self.register_buffer("samples", samples)
        self.num_samples_for_acquisition = num_samples_for_acquisition
        self._eval_mode = False

        if init_params_to_prior_and_fix_them or (only_hf and num_layer > 0):
            fix_all = init_params_to_prior_and_fix_them
            if num_layer == 0:
                self.covar_module.base_kernel.raw_lengthscale.requires_grad = False
                self.covar_module.raw_outputscale.requires_grad = False
            else:
                k1 = self.covar_module.kernels[0].kernels[0]
                kf = self.covar_module.kernels[0].kernels[1].kernels[1]
                kl = self.covar_module.kernels[0].kernels[1].kernels[0]
                k2 = self.covar_module.kernels[1]
                for p in (k1.base_kernel.raw_lengthscale, k1.raw_outputscale, kf.base_kernel.raw_lengthscale,
                          kf.raw_outputscale, kl.raw_variance):
                    p.requires_grad = False
                if fix_all:
                    k2.base_kernel.raw_lengthscale.requires_grad = False
                    k2.raw_outputscale.requires_grad = False

    # ------------------------------------------------------------------ reference helpers
    def print_lengthscales_and_outputscale(self, custom_print):
        cm = self.covar_module
        if self.num_layer == 0:
            custom_print({"l0_lengthscale:": cm.base_kernel.lengthscale.detach().cpu().numpy().flatten(),
                          "l0_outputscale:": cm.outputscale.detach().cpu().numpy().item()})
        else:
            k1, kf = cm.kernels[0].kernels[0], cm.kernels[0].kernels[1].kernels[1]
            kl, k2 = cm.kernels[0].kernels[1].kernels[0], cm.kernels[1]
            a1, af = k1.outputscale.item(), kf.outputscale.item()
            custom_print({"l1_lengthscale_x1:": k1.base_kernel.lengthscale.detach().cpu().numpy().flatten(),
                          "l1_lengthscale_f:": kf.base_kernel.lengthscale.detach().cpu().numpy().flatten(),
                          "l1_lengthscale_x2:": k2.base_kernel.lengthscale.detach().cpu().numpy().flatten(),
                          "l1_alpha_x1:": a1, "l1_alpha_f:": af, "l1_alpha_x1f:": a1 * af,
                          "l1_alpha_x2:": k2.outputscale.item(), "l1_nu_lin:": kl.variance.item()})

    def train_mode(self):
        self._eval_mode = False

    def eval_mode(self):
        self._eval_mode = True

    def __getstate__(self):
        # a copy / unpickled layer draws its own seed on first use (two copies must not replay one eps stream)
        state = self.__dict__.copy()
        state.pop("_rng_state", None)
        return state

    def _rng(self, device):
        """int64 [seed, calls, ticket] of this layer's training-sample stream on ``device`` (functional.propagate_rng): the
        seed is drawn once from torch's global CPU generator (torch.manual_seed governs it), the counters live on the device."""
        st = self.__dict__.get("_rng_state")
        if st is None or st.device != device:
            seed = int(torch.randint(1, 2 ** 62, (), dtype=torch.int64))
            st = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)
            self.__dict__["_rng_state"] = st
        return st

    # ------------------------------------------------------------------ the hot path
    def _moments(self, x, f, xdiv, want_dx):
        """(mean, var, kl) for layer rows X~ = [x[n/xdiv], f[n]] through the HIP library."""
        vs = self.variational_strategy
        vd = vs._variational_distribution
        Zx, zf = vs.Zx, vs.zf
        assert x.shape[-1] == Zx.shape[-1], "wrong input dimensionality for this layer"
        hyp = gp.pack_hypers(self.covar_module, self.kind)
        branch = 0 if self.training else 1
        if self._info is None or self._info.device != x.device:
            self._info = torch.zeros((), dtype=torch.int32, device=x.device)
        jit = vs.jitter_val
        for attempt in range(4):
            out = F.layer_forward(x, f, Zx, zf, hyp, vd.variational_mean, vd.chol_variational_covar, self.kind,
                                  xdiv=xdiv, branch=branch, jitter=jit, want_dx=want_dx, info_out=self._info)
            if not self.check_pd:
                break
            pivot = F.check_info(self._info)
            if pivot == 0:
                break
            F.raise_if_abandoned(pivot, "layer forward")      # (-1: not a pivot, no jitter helps)
            if attempt == 3:
                raise NotPSDError(f"K_mm not positive definite (pivot {pivot}) after adding jitter {jit:.1e}")
            jit = vs.jitter_val + 1e-8 * (10 ** attempt)      # psd_safe_cholesky retry ladder
            warnings.warn(f"K_mm not positive definite, retrying with jitter {jit:.1e}", RuntimeWarning)
        return out

    _info = None
    _shortcut_last = False

    def launch_chain(self, n_rows, xdiv, want_dx, main, side):
        """CHAIN half of the next call of this layer on ``n_rows`` rows (K_mm, Cholesky, L^-1, U, a, KL), issued
        under ``torch.cuda.stream(side)``; hands back what ``__call__(..., chain=...)`` needs for the PANEL half."""
        vs = self.variational_strategy
        vd = vs._variational_distribution
        Zx, zf = vs.Zx, vs.zf
        if self._info is None or self._info.device != Zx.device:
            self._info = torch.zeros((), dtype=torch.int32, device=Zx.device)
        with torch.cuda.stream(main):
            hyp = gp.pack_hypers(self.covar_module, self.kind)     # constraint transforms stay on the main stream
            ev = torch.cuda.Event()
            ev.record(main)
        side.wait_event(ev)
        P, token = F.layer_chain(Zx, zf, hyp, vd.variational_mean, vd.chol_variational_covar, self.kind, n_rows,
                                 xdiv=xdiv, branch=0 if self.training else 1, jitter=vs.jitter_val, want_dx=want_dx,
                                 info_out=self._info, main=main, side=side)
        return P, token, hyp

    def freeze_chain(self):
        """CHAIN half for the current (fixed) parameters, with the psd_safe_cholesky jitter ladder of ``_moments``."""
        vs = self.variational_strategy
        vd = vs._variational_distribution
        hyp = gp.pack_hypers(self.covar_module, self.kind)
        jit = vs.jitter_val
        for attempt in range(4):
            fc = F.freeze_chain(vs.Zx, vs.zf, hyp, vd.variational_mean, vd.chol_variational_covar, self.kind,
                                branch=0 if self.training else 1, jitter=jit)
            pivot = F.check_info(fc.info)
            if pivot == 0:
                return fc
            F.raise_if_abandoned(pivot, "chain forward")
            if attempt == 3:
                raise NotPSDError(f"K_mm not positive definite (pivot {pivot}) after adding jitter {jit:.1e}")
            jit = vs.jitter_val + 1e-8 * (10 ** attempt)
            warnings.warn(f"K_mm not positive definite, retrying with jitter {jit:.1e}", RuntimeWarning)

    def _layer_call(self, x, f, xdiv=1, want_dx=False, chain=None):
        vs = self.variational_strategy
        vd = vs._variational_distribution
        if self.training:
            vs._kl_cache = None
        # GPyTorch shortcut: inputs identical to the inducing inputs -> q(u) itself.
        # torch.equal synchronises, which a stream capture forbids: while capturing, the verdict of the last eager
        # call (the warm-up pass on the same static inputs) is reused.
        if xdiv != 1 or x.shape[0] != vs.Zx.shape[0]:
            hit = False
        elif x.is_cuda and torch.cuda.is_current_stream_capturing():
            hit = self._shortcut_last
        else:
            hit = bool(torch.equal(x, vs.Zx) and (f is None or torch.equal(f, vs.zf)))
            self._shortcut_last = hit
        if hit:
            if isinstance(chain, tuple) and isinstance(chain[0], F.ChainBatch):
                vs._kl_cache = chain[0].kls[chain[1]]      # the batch has this layer's KL already
            return vd.variational_mean, F.shortcut_var(vd.chol_variational_covar)
        if isinstance(chain, F.FrozenChain):
            return F.layer_panel_frozen(chain, x, f, xdiv=xdiv, want_dx=want_dx)
        if isinstance(chain, tuple) and isinstance(chain[0], F.ChainBatch):      # all layers' chains in one batch
            CB, z, hyp = chain
            mean, var = F.layer_panel_batched(CB, z, x, f, vs.Zx, vs.zf, hyp, xdiv=xdiv, want_dx=want_dx)
            vs._kl_cache = CB.kls[z]
            return mean, var
        if chain is not None:
            P, token, hyp = chain
            mean, var = F.layer_panel(P, token, x, f, vs.Zx, vs.zf, hyp)
            vs._kl_cache = P.kl
            return mean, var
        mean, var, kl = self._moments(x, f, xdiv, want_dx)
        vs._kl_cache = kl
        return mean, var

    def forward(self, x):
        """Reference signature (:232-243): the layer's prior N(0, k(x, x)) at rows x = [x, f] (``input_dims`` columns).
        The reference returns it lazily and GPyTorch's strategy only ever evaluates blocks of it; the variational layer
        call (``__call__``) never comes through here -- its Gram blocks are produced inside the HIP layer call.  A direct
        call gets the dense prior from the same Gram kernel (no gradient)."""
        assert x.shape[-1] == self.input_dims
        hyp = gp.pack_hypers(self.covar_module, self.kind)
        if self.kind == 0:
            K = F.gram(0, x, None, x, None, hyp)
        else:
            xs, fs = x[:, :-1].contiguous(), x[:, -1].contiguous()
            K = F.gram(1, xs, fs, xs, fs, hyp)
        return gp.MultivariateNormal(torch.zeros(x.shape[0], dtype=x.dtype, device=x.device), covariance_matrix=K)

    def __call__(self, x, *other_inputs, eps=None, xdiv=1, want_dx=False, chain=None, **kwargs):
        """Layer 0: ``layer(x)``.  Layers >= 1: ``layer(x, previous_output)`` (mfdgp_hidden_layer.py:245-286).

        ``x`` holds the base rows; this layer processes ``x.shape[0] * xdiv`` rows (row n uses x[n // xdiv]).
        ``eps`` (optional, N' values) replaces the N(0,1) draw of the training branch (:274).
        """
        if not len(other_inputs):
            mean, var = self._layer_call(x, None, xdiv, want_dx, chain)
            return gp.MultivariateNormal(mean[None, :], var[None, :])      # shape (1, N): SURVEY A.2
        inp = other_inputs[0]
        n_rows = x.shape[0] * xdiv
        if isinstance(inp, gp.MultivariateNormal):
            mean_p, var_p = inp.mean.reshape(-1), inp.variance.reshape(-1)
            if inp.batch_rows is None:
                fdiv = n_rows // mean_p.numel()
            else:
                # the previous layer holds batch_rows >= x.shape[0] rows of the batch (MFDGP.forward(rows=...)): this layer
                # propagates the first x.shape[0] of them; the kernels read that prefix and zero the rest's gradient
                fdiv = xdiv // (mean_p.numel() // inp.batch_rows)
                assert fdiv >= 1 and n_rows % fdiv == 0 and n_rows // fdiv <= mean_p.numel()
            if self._eval_mode:
                S = self.num_samples_for_acquisition
                assert n_rows % S == 0, "eval_mode expects inputs tiled num_samples_for_acquisition-fold"
                e = self.samples.reshape(-1).to(mean_p.dtype).repeat(n_rows // S)
            elif eps is not None:
                e = eps.reshape(-1)
            else:
                e = None
            # The previous layer's moments have two consumers: this propagation and the ELBO's data term of that layer's own
            # fidelity.  With gradients recorded the propagation hands back pass-through aliases of (mean, var) and the
            # distribution object is re-pointed at them, so whoever scores it afterwards goes THROUGH this node: the two
            # gradient contributions are then summed inside the propagate-backward launch (autograd would spend two
            # element-wise launches per layer and step on the sums).
            through = torch.is_grad_enabled() and (mean_p.requires_grad or var_p.requires_grad) and inp._cov is None
            if e is None:
                # the reference draws float32 N(0,1) on the CPU RNG; here float64, inside the propagation launch
                # (counter-based Philox keyed by this layer's seed + call counter: capturable, fresh at every replay)
                if through:
                    f, _, mean_t, var_t = F.propagate_rng_through(mean_p, var_p, self._rng(mean_p.device), n_rows, fdiv)
                else:
                    f, _ = F.propagate_rng(mean_p, var_p, self._rng(mean_p.device), n_rows, fdiv)
            elif through:
                f, mean_t, var_t = F.propagate_through(mean_p, var_p, e, fdiv)
            else:
                f = F.propagate(mean_p, var_p, e, fdiv)
            if through:
                inp.mean, inp._variance = mean_t.view(inp.mean.shape), var_t.view(inp._variance.shape)
        else:
            f = inp.reshape(-1)
            if f.numel() != n_rows:
                f = f.repeat_interleave(n_rows // f.numel())
        mean, var = self._layer_call(x, f, xdiv, want_dx, chain)
        return gp.MultivariateNormal(mean, var)
