#!/usr/bin/env python3
"""ELBO steps/sec of the MFDGP hot path on MI355X (BASELINE.json metric, headline config C3).

One bench "step" = one ELBO step (zero_grad + MFDGP.forward + VariationalELBOMF + backward + Adam,
mobocmf/util/blackbox_mfdgp_fitter.py:161-171) for EACH of the 3 surrogates (2 objectives + 1 constraint)
of C3 on the full batch B = N.  `value` = surrogate ELBO steps per second, whole job.  With --gpus N every
rank trains its own 3 surrogates (weak scaling, no data-path collective: the surrogates are independent,
SURVEY 8(e)); the single RCCL all-gather of posterior moments for the joint acquisition happens once after
the timed region and is reported as `exchange_ms`.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            (starts the N ranks itself, one fresh process per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The line's `value` is the median of --repeats (default 5) timed regions of exactly K steps each (all listed in
`repeat_values`); `roofline` times the GEMM instantiations the layer really launches (column-statistics and dA
epilogues included), each on its own, and `per_kernel_instep_ms` the same launches INSIDE a training step (HIP events
recorded by the library around them, mobocmf_layer_desc.probe_events); `cpu_baseline` is 3 warm-up + 10 timed oracle steps on
the box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mobocmf_amd.util import synthetic  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X dense FP64 vector == matrix peak


def algorithmic_flops(cfg):
    """F_step of SURVEY 8(d) / BASELINE.md section 5 for one surrogate."""
    d, L, M, N, S = cfg["d"], cfg["L"], cfg["M"], cfg["N"], cfg["S"]
    tot = 0.0
    for l in range(L):
        Np = N if l == 0 else N * S
        c = 3 * d + 8 if l == 0 else 3 * (2 * d + 1) + 30
        tot += 3.0 * M * M * Np + (2.0 / 3.0) * M ** 3 + 4.0 * M * Np + c * (M * (M + 1) / 2.0 + M * Np)
    return 3.0 * tot


def build_graphed(cfg, outputs, device, use_graph, shard_rows=False, prune_rows=True, top_fraction=0.25):
    """One GraphedELBOStep (HIP-graph replay of the whole step) per surrogate, each on its own stream.
    shard_rows: every rank holds the SAME surrogates and 1/W of their batch rows (SURVEY 8(e) level 2)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    if shard_rows:
        from mobocmf_amd.parallel import RowShardedELBOStep as GraphedELBOStep
    steps = []
    for o in outputs:
        prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], output=o % 3, seed=o,
                                      top_fraction=top_fraction)
        model = synthetic.model_from_problem(prob, device=device)
        elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device)
        x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
        if cfg["M"] == cfg["N"] and not shard_rows:
            # Z = the training inputs (the reference's default, C1): its shuffling loader feeds a PERMUTATION of Z, i.e.
            # GPyTorch's general branch, never the equal-inputs shortcut (blackbox_mfdgp_fitter.py:35) -- same here
            from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter
            perm = BlackBoxMFDGPFitter.shuffled_rows(cfg["N"], device)
            x, y, fid = x[perm].contiguous(), y[perm].contiguous(), fid[perm].contiguous()
        steps.append(GraphedELBOStep(model, elbo, x, y, fid, lr=1e-3, use_graph=use_graph, prune_rows=prune_rows))
    return steps


def build_surrogates(cfg, outputs, device):
    from mobocmf_amd.mlls import VariationalELBOMF
    sur = []
    for o in outputs:
        prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], output=o % 3, seed=o)
        model = synthetic.model_from_problem(prob, device=device)
        model.set_check_pd(False)          # no host sync inside the step; finiteness is checked after the timed region
        elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device)
        data = (t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None])
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        sur.append((model, elbo, opt, data))
    return sur


def one_step(sur, cfg, gens, streams):
    """One ELBO step of every surrogate.  The surrogates are independent (blackbox_mfdgp_fitter.py:134-152 loops
    over them sequentially), so each one runs on its own HIP stream: the latency-bound M x M chains (Cholesky,
    triangular inverse, Cholesky backward) of one surrogate overlap the MFMA GEMMs of the others."""
    losses = []
    for (model, elbo, opt, (x, y, fid)), gen, st in zip(sur, gens, streams):
        with torch.cuda.stream(st):
            opt.zero_grad(set_to_none=True)
            eps = [None] + [torch.randn(cfg["N"] * cfg["S"], dtype=torch.float64, device=x.device, generator=gen)
                            for _ in range(1, cfg["L"])]
            out = model(x, eps=eps)
            res = elbo(out, y.T, fid)
            (-res[0]).backward()
            opt.step()
            losses.append(res[0].detach())
    return losses


def panel_columns(cfg, layer_rows=None):
    """N' of every layer's M x N' panel work in the step: base rows the layer is evaluated on (all N, or the prefix of rows
    that can reach the loss -- GraphedELBOStep.layer_rows) times the samples per row (1 for layer 0, S above)."""
    rows = layer_rows if layer_rows is not None else [cfg["N"]] * cfg["L"]
    return [rows[l] * (1 if l == 0 else cfg["S"]) for l in range(cfg["L"])]


def executed_gemm_flops(cfg, layer_rows=None, active_frac=None):
    """Flops the step really executes in its N'-sized contractions: per layer 4 triangular products (A = L^-1 K, C = U^T A,
    dA, dK: M^2 N' each) + the weighted syrk H = A diag(gv) A^T (M^2 N') = 5 M^2 N' (DESIGN.md section 1).  layer_rows: the
    layers' row counts (dead rows pruned); active_frac[l]: share of layer l's column blocks with non-zero upstream gradient
    (the three backward products run over those only)."""
    cols = panel_columns(cfg, layer_rows)
    act = active_frac if active_frac is not None else [1.0] * cfg["L"]
    return sum((2.0 + 3.0 * act[l]) * cfg["M"] ** 2 * cols[l] for l in range(cfg["L"]))


def backward_active_fractions(cfg, fid, layer_rows, sparse):
    """Share of each layer's 128-column blocks that carry upstream gradient: rows scored at the layer's own fidelity plus the
    rows the next layer propagates gradient back to."""
    import numpy as np
    fid = np.asarray(fid).reshape(-1)
    L, N = cfg["L"], fid.size
    out = []
    need = np.zeros(N, dtype=bool)
    for l in reversed(range(L)):
        n_l = layer_rows[l] if layer_rows is not None else N
        on = (fid == l)
        on = on | need
        need = on.copy()
        if not sparse:
            out.append(1.0)
            continue
        mult = 1 if l == 0 else cfg["S"]
        cols = np.repeat(on[:n_l], mult)
        nb = (cols.size + 127) // 128
        pad = np.zeros(nb * 128, dtype=bool)
        pad[:cols.size] = cols
        out.append(float(pad.reshape(nb, 128).any(1).sum()) / nb)
    return out[::-1]


def _file_sha16(path):
    import hashlib
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def measure_gemm_variants(cfg, device, iters=20, n_cols=None):
    """Average duration (HIP events on the launch stream, each variant alone on an idle chip) of the four triangular
    M x N' products of the top layer, launched exactly as mobocmf_layer_forward / _backward launch them:
      A = L^-1 K  (lower-triangular, column-statistics epilogue: q and mean partials)
      C = U^T A   (upper-triangular, column-statistics epilogue: r partials, non-temporal stores)
      dA          (lower-triangular, dA epilogue: column scale + rank-1 + axpy, row-dot partials for da)
      dK = L^-T dA (upper-triangular, plain store)
    Algorithmic flops per launch: M^2 N' (triangular product)."""
    from mobocmf_amd import functional as F
    Mp = (cfg["M"] + 127) // 128 * 128
    n_cols = cfg["N"] * cfg["S"] if n_cols is None else n_cols
    Np = (n_cols + 127) // 128 * 128
    nrb = Mp // 128
    g = torch.Generator(device=device)
    g.manual_seed(7)
    rnd = lambda *sh: torch.randn(*sh, dtype=torch.float64, device=device, generator=g)
    Lw = torch.tril(rnd(Mp, Mp))
    Up = torch.triu(rnd(Mp, Mp))
    B = rnd(Mp, Np)
    A2 = rnd(Mp, Np)
    C = torch.empty(Mp, Np, dtype=torch.float64, device=device)
    avec, gmu, cgv, gv = rnd(Mp), rnd(Np), rnd(Np), rnd(Np)
    p1 = torch.empty(4 * nrb, Np, dtype=torch.float64, device=device)
    p2 = torch.empty(4 * nrb, Np, dtype=torch.float64, device=device)
    rdp = torch.empty(2 * (Np // 128), Mp, dtype=torch.float64, device=device)
    stream_out = Np * Mp * 8 >= (64 << 20)
    variants = [
        ("A = L^-1 K (lower, colstats q+mean)", lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=avec)),
        ("C = U^T A (upper, colstats r, stream-out)", lambda: F.gemm_f64_epilogue(Up, B, C, 2, 1, stream_out=stream_out, colsq_part=p1, avec=avec)),
        ("dA (lower, dA epilogue + row dots)", lambda: F.gemm_f64_epilogue(Lw, B, C, 1, 2, alpha=2.0, avec=avec, bscale=gv, gmu=gmu, cgv=cgv, Aaux=A2, rowdot_part=rdp)),
        ("dK = L^-T dA (upper, plain store)", lambda: F.gemm_f64_epilogue(Up, B, C, 2, 0)),
    ]
    flops = float(cfg["M"]) ** 2 * n_cols
    out = []
    for name, fn in variants:
        # the chip needs a few hundred ms of load before its clock settles (a variant timed cold reads ~15 % slow); the
        # training step runs in that settled state, so each variant is timed there too
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.25:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        sec = st.elapsed_time(en) * 1e-3 / iters
        out.append({"kernel": name, "kernel_ms": sec * 1e3, "achieved": flops / sec / 1e12,
                    "frac": flops / sec / 1e12 / FP64_PEAK_TFLOPS})
    return out, flops, (Mp, Np)


def measure_dominant_kernel(cfg, device, iters=20, n_cols=None, layer=None):
    """`roofline` of the bench line: the dominant kernel of the step is the triangular f64 MFMA GEMM; the record carries
    the instantiation with the largest share of the step (A = L^-1 K with the column-statistics epilogue) and, under
    `variants`, all four launches of that layer with their own times, plus their flop-weighted (= time-weighted, equal
    flops) fraction.  n_cols: N' of the layer with the widest panel in the step as it runs (dead rows pruned)."""
    var, flops, (Mp, Np) = measure_gemm_variants(cfg, device, iters, n_cols)
    head = var[0]
    traffic, traffic_src = None, None
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (profiles/): static, valid only for the kernel source
    # they were collected on
    try:
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if name.endswith("_pmc_gemm.json"):
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pm = json.load(fh)
                same = pm.get("kernel_source_sha16") == _file_sha16(os.path.join(ROOT, "mobocmf_amd", "csrc", "gemm_f64.hip"))
                if pm.get("shape") == [Mp, Np, Mp] and same:
                    traffic = pm["traffic_bytes_per_launch"]
                    traffic_src = "profiles/%s (static: separate rocprofv3 --pmc passes on this kernel source)" % name
                else:
                    traffic_src = "profiles/%s is for another kernel source or shape: not quoted" % name
                break
    except Exception:
        pass
    tot = sum(v["kernel_ms"] for v in var)
    return {"bound": "mfma", "achieved": head["achieved"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": head["frac"], "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "gemm_f64_kernel<NN, triangular, colstats> -- %s, %dx%dx%d" % (head["kernel"], Mp, Np, Mp),
            "kernel_ms": head["kernel_ms"], "flops_per_launch": flops, "variants": var, "layer": layer,
            "weighted_frac": len(var) * flops / (tot * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            "timing": "HIP events on the launch stream, %d launches per variant after 0.25 s of the same launches (settled "
                      "clock), each variant alone on the chip" % iters}


PROBE_SPANS = (("Gram forward K_mn", 9, 0), ("A = L^-1 K (lower, colstats q+mean)", 0, 1),
               ("C = U^T A (upper, colstats r, stream-out)", 1, 2), ("dA (lower, dA epilogue + row dots)", 3, 4),
               ("H = A diag(gv) A^T (weighted syrk + slab reduction)", 5, 6), ("dK = L^-T dA (upper, plain store)", 7, 8),
               ("Gram backward of K_mn", 8, 10))


def reference_sizes_leg(device, steps=2000):
    """The reference's OWN problem size beside the headline (BASELINE.json configs[0]: Forrester 1D, 2 fidelities, M = N = 16,
    S = 4; three surrogates): ELBO steps/s through the one-launch step (mobocmf_tiny_elbo_step) and through
    the layer entry points (HIP-graph replay on three streams), same process.  A few tenths of a second."""
    from mobocmf_amd.util import tiny_step as TS
    cfg = dict(synthetic.CONFIGS["C1"])
    gsteps = build_graphed(cfg, [0, 1, 2], device, use_graph=True)
    out = {"workload": "C1: Forrester-sized, d=%d M=%d N=%d S=%d, 3 surrogates" % (cfg["d"], cfg["M"], cfg["N"], cfg["S"])}

    def timed(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 3 * n / (time.perf_counter() - t0)

    out["layer_path_steps_per_s"] = timed(lambda: [g.step() for g in gsteps], max(steps // 8, 50))
    if all(TS.eligible(g.model, g.x, g.fid) for g in gsteps):
        tiny = TS.TinyELBOStep([g.model for g in gsteps], [cfg["N"]] * 3, [g.x for g in gsteps], [g.y for g in gsteps],
                               [g.fid for g in gsteps], lr=1e-3)
        out["one_launch_steps_per_s"] = timed(tiny.step, steps)
        tiny.check()
        out["finite"] = bool(torch.isfinite(tiny.losses).all())
    for g in gsteps:
        g.retire()
    return out


def measure_instep_kernels(gstep, cfg, steps=6, skip=2, layer=None):
    """Durations of one layer's grid-filling launches INSIDE a training step of one surrogate (the other surrogates idle): the
    library records caller-created HIP events around them (mobocmf_layer_desc.probe_events, attached to the PANEL calls of
    layer index `layer` only -- two layers with the same N' cannot be mixed up) while the step is issued eagerly on the
    surrogate's own stream -- the same launch sequence the captured graph replays.  Mean over the last steps - skip steps."""
    import ctypes

    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F_
    n_ev = _lib.PROBE_EVENTS
    acc = {name: [] for name, _, _ in PROBE_SPANS}
    for k in range(steps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
        with torch.cuda.stream(gstep.stream):
            for e in evs:
                e.record(gstep.stream)      # creates the underlying hipEvent_t (lazy in torch) before it is handed over
            table = (ctypes.c_void_p * n_ev)(*[e.cuda_event for e in evs])
            with F_.probe_events(table, layer):
                gstep._eager()
        gstep.stream.synchronize()
        if k >= skip:
            for name, a, b in PROBE_SPANS:
                acc[name].append(evs[a].elapsed_time(evs[b]))
    return {name: sum(v) / len(v) for name, v in acc.items() if v}


def usable_cores():
    """Host cores this process may actually use: min(os.cpu_count(), affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(cfg, steps=10, warmup=3, device=None, budget_s=75.0):
    """Reference-equivalent float64 torch-CPU restatement (gpytorch unavailable): the oracle executing GPyTorch's
    op sequence + autograd + torch.optim.Adam, ONE surrogate at the full size, `warmup` warm-up + `steps` timed steps
    (>= 3 + >= 10), median / min / max reported; never more than ~budget_s of CPU work."""
    import numpy as np

    from oracle import mfdgp_oracle as O
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], output=0, seed=0)
    t = lambda a, rg=False: torch.as_tensor(np.asarray(a), dtype=torch.float64).clone().requires_grad_(rg)
    layers = []
    for l, lay in enumerate(prob["layers"]):
        h = lay["hyp"]
        if l == 0:
            r = {"raw_ls": O.inv_softplus(t(h["ls"])), "raw_alpha": O.inv_softplus(t(h["alpha"]))}
        else:
            r = {"raw_ls1": O.inv_softplus(t(h["ls1"])), "raw_a1": O.inv_softplus(t(h["a1"])),
                 "raw_lsf": O.inv_softplus(t(h["lsf"])), "raw_af": O.inv_softplus(t(h["af"])),
                 "raw_nu": O.inv_softplus(t(h["nu"])), "raw_ls2": O.inv_softplus(t(h["ls2"])),
                 "raw_a2": O.inv_softplus(t(h["a2"]))}
        r = {k: v.detach().clone().requires_grad_(True) for k, v in r.items()}
        r["m"], r["L_S"] = t(lay["m"], True), t(lay["L_S"], True)
        layers.append(r)
    raw = {"Zx": t(prob["Zx"]), "layers": layers,
           "raw_noise": [O.inv_interval(t(v), 1e-8, 1.0).detach().clone().requires_grad_(True) for v in prob["noise"]],
           "noise_hi": [1.0] * cfg["L"]}
    opt = torch.optim.Adam(O.flatten_raw(raw), lr=1e-3)
    x, y, fid = t(prob["x"]), t(prob["y"]), t(prob["fid"])
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    parity = parity_vs_oracle(O, cfg, prob, raw, x, y, fid, eps, device) if device is not None else None
    times = []
    t_all = time.perf_counter()
    for k in range(warmup + steps):
        t0 = time.perf_counter()
        O.elbo_step(raw, opt, x, y, fid, eps, cfg["S"], ref_equiv=True)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(times) > warmup + 2:      # bounded sample (slow / small hosts)
            break
    timed = sorted(times[warmup:])
    med = timed[len(timed) // 2]
    steps = len(timed)
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = [ln.split(":")[1].strip() for ln in fh if ln.startswith("model name")][0]
    except Exception:
        pass
    return {"value": 1.0 / med, "unit": "ELBO steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 surrogate at full size (d=%d M=%d N=%d S=%d), %d warm-up + %d timed steps, median; "
                      "reference-equivalent CPU restatement (gpytorch unavailable)" %
                      (cfg["d"], cfg["M"], cfg["N"], cfg["S"], warmup, steps),
            "os_cpu_count": os.cpu_count(), "cpu_model": model, "sec_per_step": med,
            "sec_per_step_min": timed[0], "sec_per_step_max": timed[-1], "timed_steps": steps, "warmup_steps": warmup}, parity


def parity_vs_oracle(O, cfg, prob, raw, x, y, fid, eps, device, T=256):
    """BASELINE.json's second metric (pred-var rel-err) at the headline size, in the CPU-baseline leg: ELBO of the same
    surrogate (same parameters, same explicit eps) and predict_for_acquisition moments on T test points, HIP path vs the
    oracle.  Tolerance of the north star: 1e-4 relative."""
    from mobocmf_amd.mlls import VariationalELBOMF
    with torch.no_grad():
        state = O.state_from_raw(raw)
        state["samples"] = [None if s_ is None else torch.as_tensor(s_, dtype=torch.float64) for s_ in prob["samples"]]
        e_ref, _ = O.elbo(state, x, y, fid, eps=eps, S=cfg["S"], ref_equiv=True)
        Xt = torch.as_tensor(synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=8, N=T, S=1, seed=98)["x"], dtype=torch.float64)
        mu_ref, var_ref = O.predict_for_acquisition(state, Xt, cfg["L"] - 1, cfg["S"])
        model = synthetic.model_from_problem(prob, device=device)
        dv = lambda a: a.to(device)
        e_gpu, _ = VariationalELBOMF(model, cfg["N"], cfg["L"])(model(dv(x), eps=[None] + [dv(e) for e in eps[1:]]),
                                                                dv(y)[None, :], dv(fid)[:, None])
        # the same ELBO as the timed step evaluates it: layer l on the rows of fidelity >= l only (the synthetic batch is ordered
        # by descending fidelity, as GraphedELBOStep orders any batch), against the oracle's every-layer-at-every-row value
        e_pruned = None
        fv = fid.reshape(-1)
        if bool((fv[:-1] >= fv[1:]).all()):
            rows = [int((fv >= l).sum()) for l in range(cfg["L"])]
            if rows[-1] >= 1:
                eps_p = [None] + [dv(e).reshape(cfg["N"], cfg["S"])[:rows[l + 1]].reshape(-1).contiguous()
                                  for l, e in enumerate(eps[1:])]
                e_pruned, _ = VariationalELBOMF(model, cfg["N"], cfg["L"])(model(dv(x), eps=eps_p, rows=rows),
                                                                          dv(y)[None, :], dv(fid)[:, None])
        model.eval()
        mu, var = model.predict_for_acquisition(dv(Xt), cfg["L"] - 1)
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    return {"elbo_rel_err": abs(float(e_gpu) - float(e_ref)) / abs(float(e_ref)),
            "elbo_rel_err_dead_rows_pruned": None if e_pruned is None else abs(float(e_pruned) - float(e_ref)) / abs(float(e_ref)),
            "pred_mean_rel_err": rel(mu, mu_ref),
            "pred_var_rel_err": rel(var, var_ref), "tolerance": 1e-4,
            "against": "oracle (float64 CPU restatement, GPyTorch op order) -- %s seed 0 output 0, explicit eps; moments of "
                       "predict_for_acquisition at %d test points, top fidelity, S=%d fixed samples; errors are max |diff| / "
                       "max |oracle|" % (cfg.get("name", "headline config"), T, cfg["S"])}


def main():
    # stdout carries exactly ONE line (the JSON record): library chatter written to fd 1 from native code (the RCCL
    # banner, gloo's connection messages) is diverted to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="run the surrogates back to back on one stream")
    ap.add_argument("--surrogates", type=int, default=0, help="surrogates per GPU (default: the config's 3)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) | gloo (CPU rehearsal of the multi-rank control flow)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--eager", action="store_true", help="issue every step from Python instead of replaying HIP graphs")
    ap.add_argument("--shard", default="surrogates", choices=["surrogates", "rows"],
                    help="surrogates: each rank trains its own surrogates (weak scaling, default); rows: all ranks train "
                         "the same surrogates on 1/W of the batch rows + one gradient all-reduce per step (strong scaling)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="serialise each surrogate's layers on one stream (no chain/panel split across streams)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each; value = median")
    ap.add_argument("--tile-rows", type=int, default=0, help="A/B knob: tile height of the panel products (0 automatic, 64, 128)")
    ap.add_argument("--potrf-cols", type=int, default=None, help="A/B knob: the blocked Cholesky as one launch (0, default) or a launch pair per 64 columns (4 | 1 columns per hand-over)")
    ap.add_argument("--dense-backward", action="store_true",
                    help="A/B knob: do not skip the column blocks of a layer backward whose upstream gradients are all zero")
    ap.add_argument("--small-gemm-max", type=int, default=0, help="A/B knob: largest M x M product the small-operand kernel takes (mobocmf_tuning.small_gemm_max)")
    ap.add_argument("--mid-gemm-max", type=int, default=-1, help="A/B knob: largest M x M product on the mid-size kernel (0 = off)")
    ap.add_argument("--mid-gemm-waves", type=int, default=0, help="A/B knob: wavefronts per workgroup of the mid-size kernel (8 | 4)")
    ap.add_argument("--top-fraction", type=float, default=0.25,
                    help="sensitivity sweep only: share of the rows at the top fidelity (SURVEY 8(d) / BASELINE.md fix it at 1/4)")
    ap.add_argument("--syrk-wgs", type=int, default=0, help="A/B knob: workgroups a k-sliced weighted syrk may occupy (default 512)")
    ap.add_argument("--no-prune-rows", action="store_true",
                    help="A/B knob: evaluate every layer at every row (the reference's layout) instead of the rows that reach the loss")
    ap.add_argument("--no-dense-leg", action="store_true",
                    help="skip the second timing of the step in the reference's layout (every layer at every row, dense backward)")
    ap.add_argument("--launch", action="store_true", help="go through the rank launcher even for --gpus 1")
    ap.add_argument("--no-rccl", action="store_true", help="one GPU: do not create the one-rank process group")
    ap.add_argument("--coop-wgs", type=int, default=0, help="A/B knob: workgroups per surrogate of the cooperative one-launch step (0: automatic)")
    ap.add_argument("--layer-path", action="store_true",
                    help="small configurations: the layer entry points (HIP-graph replay) instead of the one-launch step")
    args = ap.parse_args()
    if args.no_overlap:
        from mobocmf_amd.models import MFDGP
        MFDGP.overlap_chains = False

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.launch):
        # plain `python bench.py --gpus N`: start the N ranks here -- one fresh process per GPU, RCCL.  This launcher
        # process never touches the GPU (not even torch.cuda.device_count(): without amdsmi it falls back to
        # hipGetDeviceCount, which initialises the HIP runtime): every rank checks its own LOCAL_RANK against the visible
        # devices and exits 3, which takes its siblings down (parallel.launch_ranks) -- never a silent 1-GPU line
        from mobocmf_amd import parallel
        os.dup2(json_fd, 1)
        codes = parallel.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        raise SystemExit(max(abs(c) for c in codes))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        raise SystemExit(2)
    if args.force_device >= 0:
        local_rank = args.force_device
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d needs device %d but only %d device(s) are visible\n" %
                         (rank, local_rank, torch.cuda.device_count()))
        raise SystemExit(3)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.tile_rows:
        from mobocmf_amd import functional as F_
        F_.set_tile_rows(args.tile_rows)
    if args.dense_backward:
        from mobocmf_amd import functional as F_
        F_.set_sparse_backward(False)
    if args.small_gemm_max:
        from mobocmf_amd import functional as F_
        F_.set_tuning(small_gemm_max=args.small_gemm_max)
    if args.mid_gemm_max >= 0:
        from mobocmf_amd import functional as F_
        F_.set_mid_gemm_max(args.mid_gemm_max)
    if args.mid_gemm_waves:
        from mobocmf_amd import functional as F_
        F_.set_mid_gemm_waves(args.mid_gemm_waves)
    if args.syrk_wgs:
        from mobocmf_amd import functional as F_
        F_.set_syrk_workgroups(args.syrk_wgs)
    if args.potrf_cols is not None:
        from mobocmf_amd import functional as F_
        F_.set_potrf_cols(args.potrf_cols)
    dist = None
    rccl_error = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    elif not args.no_rccl:
        # one GPU: a process group of ONE rank on the nccl (= RCCL) backend all the same, so that the exchange below runs the
        # collective an N-rank job runs (`rccl_ranks`, `exchange_ms` of the record) -- through the launcher's environment when
        # there is one, else through a rendezvous file (no port to collide on).  A box whose RCCL cannot start is reported
        # (`rccl_error`), the step timing does not depend on it.
        import torch.distributed as dist
        try:
            if "MASTER_PORT" in os.environ and "RANK" in os.environ:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(args.backend, rank=0, world_size=1,
                                        **({"device_id": device} if args.backend == "nccl" else {}))
            else:
                import tempfile
                rdv = tempfile.NamedTemporaryFile(prefix="mobocmf_rdv_", delete=False)
                rdv.close()
                os.unlink(rdv.name)
                dist.init_process_group(args.backend, init_method="file://" + rdv.name, rank=0, world_size=1,
                                        **({"device_id": device} if args.backend == "nccl" else {}))
        except Exception as e:      # noqa: BLE001 -- whatever the backend raises: report, go on without a group
            rccl_error = "%s: %s" % (type(e).__name__, str(e)[:300])
            dist = None

    cfg = dict(synthetic.CONFIGS[args.config])
    n_out = 3 if args.config == "C3" else min(cfg["outputs"], 3) if args.config != "C5" else 1
    if args.surrogates:
        n_out = args.surrogates
    rows = args.shard == "rows"
    outputs = list(range(n_out)) if rows else list(range(rank * n_out, rank * n_out + n_out))
    torch.manual_seed(1234 + rank)
    if rows and args.eager:
        raise SystemExit("--shard rows runs through RowShardedELBOStep (graph | all-reduce | graph)")
    if not args.eager:
        gsteps = build_graphed(cfg, outputs, device, use_graph=True, shard_rows=rows, prune_rows=not args.no_prune_rows,
                               top_fraction=args.top_fraction)
        sur = [(g.model, g.elbo, g.optimizer, (g.x, g.y, g.fid)) for g in gsteps]

        def one_step(*_a):
            return [g.step()[0] for g in gsteps]

        # the reference's own sizes (C1): the package trains such surrogates through ONE launch per step for all of them
        # (mobocmf_tiny_elbo_step, util/tiny_step.py) -- so does the bench, unless --layer-path asks for the launch sequence
        # -- and mid-size surrogates (M <= 128, e.g. C2) through the cooperative one-launch step (mobocmf_coop_elbo_step,
        # util/coop_step.py), as BlackBoxMFDGPFitter.train_mfdgps does
        tiny = None
        if not rows and not args.layer_path:
            from mobocmf_amd.util import coop_step as CS
            from mobocmf_amd.util import tiny_step as TS
            cls = None
            if all(TS.eligible(g.model, g.x, g.fid) for g in gsteps):
                cls = TS.TinyELBOStep
            elif all(CS.worthwhile(g.model, g.x, g.fid) for g in gsteps):
                cls = CS.CoopELBOStep
            if cls is not None:
                tiny = cls([g.model for g in gsteps], [cfg["N"]] * len(gsteps), [g.x for g in gsteps],
                           [g.y for g in gsteps], [g.fid for g in gsteps], lr=1e-3)
                if args.coop_wgs and cls is CS.CoopELBOStep:
                    tiny.wgs_per_model = args.coop_wgs

                def one_step(*_a):
                    return [tiny.step()]
    else:
        tiny = None
        sur = build_surrogates(cfg, outputs, device)
        one_step = globals()["one_step"]
    gens = []
    for i in range(len(sur)):
        gen = torch.Generator(device=device)
        gen.manual_seed(1234 + 16 * rank + i)
        gens.append(gen)
    streams = [torch.cuda.Stream(device=device) for _ in sur] if not args.single_stream else \
        [torch.cuda.current_stream(device)] * len(sur)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(sur, cfg, gens, streams)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed_all = []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            losses = one_step(sur, cfg, gens, streams)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        elapsed_all.append(el)
    elapsed = sorted(elapsed_all)[len(elapsed_all) // 2]       # median repeat; every repeat is listed in the line
    finite = all(bool(torch.isfinite(l).all()) for l in losses)

    # the path's single exchange: all-gather of the posterior moments on a shared test grid (JES)
    from mobocmf_amd import parallel
    T = 256
    Xg = torch.as_tensor(synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=8, N=T, S=1, seed=99)["x"],
                         dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.no_grad():
        local = []
        for model, _, _, _ in sur:
            model.eval()
            mus, vs = model.predict_for_acquisition(Xg, cfg["L"] - 1)
            model.train()
            local.append(torch.stack([mus, vs]))
        local = torch.stack(local)
        gathered = parallel.all_gather_moments(local)      # (the warm call: communicator set-up, first-use allocations)
        torch.cuda.synchronize()
        exchange_first_ms = (time.perf_counter() - t1) * 1e3
        ex = []
        for _ in range(5):      # the collective alone: median of 5
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gathered = parallel.all_gather_moments(local)
            torch.cuda.synchronize()
            ex.append((time.perf_counter() - t1) * 1e3)
    exchange_ms = sorted(ex)[2]
    rccl_loaded = any("librccl" in ln for ln in open("/proc/self/maps")) if os.path.exists("/proc/self/maps") else None
    finite = finite and bool(torch.isfinite(gathered).all()) and gathered.shape[0] == world * local.shape[0] and \
        bool(torch.equal(gathered[rank * local.shape[0]:(rank + 1) * local.shape[0]], local))

    # What the record says about rows / columns / executed flops is read off the STEP OBJECT (its layer_rows and the
    # fidelities of the rows it holds, in the order it holds them), never inferred from the CLI flags: with --shard rows every
    # rank holds N/W rows (strided shard, pruned inside the shard), so the per-rank figures below are for `cfg_rank`.
    sparse = not args.dense_backward
    if not args.eager:
        layer_rows = gsteps[0].layer_rows
        fid0 = gsteps[0].fid.detach().reshape(-1).cpu().numpy()
    else:
        layer_rows = None
        fid0 = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=8, N=cfg["N"], S=1, seed=0, top_fraction=args.top_fraction)["fid"]
    cfg_rank = dict(cfg, N=int(len(fid0)))            # rows this rank's step holds (N, or its shard of N)
    act = backward_active_fractions(cfg_rank, fid0, layer_rows, sparse)
    cols = panel_columns(cfg_rank, layer_rows)
    dom = max(range(cfg["L"]), key=lambda l: cols[l])
    # flops one whole step of one surrogate executes: one rank's share x the ranks that share the surrogate (row shards are
    # strided, hence equal up to one row)
    flops_step = executed_gemm_flops(cfg_rank, layer_rows, act) * (world if rows else 1)

    # the same step in the reference's layout -- every layer at every row, dense backward -- timed beside it (one GPU only)
    dense_leg = None
    if world == 1 and not args.eager and not rows and not args.no_dense_leg and (layer_rows is not None or sparse):
        from mobocmf_amd import functional as F_
        F_.set_sparse_backward(False)
        try:
            dsteps = build_graphed(cfg, outputs, device, use_graph=True, prune_rows=False, top_fraction=args.top_fraction)
        finally:
            F_.set_sparse_backward(sparse)
        for _ in range(args.warmup):
            for g in dsteps:
                g.step()
        d_el = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for g in dsteps:
                    g.step()
            torch.cuda.synchronize()
            d_el.append(time.perf_counter() - t0)
        d_med = sorted(d_el)[1]
        dense_leg = {"value": n_out * args.steps / d_med, "unit": "ELBO steps/s", "ms_per_step": d_med / args.steps * 1e3,
                     "repeat_values": [n_out * args.steps / e for e in d_el],
                     "step_flops_executed_gemm": executed_gemm_flops(cfg),
                     "what": "the same surrogates, every layer evaluated at every row and the backward dense: the reference's "
                             "op layout (mfdgp.py:174-196 + variational_elbo_mf.py:33-38 masking afterwards)"}
        for g in dsteps:
            g.retire()
        del dsteps

    small_leg = None
    if world == 1 and rank == 0 and not args.no_dense_leg and args.config == "C3":
        small_leg = reference_sizes_leg(device)
    if rank == 0:
        n_sur = n_out if rows else n_out * world
        value = n_sur * args.steps / elapsed
        line = {
            "metric": "ELBO steps/sec (MFDGP d=%d M=%d N=%d S=%d)" % (cfg["d"], cfg["M"], cfg["N"], cfg["S"]),
            "value": value, "unit": "ELBO steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if rows else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic d=%d, %d fidelities, M=%d, N=%d, S=%d, %d surrogates per GPU "
                                   "(2 objectives + 1 constraint), full batch, Adam" %
                                   (args.config, cfg["d"], cfg["L"], cfg["M"], cfg["N"], cfg["S"], n_out),
                       "dead_rows": "pruned (layer l on the rows of fidelity >= l)" if layer_rows is not None else "kept (every layer at every row)",
                       "backward": "skips zero-gradient column blocks" if sparse else "dense",
                       "surrogates_per_gpu": n_out, "parallelism": ("row-sharded x%d + grad all-reduce" if rows else "surrogate-per-rank x%d") % world},
            "per_surrogate_steps_per_s": value / n_sur,
            "repeat_values": [n_sur * args.steps / e for e in elapsed_all],
            "repeat_spread": (max(elapsed_all) - min(elapsed_all)) / elapsed,
            "step_flops_algorithmic": algorithmic_flops(cfg),
            "step_flops_executed_gemm": flops_step,
            # executed GEMM flops per second over the FP64 peak: what the chip really sustains over the whole step
            "step_executed_fp64_frac": flops_step * value / world / (FP64_PEAK_TFLOPS * 1e12),
            # work the step does NOT do because it cannot reach the loss (same ELBO, same gradients -- tests/, parity below)
            "dead_work": {"layer_rows": layer_rows, "rows_held_per_rank": cfg_rank["N"], "panel_columns": cols, "backward_active_fraction": act,
                          "dominant_layer": dom,
                          "what": "layer l runs on the rows of fidelity >= l (batch ordered once by descending fidelity); a "
                                  "layer backward skips 128-column blocks whose upstream gradients are all zero"},
            "reference_layout": dense_leg,
            "reference_sizes": small_leg,
            "value_over_reference_layout": (value / dense_leg["value"]) if dense_leg else None,
            # SURVEY 8(d)'s F_step prices the reference's solve-based op sequence at every row (~5.7x the flops executed
            # here at C3): steps/s x F_step in TFLOP/s is what a chip running the REFERENCE's operation count would have to
            # sustain to match this step rate.  It exceeds the FP64 peak because work was removed, so it is reported as an
            # equivalent rate, not as a fraction of peak; the roofline figure of the whole step is step_executed_fp64_frac.
            "reference_equivalent_tflops": algorithmic_flops(cfg) * value / world / 1e12,
            # the path's exchange: ONE all-gather of every surrogate's posterior moments on a 256-point grid;
            # `exchange_ms` = that collective alone (median of 5 after a warm call), `exchange_first_ms` = the moments of this
            # rank's surrogates + the first (warm-up) collective
            "exchange_ms": exchange_ms, "exchange_first_ms": exchange_first_ms, "exchange_repeats_ms": ex,
            "exchange_payload_bytes": int(local.numel() * 8),
            "rccl_ranks": (dist.get_world_size() if dist is not None else 0),
            "backend": (args.backend if dist is not None else None), "librccl_mapped": rccl_loaded, "rccl_error": rccl_error,
            "finite": finite, "step_issue": "eager" if args.eager else (
                ("one launch per step for all surrogates (%s)" % ("mobocmf_coop_elbo_step, %d workgroups per surrogate" % tiny.wgs_used
                                                                   if hasattr(tiny, "wgs_used") else "mobocmf_tiny_elbo_step"))
                if tiny is not None else "hip-graph replay"),
        }
        if not args.no_roofline:
            line["roofline"] = measure_dominant_kernel(cfg, device, n_cols=cols[dom], layer=dom)
            if layer_rows is not None and cols[dom] != cfg["N"] * cfg["S"]:
                # the same four launches on the panel the reference's layout gives the top layer (every row): the fraction the
                # kernel reaches when the panel is wide enough for a launch's fixed costs not to show
                var_ref, fl_ref, (Mp_r, Np_r) = measure_gemm_variants(cfg, device, 20, cfg["N"] * cfg["S"])
                tot_ref = sum(v["kernel_ms"] for v in var_ref)
                line["roofline"]["same_kernel_on_reference_layout_panel"] = {
                    "shape": [Mp_r, Np_r, Mp_r], "kernel_ms": var_ref[0]["kernel_ms"], "frac": var_ref[0]["frac"],
                    "weighted_frac": len(var_ref) * fl_ref / (tot_ref * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "variants": var_ref}
            if not args.eager and not rows:
                ins = measure_instep_kernels(gsteps[0], cfg, layer=dom)
                iso = {v["kernel"]: v["kernel_ms"] for v in line["roofline"]["variants"]}
                line["per_kernel_instep_ms"] = {
                    "kernels": ins, "instep_over_isolated": {k: ins[k] / iso[k] for k in ins if k in iso},
                    "how": "an EAGER SINGLE-STREAM PROBE, not the timed configuration: the widest layer of ONE surrogate, its step "
                           "issued eagerly on its own stream (the launch sequence the graph replays), HIP events recorded by the "
                           "library around each launch (mobocmf_layer_desc.probe_events), mean of 4 steps; `instep_over_isolated` "
                           "divides by the same launch timed alone (roofline.variants).  Per-kernel durations of the timed "
                           "configuration (3 streams, graph replay): profiles/r05_bench_C3_timed_only_kernel_summary.md"}
        if world == 1 and not args.no_cpu_baseline:
            cb, parity = cpu_baseline(cfg, device=device)
            line["cpu_baseline"] = cb
            line["parity"] = parity
            line["gpu_over_cpu"] = (value / n_sur) / cb["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
