#!/usr/bin/env python3
"""ELBO steps/sec of the MFDGP hot path on MI355X (BASELINE.json metric, headline config C3).

One bench "step" = one ELBO step (zero_grad + MFDGP.forward + VariationalELBOMF + backward + Adam,
mobocmf/util/blackbox_mfdgp_fitter.py:161-171) for EACH of the 3 surrogates (2 objectives + 1 constraint)
of C3 on the full batch B = N.  `value` = surrogate ELBO steps per second, whole job.  With --gpus N every
rank trains its own 3 surrogates (weak scaling, no data-path collective: the surrogates are independent,
SURVEY 8(e)); the single RCCL all-gather of posterior moments for the joint acquisition happens once after
the timed region and is reported as `exchange_ms`.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
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


def build_graphed(cfg, outputs, device, use_graph, shard_rows=False):
    """One GraphedELBOStep (HIP-graph replay of the whole step) per surrogate, each on its own stream.
    shard_rows: every rank holds the SAME surrogates and 1/W of their batch rows (SURVEY 8(e) level 2)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    if shard_rows:
        from mobocmf_amd.parallel import RowShardedELBOStep as GraphedELBOStep
    steps = []
    for o in outputs:
        prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], output=o % 3, seed=o)
        model = synthetic.model_from_problem(prob, device=device)
        elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device)
        steps.append(GraphedELBOStep(model, elbo, t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], lr=1e-3,
                                     use_graph=use_graph))
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


def measure_dominant_kernel(cfg, device, iters=20):
    """Average duration of the dominant kernel (gemm_f64_kernel<false>, lower-triangular left operand: A = L^-1 K_mn
    at the top layer's shape), HIP events on the stream it is launched on."""
    from mobocmf_amd import functional as F
    Mp = (cfg["M"] + 127) // 128 * 128
    Np = (cfg["N"] * cfg["S"] + 127) // 128 * 128
    A = torch.tril(torch.randn(Mp, Mp, dtype=torch.float64, device=device))
    B = torch.randn(Mp, Np, dtype=torch.float64, device=device)
    C = torch.empty(Mp, Np, dtype=torch.float64, device=device)
    for _ in range(3):
        F.gemm_f64(A, B, C, tri=1)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        F.gemm_f64(A, B, C, tri=1)
    en.record()
    torch.cuda.synchronize()
    sec = st.elapsed_time(en) * 1e-3 / iters
    flops = float(cfg["M"]) ** 2 * cfg["N"] * cfg["S"]          # algorithmic: M^2 N' (triangular product)
    traffic = None      # HBM bytes per launch from the PMC passes committed under profiles/ (same kernel, same shape)
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_gemm.json")) as fh:
            pm = json.load(fh)
        if Mp == 512 and Np == 65536:
            traffic = pm["traffic_bytes_per_launch"]
    except Exception:
        pass
    return {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / sec / 1e12 / FP64_PEAK_TFLOPS, "traffic": traffic,
            "kernel": "gemm_f64_kernel<false> (A = L^-1 K_mn, %dx%dx%d lower-triangular)" % (Mp, Np, Mp),
            "kernel_ms": sec * 1e3, "flops_per_launch": flops}


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


def cpu_baseline(cfg, steps=2, device=None):
    """Reference-equivalent float64 torch-CPU restatement (gpytorch unavailable): the oracle executing GPyTorch's
    op sequence + autograd + torch.optim.Adam, ONE surrogate at the full C3 size, 1 warm-up + `steps` timed steps."""
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
    for k in range(steps + 1):
        t0 = time.perf_counter()
        O.elbo_step(raw, opt, x, y, fid, eps, cfg["S"], ref_equiv=True)
        times.append(time.perf_counter() - t0)
        if sum(times) > 40.0 and len(times) >= 2:      # bounded sample: never more than ~1 minute of CPU work
            break
    med = sorted(times[1:])[len(times[1:]) // 2]
    steps = len(times) - 1
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = [ln.split(":")[1].strip() for ln in fh if ln.startswith("model name")][0]
    except Exception:
        pass
    return {"value": 1.0 / med, "unit": "ELBO steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 surrogate of C3 at full size (d=8 M=512 N=8192 S=8), 1 warm-up + %d timed steps, median; "
                      "reference-equivalent CPU restatement (gpytorch unavailable)" % steps,
            "os_cpu_count": os.cpu_count(), "cpu_model": model, "sec_per_step": med}, parity


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
        model.eval()
        mu, var = model.predict_for_acquisition(dv(Xt), cfg["L"] - 1)
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    return {"elbo_rel_err": abs(float(e_gpu) - float(e_ref)) / abs(float(e_ref)), "pred_mean_rel_err": rel(mu, mu_ref),
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
    ap.add_argument("--steps", type=int, default=20)
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
    args = ap.parse_args()
    if args.no_overlap:
        from mobocmf_amd.models import MFDGP
        MFDGP.overlap_chains = False

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE != --gpus")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

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
        gsteps = build_graphed(cfg, outputs, device, use_graph=True, shard_rows=rows)
        sur = [(g.model, g.elbo, g.optimizer, (g.x, g.y, g.fid)) for g in gsteps]

        def one_step(*_a):
            return [g.step()[0] for g in gsteps]
    else:
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

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = one_step(sur, cfg, gens, streams)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = all(bool(torch.isfinite(l)) for l in losses)

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
        gathered = parallel.all_gather_moments(torch.stack(local))
    torch.cuda.synchronize()
    exchange_ms = (time.perf_counter() - t1) * 1e3
    finite = finite and bool(torch.isfinite(gathered).all())

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
                       "surrogates_per_gpu": n_out, "parallelism": ("row-sharded x%d + grad all-reduce" if rows else "surrogate-per-rank x%d") % world},
            "per_surrogate_steps_per_s": value / n_sur,
            "step_flops_algorithmic": algorithmic_flops(cfg),
            "step_fp64_frac": algorithmic_flops(cfg) * value / world / (FP64_PEAK_TFLOPS * 1e12),
            "exchange_ms": exchange_ms, "finite": finite, "step_issue": "eager" if args.eager else "hip-graph replay",
        }
        if not args.no_roofline:
            line["roofline"] = measure_dominant_kernel(cfg, device)
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
