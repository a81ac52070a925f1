"""Timing of the cooperative one-launch step (mobocmf_coop_elbo_step) against the layer path (HIP-graph replay) and, where it
applies, the one-workgroup kernel: us per step of a group of surrogates.
    python tools/coop_sweep.py [--quick]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobocmf_amd.mlls import VariationalELBOMF      # noqa: E402
from mobocmf_amd.util import synthetic      # noqa: E402
from mobocmf_amd.util.coop_step import CoopELBOStep      # noqa: E402
from mobocmf_amd.util.graphed_step import GraphedELBOStep      # noqa: E402

DEV = torch.device("cuda", 0)


def build(cfg, n_sur):
    out = []
    for o in range(n_sur):
        prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], seed=o, output=o)
        model = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=DEV)
        perm = torch.as_tensor(np.random.default_rng(5 + o).permutation(cfg["N"]), device=DEV)
        out.append((model, tc(prob["x"])[perm].contiguous(), tc(prob["y"])[perm].contiguous(), tc(prob["fid"])[perm].contiguous()))
    return out


def time_steps(fn, sync, n, reps=3):
    for _ in range(5):
        fn()
    sync()
    best = 1e30
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--wgs", default="0,1,2,4,8,16,32")
    ap.add_argument("--no-layer-path", action="store_true")
    args = ap.parse_args()
    cases = [("M=N=64 S=1 d=2, 1 surrogate (reference BO loop, iteration ~50)", dict(d=2, L=2, M=64, N=64, S=1), 1),
             ("M=N=64 S=1 d=2, 4 surrogates", dict(d=2, L=2, M=64, N=64, S=1), 4),
             ("M=N=32 S=1 d=2, 4 surrogates", dict(d=2, L=2, M=32, N=32, S=1), 4),
             ("M=N=128 S=1 d=2, 4 surrogates", dict(d=2, L=2, M=128, N=128, S=1), 4),
             ("C2: d=2 M=128 N=512 S=8, 3 surrogates (bench.py --config C2)", dict(d=2, L=2, M=128, N=512, S=8), 3),
             ("C2: d=2 M=128 N=512 S=8, 4 surrogates", dict(d=2, L=2, M=128, N=512, S=8), 4),
             ("C2 alone: 1 surrogate", dict(d=2, L=2, M=128, N=512, S=8), 1)]
    if args.quick:
        cases = [cases[0], cases[4]]
    wgs = [int(v) for v in args.wgs.split(",")]
    for name, cfg, n_sur in cases:
        print("== %s" % name)
        sur = build(cfg, n_sur)
        if not args.no_layer_path:
            gs = [GraphedELBOStep(m, VariationalELBOMF(m, cfg["N"], cfg["L"]), x, y[:, None], f[:, None], lr=1e-3) for m, x, y, f in sur]
            us = time_steps(lambda: [g.step() for g in gs], torch.cuda.synchronize, args.steps)
            print("   layer path (HIP-graph replay, %d streams): %8.1f us per round of %d = %8.0f steps/s" % (n_sur, us, n_sur, n_sur / us * 1e6))
            for g in gs:
                g.retire()
        sur = build(cfg, n_sur)
        for k in wgs:
            try:
                st = CoopELBOStep([s[0] for s in sur], [cfg["N"]] * n_sur, [s[1] for s in sur], [s[2] for s in sur],
                                  [s[3] for s in sur], lr=1e-3, force=True)
                st.wgs_per_model = k
                us = time_steps(st.step, st.stream.synchronize, args.steps)
                st.check()
                print("   one launch, %2d workgroups per surrogate (asked %2d):  %8.1f us per launch = %8.0f steps/s" %
                      (st.wgs_used, k, us, n_sur / us * 1e6))
            except Exception as e:      # noqa: BLE001
                print("   wgs %d: %s" % (k, str(e)[:100]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
