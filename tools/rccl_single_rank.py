"""Every exchange of mobocmf_amd.parallel on the nccl (= RCCL) backend with ONE rank, on device tensors: the collectives an
N-rank job issues (JESMOC_MFDGP.py:125-135 coupled acquisition; blackbox_mfdgp_fitter.py:317-341 omega-factor moments), run on
the one GPU a test box has.  The process group is created BEFORE anything else touches the GPU.  Prints one JSON line:
results equal the inputs, librccl is mapped, median-of-5 timings (ms) after a warm call.
    python tools/rccl_single_rank.py            (tests/test_hip_rccl_single_rank.py runs it in a fresh process)"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rdv = os.path.join(tempfile.mkdtemp(prefix="mobocmf_rccl_"), "rdv")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="file://" + rdv, rank=0, world_size=1, device_id=dev)
    from mobocmf_amd import parallel
    rec = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, sorted(ts)[2]

    ok = {}
    # coupled acquisition: per-output moments on a grid (3 surrogates x (mean, var) x 256 points)
    mom = rnd(3, 2, 256)
    out, rec["all_gather_moments_ms"] = timed(lambda: parallel.all_gather_moments(mom))
    ok["all_gather_moments"] = bool(torch.equal(out, mom)) and out.data_ptr() != mom.data_ptr()
    # ragged shards (row counts first, then padded rows)
    rag = rnd(2, 77)
    out, rec["all_gather_ragged_ms"] = timed(lambda: parallel.all_gather_ragged(rag))
    ok["all_gather_ragged"] = len(out) == 1 and bool(torch.equal(out[0], rag))
    empty = rag[:0]
    out = parallel.all_gather_ragged(empty)
    ok["all_gather_ragged_empty"] = len(out) == 1 and out[0].shape == (0, 77)
    out, rec["coupled_acquisition_ms"] = timed(lambda: parallel.coupled_acquisition(rag))
    ok["coupled_acquisition"] = bool(torch.allclose(out, rag.sum(0), rtol=0, atol=0))
    # omega-factor moments with the layout-hash header row and global indices (own rows keep their autograd history)
    fm, fv = rnd(2, 10).requires_grad_(), rnd(2, 10).abs().requires_grad_()
    cm, cv = rnd(1, 10).requires_grad_(), rnd(1, 10).abs().requires_grad_()
    call = lambda: parallel.gather_with_local_grad(fm, fv, cm, cv, obj_index=[1, 0], con_index=[0])
    (afm, afv, acm, acv), rec["gather_with_local_grad_ms"] = timed(call)
    ok["gather_with_local_grad"] = bool(torch.equal(afm, fm[[1, 0]]) and torch.equal(afv, fv[[1, 0]]) and
                                        torch.equal(acm, cm) and torch.equal(acv, cv))
    (afm.sum() + 2 * acv.sum()).backward()
    ok["gather_with_local_grad_backward"] = bool(torch.equal(fm.grad, torch.ones_like(fm)) and
                                                 torch.equal(cv.grad, 2 * torch.ones_like(cv)))
    # float32 moments carry the same header (two 24-bit halves)
    f32 = lambda t: t.detach().float()
    parallel.reset_gather_plans()
    a32 = parallel.gather_with_local_grad(f32(fm), f32(fv), f32(cm), f32(cv), obj_index=[1, 0], con_index=[0])
    ok["gather_with_local_grad_float32"] = bool(torch.equal(a32[0], f32(fm)[[1, 0]]))
    # broadcast of the points every rank must agree on; gradient bucket all-reduce of the row-sharded step
    xt = rnd(10, 8)
    ref = xt.clone()
    _, rec["broadcast_ms"] = timed(lambda: parallel.broadcast_(xt))
    ok["broadcast"] = bool(torch.equal(xt, ref))
    params = [torch.nn.Parameter(rnd(512, 512)), torch.nn.Parameter(rnd(512))]
    bucket = parallel.GradBucket(params, extra=2)
    bucket.flat.copy_(rnd(bucket.flat.numel()))
    ref = bucket.flat.clone()
    _, rec["grad_bucket_all_reduce_ms"] = timed(bucket.all_reduce)
    ok["grad_bucket_all_reduce"] = bool(torch.equal(bucket.flat, ref)) and params[0].grad.data_ptr() == bucket.flat.data_ptr()
    rec["grad_bucket_bytes"] = int(bucket.flat.numel() * 8)
    dist.barrier()
    torch.cuda.synchronize()
    rec["librccl_mapped"] = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
    rec["ok"] = ok
    rec["all_ok"] = all(ok.values()) and bool(rec["librccl_mapped"])
    dist.destroy_process_group()
    print(json.dumps(rec))
    return 0 if rec["all_ok"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
