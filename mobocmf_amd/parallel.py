"""Multi-GPU layout of the path: one process per GPU, independent surrogates sharded over the
ranks with no data-path collective, and ONE all-gather (RCCL over xGMI; gloo on CPU for tests) per exchange
point: the per-output posterior moments for the joint acquisition (JESMOC_MFDGP.py:125-135).

Payloads are tiny (<= 0.6 MB), i.e. latency-bound: a single all-gather on the default communicator is the
whole collective; there is nothing to bucket or overlap.
"""
import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_blackboxes(names, rank=None, world_size=None):
    """Round-robin assignment of black-box (objective / constraint) names to ranks; every rank derives the
    same table.  Returns (my_names, owner_of) with owner_of[name] = rank."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    names = list(names)
    owner = {n: i % world_size for i, n in enumerate(names)}
    return [n for n in names if owner[n] == rank], owner


def all_gather_moments(local):
    """local: (k, ...) tensor of this rank's k surrogates -> (world*k, ...) in rank order.  World size 1
    degenerates to the identity (no communicator needed)."""
    import torch.distributed as dist
    _, w = world()
    if w == 1:
        return local
    local = local.contiguous()
    dev = local.device
    if dev.type == "cuda" and dist.get_backend() == "gloo":
        local = local.cpu()                       # CPU rehearsal of the multi-rank path (tests); RCCL takes HBM buffers
    out = torch.empty((w * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)      # concatenation along dim 0, rank order
    return out.to(dev)


def coupled_acquisition(local_acq):
    """Sum over ALL black-boxes of the per-black-box acquisition (JESMOC_MFDGP.py:125-135): each rank holds
    the (k_local, T) values of its own surrogates; one all-gather, then a local sum."""
    return all_gather_moments(local_acq).sum(0)


def gather_with_local_grad(fm, fv, cm, cv):
    """omega-factor coupling of the conditioned training (blackbox_mfdgp_fitter.py:317-341) when the surrogates are
    sharded over ranks: every rank needs ALL models' (mean, var) at the 10 x~ points; its own rows keep their autograd
    history, the other ranks' rows arrive as constants (one all-gather of 2 x 10 doubles per model).  Ranks must hold
    the same number of objectives and of constraints.  World size 1: identity."""
    _, w = world()
    if w == 1:
        return fm, fv, cm, cv
    r, _ = world()

    def mix(local):
        if local.shape[0] == 0:
            return local
        allv = all_gather_moments(local.detach())
        k = local.shape[0]
        return torch.cat([allv[:r * k], local, allv[(r + 1) * k:]], 0)

    return mix(fm), mix(fv), mix(cm), mix(cv)
