"""Data parallelism over parameter sets: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

Parameter sets are independent (basic_ddm_dc.py:121-122 has no cross-trial state), so a batch of B sets shards
into contiguous row blocks, one per rank, with NO collective on the simulate path: the random stream is keyed by
the GLOBAL set index (set_offset + row), so the union of the shards is bit-identical to the unsharded batch for any
number of GPUs.  The only exchange step is the optional all-gather that reassembles a training minibatch on every
rank (north_star); gathering only the fused summaries (40 B/set) instead of the trials (8 B/trial) makes it ~60x
smaller at N=300.
"""
import numpy as np


def shard_bounds(n_sets, world_size, rank):
    """Contiguous block of rows for `rank`: blocks of ceil(B/G), the last ones possibly short or empty."""
    per = (n_sets + world_size - 1) // world_size
    lo = min(rank * per, n_sets)
    hi = min(lo + per, n_sets)
    return lo, hi


def _dist():
    import torch.distributed as dist
    return dist


def all_gather_rows(local, n_sets, group=None):
    """All-gather row blocks produced under shard_bounds() into the full [n_sets, ...] tensor on every rank.
    Short blocks are padded to the common block size for the collective and trimmed afterwards."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    per = (n_sets + world - 1) // world
    pad = per - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], dim=0)
    local = local.contiguous()
    # The collective's form is chosen by the BACKEND, never by catching an error: a failed RCCL collective (a shape mismatch between
    # ranks, a broken communicator) must surface as its own message, not be followed by a second collective on the same communicator.
    if flat_all_gather(dist.get_backend(group)):
        full = local.new_empty((world * per,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(full, local, group=group)     # one RCCL all-gather into one tensor
    else:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)                # gloo (the CPU rehearsal): the list form
        full = torch.cat(parts, dim=0)
    return full[:n_sets]


def flat_all_gather(backend):
    """True for the backends that implement all_gather_into_tensor (nccl = RCCL on ROCm); gloo -- the CPU rehearsal of the
    multi-rank path -- has the list form only."""
    return "nccl" in str(backend).lower()          # ("nccl", or a per-device map such as "cpu:gloo,cuda:nccl")


class ShardedSimulator:
    """`sim(params, n_trials, ...)` over a process group.

    params: the GLOBAL [B, P] parameter matrix (identical on every rank: drawn from a shared seed, or on the device
    with the counter-based prior), or a callable (lo, hi) -> rows [hi-lo, P] that materialises only this rank's block.
    simulate_fn(model, params_rows, n_trials, seed=, set_offset=, **kw) -> dict with 'trials' / 'summary' tensors;
    the default is the HIP engine.  gather: 'trials' | 'summary' | 'both' | 'none' | 'codes' (the trials, exchanged as 2-byte
    codes + the parameter rows and decoded on arrival: basic_ddm_dc / alpha_not_scaled without the bridge, HIP engine only).
    """

    def __init__(self, model, simulate_fn=None, group=None, gather="trials"):
        if gather not in ("trials", "summary", "both", "none", "codes"):
            raise ValueError("gather must be 'trials', 'summary', 'both', 'none' or 'codes'")
        if simulate_fn is None:
            from . import engine
            simulate_fn = engine.simulate
        self.model, self.simulate_fn, self.group, self.gather = model, simulate_fn, group, gather

    def __call__(self, params, n_sets, n_trials, seed, set_offset=0, **kw):
        dist = _dist()
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        else:
            world, rank = 1, 0
        lo, hi = shard_bounds(n_sets, world, rank)
        rows = params(lo, hi) if callable(params) else params[lo:hi]
        if self.gather == "codes":
            return self._gather_codes(rows, n_sets, n_trials, seed, set_offset + lo, world, rank, (lo, hi), kw)
        want_t = self.gather in ("trials", "both", "none")
        want_s = self.gather in ("summary", "both", "none")
        res = self.simulate_fn(self.model, rows, n_trials, seed=seed, set_offset=set_offset + lo,
                               want_trials=want_t, want_summary=want_s, **kw)
        out = {"rank": rank, "world_size": world, "rows": (lo, hi)}
        for key, want in (("trials", want_t), ("summary", want_s)):
            if not want or key not in res:
                continue
            if world > 1 and self.gather != "none":
                out[key] = all_gather_rows(res[key], n_sets, self.group)
            else:
                out[key] = res[key]
        return out


    def _gather_codes(self, rows, n_sets, n_trials, seed, offset, world, rank, bounds, kw):
        import torch
        from . import engine
        res = self.simulate_fn(self.model, rows, n_trials, seed=seed, set_offset=offset, want_trials=False, want_summary=False,
                               want_codes=True, **kw)
        codes, p = res["codes"], res["params"]
        if world > 1:
            codes = all_gather_rows(codes.view(torch.uint8), n_sets, self.group).view(torch.int16)     # bytes: RCCL has no int16
            p = all_gather_rows(p, n_sets, self.group)
        return {"rank": rank, "world_size": world, "rows": bounds,
                "trials": engine.decode_codes(self.model, codes, p, kw.get("dt", 0.01))}


def shared_prior_N(seed, step, n_min=60, n_max=300):
    """The batch-shared number of trials (basic_ddm_dc.py:50-52, 131) drawn identically on every rank without
    communication: a pure function of (seed, step)."""
    return int(np.random.default_rng([int(seed), int(step)]).integers(n_min, n_max + 1))
