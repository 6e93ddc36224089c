"""world_size-2 (and 3) gloo tests of the data-parallel path: shard -> simulate -> all-gather reassembles exactly
the unsharded batch.  The simulate step is injected (the CPU oracle on the same Philox stream) because the HIP
kernels need a GPU; the sharding / set_offset / gather logic under test is the product's (distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_simulate(model, params, n_trials, seed=0, set_offset=0, want_trials=True, want_summary=True, **kw):
    import oracle
    p = params.numpy() if isinstance(params, torch.Tensor) else np.asarray(params)
    if len(p) == 0:
        return {"trials": torch.empty((0, n_trials, 2)), "summary": torch.empty((0, 10))}
    r = oracle.philox_simulate(model, p, n_trials, dt=kw.get("dt", 0.01), max_steps=kw.get("max_steps", 400.0),
                               seed=seed, set_offset=set_offset, want_trials=want_trials, want_summary=want_summary)
    return {k: torch.from_numpy(v) for k, v in r.items()}


def _worker(rank, world, port, B, gather, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import prior_util
        from bayesflow_nddms_amd.distributed import ShardedSimulator, shard_bounds
        params = torch.from_numpy(prior_util.basic_prior(B, 77))        # identical on every rank (shared seed)
        sim = ShardedSimulator(0, simulate_fn=_oracle_simulate, gather=gather)
        out = sim(params, B, 50, seed=123, set_offset=1000)
        lo, hi = shard_bounds(B, world, rank)
        assert out["rows"] == (lo, hi)
        # same thing with a callable that only materialises this rank's rows
        out2 = sim(lambda a, b: params[a:b], B, 50, seed=123, set_offset=1000)
        for k in ("trials", "summary"):
            if k in out:
                assert torch.equal(torch.nan_to_num(out[k]), torch.nan_to_num(out2[k]))
        torch.save({k: v for k, v in out.items() if isinstance(v, torch.Tensor)}, os.path.join(tmpdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,gather", [(2, 64, "trials"), (2, 37, "both"), (3, 10, "summary"), (2, 5, "none")])
def test_sharded_equals_unsharded(world, B, gather, tmp_path, oracle_mod):
    import prior_util
    mp.spawn(_worker, args=(world, _free_port(), B, gather, str(tmp_path)), nprocs=world, join=True)
    params = prior_util.basic_prior(B, 77)
    full = _oracle_simulate(0, params, 50, seed=123, set_offset=1000)
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    if gather == "none":
        for k in ("trials", "summary"):
            cat = torch.cat([o[k] for o in outs], dim=0)
            assert torch.equal(torch.nan_to_num(cat), torch.nan_to_num(full[k]))
        return
    keys = {"trials": ["trials"], "summary": ["summary"], "both": ["trials", "summary"]}[gather]
    for o in outs:                       # every rank holds the whole minibatch, bit-identical to the unsharded run
        for k in keys:
            assert o[k].shape == full[k].shape
            assert torch.equal(torch.nan_to_num(o[k]), torch.nan_to_num(full[k]))
