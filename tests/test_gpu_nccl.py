"""RCCL (backend "nccl") smoke on one GPU: a single-rank process group exercises the same collective calls the
multi-GPU path makes (all_gather_into_tensor of padded row blocks, barrier with device ids, MAX all-reduce)."""
import os
import socket

import numpy as np
import pytest

import prior_util

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_gather_matches_local():
    import torch
    import torch.distributed as dist
    from bayesflow_nddms_amd import engine
    from bayesflow_nddms_amd.distributed import ShardedSimulator, all_gather_rows
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        p = torch.as_tensor(prior_util.basic_prior(50, 3)).cuda()
        sim = ShardedSimulator(engine.BASIC_DDM_DC, gather="both")
        out = sim(p, 50, 120, seed=11, set_offset=7, dt=0.01, max_steps=400, fast=False)
        ref = engine.simulate(engine.BASIC_DDM_DC, p, 120, seed=11, set_offset=7, dt=0.01, max_steps=400, fast=False)
        assert torch.equal(out["trials"], ref["trials"])
        full = all_gather_rows(ref["trials"][:37], 37)          # the collective itself, with padding logic
        assert torch.equal(full, ref["trials"][:37])
        dist.barrier(device_ids=[0])
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()
