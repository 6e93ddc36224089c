#!/usr/bin/env python3
"""BASELINE config 5 in miniature: the flow of the reference's basic_ddm_dc.py (generative model -> configurator ->
amortizer -> trainer.train_experience_replay), with the simulator on the MI355X and a PyTorch-ROCm amortizer.

Single GPU:   python examples/train_basic_ddm_dc.py --iterations 500
              python examples/train_basic_ddm_dc.py --iterations 5000 --graph      (one hipGraph replay per iteration: ~8x faster)
              python examples/train_basic_ddm_dc.py --iterations 5000 --trainer-graph   (the same through the reference's call shape:
                                                                                      amortizer.Trainer(..., graph=True).train_experience_replay(...))
Multi GPU:    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_basic_ddm_dc.py --sharded
              (each rank simulates batch/G parameter sets; one RCCL all-gather reassembles the minibatch on every rank)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesflow_nddms_amd import basic_ddm_dc, engine  # noqa: E402
from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer  # noqa: E402
from bayesflow_nddms_amd.distributed import ShardedSimulator, shared_prior_N  # noqa: E402
from bayesflow_nddms_amd.priors import DevicePrior  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--sharded", action="store_true")
    ap.add_argument("--graph", action="store_true", help="graph_trainer.GraphTrainer: the whole iteration as one hipGraph replay")
    ap.add_argument("--trainer-graph", action="store_true", help="amortizer.Trainer(graph=True): the classic call shape, run as graph replays")
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--max-steps", type=float, default=400.0)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # device_id: the communicator is created eagerly on THIS rank's card (as bench.py does); without it the first collective
        # picks a device from the rank number, which is the wrong card whenever LOCAL_RANK != RANK % gpus
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.manual_seed(0)                                   # identical initial weights on every rank

    if a.graph:
        from bayesflow_nddms_amd.graph_trainer import GraphTrainer
        amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        t0 = time.time()
        with GraphTrainer(amortizer, batch_size=a.batch_size, total_steps=a.iterations, dt=a.dt, max_steps=a.max_steps, seed=2023,
                          world=world, rank=rank) as trainer:
            trainer.train_experience_replay(a.iterations)          # the reference's call, basic_ddm_dc.py:199-202
            h = trainer.loss_history()
        if rank == 0:
            print(f"{a.iterations} graph-replayed iterations in {time.time()-t0:.1f} s; loss first 20: {np.mean(h[:20]):.3f}, "
                  f"last 20: {np.mean(h[-20:]):.3f}")
        return

    if a.sharded:
        prior = DevicePrior("basic", seed=2023)
        sim = ShardedSimulator(engine.BASIC_DDM_DC, gather="trials")
        step = {"i": 0}

        def generative_model(batch_size):
            i = step["i"]; step["i"] += 1
            n = shared_prior_N(2023, i)                     # batch-shared N without communication
            base = i * batch_size
            rows = lambda lo, hi: prior(hi - lo, set_offset=base + lo)
            out = sim(rows, batch_size, n, seed=2023, set_offset=base, dt=a.dt, max_steps=a.max_steps)
            return {"prior_draws": prior(batch_size, set_offset=base), "sim_data": out["trials"],
                    "sim_non_batchable_context": n}
    else:
        generative_model = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False,
                                                              dt=a.dt, max_steps=a.max_steps)
    amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    trainer = Trainer(amortizer, generative_model, basic_ddm_dc.configurator, checkpoint_path=None, graph=a.trainer_graph and not a.sharded)
    t0 = time.time()
    res = trainer.train_experience_replay(epochs=1, iterations_per_epoch=a.iterations, batch_size=a.batch_size,
                                          save_checkpoint=False)
    torch.cuda.synchronize()
    h = res["train_losses"]
    if rank == 0:
        print(f"{a.iterations} iterations in {time.time()-t0:.1f} s; loss first 20: {np.mean(h[:20]):.3f}, last 20: {np.mean(h[-20:]):.3f}")


if __name__ == "__main__":
    main()
