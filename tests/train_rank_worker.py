"""Child process of tests/test_gpu_training.py: one rank of a process group (gloo with every rank on cuda:0 -- a one-GPU box --
or RCCL at world 1) that runs graph_trainer.GraphTrainer in the multi-rank forms of BASELINE configs[4]
(basic_ddm_dc.py:199-202 with the simulation sharded): `gather` (one all-gather per minibatch, replicated training step) and
`ddp` (flat-gradient all-reduce).  Every rank builds its amortizer from a DIFFERENT torch seed: the trainer must make the
replicas identical itself.  Writes what it saw to <out_dir>/rank<r>.json."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, backend, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    res = {}
    try:
        from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
        from bayesflow_nddms_amd.graph_trainer import GraphTrainer

        def digest(t):
            return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()

        for parallel in ("gather", "ddp"):
            for form in ("pipelined", "sequential", "eager"):
                torch.manual_seed(1000 + 17 * rank)                       # replicas start DIFFERENT: other weights, other permutations
                am = AmortizedPosterior(InvertibleNetwork(num_params=5, seed=rank), InvariantNetwork())
                gt = GraphTrainer(am, batch_size=32, total_steps=iters, seed=2023, learning_rate=1e-3, world=world, rank=rank,
                                  parallel=parallel, backend=backend, split=True, use_graph=form != "eager",
                                  overlap=form == "pipelined")
                r = {"w0": digest(gt.flat_p), "perm0": am.inference_net._perm_host}
                gt.train_online(iters - 6)
                gt.train_experience_replay(6, capacity_in_batches=4)
                r.update(loss=gt.loss_history(), w1=digest(gt.flat_p), offset=int(gt.offset.item()), graphs=gt.n_graphs,
                         minibatch=int(gt._bucket(gt.n_max).t_params.shape[0]))
                gt.close()
                res[f"{parallel}/{form}"] = r
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
