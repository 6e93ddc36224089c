"""Child process of tests/test_gpu_training.py::test_pipelined_loops_feed_the_training_graph_the_right_batch_soak: an INTEGRITY soak of
the pipelined loop's stream choreography (two buffer sets per bucket used in turn, the communication stream staging the gathered
minibatch behind the all-gather, the producer copying the fresh batch out for the replay buffer, free / ready events).  With the
learning rate at ZERO the weights never move, so the loss of iteration k is a function of batch k alone: the pipelined loop's history
must equal the sequential loop's iteration for iteration -- a training graph that ever read a half-written, stale or overwritten
batch shows as a mismatch at that iteration.  One rank, RCCL (world 1) for the forms with a collective.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_online, n_replay = int(sys.argv[1]), int(sys.argv[2])
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {}
    try:
        from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
        from bayesflow_nddms_amd.graph_trainer import GraphTrainer

        def run(overlap, **kw):
            torch.manual_seed(0)
            am = AmortizedPosterior(InvertibleNetwork(num_params=7 if kw.get("model") == "single" else 5), InvariantNetwork())
            with GraphTrainer(am, batch_size=32, total_steps=n_online + n_replay, seed=2023, learning_rate=0.0, dt=0.001, max_steps=4000.0,
                              overlap=overlap, **kw) as gt:
                gt.train_online(n_online)
                gt.train_experience_replay(n_replay, capacity_in_batches=16)
                return np.array(gt.loss_history()), int(gt.offset.item())

        for name, kw in (("one rank", dict()), ("gather", dict(split=True, parallel="gather")), ("ddp", dict(split=True, parallel="ddp")),
                         ("one rank, single-trial model", dict(model="single"))):
            (hp, op), (hs, os_) = run(True, **kw), run(False, **kw)
            d = np.abs(hp - hs)
            out[name] = {"iterations": int(len(hp)), "max_abs_diff": float(d.max()), "n_mismatch": int((d > 1e-5 * (1 + np.abs(hs))).sum()),
                         "first_mismatch": int(np.argmax(d > 1e-5 * (1 + np.abs(hs)))) if (d > 1e-5 * (1 + np.abs(hs))).any() else -1,
                         "offsets_equal": op == os_, "finite": bool(np.all(np.isfinite(hp))), "spread": float(hs.std())}
        print(json.dumps(out))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
