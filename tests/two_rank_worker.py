"""Child process of tests/test_gpu_bench_contract.py::test_two_fresh_ranks_union_equals_unsharded: one rank of a gloo
process group whose ranks all use cuda:0 (a one-GPU box).  Simulates its shard with the HIP engine through
distributed.ShardedSimulator, gathers trials + summaries and saves what it holds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import prior_util
        from bayesflow_nddms_amd import engine
        from bayesflow_nddms_amd.distributed import ShardedSimulator
        p = torch.as_tensor(prior_util.basic_prior(B, 21)).cuda()
        res = {}
        for gather in ("both", "none", "codes"):
            sim = ShardedSimulator(engine.BASIC_DDM_DC, gather=gather)
            out = sim(p, B, N, seed=31, set_offset=12345, dt=0.001, max_steps=4000, fast=True)
            res[gather] = {k: v.cpu() for k, v in out.items() if isinstance(v, torch.Tensor)}
            res[gather]["rows"] = out["rows"]
        torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
