#!/usr/bin/env python3
"""Where does the training stream's time go in the pipelined loop?  (No profiler: rocprofv3's per-launch host overhead makes the
graph loop host-bound, so its timelines say nothing about the unprofiled run.)  With NDDM_TRAIN_STAGE_STAMPS=1 the trainer records
four timing events per iteration ON THE TRAINING STREAM: iteration start | after the wait for the producer's event | before the
training graph | after it.  Prints the median of each stretch over the steady state, for the one-rank form and the RCCL forms at
world 1.   usage: python tools/train_stage_times.py [plain|gather|ddp] [dt]"""
import os
import sys

os.environ["NDDM_TRAIN_STAGE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    form = sys.argv[1] if len(sys.argv) > 1 else "plain"
    dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    import numpy as np
    import torch
    import torch.distributed as dist
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    torch.cuda.set_device(0)
    if form != "plain":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    with GraphTrainer(am, batch_size=32, total_steps=1000, dt=dt, max_steps=4.0 / dt, seed=2023, parallel="ddp" if form == "ddp" else "gather",
                      split=form != "plain") as gt:
        gt.train_online(200)                       # captures
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        gt.train_online(300)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 300
        st = gt.stage_stamps
        rows = []
        for k in range(20, len(st) - 1):
            a = st[k]
            if form == "ddp":
                rows.append((a[0].elapsed_time(a[1]) * 1e3, float("nan"), float("nan"), a[0].elapsed_time(st[k + 1][0]) * 1e3))
            else:
                rows.append((a[0].elapsed_time(a[1]) * 1e3, a[1].elapsed_time(a[2]) * 1e3, a[2].elapsed_time(a[3]) * 1e3,
                             a[0].elapsed_time(st[k + 1][0]) * 1e3))
        m = np.nanmedian(np.array(rows), axis=0)
        print(f"{form:7s} dt={dt:g}: {1.0 / el:7.0f} it/s ({el * 1e6:.1f} us per iteration by the host clock); training stream, medians (us): "
              f"wait for the producer {m[0]:.1f} | staging copies + fill {m[1]:.1f} | training graph {m[2]:.1f} | start to next start {m[3]:.1f}")
    if form != "plain":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
