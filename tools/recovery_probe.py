"""Config 5 end to end: train the amortizer with graph_trainer.GraphTrainer.train_experience_replay (the reference's call,
basic_ddm_dc.py:199-202) for N iterations on the MI355X, then the recovery loop of :218-223 in miniature: posterior means of 200 fresh
data sets vs their true parameters (correlation per parameter).  usage: python tools/recovery_probe.py [iterations]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesflow_nddms_amd import basic_ddm_dc
from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, posterior_recovery
from bayesflow_nddms_amd.graph_trainer import TRAIN_OFFSET_BASE, GraphTrainer
torch.manual_seed(0)
am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t0 = time.time()
with GraphTrainer(am, batch_size=32, total_steps=iters, seed=2023, offset_base=TRAIN_OFFSET_BASE) as gt:      # (evaluation below draws rows 0..)
    gt.train_experience_replay(iters)
    h = np.array(gt.loss_history())
print(f"{iters} iterations in {time.time()-t0:.1f} s; loss: first 50 {h[:50].mean():.3f}, 500-600 {h[500:600].mean():.3f}, last 100 {h[-100:].mean():.3f}; max {h.max():.2f} nan {np.isnan(h).sum()}")
np.random.seed(1)
gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
rho = posterior_recovery(am, gm, basic_ddm_dc.configurator, n_datasets=200, n_samples=500)
print("posterior-mean vs truth correlation per parameter (drift, boundary, beta, tau, dc):", np.round(rho, 3))
