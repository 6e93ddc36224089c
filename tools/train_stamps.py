#!/usr/bin/env python3
"""Where does a training kernel's time go?  Builds csrc/train_*.hip with -DNDDM_TRAIN_STAMPS (thread 0 of one workgroup writes
(phase id, 100 MHz wall clock) pairs at the kernels' barriers), runs the flow / the summary network forward and backward once
at the training loop's shape (32 sets, 300 trials) and prints the time between consecutive stamps, grouped by phase pair.
The numbers quoted in HISTORY.md section C.6 (forward 7.8 us and backward 19 -> 10 us per half-layer, ...) come from here.

usage (on the GPU box): python tools/train_stamps.py [flow|deepset] [workgroup]"""
import collections
import ctypes
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
which = sys.argv[1] if len(sys.argv) > 1 else "flow"
block = sys.argv[2] if len(sys.argv) > 2 else "0"
so = os.path.join(root, "tools", "_stamps.so")
src = " ".join(os.path.join(root, "bayesflow_nddms_amd", "csrc", f) for f in ("train_kernels.hip", "train_deepset.hip", "train_update.hip"))
if os.system(f"hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DNDDM_TRAIN_STAMPS -DNDDM_STAMP_BLOCK={block} {src} -o {so} 2>/dev/null"):
    sys.exit("hipcc failed")
import bayesflow_nddms_amd.build as b          # noqa: E402
b.build_train = lambda *a, **k: so
from bayesflow_nddms_amd import _train_lib     # noqa: E402
from bayesflow_nddms_amd.amortizer import InvariantNetwork, InvertibleNetwork   # noqa: E402

L = _train_lib.lib()
read = L.nddm_train_read_stamps if which == "flow" else L.nddm_deepset_read_stamps
read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8192)()
if which == "flow":
    net = InvertibleNetwork(num_params=5).cuda()
    theta, cond = torch.randn(32, 5, device="cuda"), torch.randn(32, 11, device="cuda")
    fwd = lambda: net.nll(theta, cond)
else:
    net = InvariantNetwork().cuda()
    x = torch.stack([0.3 + torch.rand(32, 300, device="cuda") * 2.0, (torch.rand(32, 300, device="cuda") < 0.7).float()], dim=-1)
    fwd = lambda: net(x).sum()
for _ in range(3):
    out = fwd()
    torch.cuda.synchronize()
    n = read(buf, 4096)
    st_f = [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]
    torch.autograd.grad(out, list(net.parameters()))
    torch.cuda.synchronize()
    n = read(buf, 4096)
    st_b = [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]
for name, st in (("forward", st_f), ("backward", st_b)):
    if len(st) < 2:
        continue
    per = collections.OrderedDict()
    for (i0, t0), (i1, t1) in zip(st[:-1], st[1:]):
        if t1 >= t0:
            per.setdefault((i0, i1), []).append((t1 - t0) * 10)      # 100 MHz -> ns
    print(f"{which} {name}, workgroup {block}: {len(st)} stamps, first to last {(st[-1][1] - st[0][1]) / 100:.1f} us (all launches of the pass)")
    for k, v in per.items():
        print("  %5d -> %5d  n=%3d  mean %7.0f ns  sum %7.1f us" % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1000))
