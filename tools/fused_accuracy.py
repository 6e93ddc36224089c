#!/usr/bin/env python3
"""How accurate are the fused training kernels (csrc/train_kernels.hip, csrc/train_deepset.hip) against a float64 evaluation of the same
network, next to PyTorch's own float32 composition?  Per case and tensor group: max-norm and RMS error of both, and their ratios.
The flows are the ill-conditioned ones of tests/test_gpu_training.py::test_fused_flow_equals_the_pytorch_path (weights x 1.5 .. 3) plus
networks at their initialisation and a trained-like scale.   usage: python tools/fused_accuracy.py  (MI355X; profiles/r5_fused_accuracy.txt)"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesflow_nddms_amd.amortizer import InvertibleNetwork            # noqa: E402


def err(a, e):
    d = (a.double() - e)
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def main():
    torch.manual_seed(3)
    print(f"{'layers':>6} {'rows':>5} {'D':>2} {'scale':>5} | {'tensor':10} | {'max fused':>10} {'max torch':>10} {'ratio':>6} | {'rms fused':>10} {'rms torch':>10} {'ratio':>6} | max |f64|")
    worst_max, worst_rms = {}, {}
    for scale in (1.0, 1.5, 3.0):
        for layers, R, D in ((1, 32, 5), (2, 77, 5), (6, 256, 5), (6, 32, 8), (6, 32, 7), (6, 32, 5), (2, 4001, 5), (6, 512, 7)):
            for rep in range(3):
                net = InvertibleNetwork(num_params=D, num_coupling_layers=layers, seed=layers + 10 * rep).cuda()
                with torch.no_grad():
                    for p in net.parameters():
                        p.mul_(scale)
                    for p in list(net.an_scale) + list(net.an_bias):
                        p.copy_(0.3 * torch.randn_like(p))
                net64 = copy.deepcopy(net).double()
                theta = torch.randn(R, D, device="cuda", requires_grad=True)
                cond = torch.randn(R, 11, device="cuda", requires_grad=True)
                wz, wl = torch.randn(R, D, device="cuda"), torch.randn(R, device="cuda")
                th64, c64 = theta.detach().double().requires_grad_(True), cond.detach().double().requires_grad_(True)

                def run(n, th, cd):
                    z, ld = n(th, cd)
                    g = torch.autograd.grad((z * wz.to(z.dtype)).sum() + (ld * wl.to(z.dtype)).sum(), [th, cd] + list(n.parameters()))
                    return {"z": [z.detach()], "log|det|": [ld.detach()], "d theta": [g[0]], "d cond": [g[1]], "d weights": list(g[2:])}

                out = {}
                for fused in (True, False):
                    net.fused = fused
                    out[fused] = run(net, theta, cond)
                ex = run(net64, th64, c64)
                for key in ex:
                    rows = [(err(a, e), err(b, e), float(e.abs().max())) for a, b, e in zip(out[True][key], out[False][key], ex[key])]
                    # the tensor of the group where the fused kernel is worst relative to PyTorch (RMS)
                    (fm, fr), (pm, pr), mag = max(rows, key=lambda r: r[0][1] / max(r[1][1], 1e-300))
                    rm, rr = fm / max(pm, 1e-300), fr / max(pr, 1e-300)
                    worst_max[key] = max(worst_max.get(key, 0), rm)
                    worst_rms[key] = max(worst_rms.get(key, 0), rr)
                    if rep == 0:
                        print(f"{layers:6d} {R:5d} {D:2d} {scale:5.1f} | {key:10} | {fm:10.3g} {pm:10.3g} {rm:6.2f} | {fr:10.3g} {pr:10.3g} {rr:6.2f} | {mag:.3g}")
    print("largest ratio fused / pytorch-f32 of the error against f64 over all cases (3 networks per shape):")
    print("  max-norm:", {k: round(v, 2) for k, v in worst_max.items()})
    print("  RMS     :", {k: round(v, 2) for k, v in worst_rms.items()})


if __name__ == "__main__":
    main()
