"""Minimal PyTorch-ROCm stand-in for the BayesFlow objects the reference builds at basic_ddm_dc.py:163-176 and drives
at :199-207, :223 (SURVEY f-4 / BASELINE config 5): a DeepSet summary network, a conditional affine-coupling flow, an
amortized posterior, and a trainer with online and experience-replay loops that consume `generative_model(B)` ->
`configurator(dict)` exactly as BayesFlow's Trainer does.

    summary_net = InvariantNetwork()
    inference_net = InvertibleNetwork(num_params=num_params)
    amortizer = AmortizedPosterior(inference_net, summary_net)
    trainer = Trainer(amortizer=amortizer, generative_model=generative_model, configurator=configurator,
                      checkpoint_path=f"checkpoint/{model_name}")
    losses = trainer.train_experience_replay(epochs=…, batch_size=32, iterations_per_epoch=1000)
    post = amortizer.sample(configurator(generative_model(1)), n_samples=10000)

BayesFlow/TensorFlow are not installable here, so parity with BayesFlow's networks is UNPINNED; what is pinned is
the dictionary contract on both sides ('summary_conditions', 'direct_conditions', 'parameters').  The networks are plain
PyTorch modules; on the GPU the flow and the summary network's per-trial MLPs run as hand-written kernels
(csrc/train_kernels.hip, csrc/train_deepset.hip -- at batch 32 the PyTorch composition is ~600 launches of a few
microseconds per training iteration), with the PyTorch composition as the definition they are tested against.
"""
import math
import os
import pickle

import numpy as np
import torch
from torch import nn


class _PerSetLinearFn(torch.autograd.Function):
    """y = x W^T + b for x [B, N, in] with the weight gradient computed per set and then summed: as ONE GEMM it is a
    [in, B*N] x [B*N, out] product with a 64 x 64 result -- two workgroups on a 256-CU chip, 50 us at B*N = 9600 -- as a
    batched GEMM over the B sets it is 6 us (+ a 32-row sum).  14 such products are a fifth of a graph-replayed iteration."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x.reshape(-1, x.shape[-1]), w.t()).view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = (g.reshape(-1, g.shape[-1]) @ w).view_as(x) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(g.transpose(1, 2), x).sum(0)
        return gx, gw, g.sum((0, 1))


class _Linear(nn.Linear):
    def forward(self, x):
        if x.dim() == 3 and x.is_cuda and torch.is_grad_enabled():
            return _PerSetLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _mlp(d_in, d_hidden, d_out, n_hidden=2, act=nn.ReLU, keras_init=True):
    layers, d = [], d_in
    for _ in range(n_hidden):
        layers += [_Linear(d, d_hidden), act()]
        d = d_hidden
    layers.append(_Linear(d, d_out))
    # Keras-style initialisation (the reference's networks are BayesFlow/Keras Dense layers: Glorot weights, zero biases),
    # with He scaling on the hidden layers: PyTorch's Linear default shrinks the signal by ~0.58 per layer, and through the ~15
    # stacked layers of the summary network the data-dependent part of its output was 1e-5 of the bias-driven part at
    # initialisation -- the amortizer then learns the prior long before it starts looking at the data
    # (the coupling layers' internal networks keep PyTorch's small default: a flow should start near the identity)
    if keras_init:
        for m in layers[:-1]:
            if isinstance(m, nn.Linear):
                nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
                nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(layers[-1].weight)
        nn.init.zeros_(layers[-1].bias)
    return nn.Sequential(*layers)


class _Equivariant(nn.Module):
    """x[b,n,:] -> f(x[b,n,:], mean_n g(x[b,n,:])): permutation-equivariant DeepSet block."""

    def __init__(self, d_in, d_hidden):
        super().__init__()
        self.inv = _mlp(d_in, d_hidden, d_hidden)
        self.eq = _mlp(d_in + d_hidden, d_hidden, d_hidden)

    def forward(self, x, mask=None, inv_n=None):
        g = self.inv(x)
        pooled = g.mean(dim=1, keepdim=True) if mask is None else (g * mask).sum(dim=1, keepdim=True) * inv_n
        return self.eq(torch.cat([x, pooled.expand(-1, x.shape[1], -1)], dim=-1))


class InvariantNetwork(nn.Module):
    """bf.networks.InvariantNetwork(): exchangeable trials [B, N, D] -> learned summary [B, summary_dim]."""

    def __init__(self, input_dim=2, summary_dim=10, hidden=64, num_equiv=2):
        super().__init__()
        blocks, d = [], input_dim
        for _ in range(num_equiv):
            blocks.append(_Equivariant(d, hidden))
            d = hidden
        self.equiv = nn.Sequential(*blocks)
        self.pre_pool = _mlp(d, hidden, hidden)
        self.post_pool = _mlp(hidden, hidden, summary_dim)
        self.summary_dim = summary_dim
        self.fused = True           # the hand-written kernels where they apply (False: always the PyTorch composition)

    def _fused_lib(self, x):
        """libnddm_train.so if its per-trial MLP kernels cover this network and this tensor, else None (the PyTorch composition)."""
        if not (self.fused and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[1] >= 1):
            return None
        # the kernels are the TRAINING path: their forward always stores every hidden activation for the backward (2.5 KB per
        # trial row) and the backward one partial weight-gradient block per 64 rows.  Inference (no_grad: validation_loss,
        # sample) and batches beyond FUSED_MAX_ROWS trial rows take the PyTorch composition, which keeps nothing; and the
        # kernels produce no gradient of the data.
        if not torch.is_grad_enabled() or x.shape[0] * x.shape[1] > self.FUSED_MAX_ROWS or x.requires_grad:
            return None
        mlps = [m for blk in self.equiv for m in (blk.inv, blk.eq)] + [self.pre_pool, self.post_pool]
        if any(len(m) != 5 or not isinstance(m[1], nn.ReLU) or m[2].weight.shape != (64, 64) or m[4].weight.shape[1] != 64
               for m in mlps) or any(m[4].weight.shape[0] != 64 for m in mlps[:-1]) or not 1 <= self.summary_dim <= 64:
            return None
        from . import _train_lib
        L = _train_lib.lib()
        return L if L is not None and L.nddm_deepset_supported(64, x.shape[2]) else None

    def fused_params(self):
        """The parameter tensors in the order the kernels (and their flat weight-gradient buffer) take them: per block the
        invariant then the equivariant MLP, then the pre-pooling and the post-pooling MLP; W1, b1, W2, b2, W3, b3 of each."""
        ps = [t for blk in self.equiv for mlp in (blk.inv, blk.eq) for k in (0, 2, 4) for t in (mlp[k].weight, mlp[k].bias)]
        return ps + [t for mlp in (self.pre_pool, self.post_pool) for k in (0, 2, 4) for t in (mlp[k].weight, mlp[k].bias)]

    # a GraphTrainer may point this at the slice of ITS flat gradient buffer that holds fused_params() back to back: the backward's
    # last kernel then writes the gradients where the optimizer reads them (no gather copy); None: a buffer of the call's own
    grad_sink = None

    FUSED_MAX_ROWS = 1 << 19      # sets x trials (the training loop's 32 x 300 is 9600): 1.3 GB of saved activations at the cap

    def forward(self, x, mask=None, inv_n=None, n_valid=None, direct=None):
        """direct [B, k] (or broadcastable rows): "direct conditions" (log N, basic_ddm_dc.py:151-155) appended to the summary --
        the result is then the flow's condition [B, summary_dim + k] (the kernels write both into one buffer: no concatenation,
        and the backward takes the condition's gradient as it comes).
        mask [1, N, 1] (1 = a real trial, 0 = padding) and inv_n = 1 / (number of real trials), both device tensors -- or
        n_valid, the number of real trials as a device scalar (the first n_valid trials are the real ones): the pooled means
        then run over the real trials only, so a batch padded to a fixed length (one hipGraph per length bucket,
        GraphTrainer) gives what the unpadded batch gives."""
        L = self._fused_lib(x)
        if direct is not None and (L is None or direct.dim() != 2 or direct.shape[0] != x.shape[0] or direct.dtype != torch.float32
                                   or self.summary_dim + direct.shape[1] > 64 or direct.requires_grad):
            return torch.cat([self.forward(x, mask, inv_n, n_valid), direct.to(x.device, torch.float32)], dim=-1)
        if n_valid is not None and mask is None:
            n_valid = n_valid.reshape(-1).to(torch.float32)
            if L is not None:
                return _FusedDeepSetFn.apply((L, self.grad_sink), len(self.equiv), x, n_valid, None, direct, *self.fused_params())
            mask = (torch.arange(x.shape[1], device=x.device, dtype=torch.float32) < n_valid).to(torch.float32).view(1, -1, 1)
            inv_n = 1.0 / n_valid
        if L is not None:
            params = self.fused_params()
            m = None if mask is None else mask.reshape(-1).to(torch.float32)
            inv = None if inv_n is None else (inv_n if torch.is_tensor(inv_n) else torch.full((1,), float(inv_n), device=x.device))
            if inv is not None:
                inv = inv.reshape(-1).to(torch.float32)
            return _FusedDeepSetFn.apply((L, self.grad_sink), len(self.equiv), x, m, inv, direct, *params)
        for block in self.equiv:
            x = block(x, mask, inv_n)
        h = self.pre_pool(x)
        pooled = h.mean(dim=1) if mask is None else (h * mask).sum(dim=1) * inv_n
        return self.post_pool(pooled)


class _FusedDeepSetFn(torch.autograd.Function):
    """The summary network -- every equivariant block (an MLP whose masked per-set mean is the context of a second MLP), the
    pre-pooling MLP with its masked mean, and the MLP after the pooling -- as hand-written kernels (csrc/train_deepset.hip): one
    launch per 3-layer MLP each way plus one reduction of the weight gradients, instead of ~170 PyTorch launches of 4-7
    microseconds over [sets x trials, 64].  x [B, N, d] -> summary [B, summary_dim].  params: W1, b1, W2, b2, W3, b3 of the
    blocks' (invariant, equivariant) MLPs in order, then of the pre-pooling MLP, then of the post-pooling MLP.
    mask [N] and inv_n (device scalar), or mask = ONE float, the number of real trials, and inv_n None; or both None."""

    ROWS_PER_WG = 64     # one 64-row tile per workgroup: 160 workgroups at 32 sets x 300 trials (the chip has 256 CUs)

    @staticmethod
    def _common(x_t, d, B, N, S, rpw, mask, inv_n, ctx_part, S_ctx, prm, x_part=None, S_x=0):
        count = mask is not None and mask.numel() == 1 and inv_n is None       # the number of real trials instead of a 0/1 mask
        return (None if x_t is None else x_t.data_ptr(), d, B, N, S, rpw, None if mask is None else mask.data_ptr(), int(count),
                None if inv_n is None else inv_n.data_ptr(), 1.0 / N, None if ctx_part is None else ctx_part.data_ptr(), S_ctx,
                prm[0].data_ptr(), prm[0].shape[1], prm[1].data_ptr(), prm[2].data_ptr(), prm[3].data_ptr(), prm[4].data_ptr(),
                prm[5].data_ptr(), prm[4].shape[0], None if x_part is None else x_part.data_ptr(), S_x)

    @staticmethod
    def forward(ctx, L_sink, nb, x, mask, inv_n, direct, *params):
        L, ctx.sink = L_sink
        (B, N, d0), dev, rpw = x.shape, x.device, _FusedDeepSetFn.ROWS_PER_WG
        T, S, Hd = B * N, -(-N // rpw), 64
        Sp = -(-B // rpw)                           # the post-pooling MLP runs on one row per set: B rows in all
        x = x.contiguous()
        acts = torch.empty(((2 * nb + 1) * 2 * T + 2 * B, Hd), dtype=torch.float32, device=dev)   # h1, h2 of every MLP
        act = lambda k, j: acts[(2 * k + j) * T:(2 * k + j + 1) * T]
        act_post = lambda j: acts[(2 * nb + 1) * 2 * T + j * B:(2 * nb + 1) * 2 * T + (j + 1) * B]
        xs_next = torch.empty((max(nb, 1), T, Hd), dtype=torch.float32, device=dev)
        pools = torch.empty((nb + 1, B, S, Hd), dtype=torch.float32, device=dev)
        post = params[12 * nb + 6:]
        d_sum = post[4].shape[0]
        n_extra = 0 if direct is None else direct.shape[1]
        if direct is not None and n_extra > 1 and direct.stride(1) != 1:
            direct = direct.contiguous()
        summary = torch.empty((B, d_sum + n_extra), dtype=torch.float32, device=dev)      # (+ the direct conditions' columns)
        st = torch.cuda.current_stream(dev).cuda_stream
        cm = _FusedDeepSetFn._common
        count = mask is not None and mask.numel() == 1 and inv_n is None
        # every MLP that pools (the blocks' invariant halves, the pre-pooling MLP) consumes the rows the equivariant MLP before
        # it produces: the two run as ONE launch (the first invariant MLP, on the raw trials, alone)
        cur, d, rc = x, d0, 0
        pooling = [params[12 * i:12 * i + 6] for i in range(nb)] + [params[12 * nb:12 * nb + 6]]     # inv_0 .. inv_{nb-1}, pre
        rc |= L.nddm_deepset_mlp_fwd(*cm(cur, d, B, N, S, rpw, mask, inv_n, None, 0, pooling[0]), act(0, 0).data_ptr(),
                                     act(0, 1).data_ptr(), None, pools[0].data_ptr(), 0, None, 0, 0, st)
        for i in range(nb):
            eq, nxt = params[12 * i + 6:12 * i + 12], pooling[i + 1]
            rc |= L.nddm_deepset_mlp2_fwd(*cm(cur, d, B, N, S, rpw, mask, inv_n, pools[i], S, eq), act(2 * i + 1, 0).data_ptr(),
                                          act(2 * i + 1, 1).data_ptr(), xs_next[i].data_ptr(), *[t.data_ptr() for t in nxt],
                                          act(2 * i + 2, 0).data_ptr(), act(2 * i + 2, 1).data_ptr(), pools[i + 1].data_ptr(), st)
            cur, d = xs_next[i], Hd
        # (1 / N as the host's value must be THIS launch's N, not the B rows the post-pooling MLP is launched over)
        cp = list(cm(None, Hd, 1, B, Sp, rpw, mask if count else None, inv_n, None, 0, post, pools[nb], S))
        cp[9] = 1.0 / N
        rc |= L.nddm_deepset_mlp_fwd(*cp, act_post(0).data_ptr(), act_post(1).data_ptr(), summary.data_ptr(), None, d_sum + n_extra,
                                     None if direct is None else direct.data_ptr(), n_extra, 0 if direct is None else direct.stride(0), st)
        if rc != 0:
            raise RuntimeError(f"nddm_deepset_mlp_fwd failed ({rc})")
        ctx.L, ctx.nb, ctx.has_mask, ctx.has_inv, ctx.ld_out = L, nb, mask is not None, inv_n is not None, d_sum + n_extra
        ctx.save_for_backward(x, acts, xs_next, pools, *([mask] if mask is not None else []), *([inv_n] if inv_n is not None else []),
                              *params)
        return summary

    @staticmethod
    def backward(ctx, g_summary):
        L, nb, rpw = ctx.L, ctx.nb, _FusedDeepSetFn.ROWS_PER_WG
        x, acts, xs_next, pools, *rest = ctx.saved_tensors
        mask = rest.pop(0) if ctx.has_mask else None
        inv_n = rest.pop(0) if ctx.has_inv else None
        params = rest
        (B, N, d0), dev, Hd = x.shape, x.device, 64
        T, S, Sp = B * N, pools.shape[2], -(-B // rpw)
        G = B * S
        act = lambda k, j: acts[(2 * k + j) * T:(2 * k + j + 1) * T]
        act_post = lambda j: acts[(2 * nb + 1) * 2 * T + j * B:(2 * nb + 1) * 2 * T + (j + 1) * B]
        sizes = [p.numel() for p in params]
        per_mlp = [sum(sizes[6 * k:6 * k + 6]) for k in range(2 * nb + 2)]
        offs = [sum(per_mlp[:k]) for k in range(2 * nb + 2)]
        P = sum(per_mlp)
        part = torch.empty((G, P), dtype=torch.float32, device=dev)
        sink = ctx.sink
        flat = sink if (sink is not None and sink.numel() == P and sink.device == dev) else torch.empty(P, dtype=torch.float32, device=dev)
        gxbuf = torch.empty((max(nb, 1), T, Hd), dtype=torch.float32, device=dev)
        dctx = torch.empty((max(nb, 1), B, S, Hd), dtype=torch.float32, device=dev)
        g_pooled = torch.empty((B, Hd), dtype=torch.float32, device=dev)
        g_summary = g_summary.contiguous()
        st, F = torch.cuda.current_stream(dev).cuda_stream, 4
        cm = _FusedDeepSetFn._common
        count = mask is not None and mask.numel() == 1 and inv_n is None
        pp = part.data_ptr()

        def x_of(i):                                # input of block i (and of the pre-pooling MLP for i == nb)
            return (xs_next[i - 1], Hd) if i else (x, d0)

        post = params[12 * nb + 6:]
        cp = list(cm(None, Hd, 1, B, Sp, rpw, mask if count else None, inv_n, None, 0, post, pools[nb], S))
        cp[9] = 1.0 / N
        rc = L.nddm_deepset_mlp_bwd(*cp, act_post(0).data_ptr(), act_post(1).data_ptr(), g_summary.data_ptr(), ctx.ld_out, None, 0, None, 0,
                                    g_pooled.data_ptr(), 0, None, pp + offs[2 * nb + 1] * F, P, st)
        pre = params[12 * nb:12 * nb + 6]
        if nb == 0:
            rc |= L.nddm_deepset_mlp_bwd(*cm(x, d0, B, N, S, rpw, mask, inv_n, None, 0, pre), act(0, 0).data_ptr(), act(0, 1).data_ptr(),
                                         None, 0, g_pooled.data_ptr(), 0, None, 0, None, 0, None, pp + offs[0] * F, P, st)
        # The backward of a pooling MLP X (pre-pooling, or a block's invariant half) and of the equivariant MLP Y that produced X's
        # input are ONE launch each (csrc/train_deepset.hip: mlp2_bwd_kernel): pre + eq_{nb-1}, then inv_i + eq_{i-1}, then inv_0
        # alone.  X's input gradient -- plus eq_i's, which the launch before left in gxbuf -- is Y's output gradient.
        for j in reversed(range(nb)):                       # Y = eq_j;  X = pre (j == nb - 1) or inv_{j+1}
            eq = params[12 * j + 6:12 * j + 12]
            xin, d = x_of(j)
            if j == nb - 1:
                X, hx, ox = pre, 2 * nb, offs[2 * nb]
                gpool, gp_S, gp_W, gp_ldw, gx_prev = g_pooled.data_ptr(), 0, None, 0, None
            else:
                X, hx, ox = params[12 * (j + 1):12 * (j + 1) + 6], 2 * (j + 1), offs[2 * (j + 1)]
                eq_next = params[12 * (j + 1) + 6:12 * (j + 1) + 12]          # (its context columns carry the pooled gradient)
                gpool, gp_S, gp_W, gp_ldw, gx_prev = dctx[j + 1].data_ptr(), S, eq_next[0].data_ptr() + Hd * F, 2 * Hd, gxbuf[j].data_ptr()
            rc |= L.nddm_deepset_mlp2_bwd(*cm(xin, d, B, N, S, rpw, mask, inv_n, pools[j], S, eq), act(2 * j + 1, 0).data_ptr(),
                                          act(2 * j + 1, 1).data_ptr(), gxbuf[j - 1].data_ptr() if j else None, dctx[j].data_ptr(),
                                          pp + offs[2 * j + 1] * F, xs_next[j].data_ptr(), *[t.data_ptr() for t in X],
                                          act(hx, 0).data_ptr(), act(hx, 1).data_ptr(), gpool, gp_S, gp_W, gp_ldw, gx_prev, pp + ox * F, P, st)
        if nb:
            inv, eq = params[0:6], params[6:12]
            rc |= L.nddm_deepset_mlp_bwd(*cm(x, d0, B, N, S, rpw, mask, inv_n, None, 0, inv), act(0, 0).data_ptr(), act(0, 1).data_ptr(),
                                         None, 0, dctx[0].data_ptr(), S, eq[0].data_ptr() + d0 * F, d0 + Hd, None, 0, None, pp + offs[0] * F, P, st)
        rc |= L.nddm_deepset_reduce(pp, G, P, offs[2 * nb + 1], Sp, flat.data_ptr(), st)
        if rc != 0:
            raise RuntimeError(f"nddm_deepset_mlp_bwd failed ({rc})")
        grads, o = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[o:o + n].view(p.shape))
            o += n
        return (None, None, None, None, None, None, *grads)


class _FusedFlowFn(torch.autograd.Function):
    """The whole conditional flow -- per layer an ActNorm, a fixed permutation and two conditional affine-coupling half-layers
    (concatenate, three Linear layers with two ELUs, soft clamp, exp, multiply-add) -- as ONE hand-written kernel forward and ONE
    backward (csrc/train_kernels.hip) and one autograd node: at batch 32 the PyTorch composition is ~400 launches of a few
    microseconds each way, with a slice / concatenate / select bookkeeping kernel between any two.  (The backward is two
    kernels: the chain of activation gradients, and the weight gradients of all half-layers side by side.)
    -> (z [R, D], log|det| [R]); with nll=True -> the maximum-likelihood loss mean(|z|^2 / 2 - log|det|) itself (the backward
    then derives the gradients of z and log|det| inside the kernel: ~12 more PyTorch launches gone).
    params: per layer ActNorm log-scale and bias, then weight, bias x 3 of both sub-networks."""

    @staticmethod
    def forward(ctx, L_sink, clamp, d1, perms, nll, theta, cond, *params):
        import ctypes
        L, ctx.sink = L_sink
        nl, (R, D), C = len(perms), theta.shape, cond.shape[1]
        Hd, dev = params[4].shape[0], theta.device
        theta, cond = theta.contiguous(), cond.contiguous()
        saved = torch.empty(3 * nl * R * D + 4 * nl * R * Hd, dtype=torch.float32, device=dev)
        z_all, out_all, s_all = (saved[i * nl * R * D:(i + 1) * nl * R * D].view(nl, R, D) for i in range(3))
        h_all = saved[3 * nl * R * D:]
        ld = torch.empty(R + 1, dtype=torch.float32, device=dev)           # (+ the loss)
        loss = ld[R:]
        if nll and ctx.sink is not None and ctx.sink.get("loss") is not None and ctx.sink["loss"].device == dev:
            loss = ctx.sink["loss"]                                        # the trainer's slot: no copy of the loss afterwards
        ptrs = (ctypes.c_void_p * (14 * nl))(*[p.data_ptr() for p in params])
        perm = (ctypes.c_int * (nl * D))(*[int(v) for p in perms for v in p])
        rc = L.nddm_train_flow_fwd(nl, R, D, d1, C, float(clamp), ptrs, perm, theta.data_ptr(), cond.data_ptr(), z_all.data_ptr(),
                                   out_all.data_ptr(), s_all.data_ptr(), h_all.data_ptr(), ld.data_ptr(),
                                   loss.data_ptr() if nll else None, torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"nddm_train_flow_fwd failed ({rc})")
        ctx.L, ctx.clamp, ctx.d1, ctx.perm, ctx.ptrs, ctx.nl, ctx.nll = L, float(clamp), d1, perm, ptrs, nl, bool(nll)
        ctx.save_for_backward(theta, cond, saved, *params)
        return loss.view(()) if nll else (out_all[nl - 1], ld[:R])

    @staticmethod
    def backward(ctx, *gs):
        import ctypes
        theta, cond, saved, *params = ctx.saved_tensors
        L, nl, d1 = ctx.L, ctx.nl, ctx.d1
        (R, D), C, dev = theta.shape, cond.shape[1], theta.device
        n_rd = nl * R * D
        sizes = [p.numel() for p in params]
        Hd = params[4].shape[0]
        n_scratch = [nl * R * D, R * D, R * C, R, 2 * nl * R * (2 * Hd + 16)]
        slots = None if ctx.sink is None else [ctx.sink["grads"].get(p.data_ptr()) for p in params]
        in_place = slots is not None and all(s_ is not None and s_.shape == p.shape and s_.device == dev for s_, p in zip(slots, params))
        flat = torch.empty((0 if in_place else sum(sizes)) + sum(n_scratch), dtype=torch.float32, device=dev)
        grads, o = [], 0
        for k, (p, n) in enumerate(zip(params, sizes)):      # the trainer's slots (its flat gradient buffer), or a buffer of this call's own
            if in_place:
                grads.append(slots[k])
            else:
                grads.append(flat[o:o + n].view(p.shape))
                o += n
        scratch = []
        for n in n_scratch:
            scratch.append(flat[o:o + n])
            o += n
        gz_all, gx, gcond, w_gld, work = scratch
        if ctx.nll:
            g_nll, g_z, g_ld = gs[0].contiguous(), None, w_gld
        else:
            g_nll = None
            g_z = gs[0].contiguous() if gs[0] is not None else torch.zeros_like(theta)
            g_ld = gs[1].contiguous() if gs[1] is not None else torch.zeros(R, dtype=torch.float32, device=dev)
        gptr = (ctypes.c_void_p * (14 * nl))(*[g.data_ptr() for g in grads])
        sp, F = saved.data_ptr(), 4
        rc = L.nddm_train_flow_bwd(nl, R, D, d1, C, ctx.clamp, ctx.ptrs, ctx.perm, gptr, theta.data_ptr(), cond.data_ptr(),
                                   sp, sp + n_rd * F, sp + 2 * n_rd * F, sp + 3 * n_rd * F, None if g_z is None else g_z.data_ptr(),
                                   g_ld.data_ptr(), None if g_nll is None else g_nll.data_ptr(), gz_all.data_ptr(), gx.data_ptr(),
                                   gcond.data_ptr(), work.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"nddm_train_flow_bwd failed ({rc})")
        return (None, None, None, None, None, gx.view(R, D), gcond.view(R, C), *grads)


class _AffineCoupling(nn.Module):
    def __init__(self, dim, cond_dim, hidden, clamp=1.9):
        super().__init__()
        self.d1 = dim // 2
        self.d2 = dim - self.d1
        self.clamp = clamp
        self.net1 = _mlp(self.d1 + cond_dim, hidden, 2 * self.d2, act=nn.ELU, keras_init=False)
        self.net2 = _mlp(self.d2 + cond_dim, hidden, 2 * self.d1, act=nn.ELU, keras_init=False)

    def _st(self, net, h, cond):
        s, t = net(torch.cat([h, cond], dim=-1)).chunk(2, dim=-1)
        return self.clamp * torch.tanh(s / self.clamp), t


    def forward(self, x, cond):
        """-> (y, [s_a, s_b]): the log-scales are summed ONCE by the caller for all layers (one cat + one sum instead of a
        sum and an add per half layer, forward and backward).  (The PyTorch composition; the fused form is _FusedFlowFn.)"""
        x1, x2 = x[:, :self.d1], x[:, self.d1:]
        sa, t = self._st(self.net1, x1, cond)
        y2 = torch.addcmul(t, x2, torch.exp(sa))               # x2 * exp(s) + t, one kernel fewer each way
        sb, t = self._st(self.net2, y2, cond)
        y1 = torch.addcmul(t, x1, torch.exp(sb))
        return torch.cat([y1, y2], dim=-1), [sa, sb]

    def inverse(self, y, cond):
        y1, y2 = y[:, :self.d1], y[:, self.d1:]
        s, t = self._st(self.net2, y2, cond)
        x1 = (y1 - t) * torch.exp(-s)
        s, t = self._st(self.net1, x1, cond)
        x2 = (y2 - t) * torch.exp(-s)
        return torch.cat([x1, x2], dim=-1)


class InvertibleNetwork(nn.Module):
    """bf.networks.InvertibleNetwork(num_params): conditional normalising flow of affine coupling layers with fixed
    permutations and a learnable ActNorm in front of each."""

    def __init__(self, num_params, cond_dim=11, num_coupling_layers=6, hidden=128, seed=0):
        super().__init__()
        self.num_params = num_params
        self.layers = nn.ModuleList(_AffineCoupling(num_params, cond_dim, hidden) for _ in range(num_coupling_layers))
        g = torch.Generator().manual_seed(seed)
        # fixed permutations, applied as products with 0/1 matrices: a column gather costs an index kernel forward and an
        # index_put with a radix sort backward (five launches), the 5 x 5 product one each way
        for i in range(num_coupling_layers):
            perm = torch.randperm(num_params, generator=g)
            self.register_buffer(f"perm{i}", perm)
            # (derived from perm{i}: not part of the state dict, rebuilt after a load -- a mismatched pair cannot be loaded)
            self.register_buffer(f"pmat{i}", torch.eye(num_params)[:, perm].contiguous(), persistent=False)
        # one parameter per layer (not rows of one matrix: selecting a row costs autograd a zero-fill and a copy each way)
        self.an_scale = nn.ParameterList(nn.Parameter(torch.zeros(num_params)) for _ in range(num_coupling_layers))
        self.an_bias = nn.ParameterList(nn.Parameter(torch.zeros(num_params)) for _ in range(num_coupling_layers))
        self.fused = True           # the hand-written kernels where they apply (False: always the PyTorch composition)
        self._refresh_host_perms()  # (host copies: reading the buffers back would synchronise, which a graph capture forbids)
        self.register_load_state_dict_post_hook(self._refresh_host_perms)

    def _fused_lib(self, theta, cond):
        """libnddm_train.so if the fused flow covers this network and these tensors, else None (the PyTorch composition)."""
        if not (self.fused and len(self.layers) and theta.is_cuda and theta.dtype == torch.float32 and cond.dtype == torch.float32
                and theta.shape[0] <= self.FUSED_MAX_ROWS):
            return None
        l0 = self.layers[0]
        if any(len(n) != 5 or not isinstance(n[1], nn.ELU) for l in self.layers for n in (l.net1, l.net2)):
            return None
        from . import _train_lib
        L = _train_lib.lib()
        ok = L is not None and L.nddm_train_flow_supported(l0.net1[0].weight.shape[0], len(self.layers), self.num_params, l0.d1,
                                                           cond.shape[1])
        return L if ok else None

    # (the fused kernels take 8 / 32 rows per workgroup, and one workgroup per half-layer sums the weight gradients over all rows:
    # made for the training loop's batches of 32 ... a few hundred rows)
    FUSED_MAX_ROWS = 4096
    # a GraphTrainer may set {"grads": {parameter data_ptr: its slot in the trainer's flat gradient buffer}, "loss": the loss's slot}:
    # the backward then writes every gradient (and the forward the loss) where the optimizer reads them -- no gather copies
    grad_sink = None

    def _flow_params(self):
        params = []
        for i, l in enumerate(self.layers):
            params += [self.an_scale[i], self.an_bias[i]]
            params += [t for n in (l.net1, l.net2) for k in (0, 2, 4) for t in (n[k].weight, n[k].bias)]
        return params

    def nll(self, theta, cond):
        """mean(|z|^2 / 2 - log|det|): the maximum-likelihood loss of bf.amortizers.AmortizedPosterior."""
        L = self._fused_lib(theta, cond)
        if L is not None:
            return _FusedFlowFn.apply((L, self.grad_sink), self.layers[0].clamp, self.layers[0].d1, self._perm_host, True, theta, cond, *self._flow_params())
        z, log_det = self(theta, cond)
        return (0.5 * (z ** 2).sum(-1) - log_det).mean()

    def _refresh_host_perms(self, *_):
        self._perm_host = [getattr(self, f"perm{i}").tolist() for i in range(len(self.layers))]
        with torch.no_grad():
            for i, perm in enumerate(self._perm_host):
                pm = getattr(self, f"pmat{i}")
                pm.copy_(torch.eye(self.num_params)[:, perm])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Checkpoints of earlier layouts: ActNorm's log-scale and bias were ONE [layers, D] tensor each (now one parameter per
        layer, `an_scale.0` ...), and `pmat{i}` used to be stored (now derived from `perm{i}`)."""
        for name in ("an_scale", "an_bias"):
            old = state_dict.pop(prefix + name, None)
            if old is not None:
                for i in range(old.shape[0]):
                    state_dict.setdefault(f"{prefix}{name}.{i}", old[i].clone())
        for i in range(len(self.layers)):
            state_dict.pop(f"{prefix}pmat{i}", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, theta, cond):
        L = self._fused_lib(theta, cond)
        if L is not None:
            return _FusedFlowFn.apply((L, self.grad_sink), self.layers[0].clamp, self.layers[0].d1, self._perm_host, False, theta, cond, *self._flow_params())
        z, scales = theta, []
        for i, layer in enumerate(self.layers):
            z = torch.addcmul(self.an_bias[i], z, torch.exp(self.an_scale[i]))
            z = z @ getattr(self, f"pmat{i}")                 # == z[:, perm]
            z, s2 = layer(z, cond)
            scales += s2
        return z, torch.cat(scales, dim=-1).sum(-1) + torch.cat(list(self.an_scale)).sum()

    def inverse(self, z, cond):
        x = z
        for i in reversed(range(len(self.layers))):
            x = self.layers[i].inverse(x, cond)
            x = x @ getattr(self, f"pmat{i}").t()             # == x[:, argsort(perm)]
            x = (x - self.an_bias[i]) * torch.exp(-self.an_scale[i])
        return x


class AmortizedPosterior(nn.Module):
    """bf.amortizers.AmortizedPosterior(inference_net, summary_net): maximum-likelihood training of q(theta | data)."""

    def __init__(self, inference_net, summary_net):
        super().__init__()
        self.inference_net, self.summary_net = inference_net, summary_net

    def _t(self, v):
        dev = next(self.parameters()).device
        return v.to(dev, torch.float32) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev)

    def _conditions(self, input_dict):
        pad = input_dict.get("summary_mask", None)        # (mask, inv_n) of a batch padded to a bucket length (additive keys), or
        nv = input_dict.get("summary_n", None)            # the number of real trials, a device scalar
        x = self._t(input_dict["summary_conditions"])
        direct = input_dict.get("direct_conditions", None)
        direct = None if direct is None else self._t(direct)
        if direct is not None and isinstance(self.summary_net, InvariantNetwork):
            # (the summary network appends the direct conditions itself: with the kernels, without a concatenation)
            if nv is not None:
                return self.summary_net(x, n_valid=nv, direct=direct)
            return self.summary_net(x, direct=direct) if pad is None else self.summary_net(x, pad[0], pad[1], direct=direct)
        if nv is not None:
            summ = self.summary_net(x, n_valid=nv)
        else:
            summ = self.summary_net(x) if pad is None else self.summary_net(x, pad[0], pad[1])
        return summ if direct is None else torch.cat([summ, direct], dim=-1)

    def forward(self, input_dict):
        return self.inference_net(self._t(input_dict["parameters"]), self._conditions(input_dict))

    def compute_loss(self, input_dict):
        theta, cond = self._t(input_dict["parameters"]), self._conditions(input_dict)
        if hasattr(self.inference_net, "nll"):
            return self.inference_net.nll(theta, cond)
        z, log_det = self.inference_net(theta, cond)
        return (0.5 * (z ** 2).sum(-1) - log_det).mean()

    @torch.no_grad()
    def sample(self, input_dict, n_samples, to_numpy=True, reject_outside=None, max_redraws=16):
        """amortizer.sample(dict, n_samples) (basic_ddm_dc.py:223): [n_samples, P] for a single data set,
        [B, n_samples, P] for a batch.

        reject_outside: None (the default: every draw of the flow is returned, which is what the reference's call gets) or a
        (low, high) pair of length-P sequences -- draws with a component outside [low, high] (or non-finite) are REDRAWN from fresh
        base normals, i.e. the sample is from q(theta | data) restricted to the box.  Why one might want it: a sharply trained
        coupling flow carries ~1e-6 of its mass in a far tail (one draw in 1e5-1e6 lands 1e4-1e6 prior widths away, DESIGN.md
        section 8), and a posterior MEAN over 10 000 draws -- the statistic the reference's recovery plots use, basic_ddm_dc.py:
        230-236 -- is carried off by one such draw; `priors.prior_box(model)` is a generous box (the prior's support widened by
        its own width on each side) that removes them and nothing else.  The number of redrawn draws of the last call is kept in
        `self.last_redrawn` (0 with the option off)."""
        cond = self._conditions(input_dict)
        B = cond.shape[0]
        P = self.inference_net.num_params
        z = torch.randn(B * n_samples, P, device=cond.device)
        cond_rep = cond.repeat_interleave(n_samples, dim=0)
        out = self.inference_net.inverse(z, cond_rep)
        self.last_redrawn = 0
        if reject_outside is not None:
            lo, hi = (torch.as_tensor(np.asarray(v, dtype=np.float32), device=out.device) for v in reject_outside)
            if lo.shape != (P,) or hi.shape != (P,):
                raise ValueError(f"reject_outside must be a (low, high) pair of length-{P} sequences")
            for _ in range(int(max_redraws)):
                bad = ((out < lo) | (out > hi) | ~torch.isfinite(out)).any(dim=-1)
                idx = bad.nonzero(as_tuple=False)[:, 0]
                if idx.numel() == 0:
                    break
                self.last_redrawn += int(idx.numel())
                out[idx] = self.inference_net.inverse(torch.randn(idx.numel(), P, device=out.device), cond_rep[idx])
            out = torch.minimum(torch.maximum(torch.nan_to_num(out, nan=0.0), lo), hi)      # (what max_redraws rounds did not cure: clipped)
        out = out.reshape(B, n_samples, -1)
        if B == 1:
            out = out[0]
        return out.cpu().numpy() if to_numpy else out


class Trainer:
    """bf.trainers.Trainer(amortizer, generative_model, configurator, checkpoint_path): online and experience-replay
    training (basic_ddm_dc.py:172-176, 199-202), `load_pretrained_network` (:207).  Checkpoints also carry the
    simulator's (seed, offset) stream state, so a resumed run continues the random stream."""

    def __init__(self, amortizer, generative_model, configurator=None, checkpoint_path=None, learning_rate=5e-4,
                 device=None, stream_state=None, graph=False):
        """graph=True: train_online / train_experience_replay keep the reference's call shape (basic_ddm_dc.py:199-202) and
        run as hipGraph replays (graph_trainer.GraphTrainer: device prior -> simulate -> forward/backward -> clip -> Adam, one
        batch ahead) when the generative model says how it is made (`graph_spec`, set by the model modules'
        make_generative_model): ~25 x the eager loop at batch 32.  The batches then come from the library's on-device prior
        and simulator keyed by `graph_spec['seed']` -- the same distributions, another random stream than the generative
        model's own -- and a ValueError says so if the model cannot be run that way."""
        self.graph = bool(graph)
        self.amortizer, self.generative_model = amortizer, generative_model
        self.configurator = configurator or (lambda d: d)
        self.checkpoint_path = checkpoint_path
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.amortizer.to(self.device)
        self.lr = learning_rate
        self.optimizer = torch.optim.Adam(self.amortizer.parameters(), lr=learning_rate)
        self.scheduler = None
        self._optimizer_spent = False      # set when a train_* call ends with reuse_optimizer=False (BayesFlow's default)
        self.stream_state = stream_state
        self.loss_history = []
        self.replay = []
        self._graph_pos = None             # graph=True: where the last train_* call left the random stream and the replay buffer
        self._graph_opt = None             # graph=True: the running call's optimizer / schedule state at its last epoch checkpoint
        self._graph_resume = None          # ... as load_pretrained_network found it in ckpt.pt: an interrupted call is continued

    def _simulate(self, batch_size):
        return self.configurator(self.generative_model(batch_size))

    def _step(self, conf):
        loss = self.amortizer.compute_loss(conf)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.amortizer.parameters(), 5.0)
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return float(loss.detach())

    def _setup_schedule(self, total_steps, optimizer=None, scheduler=None):
        """`optimizer` given (BayesFlow's train_*(optimizer=...)): the caller's optimizer -- and learning-rate scheduler, if
        any -- is used as it is; nothing is reset (one schedule can then span several calls).  Otherwise:
        every train_* call is a run of its own, as in BayesFlow 1.1's Trainer: a cosine decay from the Trainer's learning
        rate over THIS call's epochs x iterations (held at its final value beyond, like Keras' CosineDecay) and -- unless the
        previous call was made with reuse_optimizer=True, or a checkpoint has just been loaded -- a fresh Adam.  (Stacking a
        second CosineAnnealingLR on an optimizer whose rate the first one had annealed to 0 trained the second call, and every
        resumed run, at rate 0.)"""
        if optimizer is not None:
            self.optimizer, self.scheduler, self._optimizer_spent = optimizer, scheduler, False
            return
        if self._optimizer_spent:
            self.optimizer = torch.optim.Adam(self.amortizer.parameters(), lr=self.lr)
            self._optimizer_spent = False
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr
            g["initial_lr"] = self.lr
        total = max(1, int(total_steps))
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lambda step: 0.5 * (1.0 + math.cos(math.pi * min(step, total) / total)))

    def _prefetcher(self, batch_size, total):
        """Yields `total` configured batches, simulating batch i+1 on a side stream BEFORE batch i is trained on: the
        simulator launch (and a host-side prior) then overlaps the training step, whose loss read-back is the only
        synchronisation point of an iteration.  Batches come in the same order from the same random stream as without
        prefetching; nothing is simulated beyond `total`."""
        if self.device.type != "cuda" or total <= 0:
            for _ in range(total):
                yield self._simulate(batch_size)
            return
        side = torch.cuda.Stream(device=self.device)

        def launch():
            with torch.cuda.stream(side):
                conf = self._simulate(batch_size)
                ev = torch.cuda.Event()
                ev.record(side)
            return conf, ev

        nxt = launch()
        for i in range(total):
            conf, ev = nxt
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            for v in conf.values():          # the tensors were allocated on the side stream's pool
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            if i + 1 < total:
                nxt = launch()
            yield conf

    def _graph_loop(self, replay, epochs, iterations_per_epoch, batch_size, capacity_in_batches, validation_sims, save_checkpoint,
                    optimizer):
        """The call as graph replays: one GraphTrainer for THIS call (its cosine schedule spans epochs x iterations, a fresh Adam:
        _setup_schedule's semantics), an epoch = one train_* call of it, validation loss and checkpoint between epochs.

        The run's POSITION outlives the call (`self._graph_pos`, also in ckpt.pt): the next call -- and a run resumed through
        load_pretrained_network -- continues the random stream (next parameter sets, next batch-shared N) and the experience-replay
        buffer where this one stopped, as the eager loop does through `stream_state` and `self.replay`.  Training's parameter
        sets start at graph_trainer.TRAIN_OFFSET_BASE of the seed's index space; the generative model's own draws (validation
        sims, recovery data sets) count up from 0, so the two never meet."""
        spec = getattr(self.generative_model, "graph_spec", None)
        if spec is None or self.device.type != "cuda" or optimizer is not None:
            raise ValueError("Trainer(graph=True) needs a generative model made by a model module's make_generative_model "
                             "(it carries `graph_spec`), a ROCm device, and no caller-supplied optimizer")
        from .graph_trainer import TRAIN_OFFSET_BASE, GraphTrainer
        val = []
        total = epochs * iterations_per_epoch
        resume, self._graph_resume = self._graph_resume, None
        with GraphTrainer(self.amortizer, batch_size=batch_size, total_steps=total, learning_rate=self.lr,
                          device=self.device, offset_base=TRAIN_OFFSET_BASE, **spec) as gt:
            done_epochs = 0
            if (resume is not None and int(resume["total_steps"]) == total and int(resume["iterations_per_epoch"]) == iterations_per_epoch
                    and int(resume["batch_size"]) == batch_size and bool(resume["replay"]) == bool(replay) and 0 < int(resume["iteration"]) < total):
                # The checkpoint was written BETWEEN TWO EPOCHS of a call of this very shape: that call is continued -- Adam's moments
                # and step count, the place on the cosine schedule, the losses so far -- and only its remaining epochs run, so an
                # interrupted run + its resumption equal the uninterrupted run.  (Any other call after a load is a run of its own,
                # as every train_* call is: fresh Adam, its own schedule.)
                gt.load_optimizer_state(resume)
                done_epochs = int(resume["iteration"]) // iterations_per_epoch
                self.loss_history = self.loss_history[:max(0, len(self.loss_history) - int(resume["iteration"]))]      # (gt brings them back)
            if self._graph_pos is not None:
                gt.set_position(self._graph_pos)
            for _ in range(done_epochs, epochs):
                if replay:
                    gt.train_experience_replay(iterations_per_epoch, capacity_in_batches=capacity_in_batches)
                else:
                    gt.train_online(iterations_per_epoch)
                if validation_sims is not None:
                    val.append(gt.validation_loss(self.configurator(validation_sims)))
                if save_checkpoint and self.checkpoint_path:
                    self._graph_pos = gt.position()
                    self._graph_opt = dict(gt.optimizer_state(), iterations_per_epoch=int(iterations_per_epoch), batch_size=int(batch_size),
                                           replay=bool(replay))
                    self.loss_history_graph = gt.loss_history()
                    self.save_checkpoint(extra_losses=self.loss_history_graph)
            self._graph_pos = gt.position()
            self._graph_opt = None             # the call is complete: nothing to continue
            self.loss_history += gt.loss_history()
        if save_checkpoint and self.checkpoint_path:
            self.save_checkpoint()             # ... and the checkpoint says so (a later load starts a run of its own)
        self._optimizer_spent = True
        return val

    def train_online(self, epochs, iterations_per_epoch, batch_size, save_checkpoint=True, prefetch=True, reuse_optimizer=False,
                     optimizer=None, scheduler=None, **_):
        if self.graph:
            self._graph_loop(False, epochs, iterations_per_epoch, batch_size, 0, None, save_checkpoint, optimizer)
            return self.loss_history
        self._setup_schedule(epochs * iterations_per_epoch, optimizer, scheduler)
        for ep in range(epochs):
            batches = (self._prefetcher(batch_size, iterations_per_epoch) if prefetch
                       else (self._simulate(batch_size) for _ in range(iterations_per_epoch)))
            for conf in batches:
                self.loss_history.append(self._step(conf))
            if save_checkpoint:
                self.save_checkpoint()
        self._optimizer_spent = not (reuse_optimizer or optimizer is not None)
        return self.loss_history

    def train_experience_replay(self, epochs, iterations_per_epoch, batch_size, capacity_in_batches=100,
                                save_checkpoint=True, validation_sims=None, prefetch=True, reuse_optimizer=False,
                                optimizer=None, scheduler=None, **_):
        """Each iteration simulates one fresh batch into a ring buffer of `capacity_in_batches` batches and trains on
        a randomly chosen stored batch (BayesFlow's experience replay, used at basic_ddm_dc.py:199-202).  Batches keep
        their own N (the non-batchable context), as in BayesFlow's buffer."""
        if self.graph:
            val = self._graph_loop(True, epochs, iterations_per_epoch, batch_size, capacity_in_batches, validation_sims, save_checkpoint,
                                   optimizer)
            return {"train_losses": self.loss_history, "val_losses": val}
        self._setup_schedule(epochs * iterations_per_epoch, optimizer, scheduler)
        rng = np.random.default_rng(0)
        val = []
        for ep in range(epochs):
            batches = (self._prefetcher(batch_size, iterations_per_epoch) if prefetch
                       else (self._simulate(batch_size) for _ in range(iterations_per_epoch)))
            for conf in batches:
                if len(self.replay) < capacity_in_batches:
                    self.replay.append(conf)
                else:
                    self.replay[rng.integers(capacity_in_batches)] = conf
                self.loss_history.append(self._step(self.replay[rng.integers(len(self.replay))]))
            if validation_sims is not None:
                with torch.no_grad():
                    val.append(float(self.amortizer.compute_loss(self.configurator(validation_sims))))
            if save_checkpoint:
                self.save_checkpoint()
        self._optimizer_spent = not (reuse_optimizer or optimizer is not None)
        return {"train_losses": self.loss_history, "val_losses": val}

    # ---- checkpoint / resume -----------------------------------------------------------------------
    def save_checkpoint(self, extra_losses=()):
        if not self.checkpoint_path:
            return
        os.makedirs(self.checkpoint_path, exist_ok=True)
        state = {"model": self.amortizer.state_dict(), "optimizer": self.optimizer.state_dict(),
                 "loss_history": self.loss_history + list(extra_losses)}
        if self.stream_state is not None:
            state["stream_state"] = self.stream_state.get_state()
        if self._graph_pos is not None:
            state["graph_position"] = self._graph_pos
        if self._graph_opt is not None:        # graph=True, written between two epochs of a call: what continues that call
            state["graph_optimizer"] = self._graph_opt
        from .graph_trainer import plain_state
        torch.save(plain_state(state), os.path.join(self.checkpoint_path, "ckpt.pt"))      # plain data only: loads with weights_only=True
        with open(os.path.join(self.checkpoint_path, "history.pkl"), "wb") as f:
            pickle.dump({"loss_history": state["loss_history"]}, f)

    def load_pretrained_network(self):
        path = os.path.join(self.checkpoint_path or "", "ckpt.pt")
        if not os.path.exists(path):
            return False
        state = torch.load(path, map_location=self.device, weights_only=True)        # (plain_state(): tensors, numbers, strings -- no pickle code)
        self.amortizer.load_state_dict(state["model"])
        self.optimizer.load_state_dict(state["optimizer"])
        self._optimizer_spent = False      # the loaded moments serve the next train_* call (which sets its own schedule)
        self.loss_history = list(state.get("loss_history", []))
        if self.stream_state is not None and "stream_state" in state:
            self.stream_state.set_state(state["stream_state"])
        self._graph_pos = state.get("graph_position", self._graph_pos)      # graph=True: the next call continues the stream
        self._graph_resume = state.get("graph_optimizer", None)             # ... and, if the file was written mid-call, that call
        return True


def posterior_estimates(amortizer, generative_model, configurator, n_datasets=100, n_samples=1000):
    """n_datasets fresh data sets, one at a time as in the reference's loop (basic_ddm_dc.py:218-223): (true parameters [n, P],
    posterior means [n, P], posterior medians [n, P]) from n_samples posterior draws each."""
    true, means, meds = [], [], []
    for _ in range(n_datasets):
        conf = configurator(generative_model(1))
        post = amortizer.sample(conf, n_samples)
        p = conf["parameters"]
        true.append((p.cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p))[0])
        means.append(post.mean(axis=0))
        meds.append(np.median(post, axis=0))
    return np.array(true), np.array(means), np.array(meds)


def posterior_recovery(amortizer, generative_model, configurator, n_datasets=100, n_samples=1000, statistic="mean"):
    """The recovery loop of basic_ddm_dc.py:218-223 in miniature: posterior means vs true parameters -> per-parameter
    Pearson correlation (the reference plots R^2 / rho, pyhddmjagsutils.py:609-623).  statistic="median": the posterior median
    instead.  A sharply trained flow carries ~5e-7 of its mass at |theta| up to 1e6 (regions maximum-likelihood training never
    visits: once the inverse's intermediate vector leaves the trained range, the conditioners' log-scales flip sign and every
    remaining half-layer multiplies by e^1.9; the inverse is exact there -- DESIGN.md section 8, tools/locate_tail_draws.py), and ONE
    such draw among a data set's ten thousand moves its mean (and one such data set among hundreds the correlation) while the
    posterior's bulk sits on the truth.  The mean is the reference's statistic and the default here."""
    if statistic not in ("mean", "median"):
        raise ValueError("statistic must be 'mean' or 'median'")
    true, means, meds = posterior_estimates(amortizer, generative_model, configurator, n_datasets, n_samples)
    est = means if statistic == "mean" else meds
    return np.array([np.corrcoef(true[:, j], est[:, j])[0, 1] for j in range(true.shape[1])])
