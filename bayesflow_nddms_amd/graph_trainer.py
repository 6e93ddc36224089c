"""BASELINE config 5 GPU-bound: the online training loop of the reference (basic_ddm_dc.py:163-176, 199-202 -- a
generative model feeding `trainer.train_*(batch_size=32)`) with every iteration ONE hipGraph replay:

    device prior -> simulate (HIP kernels, C ABI) -> configurator -> DeepSet + coupling flow forward -> backward
    -> gradient clipping -> Adam (+ cosine learning-rate schedule, + the loss stored into a device buffer)

(as graphs on two or three streams: the simulation of batch i + 1 -- and, with more than one rank, the all-gather that
reassembles it -- runs beside the training step on batch i)

The eager loop (amortizer.Trainer) spends ~10 ms of host time per iteration launching ~1000 small kernels and reads the
loss back every step; the MI355X is busy for a few per cent of it.  Here the host's share of an iteration is: draw the
batch-shared N (a pure function of (seed, iteration), distributed.shared_prior_N), write it to the device, replay a graph.

What makes the iteration capturable:
  * the random stream moves WITHOUT new kernel arguments: the simulator and the prior sampler add a 64-bit offset they
    read from device memory when they run (include/nddm.h: nddm_simulate_indirect / nddm_draw_prior_indirect), and the
    graph itself advances that word;
  * N varies per batch (basic_ddm_dc.py:50-52, 131): it is BUCKETED -- 8 buckets over 60..300 (16 were 2.5 % slower: a graph that
    has not been replayed for a while starts slower, and the padding costs the summary network's kernels nothing -- their time
    is a workgroup's latency, not the number of workgroups), one graph per bucket, captured lazily -- the batch is simulated with the bucket's top number of trials and the summary network pools over
    the first N only (mask + 1/N from a device scalar), which equals pooling the unpadded batch: trial i of a set is the
    same function of (seed, set, i) whatever the launch's n_trials;
  * the loss is written to a device buffer indexed by the device-side step counter and read back when training ends (or on
    request), not every step; the learning rate is a device tensor computed in the graph from the same counter.

More than one rank (`world` > 1, one process per GPU): `parallel='gather'` keeps north_star's shape -- every rank simulates
its shard, ONE all-gather reassembles the minibatch, every rank runs the same training step (replicated); `parallel='ddp'`
shards the training too -- every rank trains on its own shard and the flat gradient buffer is all-reduced.  The collective
sits BETWEEN two graphs (simulate | all-gather | forward/backward + clip + Adam; simulate | forward/backward | all-reduce |
clip + Adam): RCCL calls are not captured.  The feed is PIPELINED at every world size: simulate (i + 1) and its all-gather go
to the simulate and communication streams while the training stream works on batch i; only the gradient all-reduce of `ddp`
stays on the critical path, between its two graphs.  Replicas are made identical at construction and after a checkpoint load
(weights, buffers, Adam's moments and the counters are broadcast from rank 0).

PyTorch is plumbing here (autograd nodes, the caching allocator, graphs): the simulator is the library's, and so -- where
libnddm_train.so builds -- are the networks' forward / backward (amortizer.py) and the optimizer step on flat buffers
(csrc/train_update.hip): 21 kernels and 0.31 ms per iteration of batch 32 on one MI355X.  Parity with BayesFlow's networks
is unpinned as for amortizer.py.
"""
import math
import os as _os

import torch

from . import engine
from .distributed import shared_prior_N

# model -> (kernel, columns of the parameter rows the kernels take, leading columns that are the network's targets):
# basic_ddm_dc.py:62-80 draws 5 parameters; single_trial_alpha_not_scaled.py:78-102 draws 7, the kernel takes an eighth (gamma = 1)
_PRIOR_MODEL = {"basic": (engine.BASIC_DDM_DC, 5, 5), "single": (engine.SINGLE_TRIAL, 8, 7)}

def plain_state(obj):
    """A checkpoint's content reduced to what torch.load(weights_only=True) accepts -- tensors, Python numbers, strings, dicts, lists,
    tuples -- so that loading a checkpoint never runs arbitrary pickle code: NumPy scalars and arrays (a generator's state, a loss
    history) become Python numbers / lists, everything else must already be plain."""
    import numpy as np
    if isinstance(obj, dict):
        return {str(k) if not isinstance(k, (str, int)) else k: plain_state(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(plain_state(v) for v in obj)
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if obj is None or isinstance(obj, (bool, int, float, str, torch.Tensor)):
        return obj
    if hasattr(obj, "items"):                              # (OrderedDict of a state_dict)
        return {k: plain_state(v) for k, v in obj.items()}
    raise TypeError(f"checkpoint content of type {type(obj).__name__} is not plain data")


# Where TRAINING runs that share a seed with a generative model start in the 60-bit space of global set indices (csrc/nddm_sim.h:
# low word + 28 high bits): the model modules' own DevicePrior counts up from 0 -- `generative_model(B)` for validation_sims and the
# recovery loop's fresh data sets (basic_ddm_dc.py:186-188, 218-223) -- so evaluation never sees a parameter row training has seen.
TRAIN_OFFSET_BASE = 1 << 59


# Cross-stream dependencies of the pipelined loop: plain events.  (Timing events -- a barrier packet with a completion signal of its
# own at the point of the record -- were tried when the simulate graph of batch i + 1 was seen starting 300-400 us into the training
# graph of batch i: no difference, A/B on one box.  What decided that start was the ORDER in which the host enqueues the two
# graphs: see _train_overlapped.)
_TIMED_EVENTS = False


class _Bucket:
    __slots__ = ("n_top", "shard", "params", "trials", "g_shard", "g_params", "g_trials", "t_params", "t_trials", "staged",
                 "r_params", "r_trials", "graphs", "n2", "free_ev")


class GraphTrainer:
    def __init__(self, amortizer, batch_size=32, total_steps=1000, n_min=60, n_max=300, n_buckets=8, dt=0.01,
                 max_steps=400.0, seed=2023, learning_rate=5e-4, clip=5.0, device=None, use_graph=True,
                 world=1, rank=0, parallel="gather", backend="nccl", split=None, model="basic", overlap=True, offset_base=0, n_base=0):
        """total_steps: length of the cosine schedule (past it the rate holds the schedule's final value) and size of the
        device-side loss ring (read out by the host before it wraps: the history is complete for any number of iterations).
        use_graph=False runs the SAME iteration eagerly (the comparator of the parity test).  split: force the two-graph
        form (the one used with a collective in the middle) at world 1.  overlap (graphs, any world size): batch i + 1 is
        simulated -- and all-gathered -- on streams of its own while batch i is trained on: same batches, same order, same
        result as the sequential loop (overlap=False).  offset_base / n_base: where this run starts in the random stream -- the
        global index of its first parameter set, and the key of its first batch-shared N (set_position() moves both later)."""
        if parallel not in ("gather", "ddp"):
            raise ValueError("parallel must be 'gather' or 'ddp'")
        torch_ = engine.require_device()
        assert torch_ is torch
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.amortizer = amortizer.to(self.dev)
        self.model_id, self.P, self.P_net = _PRIOR_MODEL[model]
        self.B, self.T = int(batch_size), int(total_steps)
        self.n_min, self.n_max, self.n_buckets = int(n_min), int(n_max), int(n_buckets)
        self.width = -(-(self.n_max - self.n_min + 1) // self.n_buckets)
        self.dt, self.max_steps, self.seed = float(dt), float(max_steps), int(seed)
        self.lr0, self.clip = float(learning_rate), float(clip)
        self.use_graph = bool(use_graph)
        self.world, self.rank, self.parallel, self.backend = int(world), int(rank), parallel, backend
        self.split = (self.world > 1) if split is None else bool(split)
        self.overlap = bool(overlap) and self.use_graph
        self.iteration = 0                 # iterations of THIS run: indexes the loss ring and the schedule
        self.n_base = int(n_base)          # key of this run's first batch-shared N: batch k of the run draws N from (seed, n_base + k)
        self._offset_base = int(offset_base)
        self._warned_past_end = False
        self._loss_host = []               # losses already read out of the device ring (_drain_losses)
        self._uses = {}                    # pipelined loop, direct form: how often each bucket was produced into (-> its buffer set)
        with torch.cuda.device(self.dev):
            # ONE flat buffer each for the parameters (the modules' tensors become views of it), the gradients (+ 1 slot that
            # carries the loss through the all-reduce) and Adam's two moments: the optimizer step is then three launches of the
            # library's (csrc/train_update.hip) instead of ~20 of PyTorch's.  Every tensor starts on a 16-byte boundary (the
            # training kernels read weights with 16-byte loads): sizes are padded to multiples of four floats, the pads stay zero.
            # order: the kernels' own (the flow's per-layer tensors, then the summary network's MLPs: amortizer.py), so that the
            # backward kernels can write every gradient straight into its slot of the flat buffer (the sinks set below)
            first = []
            for net, getter in ((getattr(self.amortizer, "inference_net", None), "_flow_params"),
                                (getattr(self.amortizer, "summary_net", None), "fused_params")):
                if net is not None and hasattr(net, getter):
                    first += [p for p in getattr(net, getter)() if p.requires_grad]
            seen = {id(p) for p in first}
            self.params = first + [p for p in self.amortizer.parameters() if p.requires_grad and id(p) not in seen]
            self._offs, o = [], 0
            for p in self.params:
                self._offs.append(o)
                o += -(-p.numel() // 4) * 4
            self.n_el = n_el = o
            self.flat_p = torch.zeros(n_el, dtype=torch.float32, device=self.dev)
            self.flat = torch.zeros(n_el + 1, dtype=torch.float32, device=self.dev)
            self._pads = []
            with torch.no_grad():
                for p, o in zip(self.params, self._offs):
                    view = self.flat_p[o:o + p.numel()].view_as(p)
                    view.copy_(p.to(self.dev))
                    p.data = view
                    p.grad = None
                    pad = -p.numel() % 4
                    self._pads.append(torch.zeros(pad, dtype=torch.float32, device=self.dev) if pad else None)
            self._slots = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self._offs)]
            self._loss_slot = self.flat[n_el:]
            self._sinks = self._make_sinks()
            self.lr_t = torch.tensor(self.lr0, dtype=torch.float32, device=self.dev)
            self._one = torch.ones((), dtype=torch.float32, device=self.dev)
            from . import _train_lib
            self._lib = _train_lib.lib()
            if self._lib is not None:
                self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
                self._partial = torch.zeros(256 + 4, dtype=torch.float32, device=self.dev)      # 256 partial sums + the update kernel's ticket word
                self.optimizer = None
            else:
                # without the library: PyTorch's fused multi-tensor Adam on the same flat layout (one tensor)
                self._flat_param = torch.nn.Parameter(self.flat_p)
                self._flat_param.grad = self.flat[:n_el]
                self.optimizer = torch.optim.Adam([self._flat_param], lr=self.lr_t, capturable=True, fused=True)
            # device-side counters and scalars
            self.offset = torch.tensor([self._offset_base + self.rank * self.B], dtype=torch.int64, device=self.dev)   # this rank's row 0 of the next batch
            self.step_i = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.step_f = torch.zeros(1, dtype=torch.float32, device=self.dev)
            self.loss_buf = torch.zeros(max(1, self.T), dtype=torch.float32, device=self.dev)
            if self.optimizer is not None:
                # Adam's state exists after a step: one with zero gradients changes no weight (update = lr * 0 / (0 + eps))
                self.optimizer.step()
                for st in self.optimizer.state.values():
                    st["step"].zero_()
        self._buckets = {}
        self._replay = None
        # everything the trainer enqueues -- warm-up passes, captures, replays -- goes to ONE stream of its own: autograd's
        # gradient-accumulation nodes remember the stream they were first used on, and capture needs a non-default one
        self._stream = torch.cuda.Stream(device=self.dev)
        # The pipelined loop needs its streams on DIFFERENT hardware queues (HIP multiplexes all streams of a priority onto four;
        # which one a stream gets is not ours to choose): a communication stream that shared the training stream's queue ran the
        # all-gather of batch i + 1 behind the training graph of batch i instead of beside it (+ 45 us per iteration,
        # HISTORY.md §C.6, "Config 5, round 4").  So the side streams are CHOSEN by a probe: independent of the streams before them.
        # (A high-priority stream has a queue of its own by construction, but its mere existence slowed every kernel that ran
        # beside the simulator by 2-7 x on this chip: measured, not used.)
        self.independent_queues = []       # one entry per probed side stream: did the probe find a queue of its own?
        self._sim_stream = self._independent_stream([self._stream]) if self.overlap else None
        self._comm = self._independent_stream([self._stream] + ([self._sim_stream] if self.overlap else [])) \
            if (self.overlap and self.world > 1) or self.split else torch.cuda.Stream(device=self.dev)
        self._pool = torch.cuda.graph_pool_handle() if self.use_graph else None
        # every graph is CAPTURED on this stream and replayed on the stream of its stretch (training / simulate): the streams that
        # carry collectives never capture, so the process group's watchdog never queries an event of a capturing stream -- and the
        # gradient all-reduce of `ddp` can sit on the training stream itself, between its two graphs, without a cross-queue hop
        self._cap = torch.cuda.Stream(device=self.dev) if self.use_graph else None
        # the library's memory behind THIS trainer's captured launches (include/nddm.h: nddm_graph_arena_*): close() frees
        # it and nothing else -- a second trainer, or a user's own captured graph, keeps replaying
        self._arena = engine.GraphArena() if self.use_graph else None
        self._closed = False
        self._sync_replicas()

    def _independent_stream(self, others, candidates=12):
        """A stream whose hardware queue is none of `others`' queues, found by experiment: a ~2 ms spin kernel goes to each
        of `others`, a 4-byte fill and an event to the candidate; if the event completes while the spins still run, the
        candidate's work does not queue behind theirs.  If no candidate passes (host jitter, a card shared with other
        processes), the last DISTINCT candidate serves -- correct, only slower (the producer then queues behind the training
        graph: + ~45 us per iteration) -- and `independent_queues` says so (bench.py reports it, a RuntimeWarning is raised)."""
        import time
        import warnings
        flag = torch.zeros(1, device=self.dev)
        fallback = None
        with torch.cuda.device(self.dev):
            for _ in range(candidates):
                cand = torch.cuda.Stream(device=self.dev)
                if any(cand.cuda_stream == o.cuda_stream for o in others):
                    continue
                fallback = cand
                torch.cuda.synchronize(self.dev)
                for o in others:
                    with torch.cuda.stream(o):
                        torch.cuda._sleep(5_000_000)
                ev = torch.cuda.Event()
                with torch.cuda.stream(cand):
                    flag.fill_(1.0)
                    ev.record(cand)
                t0 = time.perf_counter()
                while not ev.query() and time.perf_counter() - t0 < 0.7e-3:
                    pass
                ok = ev.query()
                torch.cuda.synchronize(self.dev)
                if ok:
                    self.independent_queues.append(True)
                    return cand
            self.independent_queues.append(False)
            warnings.warn("GraphTrainer: no stream with a hardware queue of its own was found for the pipelined feed; the producer of "
                          "batch i + 1 may queue behind the training graph of batch i (slower, same result)", RuntimeWarning, stacklevel=3)
            while fallback is None:               # (every candidate so far was one of `others`: keep asking for a distinct one)
                cand = torch.cuda.Stream(device=self.dev)
                if not any(cand.cuda_stream == o.cuda_stream for o in others):
                    fallback = cand
        return fallback

    # ------------------------------------------------------------------------------------------------ the iteration
    def bucket_top(self, n):
        return min(self.n_max, self.n_min + self.width * ((n - self.n_min) // self.width + 1) - 1)

    def _simulate(self, bk):
        """This rank's shard of the batch: prior draws and trials, both keyed by the global row index (offset word)."""
        engine.draw_prior_device(self.model_id, self.B, seed=self.seed, set_offset=0, set_offset_dev=self.offset,
                                 out=bk.params, device=self.dev)
        engine.simulate(self.model_id, bk.params, bk.n_top, dt=self.dt, max_steps=self.max_steps, seed=self.seed,
                        set_offset=0, set_offset_dev=self.offset, fast=True, want_summary=False, out_trials=bk.trials,
                        device=self.dev)
        self.offset += self.B * self.world

    def _forward_backward(self, params, trials, n2):
        """configurator (basic_ddm_dc.py:139-160) on device scalars (n2 = the batch-shared N and log N, :151-155) + maximum-likelihood
        loss + backward into the flat buffer."""
        for net, sink in self._sinks:
            net.grad_sink = sink
        try:
            self._forward_backward_body(params, trials, n2)
        finally:
            for net, _ in self._sinks:
                net.grad_sink = None

    def _forward_backward_body(self, params, trials, n2):
        conf = {"summary_conditions": trials, "summary_n": n2[0:1],
                "direct_conditions": n2[1:2].view(1, 1).expand(trials.shape[0], 1),     # log(N), :151-155
                "parameters": params if self.P_net == self.P else params[:, :self.P_net]}
        loss = self.amortizer.compute_loss(conf)
        # gradients straight into the flat buffer with ONE concatenation (accumulating into pre-set .grad views costs one add
        # kernel per parameter tensor plus the zero fill; torch._foreach_copy_ runs as one copy per tensor here: ~90 launches)
        grads = torch.autograd.grad(loss, self.params, grad_outputs=self._one)      # (a constant seed: autograd's own ones_like is a fill launch)
        placed = [g.data_ptr() == s_.data_ptr() for g, s_ in zip(grads, self._slots)]
        if not any(placed):                                  # no sink took effect: ONE concatenation into the flat buffer
            pieces = []
            for g, pad in zip(grads, self._pads):
                pieces.append(g.reshape(-1))
                if pad is not None:
                    pieces.append(pad)
            torch.cat(pieces, out=self.flat[:self.n_el])
        elif not all(placed):                                # some gradients are in their slots already (a concatenation would read
            for g, s_, ok in zip(grads, self._slots, placed):     # and write the same memory): copy the others one by one
                if not ok:
                    s_.copy_(g)
        if loss.data_ptr() != self._loss_slot.data_ptr():
            self._loss_slot.copy_(loss.detach().view(1))

    def _set_n(self, bk, n):
        """The batch's number of real trials and its logarithm into the bucket's device scalars -- every graph of a bucket reads its
        bucket's pair -- in one launch (eager, on the current stream)."""
        if self._lib is not None:
            if self._lib.nddm_train_set2(bk.n2.data_ptr(), float(n), math.log(float(n)), torch.cuda.current_stream(self.dev).cuda_stream) != 0:
                raise RuntimeError("nddm_train_set2 failed")
        else:
            bk.n2[0:1].fill_(float(n)); bk.n2[1:2].fill_(math.log(float(n)))

    def _make_sinks(self):
        """Where the networks' fused backward (and the flow's loss) may write straight into this trainer's flat gradient buffer
        (amortizer.py: `grad_sink`): [(network, sink)].  Only where the layouts agree -- the summary network's tensors must lie back
        to back -- otherwise the gather copy in _forward_backward serves.  The sinks are set for the duration of a
        _forward_backward only: gradients that alias a buffer are this loop's business, not that of whoever else runs the networks."""
        inf, summ = getattr(self.amortizer, "inference_net", None), getattr(self.amortizer, "summary_net", None)
        slot = {id(p): (s_, o) for p, s_, o in zip(self.params, self._slots, self._offs)}
        sinks = []
        if inf is not None and hasattr(inf, "_flow_params") and hasattr(inf, "grad_sink"):
            sinks.append((inf, {"grads": {p.data_ptr(): slot[id(p)][0] for p in inf._flow_params() if id(p) in slot}, "loss": self._loss_slot}))
        if summ is not None and hasattr(summ, "fused_params") and hasattr(summ, "grad_sink"):
            ps = summ.fused_params()
            ok = bool(ps) and all(id(p) in slot for p in ps)
            o0 = run = slot[id(ps[0])][1] if ok else 0
            for p in ps if ok else []:
                ok = ok and slot[id(p)][1] == run
                run += p.numel()
            if ok:
                sinks.append((summ, self.flat[o0:run]))
        return sinks

    def _update(self, scale):
        """cosine schedule, clip_grad_norm_(5.0) on the flat gradient, Adam, loss into the history buffer, counters."""
        if self._lib is not None:
            rc = self._lib.nddm_train_adam_step(self.flat_p.data_ptr(), self.flat.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), self.n_el, self._partial.data_ptr(), float(scale), self.clip,
                                                self.lr0, float(max(1, self.T)), 0.9, 0.999, 1e-8, self.step_i.data_ptr(),
                                                self.step_f.data_ptr(), self.lr_t.data_ptr(), self.loss_buf.data_ptr(),
                                                self.loss_buf.numel(), self.flat[self.n_el:].data_ptr(),
                                                torch.cuda.current_stream(self.dev).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"nddm_train_adam_step failed ({rc})")
            return
        g = self.flat[:self.n_el]
        if scale != 1.0:
            self.flat.mul_(scale)                                     # mean over ranks (gradients and the loss slot)
        total = float(max(1, self.T))                                 # (held at the final value past `total`, as the kernel does)
        self.lr_t.copy_((0.5 * self.lr0 * (1.0 + torch.cos(torch.clamp(self.step_f, max=total) * (math.pi / total)))).view(()))
        coef = torch.clamp(self.clip / (g.norm() + 1e-6), max=1.0)
        g.mul_(coef)
        self.optimizer.step()
        self.loss_buf.index_copy_(0, torch.remainder(self.step_i, self.loss_buf.numel()), self.flat[self.n_el:])
        self.step_i += 1
        self.step_f += 1.0

    def _on_comm_stream(self, fn):
        """An exchange step on the communication stream (the sequential loop's all-gather).  Collectives never go to a stream
        that captures: the process group's watchdog thread polls the events of its collectives (recorded on the stream they
        were issued on), and HIP refuses a query of an event whose stream is capturing (hipErrorCapturedEvent aborts the
        process) -- which is why every graph is captured on a stream of its own (self._cap)."""
        self._comm.wait_stream(self._stream)
        with torch.cuda.stream(self._comm):
            fn()
        self._stream.wait_stream(self._comm)

    def _gather(self, bk):
        """ONE all-gather reassembles the minibatch (north_star): parameter rows and trials travel in one packed shard."""
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.all_gather_into_tensor(bk.g_shard, bk.shard)
        else:
            dist.all_gather(list(bk.g_shard.unbind(0)), bk.shard)

    def _copy2(self, dst_p, src_p, dst_t, src_t, n_real=None, bk=None):
        """dst_p <- src_p, dst_t <- src_t and (n_real given) the N / log N scalars: ONE launch where the library's staging kernel
        applies (contiguous, 16-byte aligned, multiples of four floats), else PyTorch copies."""
        if (self._lib is not None and src_p.is_contiguous() and src_t.is_contiguous() and dst_p.is_contiguous() and dst_t.is_contiguous()
                and src_p.numel() % 4 == 0 and src_t.numel() % 4 == 0
                and not (src_p.data_ptr() | src_t.data_ptr() | dst_p.data_ptr() | dst_t.data_ptr()) & 15):
            if self._lib.nddm_train_stage(dst_p.data_ptr(), src_p.data_ptr(), src_p.numel(), dst_t.data_ptr(), src_t.data_ptr(), src_t.numel(),
                                          bk.n2.data_ptr() if n_real is not None else None, float(n_real or 1),
                                          math.log(float(n_real or 1)), torch.cuda.current_stream(self.dev).cuda_stream) != 0:
                raise RuntimeError("nddm_train_stage failed")
            return
        dst_p.view(src_p.shape).copy_(src_p); dst_t.view(src_t.shape).copy_(src_t)
        if n_real is not None:
            self._set_n(bk, n_real)

    def _produced(self, bk):
        """What the producer of a batch leaves: this rank's shard, or the gathered minibatch (rank-major views of the packed buffer)."""
        return (bk.g_params, bk.g_trials) if bk.g_shard is not None else (bk.params, bk.trials)

    def _stage(self, bk, n_real=None):
        """The produced batch -- this rank's shard, or the gathered minibatch (rank-major strided views of the packed buffer)
        -- into the contiguous tensors the training graph reads; with n_real also the batch's N and log N into their device
        scalars (one launch for all of it where the sources are contiguous)."""
        if bk.staged:
            self._copy2(bk.t_params, self._produced(bk)[0], bk.t_trials, self._produced(bk)[1], n_real, bk)
        elif n_real is not None:
            self._set_n(bk, n_real)

    def _all_reduce_gradients(self):
        import torch.distributed as dist
        dist.all_reduce(self.flat)

    def _has_collective(self):
        import torch.distributed as dist
        return self.world > 1 or (self.split and dist.is_available() and dist.is_initialized())

    def _sync_replicas(self):
        """More than one rank: every replica starts from RANK 0's weights, buffers (the flow's permutations), Adam moments,
        counters and learning rate.  `gather` is replicated training and `ddp` all-reduces gradients only -- ranks that
        built their amortizer from different seeds, or of which only one loaded a checkpoint, would otherwise train
        different models without any error."""
        import torch.distributed as dist
        if not (self.world > 1 and dist.is_available() and dist.is_initialized()):
            return
        m, v = self._moments()
        ts = [self.flat_p, m, v, self.step_i, self.step_f, self.lr_t] + [b for b in self.amortizer.buffers()]
        if self.optimizer is not None:
            ts.append(self.optimizer.state[self._flat_param]["step"])
        torch.cuda.synchronize(self.dev)
        with torch.cuda.device(self.dev), torch.cuda.stream(self._comm), torch.no_grad():
            for t in ts:
                if t.is_cuda:
                    dist.broadcast(t, 0)
                else:                                        # (a buffer left on the host)
                    d = t.to(self.dev)
                    dist.broadcast(d, 0)
                    t.copy_(d)
        self._comm.synchronize()
        for mod in self.amortizer.modules():                 # host copies derived from buffers (the flow's permutations)
            if hasattr(mod, "_refresh_host_perms"):
                mod._refresh_host_perms()

    # ------------------------------------------------------------------------------------------------ graphs
    def _mutable(self):
        # (the flat gradient buffer too: a stretch that starts with the update clips it in place)
        ts = [self.flat_p, self.offset, self.step_i, self.step_f, self.lr_t, self.flat, self.loss_buf]
        if self.optimizer is None:
            return ts + [self.exp_avg, self.exp_avg_sq]
        for st in self.optimizer.state.values():
            ts += [st["step"], st["exp_avg"], st["exp_avg_sq"]]
        return ts

    def _bucket(self, n_top, parity=None):
        """The static tensors of one n_trials bucket: this rank's simulated shard, (world > 1, gather) the reassembled
        minibatch, and the training inputs `t_*` -- the shard, the gathered minibatch, or (experience replay) a staging
        copy of a stored batch.  parity 0 / 1: one of the TWO buffer sets of a bucket in the pipelined loop's direct form -- the
        training graph reads what the simulate graph wrote, no staging copy; each set has graphs of its own."""
        key = n_top if parity is None else (n_top, parity)
        bk = self._buckets.get(key)
        if bk is not None:
            return bk
        bk = _Bucket()
        bk.n_top = n_top
        bk.free_ev = None
        with torch.cuda.device(self.dev):
            # parameter rows and trials of this rank's shard live in ONE buffer (params padded to 16 bytes), so that the
            # exchange step is one collective
            bp = -(-self.B * self.P // 4) * 4
            bk.shard = torch.empty(bp + self.B * n_top * 2, dtype=torch.float32, device=self.dev)
            bk.params = bk.shard[:self.B * self.P].view(self.B, self.P)
            bk.trials = bk.shard[bp:].view(self.B, n_top, 2)
            bk.g_shard = bk.g_params = bk.g_trials = None
            rows = self.B
            if self.parallel == "gather" and self._has_collective():
                bk.g_shard = torch.empty((self.world, bk.shard.numel()), dtype=torch.float32, device=self.dev)
                bk.g_params = bk.g_shard[:, :self.B * self.P].view(self.world, self.B, self.P)
                bk.g_trials = bk.g_shard[:, bp:].view(self.world, self.B, n_top, 2)
                rows = self.world * self.B
            # staged: the producer (simulator, all-gather) refills its buffers while the training graph reads these
            bk.n2 = torch.tensor([float(n_top), math.log(float(n_top))], dtype=torch.float32, device=self.dev)   # N, log N of the batch in flight
            bk.staged = (self.overlap and parity is None) or bk.g_shard is not None
            if bk.staged:
                bk.t_params = torch.empty((rows, self.P), dtype=torch.float32, device=self.dev)
                bk.t_trials = torch.empty((rows, n_top, 2), dtype=torch.float32, device=self.dev)
            else:
                bk.t_params, bk.t_trials = bk.params, bk.trials
            bk.r_params = bk.r_trials = None          # staging of a replayed batch: allocated by the first replay iteration
            bk.graphs = {}
        self._buckets[key] = bk
        return bk

    def _run(self, bk, key, fn):
        """One graph-able stretch of the iteration on one bucket: captured at its first use -- after one eager pass at this
        shape (GEMM heuristics, workspaces, autograd buffers) whose every effect on the trainer's state is rolled back, so
        that capturing does not cost an iteration -- and replayed from then on."""
        if not self.use_graph:
            fn()
            return
        g = bk.graphs.get(key)
        if g is None:
            if self.overlap:
                torch.cuda.synchronize(self.dev)    # the other stream's work in flight must not see the warm-up pass's state
            with torch.no_grad():
                snap = [t.clone() for t in self._mutable()]
            fn()
            with torch.no_grad():
                for t, s0 in zip(self._mutable(), snap):
                    t.copy_(s0)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            # thread_local: a process group's watchdog thread polls its events while this thread captures, which the
            # default (global) capture mode turns into an error that kills the process
            with self._arena.bound(), torch.cuda.graph(g, pool=self._pool, stream=self._cap, capture_error_mode="thread_local"):
                fn()
            bk.graphs[key] = g
        g.replay()

    def _iteration(self, n, replay=None):
        """simulate -> [all-gather] -> [experience replay: store, draw, stage] -> forward/backward -> [gradient all-reduce]
        -> update.  Consecutive stretches with nothing eager between them are ONE graph: the whole iteration online at one
        rank; simulate | forward/backward/update with an all-gather or the replay buffer in the middle; simulate +
        forward/backward | update around a gradient all-reduce."""
        coll, ddp = self._has_collective(), self.parallel == "ddp"
        gather = coll and not ddp
        self._keep_loss_ring()
        bk = self._bucket(self.bucket_top(n))
        self._set_n(bk, n)
        sim = lambda: self._simulate(bk)
        if replay is None:
            fb = lambda: self._forward_backward(bk.t_params, bk.t_trials, bk.n2)
            up = lambda: self._update(1.0 / self.world if (coll and ddp) else 1.0)
            if not coll and not self.split:
                self._run(bk, "sim+fb+up", lambda: (sim(), fb(), up()))
            elif gather:
                self._run(bk, "sim", sim)
                self._on_comm_stream(lambda: self._gather(bk))
                self._stage(bk)
                self._run(bk, "fb+up", lambda: (fb(), up()))
            else:                                   # ddp, or the two-graph form forced at one rank without a process group
                self._run(bk, "sim+fb", lambda: (sim(), fb()))
                if coll:
                    self._all_reduce_gradients()                # (on the training stream itself: it never captures)
                self._run(bk, "up", up)
            return
        # experience replay (basic_ddm_dc.py:199-202 calls trainer.train_experience_replay): the fresh batch goes into the
        # buffer, the step trains on a stored one -- of ITS bucket, with ITS N
        self._run(bk, "sim", sim)
        if gather:
            self._on_comm_stream(lambda: self._gather(bk))
            self._stage(bk)
        bt = self._replay_stage((bk.t_params.clone(), bk.t_trials.clone(), n), replay)
        fb = lambda: self._forward_backward(bt.r_params, bt.r_trials, bt.n2)
        up = lambda: self._update(1.0 / self.world if (coll and ddp) else 1.0)
        if coll and ddp:
            self._run(bt, "r:fb", fb)
            self._all_reduce_gradients()                # (on the training stream itself: it never captures)
            self._run(bt, "up", up)
        else:
            self._run(bt, "r:fb+up", lambda: (fb(), up()))

    def _train_overlapped(self, iterations, replay):
        """The pipelined loop (graphs, any world size): the PRODUCER of batch i + 1 -- its `sim` graph on the simulate stream
        and, in `gather` mode, the all-gather that reassembles it on the communication stream -- runs beside the training
        graph(s) of batch i on the training stream.  (A simulate launch of 32 sets is a few dozen waves for 30-200
        microseconds: alone on the chip it is pure latency, and so is a small all-gather.)  The gradient all-reduce of `ddp`
        stays between its two graphs on the training side.  Within this call only: nothing is simulated beyond the last
        iteration, so the random stream's position after the call is what the sequential loop leaves."""
        T, S, C = self._stream, self._sim_stream, self._comm
        coll, ddp = self._has_collective(), self.parallel == "ddp"
        gather = coll and not ddp
        two = (coll and ddp) or (self.split and not gather)          # forward/backward | [all-reduce] | update
        cur = torch.cuda.current_stream(self.dev)
        T.wait_stream(cur)
        S.wait_stream(T)
        C.wait_stream(T)
        ns = [shared_prior_N(self.seed, self.n_base + self.iteration + k, self.n_min, self.n_max) for k in range(int(iterations))]
        up = lambda: self._update(1.0 / self.world if (coll and ddp) else 1.0)
        stamps = self.stage_stamps = [] if _os.environ.get("NDDM_TRAIN_STAGE_STAMPS") else None     # developer aid (tools/train_stage_times.py)
        # Host order of an iteration.  The producer of batch i + 1 goes to its streams BEFORE the training graph of batch i goes
        # to its own: a cross-stream wait is resolved against what the other stream holds when the wait is ISSUED -- enqueued
        # after the training graph, the simulate graph (which only depends on batch i's staging copies) was seen starting
        # 300-400 us into it, and at dt=.001 (a 200 us launch) finishing after it (HISTORY.md §C.6, "Config 5, round 4"; the pipelined timelines: profiles/r4_train_timeline*.txt).  Only a
        # HOST-blocking exchange (gloo) turns the order round: the host then waits in the collective while the device trains.
        produce_first = not (gather and self.backend != "nccl")

        # The DIRECT form (online training): a bucket has TWO buffer sets (and graphs for each), used in turn, so that the
        # PRODUCER of batch i + 1 fills the training graph's inputs while the training graph of batch i reads the other set (or
        # another bucket's): the simulate graph writes them itself, or -- with an all-gather, whose result is rank-major and
        # strided -- the communication stream stages them right behind the collective; the producer also sets the batch's N /
        # log N.  Nothing but the wait for the producer's event stands between two training graphs.  Experience replay stages on
        # the training stream (the fresh batch goes into the buffer, a drawn one into the graph's inputs).
        direct = replay is None

        def bucket_of(n):
            n_top = self.bucket_top(n)
            if not direct:
                return self._bucket(n_top)
            use = self._uses.get(n_top, 0)
            self._uses[n_top] = use + 1
            return self._bucket(n_top, use & 1)

        last_c = [None]                                 # the event behind the previous producer's work on the communication stream

        def produce(n):
            bk = bucket_of(n)
            with torch.cuda.stream(S):
                if gather and last_c[0] is not None:
                    S.wait_event(last_c[0])             # the previous all-gather has read the shard this simulate may overwrite
                if direct:
                    if bk.free_ev is not None:
                        S.wait_event(bk.free_ev)        # the training graph that read this buffer set last (two uses ago) is done
                    if not gather:
                        self._set_n(bk, n)
                self._run(bk, "sim", lambda: self._simulate(bk))
                ev = torch.cuda.Event(enable_timing=_TIMED_EVENTS)
                ev.record(S)
            if gather:
                C.wait_event(ev)
                with torch.cuda.stream(C):
                    self._gather(bk)
                    if direct:                          # ... and the gathered minibatch into the training graph's inputs
                        if bk.free_ev is not None:
                            C.wait_event(bk.free_ev)
                        self._stage(bk, n)
                    ev = torch.cuda.Event(enable_timing=_TIMED_EVENTS)
                    ev.record(C)
            entry = None
            if replay is not None:                      # experience replay: the fresh batch is copied out for the buffer by its producer
                with torch.cuda.stream(C if gather else S):
                    src_p, src_t = self._produced(bk)
                    entry = (torch.empty_like(bk.t_params), torch.empty_like(bk.t_trials), n)
                    self._copy2(entry[0], src_p, entry[1], src_t)
                    ev = torch.cuda.Event(enable_timing=_TIMED_EVENTS)
                    ev.record(C if gather else S)
            if gather:
                last_c[0] = ev
            return ev, bk, entry

        with torch.cuda.device(self.dev):
            nxt = produce(ns[0]) if ns else None
            for k, n in enumerate(ns):
                ev, bk, entry = nxt
                with torch.cuda.stream(T):
                    if stamps is not None:
                        stamps.append([torch.cuda.Event(enable_timing=True) for _ in range(4)])
                        stamps[-1][0].record(T)
                    T.wait_event(ev)
                    if stamps is not None:
                        stamps[-1][1].record(T)
                # (the producer's own buffers are read on the producer's streams only -- by the all-gather, the staging, the copy for
                #  the replay buffer -- so the next producer, which follows on the same streams, needs no event from this one)
                if produce_first and k + 1 < len(ns):
                    nxt = produce(ns[k + 1])
                with torch.cuda.stream(T):
                    self._keep_loss_ring()
                    if replay is None:
                        b, pre = bk, ""
                        fb = lambda: self._forward_backward(bk.t_params, bk.t_trials, bk.n2)
                    else:
                        bt = self._replay_stage(entry, replay)
                        b, pre = bt, "r:"
                        fb = lambda: self._forward_backward(bt.r_params, bt.r_trials, bt.n2)
                    if two:
                        self._run(b, pre + "fb", fb)
                        if coll:
                            self._all_reduce_gradients()                # (on the training stream itself: it never captures)
                        self._run(b, "up", up)
                    else:
                        if stamps is not None:
                            stamps[-1][2].record(T)
                        self._run(b, pre + "fb+up", lambda: (fb(), up()))
                        if stamps is not None:
                            stamps[-1][3].record(T)
                    if direct:
                        bk.free_ev = torch.cuda.Event(enable_timing=_TIMED_EVENTS)
                        bk.free_ev.record(T)
                if not produce_first and k + 1 < len(ns):
                    nxt = produce(ns[k + 1])
                self.iteration += 1
        cur.wait_stream(T)
        cur.wait_stream(S)
        cur.wait_stream(C)

    def _replay_stage(self, entry, replay):
        """Experience replay's host side: the fresh batch into the buffer (overwriting a random slot once it is full), a stored
        batch drawn at random into ITS bucket's staging tensors, with ITS N (-> that bucket)."""
        ring, rng, cap = replay
        if len(ring) < cap:
            ring.append(entry)
        else:
            ring[int(rng.integers(cap))] = entry
        p_s, t_s, n_s = ring[int(rng.integers(len(ring)))]
        bt = self._bucket(self.bucket_top(n_s))
        if bt.r_params is None:
            bt.r_params, bt.r_trials = torch.empty_like(p_s), torch.empty_like(t_s)
        self._copy2(bt.r_params, p_s, bt.r_trials, t_s, n_s, bt)        # (+ ITS N and log N: one launch)
        cur = torch.cuda.current_stream(self.dev)                       # (stored batches are allocated on their producer's stream and read
        p_s.record_stream(cur); t_s.record_stream(cur)                  #  here: the allocator must not hand them out again before this copy)
        return bt

    def train_online(self, iterations):
        """`iterations` training steps, every batch fresh; returns nothing -- losses stay on the device until loss_history()."""
        self._note_schedule_end(iterations)
        if self.overlap:
            return self._train_overlapped(iterations, None)
        self._stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.device(self.dev), torch.cuda.stream(self._stream):
            for _ in range(int(iterations)):
                self._iteration(shared_prior_N(self.seed, self.n_base + self.iteration, self.n_min, self.n_max))   # batch-shared N (basic_ddm_dc.py:50-52, 131)
                self.iteration += 1
        torch.cuda.current_stream(self.dev).wait_stream(self._stream)

    def train_experience_replay(self, iterations, capacity_in_batches=100, replay_seed=0):
        """The reference's call (basic_ddm_dc.py:199-202): each iteration simulates one fresh batch into a buffer of
        `capacity_in_batches` batches (every batch keeps its own N) and trains on a stored batch drawn at random -- the same
        draws, in the same order, as amortizer.Trainer.train_experience_replay."""
        import numpy as np
        if self._replay is None:
            self._replay = ([], np.random.default_rng(replay_seed), int(capacity_in_batches))
        elif self._replay[2] != int(capacity_in_batches):      # a continued buffer (set_position) under another capacity
            ring, rng, _ = self._replay
            del ring[int(capacity_in_batches):]
            self._replay = (ring, rng, int(capacity_in_batches))
        self._note_schedule_end(iterations)
        if self.overlap:
            return self._train_overlapped(iterations, self._replay)
        self._stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.device(self.dev), torch.cuda.stream(self._stream):
            for _ in range(int(iterations)):
                self._iteration(shared_prior_N(self.seed, self.n_base + self.iteration, self.n_min, self.n_max), self._replay)
                self.iteration += 1
        torch.cuda.current_stream(self.dev).wait_stream(self._stream)

    # ---- the run's POSITION in the random stream, apart from its weights / optimizer / schedule: what a LATER run (the next
    # train_* call of amortizer.Trainer(graph=True), or a resumed one) needs to continue with fresh data instead of replaying this one's
    def position(self):
        """{'offset': global index of the next batch's first parameter set (rank 0's row 0), 'n_key': key of the next batch-shared N,
        'replay': the experience-replay buffer and its generator, or None} -- one synchronisation of the training stream."""
        self._stream.synchronize()
        pos = {"offset": int(self.offset.item()) - self.rank * self.B, "n_key": self.n_base + self.iteration, "replay": None}
        if self._replay is not None:
            ring, rng, cap = self._replay
            pos["replay"] = {"ring": [(p.cpu(), t.cpu(), n) for p, t, n in ring], "rng": rng.bit_generator.state, "capacity": cap}
        return pos

    def set_position(self, pos):
        """Continue the stream a previous run left at `pos` (position()): the next batch takes the next parameter sets and the next
        N, experience replay goes on with the stored batches.  Weights, Adam's state and the schedule are not touched."""
        import numpy as np
        self._stream.synchronize()
        with torch.no_grad():
            self.offset.fill_(int(pos["offset"]) + self.rank * self.B)
        self.n_base = int(pos["n_key"]) - self.iteration
        self._replay = None
        if pos.get("replay") is not None:
            rng = np.random.default_rng(0)
            rng.bit_generator.state = pos["replay"]["rng"]
            self._replay = ([(p.to(self.dev), t.to(self.dev), n) for p, t, n in pos["replay"]["ring"]], rng, int(pos["replay"]["capacity"]))
        torch.cuda.synchronize(self.dev)

    def _note_schedule_end(self, iterations):
        """A run that goes past `total_steps` trains on at the schedule's final rate (0 for the cosine): legal, and what
        BayesFlow's decay does past `decay_steps`, but almost never what was meant -- say so once."""
        if self.iteration + int(iterations) > self.T and not self._warned_past_end:
            import warnings
            self._warned_past_end = True
            warnings.warn(f"GraphTrainer: iteration {self.iteration} + {int(iterations)} goes past total_steps={self.T}; the cosine schedule "
                          "holds its final learning rate (0) from there on", RuntimeWarning, stacklevel=3)

    def _drain_losses(self):
        """Read the losses the host does not hold yet out of the device ring (one synchronisation of the training stream)."""
        have, done, cap = len(self._loss_host), self.iteration, self.loss_buf.numel()
        if done > have:
            self._stream.synchronize()
            buf = self.loss_buf.cpu().numpy()
            self._loss_host += [float(buf[i % cap]) for i in range(have, done)]

    def _keep_loss_ring(self):
        """Before the iteration that would overwrite a loss the host has not read: read the ring out (once per
        `total_steps` iterations, so the history is complete however long the run is)."""
        if self.iteration - len(self._loss_host) >= self.loss_buf.numel():
            self._drain_losses()

    def loss_history(self):
        """The losses of ALL iterations run so far (one device synchronisation)."""
        self._drain_losses()
        return list(self._loss_host)

    @torch.no_grad()
    def validation_loss(self, conf):
        """Loss on a configured data set (`validation_sims` of basic_ddm_dc.py:186-188, 199-202), computed eagerly."""
        self._stream.synchronize()
        return float(self.amortizer.compute_loss(conf))

    # ---- checkpoint / resume: weights, Adam state, and the POSITION of the run -- iteration (it keys the batch-shared N),
    # the device-side set offset, step counters and learning rate, the loss history, the replay buffer and its generator --
    # so that a resumed run continues the random stream and reproduces the uninterrupted one
    def state_dict(self):
        self._stream.synchronize()
        st = {"model": self.amortizer.state_dict(), "optimizer": self._optimizer_state(), "iteration": self.iteration, "n_base": self.n_base,
              "offset": self.offset.cpu(), "step_i": self.step_i.cpu(), "step_f": self.step_f.cpu(), "lr": self.lr_t.cpu(),
              "loss_buf": self.loss_buf.cpu(), "loss_host": self.loss_history(), "replay": None}
        if self._replay is not None:
            ring, rng, cap = self._replay
            st["replay"] = {"ring": [(p.cpu(), t.cpu(), n) for p, t, n in ring], "rng": rng.bit_generator.state, "capacity": cap}
        return st

    def _moments(self):
        if self.optimizer is None:
            return self.exp_avg, self.exp_avg_sq
        state = self.optimizer.state[self._flat_param]
        return state["exp_avg"], state["exp_avg_sq"]

    def _optimizer_state(self):
        m, v = self._moments()
        return {"exp_avg": m.cpu(), "exp_avg_sq": v.cpu(), "layout": "flat, every parameter tensor padded to a multiple of 4 floats"}

    def load_state_dict(self, st):
        import numpy as np
        self._stream.synchronize()
        with torch.no_grad():
            self.amortizer.load_state_dict(st["model"])
            # Adam's state tensors are the ones the captured graphs hold: copy INTO them (Adam's step count is step_i)
            m, v = self._moments()
            m.copy_(st["optimizer"]["exp_avg"])
            v.copy_(st["optimizer"]["exp_avg_sq"])
            if self.optimizer is not None:
                self.optimizer.state[self._flat_param]["step"].copy_(st["step_i"].to(torch.float32).view(()))
            self.offset.copy_(st["offset"]); self.step_i.copy_(st["step_i"]); self.step_f.copy_(st["step_f"])
            self.lr_t.copy_(st["lr"])
            n = min(self.loss_buf.numel(), st["loss_buf"].numel())
            self.loss_buf[:n].copy_(st["loss_buf"][:n])
        self.iteration = int(st["iteration"])
        self.n_base = int(st.get("n_base", 0))
        # (a checkpoint written before the ring existed holds the first min(iteration, capacity) losses in its buffer)
        self._loss_host = list(st["loss_host"]) if "loss_host" in st else st["loss_buf"][:min(self.iteration, n)].tolist()
        self._replay = None
        if st.get("replay") is not None:
            rng = np.random.default_rng(0)
            rng.bit_generator.state = st["replay"]["rng"]
            self._replay = ([(p.to(self.dev), t.to(self.dev), n) for p, t, n in st["replay"]["ring"]], rng, st["replay"]["capacity"])
        self._sync_replicas()          # more than one rank: whatever each rank loaded, all continue from rank 0's state

    def optimizer_state(self):
        """The part of state_dict() that is neither weights nor position: Adam's two moments, the step counters, the learning rate,
        this run's iteration count and loss history, and the length of its schedule -- what amortizer.Trainer(graph=True) keeps in
        ckpt.pt so that a run interrupted between two epochs is CONTINUED (same moments, same place on the cosine schedule) rather
        than restarted with a fresh Adam."""
        st = self.state_dict()
        out = {k: st[k] for k in ("optimizer", "iteration", "step_i", "step_f", "lr", "loss_buf", "loss_host")}
        out["total_steps"] = int(self.T)
        return out

    def load_optimizer_state(self, st):
        """optimizer_state() of an interrupted run with the same total_steps: moments, counters, rate, iteration and losses.  Call
        BEFORE set_position (the key of the batch-shared N is counted from the iteration)."""
        self._stream.synchronize()
        with torch.no_grad():
            m, v = self._moments()
            m.copy_(st["optimizer"]["exp_avg"])
            v.copy_(st["optimizer"]["exp_avg_sq"])
            if self.optimizer is not None:
                self.optimizer.state[self._flat_param]["step"].copy_(st["step_i"].to(torch.float32).view(()))
            self.step_i.copy_(st["step_i"]); self.step_f.copy_(st["step_f"]); self.lr_t.copy_(st["lr"])
            n = min(self.loss_buf.numel(), st["loss_buf"].numel())
            self.loss_buf[:n].copy_(st["loss_buf"][:n])
        self.iteration = int(st["iteration"])
        self._loss_host = list(st["loss_host"])
        self._sync_replicas()

    def save_checkpoint(self, path):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(plain_state(self.state_dict()), path)

    def load_checkpoint(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))       # (plain_state: tensors, numbers, strings only)

    @property
    def n_graphs(self):
        return sum(len(b.graphs) for b in self._buckets.values())

    # ------------------------------------------------------------------------------------------------ ownership
    def close(self):
        """Destroy the graphs, then hand the library's memory behind THEIR captured launches back: the trainer's own graph
        arena (include/nddm.h: nddm_graph_arena_release) -- other trainers' and the user's graphs are not touched."""
        if self._closed:
            return
        self._closed = True
        torch.cuda.synchronize(self.dev)
        self._drain_losses()
        self._buckets.clear()
        self._replay = None
        if self._arena is not None:
            self._arena.release()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass
