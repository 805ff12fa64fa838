"""hipGraph capture of the image encoder for the train step.

The episode step calls the same encoder `calls` times on images of one fixed shape and keeps every call's
activations alive until the backward, so each call gets its own forward graph and its own backward graph
(one shared graph memory pool, captured in the order they are replayed: forwards 0..n-1, backwards n-1..0).

Different from torch.cuda.make_graphed_callables, by design:
  * the backward graph ADDS the parameter gradients into the parameters' existing `.grad` tensors (the views
    of the Trainer's flat all-reduce buffer) inside the graph: the autograd engine never sees the encoder's 62
    parameters, so the ~310 AccumulateGrad add launches per step (5 calls x 62 tensors) disappear;
  * capture runs in `thread_local` error mode: a process group's watchdog thread polling events does not
    invalidate the capture, so the data-parallel run (N > 1) keeps the graphs;
  * batch-norm running statistics touched by the warm-up iterations are restored even when capture fails;
  * every captured graph is rewritten before it is instantiated: its memset nodes become kernel nodes
    (t2o_graph_memsets_to_kernels).  On this stack a memset node ran out of order with the kernels around it on
    replay; the library's atomic weight-gradient solvers clear their output with hipMemsetAsync, so, depending on
    which solver its find step picked on a box, the stem's gradients came out as garbage from the second step on.
"""
import ctypes

import torch

from . import _lib


class _Slot:
    __slots__ = ('static_in', 'static_out', 'static_gout', 'static_gin', 'fwd', 'bwd', 'needs_gin', 'plan')


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, _anchor, slot):
        slot.static_in.copy_(img)
        if slot.plan is not None:                      # (Winograd-transformed filters: refreshed once per optimiser step)
            slot.plan.wino_forward(_lib.load(), torch.cuda.current_stream(img.device).cuda_stream)
        slot.fwd.replay()
        ctx.slot = slot
        return slot.static_out.detach()

    @staticmethod
    def backward(ctx, gout):
        slot = ctx.slot
        slot.static_gout.copy_(gout)
        if slot.plan is not None:                      # transformed weights at fixed addresses, refreshed once per optimiser step
            slot.plan.transformed(_lib.load(), torch.cuda.current_stream(gout.device).cuda_stream)
        slot.bwd.replay()                              # parameter gradients are accumulated inside the graph
        return (slot.static_gin.detach() if slot.needs_gin else None), None, None


def _harden(graph):
    """Memset nodes -> kernel nodes in the captured hipGraph, then instantiate it.  Returns the number replaced."""
    n = ctypes.c_int(0)
    rc = _lib.load().t2o_graph_memsets_to_kernels(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(n))
    _lib.check(rc, 't2o_graph_memsets_to_kernels')
    graph.instantiate()                                # (keep_graph=True: not instantiated by capture_end)
    return n.value


class GraphedEncoder:
    """module(img) for call index k in [0, calls) as hipGraph replays.  Training mode, fixed input shape; the
    parameters' .grad tensors must exist (and stay the same tensors) -- Trainer's FlatGradients provides that."""

    def __init__(self, module, sample_img, calls, warmup=3):
        self.module = module
        self.shape = tuple(sample_img.shape)
        self.params = [p for p in module.parameters() if p.requires_grad]
        if any(p.grad is None for p in self.params):
            raise RuntimeError('GraphedEncoder: every parameter needs a persistent .grad tensor before capture')
        self.grads = [p.grad for p in self.params]
        self._grad_ptrs = [g.data_ptr() for g in self.grads]
        self._param_ptrs = [p.data_ptr() for p in self.params]
        # a grad-requiring scalar keeps the replay node in the autograd graph when the image itself needs no gradient
        self.anchor = torch.zeros((), device=sample_img.device, requires_grad=True)
        self.memsets_replaced = 0                      # memset nodes of the captured graphs turned into kernel nodes
        buffers = list(module.buffers())
        saved = [t.clone() for t in buffers]
        # With the plan's transformed-weight cache on (the Trainer: it calls weights_changed() after every optimiser step) the
        # backward graphs READ the cache's fixed buffers and _Replay.backward refreshes them eagerly, once per step;
        # without it every backward graph carries its own weight transforms (any subset of slots may be replayed in a step).
        plan = module.trunk_plan() if hasattr(module, 'trunk_plan') else None
        self.static_plan = plan if (plan is not None and plan.persistent_wt) else None
        if plan is not None:
            plan.weights_changed()
        # the warm-up passes run real backwards: the trunk adds its parameter gradients into .grad inside its kernels
        # (encoder.py into_grad) -- put the gradient buffers back as they were
        saved_grads = [t.clone() for t in self.grads]
        try:
            self.slots = self._capture(sample_img, calls, warmup)
        finally:
            with torch.no_grad():
                for t, keep in zip(self.grads, saved_grads):
                    t.copy_(keep)
            if plan is not None:
                plan.weights_changed()
            with torch.no_grad():                      # warm-up and capture ran real forwards: undo their running statistics
                for t, keep in zip(buffers, saved):
                    t.copy_(keep)

    def _capture(self, sample_img, calls, warmup):
        dev = sample_img.device
        module, params = self.module, self.params
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            x = sample_img.detach().clone().requires_grad_(True)
            for _ in range(warmup):                    # kernel selection / lazy initialisation outside the capture
                out = module(x)
                torch.autograd.grad((out,), [x] + params, (torch.ones_like(out),), allow_unused=True)
            del out
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        pool = torch.cuda.graph_pool_handle()
        slots = []
        for k in range(calls):
            s = _Slot()
            s.plan = self.static_plan
            s.needs_gin = k > 0                        # call 0 sees the input image: no gradient needed
            s.static_in = sample_img.detach().clone().requires_grad_(True)
            s.fwd = torch.cuda.CUDAGraph(keep_graph=True)
            with _no_gc(), torch.cuda.graph(s.fwd, pool=pool, capture_error_mode='thread_local'):
                s.static_out = module(s.static_in)
            self.memsets_replaced += _harden(s.fwd)
            slots.append(s)
        for s in reversed(slots):
            s.static_gout = torch.zeros_like(s.static_out)
            s.bwd = torch.cuda.CUDAGraph(keep_graph=True)
            with _no_gc(), torch.cuda.graph(s.bwd, pool=pool, capture_error_mode='thread_local'):
                wrt = ([s.static_in] if s.needs_gin else []) + params
                g = torch.autograd.grad((s.static_out,), wrt, (s.static_gout,), allow_unused=True)
                if s.needs_gin:
                    s.static_gin, g = g[0], g[1:]
                else:
                    s.static_gin = None
                have = [(acc, gi) for acc, gi in zip(self.grads, g) if gi is not None]
                if have:                                # (all in place already: trunk kernels + functional.linear_acc)
                    torch._foreach_add_([a for a, _ in have], [b for _, b in have])
            self.memsets_replaced += _harden(s.bwd)
        for s in slots:
            # keep the buffers, drop the autograd graph: it holds the parameters' AccumulateGrad nodes, which were
            # created on the capture stream -- an eager encoder call reusing them would add into .grad on THAT
            # stream, racing with the graph replays' in-graph accumulation on the current stream
            s.static_out = s.static_out.detach()
            s.static_in = s.static_in.detach()
        return slots

    def usable(self, img, call):
        return (call is not None and call < len(self.slots) and tuple(img.shape) == self.shape
                and self.module.training and torch.is_grad_enabled())

    def __call__(self, img, call):
        for p, pp, gp in zip(self.params, self._param_ptrs, self._grad_ptrs):     # (62 integer compares per call)
            if p.data_ptr() != pp:
                raise RuntimeError('GraphedEncoder: a parameter was re-allocated after capture (module.to(...), '
                                   'use_channels_last() ...): the captured graphs still read the old storage')
            if p.grad is None or p.grad.data_ptr() != gp:
                raise RuntimeError('GraphedEncoder: a parameter .grad tensor was replaced after capture (use '
                                   'zero_grad(set_to_none=False) / the Trainer\'s flat buffer)')
        return _Replay.apply(img, self.anchor, self.slots[call])


class _no_gc:
    """No cyclic garbage collection while a stream is capturing.  The collector may run on ANY thread that allocates -- the autograd
    worker in the middle of a captured backward -- and finalise whatever garbage is around: an earlier Trainer with its own
    hipGraphs, side streams and pool memory.  Seen once in seven full test runs (round 5): `Fatal Python error: Aborted` inside
    a capture, the collector on the stack.  Collect first, then keep the collector off until the capture has ended."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class GraphedEpisodeStep:
    """The episode/L1 train step (train_seq2seqL1.py:74-88) -- request encoder, 5 x (image encoder, decoder step,
    sampling, parameter heads, operator), END select, L1 and the whole backward -- as ONE hipGraph.  Outside stay only
    the gradient all-reduce and Adam.  Per step the host issues a handful of calls instead of ~3,000 kernel launches; the
    ~1,500 small launches of the decoder steps between the encoder passes, which ran ~4 us apart when enqueued one by
    one, are graph nodes, and so is the request encoder (static shapes for a given request
    length), the step kernels of functional.lstm_layer, forked onto a second stream inside the capture (Actor._encode_request), i.e. a parallel branch of the
    graph beside the first image-encoder pass and, in the backward, beside the last one.

    The graph zeroes the flat gradient buffer first.  Sampling (`torch.rand` in actor.sample_categorical) and the
    request encoder's dropout use the generator's graph-safe state: every replay draws fresh numbers.  One instance
    per (image shape, request length L, reinforce_sample)."""

    def __init__(self, trainer, x, lengths, longest, img, target, reinforce_sample, warmup=2):
        model = trainer.model
        dev = img.device
        self.trainer = trainer
        self.reinforce_sample = reinforce_sample
        self.longest = int(longest)
        self.s_x = x.detach().clone()
        self.s_img = img.detach().clone()
        self.s_target = target.detach().clone()
        self.memsets_replaced = 0
        buffers = [t for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for t in m.buffers()]
        saved = [t.clone() for t in buffers]
        side = torch.cuda.Stream(device=dev)
        try:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):                        # lazy initialisation (kernel selection, zero regions) outside the capture
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            with _no_gc(), torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'):
                self.loss = self._body()
            self.memsets_replaced = _harden(self.graph)
        finally:
            with torch.no_grad():                              # warm-up ran real steps: undo their running statistics
                for t, keep in zip(buffers, saved):
                    t.copy_(keep)
            trainer.grads.zero()
            self.trainer = None                                # (used by _body only: no Trainer <-> graph reference cycle, so a finished
                                                               # Trainer and its graphs are freed at once, not by a later collection)

    def _body(self):
        from .train import end_l1_loss
        tr = self.trainer
        model = tr.model
        if tr._trunk is not None:
            tr._trunk.weights_changed()                        # the weights were updated since the last step: transform them again
        tr.grads.zero()
        try:
            tr._tape(self.s_img.shape[0])
            tr._arena(self.s_img, tr.opt.decoder_max_len)
            # the captured kernels read and write the tape's and the arena's storage: this graph keeps both alive whatever the
            # trainer's caches do later (another batch size, the arena cache's eviction)
            self._keep = (tr.__dict__.get('_tape_obj'), tr._trunk.__dict__.get('arena') if tr._trunk is not None else None)
            lengths = (self.s_x != tr.opt.null_id).sum(1)     # on the device, inside the graph: no host-side lengths to copy
            _, imgs, ops, _ = model.episode_forward(self.s_x, self.s_img, None, self.reinforce_sample, lengths, self.longest, stack=False)
            loss = end_l1_loss(imgs, ops, tr.opt.end_id, self.s_target)
            loss.backward()
            tr._flush_tape()                                   # the decoder's weight gradients: one product per weight, in the graph
        finally:
            tr._release()
        return loss.detach()

    def run(self, x, img, target):
        """One step for a batch whose longest request has this instance's length; returns the loss (a fresh tensor)."""
        self.s_x.copy_(x)
        self.s_img.copy_(img)
        self.s_target.copy_(target)
        self.graph.replay()
        return self.loss.clone()
