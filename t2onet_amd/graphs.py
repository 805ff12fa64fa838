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
    __slots__ = ('static_in', 'static_out', 'static_gout', 'static_gin', 'fwd', 'bwd', 'needs_gin')


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, _anchor, slot):
        slot.static_in.copy_(img)
        slot.fwd.replay()
        ctx.slot = slot
        return slot.static_out.detach()

    @staticmethod
    def backward(ctx, gout):
        slot = ctx.slot
        slot.static_gout.copy_(gout)
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
        self._grad_ptrs = (self.grads[0].data_ptr(), self.grads[-1].data_ptr())
        # a grad-requiring scalar keeps the replay node in the autograd graph when the image itself needs no gradient
        self.anchor = torch.zeros((), device=sample_img.device, requires_grad=True)
        self.memsets_replaced = 0                      # memset nodes of the captured graphs turned into kernel nodes
        buffers = list(module.buffers())
        saved = [t.clone() for t in buffers]
        try:
            self.slots = self._capture(sample_img, calls, warmup)
        finally:
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
            s.needs_gin = k > 0                        # call 0 sees the input image: no gradient needed
            s.static_in = sample_img.detach().clone().requires_grad_(True)
            s.fwd = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(s.fwd, pool=pool, capture_error_mode='thread_local'):
                s.static_out = module(s.static_in)
            self.memsets_replaced += _harden(s.fwd)
            slots.append(s)
        for s in reversed(slots):
            s.static_gout = torch.zeros_like(s.static_out)
            s.bwd = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(s.bwd, pool=pool, capture_error_mode='thread_local'):
                wrt = ([s.static_in] if s.needs_gin else []) + params
                g = torch.autograd.grad((s.static_out,), wrt, (s.static_gout,), allow_unused=True)
                if s.needs_gin:
                    s.static_gin, g = g[0], g[1:]
                else:
                    s.static_gin = None
                have = [(acc, gi) for acc, gi in zip(self.grads, g) if gi is not None]
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
        if (self.params[0].grad is None or self.params[0].grad.data_ptr() != self._grad_ptrs[0]
                or self.params[-1].grad is None or self.params[-1].grad.data_ptr() != self._grad_ptrs[1]):
            raise RuntimeError('GraphedEncoder: a parameter .grad tensor was replaced after capture (use '
                               'zero_grad(set_to_none=False) / the Trainer\'s flat buffer)')
        return _Replay.apply(img, self.anchor, self.slots[call])
