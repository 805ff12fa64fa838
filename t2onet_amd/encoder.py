"""The image encoder's convolutional trunk (models/actor_resnet.py:98-105: stem + 4 stages of 2 BasicBlocks) as ONE
autograd node with an explicit forward / backward schedule over the C ABI -- what the per-layer autograd functions of
`functional.py` did, without the passes autograd had to insert between them:

  * a block input that feeds both the first convolution and the shortcut gets ONE gradient: the data gradient of the
    first convolution takes the identity shortcut's gradient as an addend in its epilogue
    (t2o_conv3x3_dgrad_pre_nhwc), the 1x1 shortcut's data gradient is added into it in place
    (t2o_conv1x1s2_dgrad_acc_nhwc) -- 8 full-tensor `add` launches per pass (up to 805 MB each) are gone;
  * the 1x1 stride-2 shortcut convolutions run on this library's kernels (t2o_conv1x1.hip), all three directions:
    the trunk contains no library convolution, no atomic (non-deterministic) gradient and no memset;
  * the image is read and its gradient written in the image's own NCHW layout (t2o_stem_*: planar), the stem's data
    gradient ADDS into the image gradient the caller already holds (operator backward), no channels-last copies;
  * with persistent gradient buffers (`into_grad`: every parameter has a dense .grad -- the Trainer's flat buffer) all
    weight / batch-norm gradients are accumulated by the kernels that produce them; autograd sees no parameter
    gradient of the trunk (62 tensors x 5 encoder calls of accumulation launches per step);
  * transformed weights (tap-mirrored transposes for the data gradients) are made once per optimiser step
    (`TrunkPlan.weights_changed()` invalidates them), not once per data-gradient call.

Arithmetic is that of the per-layer path (same kernels, same order inside every sum); tests/test_gpu_encoder.py holds
both against fp64 autograd of the oracle's ResNet.
"""
import os

import torch

from . import _lib
from .functional import (_need_gpu, _persistent_grad, _ptr, _stream, _conv_workspace, _zero_block, wino_conv_nhwc, wino_fused_conv_nhwc, wino_wgrad_nhwc, wino_input,
                         wino_backward_nhwc, wino_dw_from, wino_fused_wgrad_nhwc)

# Winograd F(2x2,3x3) for the stride-1 layers with >= 256 channels (t2o_winograd.hip); T2O_WINOGRAD=0: the direct kernels everywhere
_WINOGRAD = os.environ.get('T2O_WINOGRAD', '1') != '0'
_WINO_MIN_C = int(os.environ.get('T2O_WINOGRAD_MIN_C', '256'))
_DUAL_BN = True      # a shortcut block's two batch norms in one pass each way (t2o_bn_dual_*); module switch for the tests
# Winograd F(2x2,3x3) with V and M kept on chip for the stride-1 layers of the 64- / 128-channel stages (t2o_wino_fused.hip);
# T2O_WINOGRAD_FUSED=0: the direct kernels there (A/B)
_WINO_FUSED = os.environ.get('T2O_WINOGRAD_FUSED', '1') != '0'
_BN_SUMS_EPILOGUE = True     # bn1's backward sums from the epilogue of conv2's data gradient (direct kernels); switch for the tests
# the weight gradient of the on-chip Winograd layers in the Winograd domain too, both transforms on chip (t2o_wino_wgrad.hip);
# T2O_WINOGRAD_WGRAD=0: the direct weight-gradient kernel there (A/B)
_WINO_WGRAD = os.environ.get('T2O_WINOGRAD_WGRAD', '1') != '0'


def _fast_direct(stride, Hi, Wi, Wo):
    """The LDS-DMA forward / data-gradient kernels take this layer: the OUTPUT-side width a multiple of 8 and, under
    stride 2, an even input."""
    return Wo % 8 == 0 and (stride == 1 or (Hi % 2 == 0 and Wi % 2 == 0))


def _batched(fn, name, st, srcs, dsts, int_lists):
    """fn(src pointers, dst pointers, int arrays..., n, stream) for up to 32 jobs per launch (host arrays)."""
    import ctypes
    for i in range(0, len(srcs), 32):
        n = min(32, len(srcs) - i)
        a = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs[i:i + n]])
        b = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts[i:i + n]])
        ints = [(ctypes.c_int * n)(*lst[i:i + n]) for lst in int_lists]
        _lib.check(fn(a, b, *ints, n, st), name)


def _nhwc(N, H, W, C, dev):
    return torch.empty((N, H, W, C), dtype=torch.float32, device=dev)


class TrunkPlan:
    """Static description of a ResNet trunk for _TrunkFn: layer list, parameter order, transformed-weight cache."""

    def __init__(self, resnet):
        self.net = resnet
        blocks = [b for layer in (resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4) for b in layer]
        self.blocks = blocks
        self.params = [resnet.conv1.weight, resnet.bn1.weight, resnet.bn1.bias]
        self.index = {}                                        # id(parameter) -> position in self.params
        for b in blocks:
            self.params += [b.conv1.weight, b.bn1.weight, b.bn1.bias, b.conv2.weight, b.bn2.weight, b.bn2.bias]
            if len(b.shortcut):
                self.params += [b.shortcut[0].weight, b.shortcut[1].weight, b.shortcut[1].bias]
        for i, p in enumerate(self.params):
            self.index[id(p)] = i
        self.bns = [resnet.bn1] + [m for b in blocks for m in ([b.bn1, b.bn2] + ([b.shortcut[1]] if len(b.shortcut) else []))]
        self.pool = False                                      # trunk_forward(..., pool=True): the node returns the pooled (N, C) features
        self.persistent_wt = False                             # Trainer: transformed weights live until weights_changed()
        self._wt = None                                        # persistent buffers (fixed addresses: captured graphs read them)
        self._wt_valid = False
        self._uf, self._uf_valid = None, False                 # Winograd-transformed filters of the forward (wino layers)
        self._ub = {}                                          # ... of the data gradient (refreshed with _wt)

    def weights_changed(self):
        self._wt_valid = False
        self._uf_valid = False

    def _versions(self):
        """torch's in-place version counters of the convolution weights: an update made OUTSIDE the Trainer after a forward
        (load_state_dict, a broadcast, an EMA, clipping) bumps them, and the cached transforms are then stale.  (The
        Trainer's raw-pointer Adam kernel does not touch them: it calls weights_changed() instead -- both are checked.)"""
        return tuple(p._version for p in self.params if p.dim() == 4)

    def wino(self, conv, H, W):
        """Winograd F(2x2,3x3) for this layer?  Stride 1, even maps, >= 256 channels: where 16 GEMMs of 8.6 GFLOP plus two
        transform passes over 4x the activation beat the direct kernel's 19.3 GFLOP (t2o_winograd.hip)."""
        w = conv.weight
        return (_WINOGRAD and conv.stride[0] == 1 and H % 2 == 0 and W % 2 == 0 and w.shape[0] >= _WINO_MIN_C and w.shape[1] >= _WINO_MIN_C
                and w.shape[0] <= 1024 and w.shape[1] <= 1024 and (w.shape[0] & (w.shape[0] - 1)) == 0 and (w.shape[1] & (w.shape[1] - 1)) == 0
                and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0)          # (t2o_gemm_tn_batched: 128-wide tiles)

    def fused_wino(self, conv, H, W):
        """The on-chip Winograd kernel for this layer?  Stride 1, <= 128 channels (where the separate-pass pipeline loses: V and
        M are 4x the activation each), maps that are multiples of 16 (8 x 8 tiles per workgroup)."""
        w = conv.weight
        return (_WINO_FUSED and _WINOGRAD and conv.stride[0] == 1 and not self.wino(conv, H, W) and w.shape[0] in (64, 128)
                and w.shape[1] in (64, 128) and H % 16 == 0 and W % 16 == 0)

    def onchip_wgrad(self, conv, H, W):
        """The weight gradient of this layer in the Winograd domain with both transforms on chip (t2o_wino_wgrad.hip)?  Every
        stride-1 layer the on-chip forward kernel runs (64 / 128 channels: fused_wino; 256 channels on maps that are multiples of
        16: wino + the chunk-major filters) -- there the forward then forms no V and the backward no A dY A^T at all."""
        w = conv.weight
        return (_WINO_WGRAD and _WINO_FUSED and _WINOGRAD and conv.stride[0] == 1 and (self.fused_wino(conv, H, W) or self.wino(conv, H, W))
                and bool(_lib.load().t2o_wino_fused_wgrad_supported(1, H, W, w.shape[1], w.shape[0]))
                and bool(_lib.load().t2o_wino_fused_supported(1, H, W, w.shape[1], w.shape[0])))

    # ---- the ONE place a layer's kernel family is chosen (the forward / backward schedules below and bench.py's executed-FLOP
    # accounting both ask here)
    def kernel_for(self, conv, N, Hi, Wi, direction):
        """Kernel family that runs `direction` ('fwd' | 'dgrad' | 'wgrad') of the 3x3 layer `conv` on an (N, Hi, Wi, Ci) input:
        'wino_fused' (t2o_wino_fused.hip, forward / data gradient), 'wino_wgrad' (t2o_wino_wgrad.hip), 'wino_sep' (the separate-pass
        pipeline of t2o_winograd.hip: transforms + 16 GEMMs), 'direct' (t2o_conv.hip, LDS-DMA implicit GEMM), 'generic'
        (t2o_conv_generic.hip, gathered rows)."""
        # (memoised: the schedule asks ~60 times per encoder pass, and every answer costs C-ABI `*_supported` calls; the module
        # switches the tests flip are part of the key)
        key = (id(conv), N, Hi, Wi, direction, _WINOGRAD, _WINO_MIN_C, _WINO_FUSED, _WINO_WGRAD)
        memo = self.__dict__.setdefault('_kernel_memo', {})
        fam = memo.get(key)
        if fam is None:
            fam = memo[key] = self._kernel_for(conv, N, Hi, Wi, direction)
        return fam

    def _kernel_for(self, conv, N, Hi, Wi, direction):
        w = conv.weight
        Co, Ci = w.shape[0], w.shape[1]
        s = conv.stride[0]
        Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
        lib = _lib.load()
        if direction == 'wgrad':
            if self.onchip_wgrad(conv, Hi, Wi):
                return 'wino_wgrad'
            if self.wino(conv, Hi, Wi):
                return 'wino_sep'
            return 'direct' if (Wo % 4 == 0 and (s == 1 or (Hi % 2 == 0 and Wi % 2 == 0))) else 'generic'
        if self.wino(conv, Hi, Wi):
            # (forward: Ci -> Co; data gradient: the same kernel with the channel roles swapped)
            a, b = (Ci, Co) if direction == 'fwd' else (Co, Ci)
            return 'wino_fused' if (_WINO_FUSED and bool(lib.t2o_wino_fused_supported(N, Hi, Wi, a, b))) else 'wino_sep'
        if self.fused_wino(conv, Hi, Wi):
            return 'wino_fused'
        return 'direct' if _fast_direct(s, Hi, Wi, Wo) else 'generic'

    def flop_table(self, N, H, W):
        """One encoder pass (forward + backward) over an (N, 3, H, W) batch, row by row: (layer, direction, kernel family,
        algorithmic FLOP = the direct convolution's 2 * taps * Ci * Co * output pixels, executed FLOP = what the chosen kernel's
        matrix instructions do: 16 of 36 multiplies per 2 x 2 output tile for the Winograd families, over the PADDED tile count
        for the separate-pass GEMMs).  bench.py multiplies by the passes of a train step; tests/test_actor_cpu.py pins the sums
        and tests/test_gpu_encoder.py checks the families against the entry points a real pass calls."""
        lib = _lib.load()
        rows = []
        Hc, Wc = H // 2, W // 2
        C0 = self.net.conv1.weight.shape[0]
        f0 = 2.0 * 27 * C0 * N * Hc * Wc
        for d in ('fwd', 'dgrad', 'wgrad'):
            rows.append(('stem', d, 'stem', f0, f0))
        for i, b in enumerate(self.blocks):
            s = b.conv1.stride[0]
            Hn, Wn = (Hc - 1) // s + 1, (Wc - 1) // s + 1
            for name, conv, hi, wi, ho, wo in (('block%d.conv1' % i, b.conv1, Hc, Wc, Hn, Wn), ('block%d.conv2' % i, b.conv2, Hn, Wn, Hn, Wn)):
                Co, Ci = conv.weight.shape[0], conv.weight.shape[1]
                algo = 2.0 * 9 * Ci * Co * N * ho * wo
                for d in ('fwd', 'dgrad', 'wgrad'):
                    fam = self.kernel_for(conv, N, hi, wi, d)
                    if fam == 'wino_sep':
                        ex = 2.0 * 16 * lib.t2o_wino_padded_tiles(N, hi, wi) * Ci * Co
                    elif fam in ('wino_fused', 'wino_wgrad'):
                        ex = 2.0 * 16 * (N * (hi // 2) * (wi // 2)) * Ci * Co
                    else:
                        ex = algo
                    rows.append((name, d, fam, algo, ex))
            if len(b.shortcut):
                sc = b.shortcut[0]
                f = 2.0 * sc.weight.shape[0] * sc.weight.shape[1] * N * Hn * Wn
                for d in ('fwd', 'dgrad', 'wgrad'):
                    rows.append(('block%d.shortcut' % i, d, 'conv1x1', f, f))
            Hc, Wc = Hn, Wn
        return rows

    def fused_convs(self):
        return [c for b in self.blocks for c in (b.conv1, b.conv2)
                if c.stride[0] == 1 and c.weight.shape[0] in (64, 128, 256, 512) and c.weight.shape[1] in (64, 128, 256, 512)] if (_WINO_FUSED and _WINOGRAD) else []

    def _chunked(self, lib, st, store, convs, src, swap):
        """store[('c', id(conv))] = chunk-major Winograd filters (Ck/8, 16, Cn, 8) of src(conv) (Cn,3,3,Ck) for the on-chip kernel:
        every layer's in ONE launch (t2o_wino_weight_transform_chunked_batch)."""
        ujobs = []
        for conv in convs:
            w = conv.weight
            Cn, Ck = (w.shape[1], w.shape[0]) if swap else (w.shape[0], w.shape[1])
            uc = store.get(('c', id(conv)))
            if uc is None or uc.device != w.device:
                uc = store[('c', id(conv))] = torch.empty((Ck // 8, 16, Cn, 8), dtype=torch.float32, device=w.device)
            ujobs.append((src(conv), uc, Cn, Ck))
        if ujobs:
            _batched(lib.t2o_wino_weight_transform_chunked_batch, 't2o_wino_weight_transform_chunked_batch', st,
                     [j[0] for j in ujobs], [j[1] for j in ujobs], [[j[2] for j in ujobs], [j[3] for j in ujobs]])

    def wino_convs(self):
        return [c for b in self.blocks for c in (b.conv1, b.conv2) if c.stride[0] == 1 and c.weight.shape[0] >= _WINO_MIN_C and c.weight.shape[1] >= _WINO_MIN_C
                and c.weight.shape[0] % 128 == 0 and c.weight.shape[1] % 128 == 0]

    def wino_forward(self, lib, st):
        """{id(conv): U (16,Co,Ci)} for the forward, refreshed once per weight update (persistent_wt) or per call."""
        if self.persistent_wt and self._uf_valid and self.__dict__.get('_uf_ver') == self._versions():
            return self._uf
        uf = self._uf if (self.persistent_wt and self._uf is not None) else {}
        ujobs = []
        for conv in self.wino_convs():
            w = conv.weight
            Co, Ci = w.shape[0], w.shape[1]
            u = uf.get(id(conv))
            if u is None or u.device != w.device:
                u = torch.empty((16, Co, Ci), dtype=torch.float32, device=w.device)
            ujobs.append((w, u, Co, Ci))
            uf[id(conv)] = u
        if ujobs:
            _batched(lib.t2o_wino_weight_transform_batch, 't2o_wino_weight_transform_batch', st,
                     [j[0] for j in ujobs], [j[1] for j in ujobs], [[j[2] for j in ujobs], [j[3] for j in ujobs]])
        self._chunked(lib, st, uf, self.fused_convs(), lambda conv: conv.weight, False)
        if self.persistent_wt:
            self._uf, self._uf_valid, self._uf_ver = uf, True, self._versions()
        return uf

    def supported(self, img):
        """fp32 GPU image (N,3,H,W) with even H and W (the stem's gradient kernels).  Every later stage takes any size:
        the LDS-DMA kernels where the stage width allows (multiples of 8 / 4, even sizes under stride 2 -- every stage
        of a 256 x 256 image), the gathered-row kernels (t2o_conv_generic.hip) elsewhere (e.g. the 4 x 4 stage of a
        128 x 128 image)."""
        net = self.net
        if not (img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3):
            return False
        if img.shape[2] % 2 or img.shape[3] % 2 or net.conv1.weight.shape[0] != 64:
            return False
        if not (img.is_contiguous() or img.is_contiguous(memory_format=torch.channels_last)):
            return False
        for p in self.params:
            if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
                return False
        return all(bn.training and bn.momentum is not None and bn.affine and bn.track_running_stats for bn in self.bns)

    def transformed(self, lib, st):
        """{id(conv): wt}: 3x3 stride-1 -> tap-mirrored transpose, 3x3 stride-2 -> plain transpose per tap, 1x1 -> transpose."""
        if self.persistent_wt and self._wt_valid and self.__dict__.get('_wt_ver') == self._versions():
            return self._wt
        wt = self._wt if (self.persistent_wt and self._wt is not None) else {}
        jobs = []                                              # every layer's transform in ONE launch
        for b in self.blocks:
            convs = [(b.conv1, 9, 1 if b.conv1.stride[0] == 1 else 0), (b.conv2, 9, 1)]
            if len(b.shortcut):
                convs.append((b.shortcut[0], 1, 0))
            for conv, taps, flip in convs:
                w = conv.weight
                Co, Ci = w.shape[0], w.shape[1]
                t = wt.get(id(conv))
                if t is None or t.device != w.device:
                    t = torch.empty(Ci * taps * Co, dtype=torch.float32, device=w.device)
                jobs.append((w, t, Co, Ci, taps, flip))
                wt[id(conv)] = t
        _batched(lib.t2o_conv_weight_transform_batch, 't2o_conv_weight_transform_batch', st,
                 [j[0] for j in jobs], [j[1] for j in jobs], [[j[2] for j in jobs], [j[3] for j in jobs], [j[4] for j in jobs], [j[5] for j in jobs]])
        ub = self._ub if self.persistent_wt else {}
        if _WINOGRAD and self.wino_convs():
            ujobs = []
            for conv in self.wino_convs():                     # data gradient: the same transform of the mirrored transpose
                w = conv.weight
                Co, Ci = w.shape[0], w.shape[1]
                u = ub.get(id(conv))
                if u is None or u.device != w.device:
                    u = torch.empty((16, Ci, Co), dtype=torch.float32, device=w.device)
                ujobs.append((wt[id(conv)], u, Ci, Co))
                ub[id(conv)] = u
            _batched(lib.t2o_wino_weight_transform_batch, 't2o_wino_weight_transform_batch', st,
                     [j[0] for j in ujobs], [j[1] for j in ujobs], [[j[2] for j in ujobs], [j[3] for j in ujobs]])
        self._chunked(lib, st, ub, self.fused_convs(), lambda conv: wt[id(conv)], True)     # data gradient: of the mirrored transpose
        wt['wino'] = ub
        if self.persistent_wt:
            self._wt, self._wt_valid, self._ub, self._wt_ver = wt, True, ub, self._versions()
        return wt


class WgradArena:
    """Weight gradients of the direct (non-Winograd) 3x3 and 1x1 layers ONCE per train step instead of once per encoder pass
    (Trainer opt-in, like the decoder tape).  A train step runs the trunk P = 5 or 6 times; per pass and layer the weight
    gradient kernel writes and re-reads 25-50 MB of split-K partials and launches a reduce.  With every pass's convolution
    input x and output gradient dy stored as slice [p] of per-layer arenas (P*N, H, W, C), the P launches of a layer become
    ONE launch of the unchanged kernel over P*N images after the last backward pass (flush()): one fifth of the partial
    traffic, 4 of 5 reduce launches gone, nothing of it on the backward's dependent chain.  The x arenas replace buffers the
    passes hold anyway; the dy arenas cost P x 0.55 GB at bs = 64, 256 x 256.

    Keys: an activation is named by its producer -- 'a0' (stem output), (i, 'a1') / (i, 'out') of block i; a layer by its
    module id.  begin() starts a train step; the trunk's forward takes pass numbers in call order."""

    def __init__(self, plan, N, H, W, passes, device):
        self.shape, self.P, self.N = (N, H, W), passes, N
        self.x, self.dy, self.layers = {}, {}, []
        self.V, self.Ad, self.wino = {}, {}, []
        Hc, Wc = H // 2, W // 2
        prev = 'a0'
        first_c = plan.net.conv1.weight.shape[0]
        dims = {'a0': (Hc, Wc, first_c)}

        def arena(h, w, c):
            return torch.empty((passes * N, h, w, c), dtype=torch.float32, device=device)
        for i, b in enumerate(plan.blocks):
            s = b.conv1.stride[0]
            Co, Ci = b.conv1.weight.shape[0], b.conv1.weight.shape[1]
            Hn, Wn = (Hc - 1) // s + 1, (Wc - 1) // s + 1
            dims[(i, 'a1')] = dims[(i, 'out')] = (Hn, Wn, Co)
            for conv, xkey, hi, wi in ((b.conv1, prev, Hc, Wc), (b.conv2, (i, 'a1'), Hn, Wn)):
                st = conv.stride[0]
                if plan.wino(conv, hi, wi) and not plan.onchip_wgrad(conv, hi, wi):
                    # Winograd layer: the passes' transformed inputs V and output gradients A dY A^T side by side, one
                    # batch of 16 GEMMs over all their tiles + one back-transform per train step
                    Tpad = _lib.load().t2o_wino_padded_tiles(N, hi, wi)
                    self.wino.append((conv, hi, wi, Tpad))
                    self.V[id(conv)] = torch.empty((16, passes * Tpad, conv.weight.shape[1]), dtype=torch.float32, device=device)
                    self.Ad[id(conv)] = torch.empty((16, passes * Tpad, conv.weight.shape[0]), dtype=torch.float32, device=device)
                    continue
                if not (Wn % 4 == 0 and (st == 1 or (hi % 2 == 0 and wi % 2 == 0))):
                    continue                                   # odd shapes use the gathered-row kernels, pass by pass
                self.layers.append(('3x3', conv, xkey, hi, wi, Hn, Wn))
                self.dy[id(conv)] = arena(Hn, Wn, conv.weight.shape[0])
                if xkey not in self.x:
                    self.x[xkey] = arena(*dims[xkey])
            if len(b.shortcut):
                sc = b.shortcut[0]
                self.layers.append(('1x1', sc, prev, Hc, Wc, Hn, Wn))
                self.dy[id(sc)] = arena(Hn, Wn, Co)
                if prev not in self.x:
                    self.x[prev] = arena(*dims[prev])
            prev, Hc, Wc = (i, 'out'), Hn, Wn
        self.begin()

    def begin(self):
        self.n_passes = 0
        self.done = set()                                      # passes whose backward ran
        self.rec = {}                                          # id(layer) -> passes whose backward left this layer's dy / A dY A^T here

    def take_pass(self):
        if self.n_passes >= self.P:
            return None
        self.n_passes += 1
        return self.n_passes - 1

    def x_slot(self, key, p):
        a = self.x.get(key)
        return None if a is None else a[p * self.N:(p + 1) * self.N]

    def dy_slot(self, conv, p):
        """Pass p's slice of the layer's dy arena for the backward to fill (the layer's weight gradient is then flush()'s), or
        None when the layer is not deferred."""
        a = self.dy.get(id(conv))
        if a is None:
            return None
        self.rec.setdefault(id(conv), set()).add(p)
        return a[p * self.N:(p + 1) * self.N]

    def wino_slot(self, which, conv, p):
        """Pass p's (16, Tpad, C) row range of the layer's V ('V', forward) or A dY A^T ('Ad', backward) arena, or None."""
        a = (self.V if which == 'V' else self.Ad).get(id(conv))
        if a is None:
            return None
        if which == 'Ad':
            self.rec.setdefault(id(conv), set()).add(p)
        Tpad = a.shape[1] // self.P
        return a[:, p * Tpad:(p + 1) * Tpad]

    def flush(self, plan):
        """One weight-gradient launch per layer over every recorded pass, ADDED into the parameters' .grad."""
        if not self.done:
            self.begin()
            return
        lib = _lib.load()
        N = self.N
        def runs_of(passes):
            """Maximal runs of consecutive passes: [first, count]."""
            runs = []
            for p in sorted(passes):
                if runs and runs[-1][0] + runs[-1][1] == p:
                    runs[-1][1] += 1
                else:
                    runs.append([p, 1])
            return runs
        for kind, conv, xkey, Hi, Wi, Hn, Wn in self.layers:
            w = conv.weight
            Co, Ci = w.shape[0], w.shape[1]
            dev = w.device
            st = _stream(dev)
            for first, count in runs_of(self.rec.get(id(conv), ())):      # (a layer the backward did not leave here -- e.g. the plan's
                                                                          # Winograd choice changed since the arena was built -- has none)
                x = self.x[xkey][first * N:(first + count) * N]
                dy = self.dy[id(conv)][first * N:(first + count) * N]
                n = count * N
                if kind == '3x3' and plan.kernel_for(conv, N, Hi, Wi, 'wgrad') == 'wino_wgrad':
                    if not wino_fused_wgrad_nhwc(x, dy, w.grad, n, Hn, Wn, True):     # Winograd domain, both transforms on chip: 16 of 36 multiplies
                        raise RuntimeError('WgradArena: the on-chip Winograd weight gradient refused a layer the plan gave it')
                    continue
                if kind == '3x3':
                    s = conv.stride[0]
                    need = (lib.t2o_conv3x3_wgrad_workspace_bytes if s == 1 else lib.t2o_conv3x3s2_wgrad_workspace_bytes)(n, Hn, Wn, Ci, Co)
                    if need == 0:                              # (more pixels than the kernel's 32-bit indices: pass by pass)
                        for q in range(first, first + count):
                            need1 = (lib.t2o_conv3x3_wgrad_workspace_bytes if s == 1 else lib.t2o_conv3x3s2_wgrad_workspace_bytes)(N, Hn, Wn, Ci, Co)
                            ws = torch.empty(need1, dtype=torch.uint8, device=dev)
                            rc = lib.t2o_conv3x3_wgrad_acc_nhwc(_ptr(self.x[xkey][q * N:(q + 1) * N]), _ptr(self.dy[id(conv)][q * N:(q + 1) * N]),
                                                                _ptr(w.grad), _ptr(ws), need1, N, Hn, Wn, Ci, Co, s, 1, st)
                            _lib.check(rc, 't2o_conv3x3_wgrad_acc_nhwc')
                        continue
                    ws = torch.empty(need, dtype=torch.uint8, device=dev)
                    rc = lib.t2o_conv3x3_wgrad_acc_nhwc(_ptr(x), _ptr(dy), _ptr(w.grad), _ptr(ws), need, n, Hn, Wn, Ci, Co, s, 1, st)
                    _lib.check(rc, 't2o_conv3x3_wgrad_acc_nhwc')
                else:
                    need = lib.t2o_conv1x1s2_wgrad_workspace_bytes(n, Hi, Wi, Ci, Co)
                    ws = torch.empty(need, dtype=torch.uint8, device=dev)
                    rc = lib.t2o_conv1x1s2_wgrad_nhwc(_ptr(x), _ptr(dy), _ptr(w.grad), _ptr(ws), need, n, Hi, Wi, Ci, Co, 1, st)
                    _lib.check(rc, 't2o_conv1x1s2_wgrad_nhwc')
        for conv, Hi, Wi, Tpad in self.wino:
            V, Ad, dw = self.V[id(conv)], self.Ad[id(conv)], conv.weight.grad
            for first, count in runs_of(self.rec.get(id(conv), ())):
                # a run of passes = a row range of every plane: ONE batch of GEMMs over its tiles (a step that made fewer passes
                # than the arena holds -- the episode step in an arena sized for the teacher-forced one -- uses the first rows)
                rows = slice(first * Tpad, (first + count) * Tpad)
                if not wino_dw_from(Ad[:, rows], V[:, rows], dw, True):
                    raise RuntimeError('WgradArena: no split for the Winograd weight gradient of %d tiles' % (count * Tpad))
        self.begin()


def _bn_fwd(lib, st, ws, bn, x, out, res, relu, partial, M, C):
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    if partial is not None:
        rc = lib.t2o_bn_relu_nhwc_fwd_partials(_ptr(x), _ptr(res), _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean),
                                               _ptr(bn.running_var), _ptr(mean), _ptr(invstd), _ptr(out), float(bn.momentum),
                                               float(bn.eps), relu, _ptr(partial), partial.shape[0], _ptr(ws), ws.numel(), M, C, st)
    else:
        rc = lib.t2o_bn_relu_nhwc_fwd(_ptr(x), _ptr(res), _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean),
                                      _ptr(bn.running_var), _ptr(mean), _ptr(invstd), _ptr(out), float(bn.momentum),
                                      float(bn.eps), relu, _ptr(ws), ws.numel(), M, C, st)
    _lib.check(rc, 't2o_bn_relu_nhwc_fwd')
    return mean, invstd


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, img, *params):
        net = plan.net
        _need_gpu(img, *params)
        lib = _lib.load()
        dev = img.device
        st = _stream(dev)
        planar = 1 if img.is_contiguous() else 0
        N, _, H, W = img.shape
        bn_ws = torch.empty(lib.t2o_bn_nhwc_workspace_bytes(1, 512), dtype=torch.uint8, device=dev)
        conv_ws = _conv_workspace(dev, 64 << 10)               # zero region only (registered once per device: not cleared per call)
        saved = []                                             # per layer: what the backward needs
        uf = plan.wino_forward(lib, st) if _WINOGRAD else {}
        kept_v = {}
        # deferred weight gradients (WgradArena, installed by the Trainer): this call's pass number, or None
        arena = plan.__dict__.get('arena')
        into_grad = all(_persistent_grad(p) for p in plan.params)
        apass = arena.take_pass() if (arena is not None and into_grad and arena.shape == (N, H, W) and any(ctx.needs_input_grad)) else None   # (a call under no_grad records nothing)

        def act_like(t, key):
            """Buffer for an activation: its slice of the arena when a deferred weight gradient reads it, else a fresh tensor."""
            slot = arena.x_slot(key, apass) if apass is not None else None
            return slot if slot is not None else torch.empty_like(t)

        def conv3(x, conv, Nn, Hi, Wi, want_stats):
            """3x3, padding 1, stride 1 / 2 on (Nn,Hi,Wi,Ci) -> (y (Nn,Ho,Wo,Co), stats)."""
            w = conv.weight
            Co, Ci = w.shape[0], w.shape[1]
            s = conv.stride[0]
            Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
            if plan.wino(conv, Hi, Wi):
                if plan.kernel_for(conv, Nn, Hi, Wi, 'wgrad') == 'wino_wgrad':     # forward, data and weight gradient all on chip: V is never formed
                    return wino_fused_conv_nhwc(x, uf[('c', id(conv))], Nn, Hi, Wi, None, True)
                keep = []
                y, stats = wino_conv_nhwc(x, uf[id(conv)], Nn, Hi, Wi, None, True, keep_v=keep,
                                          v_out=arena.wino_slot('V', conv, apass) if apass is not None else None,
                                          uc=uf.get(('c', id(conv))))      # (the on-chip kernel where the map is a multiple of 16)
                kept_v[id(conv)] = keep[0]                     # (4x the layer's input: its weight gradient starts from it)
                return y, stats
            fam = plan.kernel_for(conv, Nn, Hi, Wi, 'fwd')
            if fam == 'wino_fused':
                return wino_fused_conv_nhwc(x, uf[('c', id(conv))], Nn, Hi, Wi, None, True)
            y = _nhwc(Nn, Ho, Wo, Co, dev)
            if fam == 'direct':
                stats = torch.empty((lib.t2o_conv3x3_fwd_stats_rows(Nn, Ho, Wo, Co, s), 2, Co), dtype=torch.float32, device=dev)
                rc = lib.t2o_conv3x3_fwd_stats_nhwc(_ptr(x), _ptr(w), _ptr(y), _ptr(stats), _ptr(conv_ws), conv_ws.numel(), Nn, Ho, Wo,
                                                    Ci, Co, s, st)
                _lib.check(rc, 't2o_conv3x3_fwd_stats_nhwc')
                return y, stats
            _lib.check(lib.t2o_conv3x3_any_fwd_nhwc(_ptr(x), _ptr(w), _ptr(y), Nn, Hi, Wi, Ci, Co, s, st), 't2o_conv3x3_any_fwd_nhwc')
            return y, None                                     # (the batch norm makes its own statistics pass)

        # ---- stem: conv 3 -> 64 stride 2 + bn + relu
        Ho, Wo = H // 2, W // 2
        C0 = net.conv1.weight.shape[0]
        y0 = _nhwc(N, Ho, Wo, C0, dev)
        st0 = torch.empty((lib.t2o_stem_fwd_stats_rows(N, Ho, Wo), 2, C0), dtype=torch.float32, device=dev)
        _lib.check(lib.t2o_stem_fwd(_ptr(img), _ptr(net.conv1.weight), _ptr(y0), _ptr(st0), N, Ho, Wo, C0, planar, st), 't2o_stem_fwd')
        a0 = act_like(y0, 'a0')
        m0, i0 = _bn_fwd(lib, st, bn_ws, net.bn1, y0, a0, None, 1, st0, N * Ho * Wo, C0)
        stem = (y0, m0, i0)
        x, Hc, Wc = a0, Ho, Wo
        for bi, b in enumerate(plan.blocks):
            s = b.conv1.stride[0]
            Co = b.conv1.weight.shape[0]
            Hn, Wn = (Hc - 1) // s + 1, (Wc - 1) // s + 1
            M = N * Hn * Wn
            y1, s1 = conv3(x, b.conv1, N, Hc, Wc, True)
            a1 = act_like(y1, (bi, 'a1'))
            m1, i1 = _bn_fwd(lib, st, bn_ws, b.bn1, y1, a1, None, 1, s1, M, Co)
            rec = {'x': x, 'y1': y1, 'm1': m1, 'i1': i1, 'a1': a1, 'H': Hc, 'W': Wc}
            dual = bool(len(b.shortcut)) and _DUAL_BN
            if len(b.shortcut):
                sc_conv, sc_bn = b.shortcut[0], b.shortcut[1]
                ys = _nhwc(N, Hn, Wn, Co, dev)
                rc = lib.t2o_conv1x1s2_fwd_nhwc(_ptr(x), _ptr(sc_conv.weight), _ptr(ys), N, Hc, Wc, sc_conv.weight.shape[1], Co, st)
                _lib.check(rc, 't2o_conv1x1s2_fwd_nhwc')
                if not dual:
                    sc = torch.empty_like(ys)
                    ms, is_ = _bn_fwd(lib, st, bn_ws, sc_bn, ys, sc, None, 0, None, M, Co)
                    rec.update(ys=ys, ms=ms, is_=is_)
            else:
                sc = x
            y2, s2 = conv3(a1, b.conv2, N, Hn, Wn, True)
            out = act_like(y2, (bi, 'out'))
            if dual:
                # out = relu(bn2(y2) + bn_s(ys)) in one pass: the normalised shortcut is never stored
                m2, i2, ms, is_ = (torch.empty(Co, dtype=torch.float32, device=dev) for _ in range(4))
                need = lib.t2o_bn_dual_nhwc_workspace_bytes(M, Co)
                dws = torch.empty(need, dtype=torch.uint8, device=dev)
                rc = lib.t2o_bn_dual_relu_nhwc_fwd(_ptr(y2), _ptr(s2), 0 if s2 is None else s2.shape[0], _ptr(ys),
                                                   _ptr(b.bn2.weight), _ptr(b.bn2.bias), _ptr(b.bn2.running_mean), _ptr(b.bn2.running_var),
                                                   _ptr(m2), _ptr(i2), _ptr(sc_bn.weight), _ptr(sc_bn.bias), _ptr(sc_bn.running_mean),
                                                   _ptr(sc_bn.running_var), _ptr(ms), _ptr(is_), _ptr(out), float(b.bn2.momentum),
                                                   float(b.bn2.eps), float(sc_bn.momentum), float(sc_bn.eps), _ptr(dws), need, M, Co, st)
                _lib.check(rc, 't2o_bn_dual_relu_nhwc_fwd')
                rec.update(ys=ys, ms=ms, is_=is_, dual=True)
            else:
                m2, i2 = _bn_fwd(lib, st, bn_ws, b.bn2, y2, out, sc, 1, s2, M, Co)
            rec.update(y2=y2, m2=m2, i2=i2, out=out)
            saved.append(rec)
            x, Hc, Wc = out, Hn, Wn
        ctx.plan, ctx.img, ctx.stem, ctx.saved, ctx.planar = plan, img, stem, saved, planar
        ctx.a0 = a0
        ctx.kept_v = kept_v
        ctx.arena, ctx.apass = (arena, apass) if apass is not None else (None, None)
        # persistent, dense gradient buffers registered for every parameter (functional.enable_grad_accumulation -- the
        # Trainer's flat buffer): the kernels accumulate into them.  Checked again in the backward.
        ctx.into_grad = into_grad
        ctx.grads = [p.grad for p in plan.params] if ctx.into_grad else None
        ctx.pooled = plan.pool
        if plan.pool:                                          # global average pool inside the node (models/actor_resnet.py:106):
            ctx.hw = (x.shape[1], x.shape[2])                  # its backward then builds the NHWC gradient directly instead of
            return x.mean((1, 2))                              # autograd's NCHW expand + a permuting copy (20 -> 7 us per pass)
        return x.permute(0, 3, 1, 2)                          # (N,C,h,w) view of the NHWC buffer = channels_last

    @staticmethod
    def backward(ctx, dout):
        plan, net = ctx.plan, ctx.plan.net
        lib = _lib.load()
        dev = dout.device
        st = _stream(dev)
        acc = 1 if (ctx.into_grad and all(p.grad is g_ for p, g_ in zip(plan.params, ctx.grads))) else 0
        N = ctx.img.shape[0]
        grads = ctx.grads if acc else [torch.empty_like(p) for p in plan.params]

        def g(p):
            return grads[plan.index[id(p)]]

        bn_ws = torch.empty(lib.t2o_bn_nhwc_workspace_bytes(1, 512), dtype=torch.uint8, device=dev)
        conv_ws = _conv_workspace(dev, 64 << 10)
        wt = plan.transformed(lib, st)

        # deferred weight gradients: only while the gradients really are accumulated in place (acc)
        arena, apass = (ctx.arena, ctx.apass) if acc else (None, None)

        def deferred(conv):
            """This layer's dy slot in the arena (its weight gradient is then formed by arena.flush()), or None."""
            return arena.dy_slot(conv, apass) if arena is not None else None

        def bn_bwd(bn, x, y, dy, mean, invstd, has_res, relu, want_dres, M, C, dx=None):
            dx = torch.empty_like(x) if dx is None else dx
            dres = torch.empty_like(x) if want_dres else None
            rc = lib.t2o_bn_relu_nhwc_bwd_acc(_ptr(x), _ptr(y), _ptr(dy), _ptr(bn.weight), _ptr(bn.bias), _ptr(mean), _ptr(invstd),
                                              _ptr(dx), _ptr(dres), _ptr(g(bn.weight)), _ptr(g(bn.bias)), has_res, relu, acc,
                                              _ptr(bn_ws), bn_ws.numel(), M, C, st)
            _lib.check(rc, 't2o_bn_relu_nhwc_bwd_acc')
            return dx, dres

        def wgrad3(conv, x, dy, Hi, Wi, Hn, Wn):
            """(Hi, Wi): the convolution's input grid, (Hn, Wn): its output grid."""
            w = conv.weight
            Co, Ci = w.shape[0], w.shape[1]
            s = conv.stride[0]
            fam = plan.kernel_for(conv, N, Hi, Wi, 'wgrad')
            if fam == 'wino_wgrad':
                if not wino_fused_wgrad_nhwc(x, dy, g(w), N, Hi, Wi, bool(acc)):
                    raise RuntimeError('trunk: the on-chip Winograd weight gradient refused a layer the plan gave it')
                return
            if fam == 'wino_sep':
                V = ctx.kept_v.pop(id(conv), None)
                wino_wgrad_nhwc(V if V is not None else wino_input(x, N, Hi, Wi), dy, g(w), N, Hi, Wi, acc)
                return
            if fam == 'direct':
                need = (lib.t2o_conv3x3_wgrad_workspace_bytes if s == 1 else lib.t2o_conv3x3s2_wgrad_workspace_bytes)(N, Hn, Wn, Ci, Co)
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                rc = lib.t2o_conv3x3_wgrad_acc_nhwc(_ptr(x), _ptr(dy), _ptr(g(w)), _ptr(ws), need, N, Hn, Wn, Ci, Co, s, acc, st)
                _lib.check(rc, 't2o_conv3x3_wgrad_acc_nhwc')
                return
            need = lib.t2o_conv3x3_any_wgrad_workspace_bytes(N, Hi, Wi, Ci, Co, s)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            rc = lib.t2o_conv3x3_any_wgrad_nhwc(_ptr(x), _ptr(dy), _ptr(g(w)), _ptr(ws), need, N, Hi, Wi, Ci, Co, s, acc, st)
            _lib.check(rc, 't2o_conv3x3_any_wgrad_nhwc')

        def wino_bwd(conv, x, dy, dx, addend, Hi, Wi):
            """Both gradients of a Winograd layer: ONE transform pass over dy feeds the data gradient's and the weight gradient's GEMMs."""
            V = ctx.kept_v.pop(id(conv), None)
            if V is None:
                V = wino_input(x, N, Hi, Wi)
            ad_out = arena.wino_slot('Ad', conv, apass) if arena is not None else None
            wino_backward_nhwc(dy, V, wt['wino'][id(conv)], g(conv.weight), dx, N, Hi, Wi, addend, acc, ad_out,
                               uc=wt['wino'].get(('c', id(conv))))

        def dgrad3(conv, dy, dx, addend, Hi, Wi, Hn, Wn):
            """dx (N,Hi,Wi,Ci) = data gradient of conv for dy (N,Hn,Wn,Co) (+ addend, stride 1 only)."""
            w = conv.weight
            Co, Ci = w.shape[0], w.shape[1]
            s = conv.stride[0]
            fam = plan.kernel_for(conv, N, Hi, Wi, 'dgrad')
            if fam == 'wino_fused':
                wino_fused_conv_nhwc(dy, wt['wino'][('c', id(conv))], N, Hi, Wi, addend, False, out=dx)
                return
            if fam == 'wino_sep':
                wino_conv_nhwc(dy, wt['wino'][id(conv)], N, Hi, Wi, addend, False, out=dx)
                return
            if fam == 'direct':
                if s == 1:
                    rc = lib.t2o_conv3x3_dgrad_pre_nhwc(_ptr(dy), _ptr(wt[id(conv)]), _ptr(addend), _ptr(dx), _ptr(conv_ws), conv_ws.numel(),
                                                        N, Hi, Wi, Ci, Co, st)
                else:
                    rc = lib.t2o_conv3x3s2_dgrad_pre_nhwc(_ptr(dy), _ptr(wt[id(conv)]), _ptr(dx), _ptr(conv_ws), conv_ws.numel(),
                                                          N, Hn, Wn, Ci, Co, st)
                _lib.check(rc, 't2o_conv3x3_dgrad_pre_nhwc')
            else:
                rc = lib.t2o_conv3x3_any_dgrad_nhwc(_ptr(dy), _ptr(wt[id(conv)]), _ptr(addend), _ptr(dx), N, Hi, Wi, Ci, Co, s, st)
                _lib.check(rc, 't2o_conv3x3_any_dgrad_nhwc')

        if ctx.pooled:
            h, w = ctx.hw
            d = (dout * (1.0 / (h * w))).view(N, 1, 1, -1).expand(N, h, w, dout.shape[1]).contiguous()
        else:
            d = dout.permute(0, 2, 3, 1).contiguous()          # NHWC (a no-op for a channels_last gradient)
        for b, rec in zip(reversed(plan.blocks), reversed(ctx.saved)):
            s = b.conv1.stride[0]
            Co, Ci = b.conv1.weight.shape[0], b.conv1.weight.shape[1]
            Hc, Wc = rec['H'], rec['W']
            Hn, Wn = (Hc - 1) // s + 1, (Wc - 1) // s + 1
            M = N * Hn * Wn
            # out = relu(bn2(y2) + sc)
            slot2 = deferred(b.conv2)
            dys = None
            if rec.get('dual'):
                # both batch norms of the shortcut block from one pass over the gated gradient (never stored)
                sc_conv, sc_bn = b.shortcut[0], b.shortcut[1]
                slots = deferred(sc_conv)
                dy2 = slot2 if slot2 is not None else torch.empty_like(rec['y2'])
                dys = slots if slots is not None else torch.empty_like(rec['ys'])
                need = lib.t2o_bn_dual_nhwc_workspace_bytes(M, Co)
                dws = torch.empty(need, dtype=torch.uint8, device=dev)
                rc = lib.t2o_bn_dual_relu_nhwc_bwd_acc(_ptr(rec['y2']), _ptr(rec['ys']), _ptr(rec['out']), _ptr(d), _ptr(b.bn2.weight),
                                                       _ptr(b.bn2.bias), _ptr(rec['m2']), _ptr(rec['i2']), _ptr(sc_bn.weight), _ptr(sc_bn.bias),
                                                       _ptr(rec['ms']), _ptr(rec['is_']), _ptr(dy2), _ptr(dys), _ptr(g(b.bn2.weight)),
                                                       _ptr(g(b.bn2.bias)), _ptr(g(sc_bn.weight)), _ptr(g(sc_bn.bias)), acc, _ptr(dws), need,
                                                       M, Co, st)
                _lib.check(rc, 't2o_bn_dual_relu_nhwc_bwd_acc')
                dsc = None
            else:
                dy2, dsc = bn_bwd(b.bn2, rec['y2'], rec['out'], d, rec['m2'], rec['i2'], 1, 1, True, M, Co, slot2)
            da1 = torch.empty_like(rec['a1'])
            rows1 = None                                   # bn1's backward sums, when conv2's data gradient leaves them
            if plan.wino(b.conv2, Hn, Wn) and plan.kernel_for(b.conv2, N, Hn, Wn, 'wgrad') != 'wino_wgrad':
                wino_bwd(b.conv2, rec['a1'], dy2, da1, None, Hn, Wn)
            else:
                C2o, C2i = b.conv2.weight.shape[0], b.conv2.weight.shape[1]
                fam2 = plan.kernel_for(b.conv2, N, Hn, Wn, 'dgrad')
                fused2 = fam2 == 'wino_fused'
                n_rows = lib.t2o_conv3x3_dgrad_bnsums_rows(N, Hn, Wn, C2i, C2o) if (_BN_SUMS_EPILOGUE and fam2 == 'direct') else 0
                if fused2 and _BN_SUMS_EPILOGUE:
                    # the on-chip Winograd data gradient with bn1's backward sums in its epilogue
                    rows1 = torch.empty(lib.t2o_wino_fused_stats_rows(N, Hn, Wn) * 2 * C2i, dtype=torch.float32, device=dev)
                    rc = lib.t2o_wino_fused_conv_bnsums_nhwc(_ptr(dy2), _ptr(wt['wino'][('c', id(b.conv2))]), _ptr(da1), _ptr(rec['y1']),
                                                             _ptr(rec['m1']), _ptr(rec['i1']), _ptr(b.bn1.weight), _ptr(b.bn1.bias), _ptr(rows1),
                                                             _ptr(_zero_block(dev)), N, Hn, Wn, C2o, C2i, st)
                    _lib.check(rc, 't2o_wino_fused_conv_bnsums_nhwc')
                elif n_rows > 0:
                    # a1 = relu(bn1(y1)) is this data gradient's only consumer: its epilogue forms bn1's backward sums
                    rows1 = torch.empty(n_rows * 2 * C2i, dtype=torch.float32, device=dev)
                    rc = lib.t2o_conv3x3_dgrad_pre_bnsums_nhwc(_ptr(dy2), _ptr(wt[id(b.conv2)]), _ptr(da1), _ptr(rec['y1']), _ptr(rec['m1']),
                                                               _ptr(rec['i1']), _ptr(b.bn1.weight), _ptr(b.bn1.bias), _ptr(rows1),
                                                               _ptr(conv_ws), conv_ws.numel(), N, Hn, Wn, C2i, C2o, st)
                    _lib.check(rc, 't2o_conv3x3_dgrad_pre_bnsums_nhwc')
                else:
                    dgrad3(b.conv2, dy2, da1, None, Hn, Wn, Hn, Wn)
                if slot2 is None:
                    wgrad3(b.conv2, rec['a1'], dy2, Hn, Wn, Hn, Wn)
            del dy2
            # a1 = relu(bn1(y1))
            slot1 = deferred(b.conv1)
            if rows1 is not None:
                dy1 = slot1 if slot1 is not None else torch.empty_like(rec['y1'])
                rc = lib.t2o_bn_relu_nhwc_bwd_partials_acc(_ptr(rec['y1']), _ptr(da1), _ptr(b.bn1.weight), _ptr(b.bn1.bias), _ptr(rec['m1']),
                                                           _ptr(rec['i1']), _ptr(dy1), _ptr(g(b.bn1.weight)), _ptr(g(b.bn1.bias)), 1, acc,
                                                           _ptr(rows1), rows1.numel() // (2 * Co), _ptr(bn_ws), bn_ws.numel(), M, Co, st)
                _lib.check(rc, 't2o_bn_relu_nhwc_bwd_partials_acc')
            else:
                dy1, _ = bn_bwd(b.bn1, rec['y1'], None, da1, rec['m1'], rec['i1'], 0, 1, False, M, Co, slot1)
            del da1
            dx = torch.empty_like(rec['x'])
            if plan.wino(b.conv1, Hc, Wc) and not len(b.shortcut) and plan.kernel_for(b.conv1, N, Hc, Wc, 'wgrad') != 'wino_wgrad':
                wino_bwd(b.conv1, rec['x'], dy1, dx, dsc, Hc, Wc)
                d = dx
                continue
            if slot1 is None:
                wgrad3(b.conv1, rec['x'], dy1, Hc, Wc, Hn, Wn)
            if len(b.shortcut):
                sc_conv, sc_bn = b.shortcut[0], b.shortcut[1]
                if dys is None:
                    slots = deferred(sc_conv)
                    dys, _ = bn_bwd(sc_bn, rec['ys'], None, dsc, rec['ms'], rec['is_'], 0, 0, False, M, Co, slots)
                if slots is None:
                    need = lib.t2o_conv1x1s2_wgrad_workspace_bytes(N, Hc, Wc, Ci, Co)
                    ws = torch.empty(need, dtype=torch.uint8, device=dev)
                    rc = lib.t2o_conv1x1s2_wgrad_nhwc(_ptr(rec['x']), _ptr(dys), _ptr(g(sc_conv.weight)), _ptr(ws), need, N, Hc, Wc, Ci, Co,
                                                      acc, st)
                    _lib.check(rc, 't2o_conv1x1s2_wgrad_nhwc')
                dgrad3(b.conv1, dy1, dx, None, Hc, Wc, Hn, Wn)
                rc = lib.t2o_conv1x1s2_dgrad_acc_nhwc(_ptr(dys), _ptr(wt[id(sc_conv)]), _ptr(dx), N, Hc, Wc, Ci, Co, st)
                _lib.check(rc, 't2o_conv1x1s2_dgrad_acc_nhwc')
            else:
                dgrad3(b.conv1, dy1, dx, dsc, Hc, Wc, Hn, Wn)
            d = dx
        # ---- stem
        y0, m0, i0 = ctx.stem
        _, Ho, Wo, C0 = y0.shape
        dy0, _ = bn_bwd(net.bn1, y0, None, d, m0, i0, 0, 1, False, N * Ho * Wo, C0)
        need = lib.t2o_stem_wgrad_workspace_bytes(N, Ho, Wo, C0)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        rc = lib.t2o_stem_wgrad(_ptr(ctx.img), _ptr(dy0), _ptr(g(net.conv1.weight)), _ptr(ws), need, N, Ho, Wo, C0, ctx.planar, acc, st)
        _lib.check(rc, 't2o_stem_wgrad')
        dimg = None
        if ctx.needs_input_grad[1]:
            dimg = torch.empty_like(ctx.img)                   # same layout as the image (planar NCHW or channels-last)
            rc = lib.t2o_stem_dgrad(_ptr(dy0), _ptr(net.conv1.weight), _ptr(dimg), N, Ho, Wo, C0, ctx.planar, 0, st)
            _lib.check(rc, 't2o_stem_dgrad')
        if arena is not None:
            arena.done.add(apass)
        return (None, dimg) + (tuple(None for _ in plan.params) if acc else tuple(grads))


def trunk_forward(plan, img, pool=False):
    """relu(bn(conv...)) trunk output (N,512,H/32,W/32), channels_last, for a supported training-mode call."""
    plan.pool = bool(pool)                                 # (read by the node's forward only)
    return _TrunkFn.apply(plan, img, *plan.params)
