"""Image encoder of the actor (models/actor_resnet.py): ResNet-18 layout with a 3x3 stride-2
stem, no max-pool, every stage stride 2 (/32), global mean, fc.  The only dense contraction of
the hot path.  Three ways through it, same arithmetic:
  * training mode, channels-last weights, a GPU image whose stages the matrix-core kernels take (W % 256 == 0,
    H % 32 == 0): the whole convolutional trunk is ONE autograd node with an explicit forward / backward schedule over
    this library's kernels (encoder.py: every convolution incl. the 1x1 shortcuts, fused batch norm + add + ReLU,
    in-kernel gradient accumulation) -- no library convolution, no autograd between the layers;
  * other training-mode GPU calls: per-layer autograd functions (functional.py) -- own kernels where the shape allows,
    library (MIOpen) convolutions elsewhere, fused batch-norm kernels;
  * evaluation mode / CPU tensors: plain PyTorch modules.
Module names follow the reference so its checkpoints load."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as T


# Module-level switches for the TESTS, which flip them with monkeypatch to compare the paths with one another (there is no
# environment knob: the product has ONE path per shape -- a framework-convolution A/B is something tests / tools build themselves).
_FUSED = True                  # fused training-mode batch norm (+ add + ReLU) kernels (False: PyTorch's batch norm; tests only)
_OWN_WGRAD = True              # the hand-written convolution kernels (t2o_conv*.hip), all three directions
_CONV_STATS = True             # the forward convolution leaves the batch-norm statistics of its output (from its accumulators)
_TRUNK = True                  # the one-node trunk (encoder.py); False: the per-layer path everywhere (what the tests compare it with)


def _conv(conv, x, bn=None):
    """conv(x).  bn: the BatchNorm2d applied to the result next; returns (y, stats) then, stats = the partial sums
    for T.batch_norm_relu(..., partial=stats) where the own forward kernel ran and bn will take the fused path,
    else None."""
    want = bn is not None and _CONV_STATS and _FUSED and bn.training
    y = stats = None
    if _OWN_WGRAD and conv.bias is None and T.conv3x3_supported(x, conv.weight, conv.stride, conv.padding) \
            and conv.weight.is_contiguous(memory_format=torch.channels_last):
        y = T.conv3x3(x, conv.weight, True) if want else T.conv3x3(x, conv.weight)
    elif _OWN_WGRAD and conv.bias is None and T.conv3x3s2_supported(x, conv.weight, conv.stride, conv.padding) \
            and conv.weight.is_contiguous(memory_format=torch.channels_last):
        y = T.conv3x3s2(x, conv.weight, True) if want else T.conv3x3s2(x, conv.weight)
    else:
        y, want = conv(x), False
    if want:
        y, stats = y
    return (y, stats) if bn is not None else y


def _bn_relu(bn, x, residual=None, counted=False, partial=None):
    """relu(bn(x) (+ residual)): fused kernels in training mode on the GPU.  counted: num_batches_tracked of the
    fused layers was already advanced for this forward (ResNet.forward, one launch for all of them).  partial: the
    producing convolution's statistics (see _conv)."""
    if _FUSED and bn.training and x.is_cuda:
        return T.batch_norm_relu(x, bn, residual, count=not counted, partial=partial if T._is_nhwc(x) else None)
    if counted and bn.training and bn.track_running_stats:
        bn.num_batches_tracked.sub_(1)                       # ResNet.forward advanced it already; bn(x) does so again
    out = bn(x)
    return F.relu(out if residual is None else out + residual)


def _bn_plain(bn, x, counted=False):
    """bn(x) of the shortcut branch: the fused kernels without the activation on channels-last activations."""
    if _FUSED and bn.training and x.is_cuda and T._is_nhwc(x):
        return T.batch_norm_relu(x, bn, None, relu=False, count=not counted)
    if counted and bn.training and bn.track_running_stats:
        bn.num_batches_tracked.sub_(1)                       # ResNet.forward advanced it already; bn(x) does so again
    return bn(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x, counted=False):
        y, st = _conv(self.conv1, x, self.bn1)
        out = _bn_relu(self.bn1, y, None, counted, st)
        if len(self.shortcut):
            sconv = self.shortcut[0]
            ys = T.conv1x1s2(x, sconv.weight) if (_OWN_WGRAD and T.conv1x1s2_supported(x, sconv)) else sconv(x)
            sc = _bn_plain(self.shortcut[1], ys, counted)
        else:
            sc = x
        y, st = _conv(self.conv2, out, self.bn2)
        return _bn_relu(self.bn2, y, sc, counted, st)


class ResNet(nn.Module):
    def __init__(self, num_inputs=3, depth=18, num_outputs=512):
        super().__init__()
        if depth != 18:
            raise NotImplementedError('the actor uses depth 18 (models/actor.py:74)')
        self.in_planes = 64
        self.conv1 = nn.Conv2d(num_inputs, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(64, 2, 2)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        self.layer4 = self._make_layer(512, 2, 2)
        self.fc = nn.Linear(512, num_outputs)

    def _batch_counters(self):
        """num_batches_tracked of the batch norms that are in training mode (a layer frozen with bn.eval() keeps its count)."""
        bns = self.__dict__.get('_bns')
        if bns is None:                                      # (the module tree is fixed after construction: walked once, not per call)
            bns = self.__dict__['_bns'] = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        cached = self.__dict__.get('_nbt')
        key = tuple(m.training for m in bns)
        if cached is None or cached[0] is not self.bn1.num_batches_tracked or cached[1] != key:      # (.to() replaces the buffer tensors)
            cached = self.__dict__['_nbt'] = (self.bn1.num_batches_tracked, key,
                                              [m.num_batches_tracked for m in bns if m.training and m.track_running_stats])
        return cached[2]

    def _make_layer(self, planes, num_blocks, stride):
        blocks = []
        for s in [stride] + [1] * (num_blocks - 1):
            blocks.append(BasicBlock(self.in_planes, planes, s))
            self.in_planes = planes
        return nn.Sequential(*blocks)

    def trunk_plan(self):
        """The explicit-schedule trunk (encoder.TrunkPlan), built once (not a submodule: state_dict unchanged)."""
        plan = self.__dict__.get('_trunk_plan')
        if plan is None or plan.params[0] is not self.conv1.weight:       # (.to() / a re-homed parameter: rebuild)
            from .encoder import TrunkPlan
            plan = self.__dict__['_trunk_plan'] = TrunkPlan(self)
        return plan

    def forward(self, x):
        pooled = self.pooled_features(x)
        if pooled.is_cuda and pooled.dtype == torch.float32 and torch.is_grad_enabled():
            return T.linear_acc(pooled, self.fc.weight, self.fc.bias)
        return self.fc(pooled)

    def pooled_features(self, x):
        """Everything before `fc` (models/actor_resnet.py:98-106): the convolutional trunk and the global mean, (N, 512)."""
        if x.is_cuda and x.dim() == 4 and not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            # a batch-strided view (the teacher-forced step feeds img_y[:, i]: actor.py:172): one 50 MB copy instead of the
            # per-layer path the strided image would otherwise take (no Winograd, no once-per-step weight gradients)
            x = x.contiguous()
        if _TRUNK and _FUSED and _OWN_WGRAD and self.training and x.is_cuda and torch.is_grad_enabled():
            plan = self.trunk_plan()
            if plan.supported(x):
                from .encoder import trunk_forward
                torch._foreach_add_(self._batch_counters(), 1)
                x = trunk_forward(plan, x, pool=True)
                return x.view(x.size(0), -1)
        if self.conv1.weight.is_contiguous(memory_format=torch.channels_last) and x.is_cuda:
            # channels-last encoder (Actor.use_channels_last): one packed NHWC copy of the 3-channel image, then every
            # convolution and every fused batch-norm pass runs NHWC -- no layout transposes inside the encoder
            x = x.contiguous(memory_format=torch.channels_last)
        counted = False
        if _FUSED and self.training and x.is_cuda and self.conv1.weight.is_contiguous(memory_format=torch.channels_last):
            # num_batches_tracked of all 21 batch norms: one multi-tensor launch instead of 21 one-element kernels
            torch._foreach_add_(self._batch_counters(), 1)
            counted = True
        y, st = _conv(self.conv1, x, self.bn1)
        x = _bn_relu(self.bn1, y, None, counted, st)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for block in layer:
                x = block(x, counted)
        x = x.mean((2, 3))
        return x.view(x.size(0), -1)
