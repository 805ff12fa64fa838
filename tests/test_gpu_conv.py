"""Hand-written fp32 MFMA kernels of the encoder's 3x3 stride-1 convolutions (t2o_conv.hip: forward, data gradient,
weight gradient) against F.conv2d and its gradients in fp64 (models/actor_resnet.py:27-44)."""
import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


def _ref_wgrad(x, dy, co, ci):
    x64 = x.double()
    w = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x64, w, None, 1, 1)
    (y * dy.double()).sum().backward()
    return w.grad


# (N, Ci, Co, H, W), W a multiple of 4: every tile template, ragged pixel counts (stage tail), one-row images, rows
# shorter / longer than a 32-pixel stage and not dividing it, deep split-K
SHAPES = [(2, 64, 64, 8, 8), (3, 128, 128, 5, 12), (2, 64, 128, 6, 4), (2, 128, 64, 9, 4), (1, 256, 192, 4, 4),
          (4, 64, 64, 1, 8), (2, 64, 64, 7, 20), (2, 128, 256, 16, 16), (3, 64, 64, 33, 20), (1, 128, 128, 3, 40),
          (2, 64, 64, 5, 64)]


@pytest.mark.parametrize('shape', SHAPES)
def test_wgrad_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 701, -1.0, 1.0)
    dy = synth.uniform((N, Co, H, W), 702, -1.0, 1.0)
    ref = _ref_wgrad(x, dy, Co, Ci)
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    dg = dy.to(dev).contiguous(memory_format=torch.channels_last)
    dw = T.conv3x3_wgrad(xg, dg)
    assert dw.shape == (Co, Ci, 3, 3) and dw.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)
    # deterministic: split-K partials are added in a fixed order
    assert torch.equal(dw, T.conv3x3_wgrad(xg, dg))


# (N, Ci, Co, H, W), W a multiple of 8: one and several channel tiles, ragged pixel counts (tile tail), one-row images,
# tiles that start in the middle of an image row, rows longer than a tile, Ci = 32
DIRECT_SHAPES = [(2, 64, 64, 8, 8), (3, 128, 64, 5, 16), (1, 64, 128, 3, 24), (5, 32, 64, 1, 8), (2, 64, 192, 40, 8),
                 (1, 128, 128, 2, 264), (3, 64, 64, 9, 32), (2, 256, 64, 12, 16)]


@pytest.mark.parametrize('shape', DIRECT_SHAPES)
def test_forward_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 721, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 722, -1.0, 1.0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    dev = torch.device('cuda:0')
    y = T.conv3x3_forward(x.to(dev).contiguous(memory_format=torch.channels_last),
                          w.to(dev).contiguous(memory_format=torch.channels_last))
    assert y.shape == (N, Co, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', DIRECT_SHAPES)
def test_dgrad_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Co, Ci, H, W = shape                     # (the data gradient wants Co % 32 == 0 and Ci % 64 == 0)
    dy = synth.uniform((N, Co, H, W), 731, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 732, -1.0, 1.0)
    x64 = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x64, w.double(), None, 1, 1).backward(dy.double())
    ref = x64.grad
    dev = torch.device('cuda:0')
    dx = T.conv3x3_dgrad(dy.to(dev).contiguous(memory_format=torch.channels_last),
                         w.to(dev).contiguous(memory_format=torch.channels_last))
    assert dx.shape == (N, Ci, H, W) and dx.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dx.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('layer', [(64, 64), (128, 32), (256, 16), (512, 8)])
def test_encoder_layer_shapes_at_batch_64_match_the_library(layer):
    """BASELINE.json's sizes (bs=64, 256x256 input: the four stages of the encoder) are too large for an fp64 CPU
    reference in a test; at these sizes the three kernels are held against the library's fp32 convolution (itself
    1e-6 from fp64 on the small shapes above) and against two size-independent properties: linearity in the
    weight, and the adjoint identity <conv(x, w), dy> == <x, dgrad(dy, w)> == <w, wgrad(x, dy)>."""
    import t2onet_amd.functional as T
    c, h = layer
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(741 + c)
    x = (torch.rand(64, c, h, h, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    w = ((torch.rand(c, c, 3, 3, generator=g) * 2 - 1) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    dy = (torch.rand(64, c, h, h, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    y = T.conv3x3_forward(x, w)
    dx = T.conv3x3_dgrad(dy, w)
    dw = T.conv3x3_wgrad(x, dy)
    lib = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
    y_lib = torch.nn.functional.conv2d(x, w, None, 1, 1)
    for own, ref in ((y, y_lib), (dx, lib[0]), (dw, lib[1])):
        assert float((own - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # linearity in the weight
    y2 = T.conv3x3_forward(x, (w * 2.0).contiguous(memory_format=torch.channels_last))
    assert float((y2 - 2.0 * y).abs().max()) <= 1e-6 * float(y.abs().max())
    # adjoint identities, sums in fp64
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    d = float((w.double() * dw.double()).sum())
    scale = float((y.double().abs() * dy.double().abs()).sum())
    assert abs(a - b) <= 1e-6 * scale and abs(a - d) <= 1e-6 * scale
    # the weight gradient is deterministic
    assert torch.equal(dw, T.conv3x3_wgrad(x, dy))


@pytest.mark.parametrize('layer', [(64, 64), (128, 32), (256, 16), (512, 8)])
def test_encoder_layer_shapes_at_batch_64_vs_fp64_samples(layer):
    """The same bs=64 launches against fp64 on the CPU, where that costs seconds: forward and data gradient are
    per-sample independent -- samples 0 and 63 of the batch (first / last tiles, a tile that starts mid-image) against
    conv2d in fp64; the weight gradient of the 512-channel 8 x 8 stage (a full reduction over the batch) entirely."""
    import t2onet_amd.functional as T
    c, h = layer
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(841 + c)
    x = torch.rand(64, c, h, h, generator=g) * 2 - 1
    w = (torch.rand(c, c, 3, 3, generator=g) * 2 - 1) * 0.05
    dy = torch.rand(64, c, h, h, generator=g) * 2 - 1
    xg, wg, dg = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in (x, w, dy))
    y = T.conv3x3_forward(xg, wg)
    dx = T.conv3x3_dgrad(dg, wg)
    for n in (0, 63):
        x64 = x[n:n + 1].double().requires_grad_(True)
        ref = torch.nn.functional.conv2d(x64, w.double(), None, 1, 1)
        ref.backward(dy[n:n + 1].double())
        scale = float(ref.abs().max())
        np.testing.assert_allclose(y[n:n + 1].cpu().numpy(), ref.detach().float().numpy(), rtol=1e-5, atol=1e-5 * scale)
        gscale = float(x64.grad.abs().max())
        np.testing.assert_allclose(dx[n:n + 1].cpu().numpy(), x64.grad.float().numpy(), rtol=1e-5, atol=1e-5 * gscale)
    if c == 512:
        dw = T.conv3x3_wgrad(xg, dg)
        ref = _ref_wgrad(x, dy, c, c)
        np.testing.assert_allclose(dw.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


# (N, Ci, Co, Ho, Wo) of the stride-2 layers: dx is (N, Ci, 2Ho, 2Wo)
# (the data gradient has two tilings: 8 waves x 64 input channels, and -- when those tiles number <= 128 -- 4 waves x 32
# channels; (40, 64, 64, 16, 32) and (12, 128, 64, 16, 32) take the first, the rest the second)
S2_SHAPES = [(2, 64, 64, 8, 8), (3, 64, 128, 5, 16), (1, 128, 64, 3, 24), (2, 64, 32, 1, 8), (2, 128, 256, 20, 8), (1, 64, 64, 2, 136),
             (40, 64, 64, 16, 32), (12, 128, 64, 16, 32), (8, 256, 512, 8, 8),
             (2, 3, 64, 16, 16), (3, 3, 64, 9, 37), (1, 3, 32, 1, 1), (2, 3, 64, 8, 64)]      # (the last four: the 3-channel stem)


@pytest.mark.parametrize('shape', S2_SHAPES)
def test_stride2_dgrad_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, Ho, Wo = shape
    dy = synth.uniform((N, Co, Ho, Wo), 751, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 752, -1.0, 1.0)
    x64 = torch.zeros(N, Ci, 2 * Ho, 2 * Wo, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x64, w.double(), None, 2, 1).backward(dy.double())
    ref = x64.grad
    dev = torch.device('cuda:0')
    dx = T.conv3x3s2_dgrad(dy.to(dev).contiguous(memory_format=torch.channels_last),
                           w.to(dev).contiguous(memory_format=torch.channels_last))
    assert dx.shape == (N, Ci, 2 * Ho, 2 * Wo) and dx.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dx.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', [(2, 64, 64, 8, 8), (3, 64, 128, 5, 16), (1, 128, 64, 3, 24), (2, 32, 64, 1, 8), (2, 128, 256, 20, 8), (1, 64, 64, 2, 136)])
def test_stride2_forward_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, Ho, Wo = shape
    x = synth.uniform((N, Ci, 2 * Ho, 2 * Wo), 781, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 782, -1.0, 1.0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    dev = torch.device('cuda:0')
    y = T.conv3x3s2_forward(x.to(dev).contiguous(memory_format=torch.channels_last),
                            w.to(dev).contiguous(memory_format=torch.channels_last))
    assert y.shape == (N, Co, Ho, Wo) and y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', [(2, 64, 64, 8, 8), (3, 64, 128, 5, 12), (1, 128, 64, 3, 4), (2, 64, 64, 33, 20), (1, 128, 256, 2, 40), (2, 64, 64, 5, 64)])
def test_stride2_wgrad_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, Ho, Wo = shape
    x = synth.uniform((N, Ci, 2 * Ho, 2 * Wo), 771, -1.0, 1.0)
    dy = synth.uniform((N, Co, Ho, Wo), 772, -1.0, 1.0)
    w64 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w64, None, 2, 1) * dy.double()).sum().backward()
    ref = w64.grad
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    dg = dy.to(dev).contiguous(memory_format=torch.channels_last)
    dw = T.conv3x3s2_wgrad(xg, dg)
    assert dw.shape == (Co, Ci, 3, 3) and dw.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)
    assert torch.equal(dw, T.conv3x3s2_wgrad(xg, dg))


def test_stride2_autograd_function_matches_library():
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    x = synth.uniform((2, 64, 12, 32), 761, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = synth.uniform((128, 64, 3, 3), 762, -0.1, 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    gy = synth.uniform((2, 128, 6, 16), 763, -1.0, 1.0).to(dev)
    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    assert T.conv3x3s2_supported(x1, w1, (2, 2), (1, 1))
    T.conv3x3s2(x1, w1).backward(gy)
    torch.nn.functional.conv2d(x2, w2, None, 2, 1).backward(gy)
    np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(w1.grad.cpu().numpy(), w2.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(w2.grad.abs().max()))


def test_kernels_inside_replayed_graphs():
    """The train step runs these kernels inside hipGraph replays (t2onet_amd/graphs.py).  Captured one graph per
    direction from one memory pool (a later capture re-uses memory an earlier one freed, as the encoder's graphs do),
    replayed with new inputs and with the allocator's free memory poisoned in between: every replay must match the
    library.  (With hipMemsetAsync clearing the workspaces' zero regions this failed from the second or third replay
    on -- the memset node ran out of order with the kernels around it; a kernel clears them now.)"""
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    c, h = 256, 16
    g = torch.Generator(device='cpu').manual_seed(771)
    x = torch.randn(16, c, h, h, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(16, c, h, h, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gy2 = torch.randn(16, c, h // 2, h // 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    fns = {'fwd': lambda: T.conv3x3_forward(x, w), 'dgrad': lambda: T.conv3x3_dgrad(gy, w),
           'wgrad': lambda: T.conv3x3_wgrad(x, gy), 's2': lambda: T.conv3x3s2_dgrad(gy2, w),
           's2w': lambda: T.conv3x3s2_wgrad(x, gy2), 's2f': lambda: T.conv3x3s2_forward(x, w)}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns.values():
            f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pool = torch.cuda.graph_pool_handle()
    graphs, outs = {}, {}
    for name, f in fns.items():
        graphs[name] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graphs[name], pool=pool, capture_error_mode='thread_local'):
            outs[name] = f()

    def lib_bwd(dy, stride, mask):
        return torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, mask)

    for trial in range(6):
        x.normal_(), gy.normal_(), gy2.normal_(), w.normal_().mul_(0.05)
        junk = torch.full((32 << 20,), float('inf'), device=dev)
        del junk
        for name in fns:
            graphs[name].replay()
        torch.cuda.synchronize()
        ref = {'fwd': torch.nn.functional.conv2d(x, w, None, 1, 1), 'dgrad': lib_bwd(gy, 1, [True, False, False])[0],
               'wgrad': lib_bwd(gy, 1, [False, True, False])[1], 's2': lib_bwd(gy2, 2, [True, False, False])[0],
               's2w': lib_bwd(gy2, 2, [False, True, False])[1], 's2f': torch.nn.functional.conv2d(x, w, None, 2, 1)}
        for name in fns:
            err = float((outs[name] - ref[name]).abs().max() / ref[name].abs().max())
            assert err < 2e-5, (trial, name, err)


def test_wgrad_refuses_widths_it_does_not_take():
    """W % 4 != 0 is not a shape of the LDS-DMA weight-gradient kernel: the library says so; the autograd wrapper takes
    the layer all the same and routes it to the any-size kernels (round 3: no library convolution at any width)."""
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    x = synth.uniform((1, 64, 6, 6), 905, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = synth.uniform((64, 64, 3, 3), 906, -0.1, 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    assert T.conv3x3_supported(x, w, (1, 1), (1, 1))
    with pytest.raises(RuntimeError):
        T.conv3x3_wgrad(x, x)
    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gy = synth.uniform((1, 64, 6, 6), 907, -1.0, 1.0).to(dev)
    T.conv3x3(x1, w1).backward(gy)
    torch.nn.functional.conv2d(x2, w2, None, 1, 1).backward(gy)
    np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(w1.grad.cpu().numpy(), w2.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(w2.grad.abs().max()))


def test_conv3x3_autograd_function_matches_library():
    """The autograd wrapper (own forward, data gradient and weight gradient) vs plain F.conv2d autograd."""
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    x = synth.uniform((2, 64, 10, 16), 711, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = synth.uniform((128, 64, 3, 3), 712, -0.1, 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    gy = synth.uniform((2, 128, 10, 16), 713, -1.0, 1.0).to(dev)
    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    assert T.conv3x3_supported(x1, w1, (1, 1), (1, 1))
    T.conv3x3(x1, w1).backward(gy)
    torch.nn.functional.conv2d(x2, w2, None, 1, 1).backward(gy)
    np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(w1.grad.cpu().numpy(), w2.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(w2.grad.abs().max()))


# (N, Ci, Co, H, W, stride); H, W = OUTPUT grid: 128- and 256-pixel tiles, a ragged last tile, several channel tiles,
# the encoder's first stage at batch 64 (1024 tiles)
STATS_SHAPES = [(2, 64, 64, 8, 8, 1), (3, 128, 192, 5, 16, 1), (9, 64, 64, 40, 40, 1), (2, 64, 128, 6, 8, 2), (3, 128, 64, 10, 24, 2),
                (64, 64, 64, 64, 64, 1)]


@pytest.mark.parametrize('shape', STATS_SHAPES)
def test_forward_leaves_the_batch_norm_statistics_of_its_output(shape):
    """conv3x3(..., want_stats=True): y unchanged, per-tile channel sums / sums of squares from the accumulators add up
    to those of y (fp64), and the batch norm that starts from them equals the one that reads y itself
    (models/actor_resnet.py:38-44 bn(conv(x)))."""
    import copy
    import t2onet_amd.functional as T
    N, Ci, Co, H, W, stride = shape
    dev = torch.device('cuda:0')
    x = synth.uniform((N, Ci, H * stride, W * stride), 761, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = synth.uniform((Co, Ci, 3, 3), 762, -0.1, 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    fwd = T.conv3x3_forward if stride == 1 else T.conv3x3s2_forward
    y0 = fwd(x, w)
    y, st = fwd(x, w, True)
    assert torch.equal(y, y0)
    assert st.shape[1:] == (2, Co) and st.shape[0] in (-(-N * H * W // 128), -(-N * H * W // 256))       # one row per pixel tile
    got = st.double().sum(0).cpu().numpy()
    y64 = y.double()
    ref = torch.stack([y64.sum((0, 2, 3)), (y64 * y64).sum((0, 2, 3))]).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())
    torch.manual_seed(3)
    bn_a = torch.nn.BatchNorm2d(Co).to(dev).train()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5)
        bn_a.bias.uniform_(-0.5, 0.5)
    bn_b = copy.deepcopy(bn_a)
    res = synth.uniform(tuple(y.shape), 763, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    out_a = T.batch_norm_relu(y, bn_a, res, partial=st)
    out_b = T.batch_norm_relu(y, bn_b, res)
    assert float((out_a - out_b).detach().abs().max()) <= 1e-5
    np.testing.assert_allclose(bn_a.running_mean.cpu().numpy(), bn_b.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(bn_a.running_var.cpu().numpy(), bn_b.running_var.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(st, fwd(x, w, True)[1])              # fixed summation order


def test_encoder_with_and_without_convolution_statistics_agree(monkeypatch):
    """The image encoder's forward and gradients with the batch norms fed from the convolutions' accumulators against
    the same encoder making its own statistics passes.  The two differ in the rounding of the batch statistics
    (1e-7 relative); a ReLU whose input is that close to zero may then switch, which changes single gradient entries
    by their full size: gradients are compared in the L2 norm."""
    import copy
    import t2onet_amd.actor_resnet as R
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    net_a = R.ResNet().to(dev).train().to(memory_format=torch.channels_last)
    net_b = copy.deepcopy(net_a)
    x = synth.uniform((8, 3, 128, 128), 771, 0.0, 1.0).to(dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    monkeypatch.setattr(R, '_CONV_STATS', True)
    fa = net_a(xa)
    monkeypatch.setattr(R, '_CONV_STATS', False)
    fb = net_b(xb)
    assert float((fa - fb).detach().abs().max()) <= 2e-5 * max(1.0, float(fb.detach().abs().max()))
    for (n, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        if 'running' in n:
            np.testing.assert_allclose(ba.cpu().numpy(), bb.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)
    g = synth.uniform(tuple(fa.shape), 772, -1.0, 1.0).to(dev)
    fa.backward(g)
    fb.backward(g)

    def rel(a, b):
        return float((a - b).double().norm()) / max(float(b.double().norm()), 1e-12)
    assert rel(xa.grad, xb.grad) <= 2e-3
    for (n, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert rel(pa.grad, pb.grad) <= 2e-3, n


# (N, Co, Ho, Wo): full and ragged tiles (8 x 32 outputs), one-row / one-column images, both channel counts, several
# tiles per workgroup
@pytest.mark.parametrize('shape', [(2, 64, 8, 32), (3, 32, 5, 12), (1, 64, 1, 40), (2, 64, 17, 1), (5, 64, 16, 64), (2, 32, 9, 70)])
def test_stem_forward_matches_conv2d_fp64_and_leaves_statistics(shape):
    """t2o_stem_fwd_nhwc: conv2d(x, w, None, 2, 1) with 3 input channels (models/actor_resnet.py:99) and the channel
    sums / sums of squares of its output."""
    import t2onet_amd.functional as T
    N, Co, Ho, Wo = shape
    x = synth.uniform((N, 3, 2 * Ho, 2 * Wo), 795, 0.0, 1.0)
    w = synth.uniform((Co, 3, 3, 3), 796, -1.0, 1.0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last)
    y = T.stem_forward(xg, wg)
    assert y.shape == (N, Co, Ho, Wo) and y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    y2, st = T.stem_forward(xg, wg, True)
    assert torch.equal(y, y2) and st.shape[1:] == (2, Co)
    y64 = y.double()
    want = torch.stack([y64.sum((0, 2, 3)), (y64 * y64).sum((0, 2, 3))]).cpu().numpy()
    np.testing.assert_allclose(st.double().sum(0).cpu().numpy(), want, rtol=2e-6, atol=2e-6 * np.abs(want).max())
    assert torch.equal(st, T.stem_forward(xg, wg, True)[1])


def test_stem_autograd_function_matches_library():
    """The stem through conv3x3s2 (own forward, data gradient and weight gradient) against F.conv2d."""
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    x = synth.uniform((3, 3, 32, 48), 797, 0.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = synth.uniform((64, 3, 3, 3), 798, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    g = synth.uniform((3, 64, 16, 24), 799, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    T.conv3x3s2(xa, wa).backward(g)
    torch.nn.functional.conv2d(xb, wb, None, 2, 1).backward(g)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(xb.grad.abs().max()))
    np.testing.assert_allclose(wa.grad.cpu().numpy(), wb.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(wb.grad.abs().max()))


@pytest.mark.parametrize('shape', [(2, 64, 8, 32), (3, 32, 5, 12), (1, 64, 1, 40), (2, 64, 17, 1), (5, 64, 16, 64), (2, 32, 9, 70), (40, 64, 32, 32)])
def test_stem_wgrad_matches_conv2d_fp64(shape):
    """t2o_stem_wgrad_nhwc against the weight gradient of F.conv2d(x, w, None, 2, 1) in fp64; bit-identical on a rerun."""
    import t2onet_amd.functional as T
    N, Co, Ho, Wo = shape
    x = synth.uniform((N, 3, 2 * Ho, 2 * Wo), 793, 0.0, 1.0)
    dy = synth.uniform((N, Co, Ho, Wo), 794, -1.0, 1.0)
    w = torch.zeros(Co, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w, None, 2, 1) * dy.double()).sum().backward()
    ref = w.grad
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    dg = dy.to(dev).contiguous(memory_format=torch.channels_last)
    dw = T.stem_wgrad(xg, dg)
    assert dw.shape == (Co, 3, 3, 3) and dw.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(dw.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert torch.equal(dw, T.stem_wgrad(xg, dg))


# (N, Ci, Co, H, W): odd sizes, widths that are no multiple of 4 / 8, one-pixel images, ragged tile tails, both strides
ANY_SHAPES = [(2, 64, 64, 4, 4), (3, 64, 128, 5, 7), (1, 128, 64, 19, 13), (2, 32, 64, 1, 1), (2, 256, 128, 3, 5),
              (1, 64, 64, 38, 57), (5, 64, 64, 2, 9)]


@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('shape', ANY_SHAPES)
def test_any_size_kernels_match_conv2d_fp64(shape, stride):
    """t2o_conv_generic.hip (gathered-row kernels for the layers the LDS-DMA kernels cannot take): forward, data gradient
    (+ addend) and weight gradient against fp64 conv2d, strides 1 and 2, odd and tiny images."""
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 901, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 902, -1.0, 1.0)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, None, stride, 1)
    dy = synth.uniform(tuple(y64.shape), 903, -1.0, 1.0)
    y64.backward(dy.double())
    dev = torch.device('cuda:0')
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)      # noqa: E731
    y = T.conv3x3_any_forward(cl(x), cl(w), stride)
    assert y.shape == y64.shape
    np.testing.assert_allclose(y.cpu().numpy(), y64.detach().float().numpy(), rtol=1e-5, atol=1e-5 * float(y64.detach().abs().max()))
    if Ci % 64 == 0:                                        # (data / weight gradient: 64-channel granularity on the reduced side)
        dx = T.conv3x3_any_dgrad(cl(dy), cl(w), (H, W), stride)
        gs = float(x64.grad.abs().max()) or 1.0
        np.testing.assert_allclose(dx.cpu().numpy(), x64.grad.float().numpy(), rtol=1e-5, atol=1e-5 * gs)
        if stride == 1:
            add = synth.uniform((N, Ci, H, W), 904, -1.0, 1.0)
            dxa = T.conv3x3_any_dgrad(cl(dy), cl(w), (H, W), 1, addend=cl(add))
            np.testing.assert_allclose(dxa.cpu().numpy(), (x64.grad + add.double()).float().numpy(), rtol=1e-5, atol=1e-5 * (gs + 1.0))
        dw = T.conv3x3_any_wgrad(cl(x), cl(dy), stride)
        np.testing.assert_allclose(dw.cpu().numpy(), w64.grad.float().numpy(), rtol=1e-5, atol=1e-5 * float(w64.grad.abs().max()))
        assert torch.equal(dw, T.conv3x3_any_wgrad(cl(x), cl(dy), stride))
        acc = dw.clone(memory_format=torch.channels_last)
        T.conv3x3_any_wgrad(cl(x), cl(dy), stride, into=acc)
        np.testing.assert_allclose(acc.cpu().numpy(), 2 * w64.grad.float().numpy(), rtol=1e-5, atol=2e-5 * float(w64.grad.abs().max()))
