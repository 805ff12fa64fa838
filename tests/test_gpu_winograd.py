"""Winograd F(2x2,3x3) path of the deep encoder stages (t2o_winograd.hip: input / filter / output transforms around 16
plain GEMMs) against F.conv2d and its data gradient in fp64 (models/actor_resnet.py:27-44, the 256- / 512-channel layers)."""
import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu

# (N, Ci, Co, H, W): the two encoder stages at small batch, non-square maps, one tile per image, unequal channel counts,
# more tiles than the output transform's workgroup cap covers in one sweep
SHAPES = [(4, 256, 256, 16, 16), (3, 512, 512, 8, 8), (2, 256, 512, 6, 10), (5, 64, 128, 2, 2), (1, 128, 64, 2, 12), (40, 256, 256, 16, 16)]


@pytest.mark.parametrize('shape', SHAPES)
def test_winograd_forward_matches_conv2d_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 1701, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 1702, -1.0, 1.0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    dev = torch.device('cuda:0')
    y = T.conv3x3_winograd(x.to(dev), w.to(dev))
    assert y.shape == (N, Co, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    # (Winograd's transforms add a few roundings to the direct kernel's: 1e-5 of the output scale still holds)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', SHAPES[:4])
def test_winograd_data_gradient_with_addend_matches_fp64(shape):
    """dx = conv_transpose(dy, w) + addend through the same pipeline on dy with the mirrored, transposed filter."""
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    dy = synth.uniform((N, Co, H, W), 1711, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 1712, -1.0, 1.0)
    ad = synth.uniform((N, Ci, H, W), 1713, -1.0, 1.0)
    x64 = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x64, w.double(), None, 1, 1) * dy.double()).sum().backward()
    ref = x64.grad + ad.double()
    dev = torch.device('cuda:0')
    wg = w.to(dev).contiguous(memory_format=torch.channels_last)
    wt = T.conv_weight_transform(wg, 9, True)                                  # (Ci,3,3,Co) mirrored transpose
    U = T.wino_weight(wt, Ci, Co)
    dyh = dy.to(dev).permute(0, 2, 3, 1).contiguous()
    adh = ad.to(dev).permute(0, 2, 3, 1).contiguous()
    dx, _ = T.wino_conv_nhwc(dyh, U, N, H, W, adh)
    got = dx.permute(0, 3, 1, 2).cpu().numpy()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got, ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', [(4, 256, 256, 16, 16), (3, 512, 512, 8, 8), (40, 256, 256, 16, 16)])
def test_winograd_output_leaves_the_batch_norm_statistics(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    dev = torch.device('cuda:0')
    x = synth.uniform((N, H, W, Ci), 1721, -1.0, 1.0).to(dev)
    w = synth.uniform((Co, 3, 3, Ci), 1722, -0.1, 0.1).to(dev)
    U = T.wino_weight(w, Co, Ci)
    y0, none = T.wino_conv_nhwc(x, U, N, H, W)
    y, st = T.wino_conv_nhwc(x, U, N, H, W, None, True)
    assert none is None and torch.equal(y, y0) and st.shape[1:] == (2, Co)
    y64 = y.double()
    ref = torch.stack([y64.sum((0, 1, 2)), (y64 * y64).sum((0, 1, 2))]).cpu().numpy()
    np.testing.assert_allclose(st.double().sum(0).cpu().numpy(), ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())
    assert torch.equal(st, T.wino_conv_nhwc(x, U, N, H, W, None, True)[1])      # fixed summation order


def test_transforms_refuse_shapes_they_do_not_take():
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    U = torch.zeros(16, 64, 64, device=dev)
    for (N, H, W) in ((1, 3, 4), (1, 4, 5)):
        with pytest.raises(RuntimeError):
            T.wino_conv_nhwc(torch.zeros(N, H, W, 64, device=dev), U, N, H, W)
    with pytest.raises(RuntimeError):                                          # channel count not a power of two
        T.wino_conv_nhwc(torch.zeros(1, 4, 4, 96, device=dev), torch.zeros(16, 64, 96, device=dev), 1, 4, 4)


# (the weight gradient's GEMM takes channel counts that are multiples of 128; T = 5 * 1 and 3 * 15 tiles: all padding / ragged)
WGRAD_SHAPES = [(4, 256, 256, 16, 16), (3, 512, 512, 8, 8), (2, 256, 512, 6, 10), (5, 128, 128, 2, 2), (1, 128, 256, 2, 12), (40, 256, 256, 16, 16)]


@pytest.mark.parametrize('shape', WGRAD_SHAPES)
@pytest.mark.parametrize('accumulate', [False, True])
def test_winograd_weight_gradient_matches_fp64(shape, accumulate):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 1731, -1.0, 1.0)
    dy = synth.uniform((N, Co, H, W), 1732, -1.0, 1.0)
    w64 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w64, None, 1, 1) * dy.double()).sum().backward()
    ref = w64.grad
    dev = torch.device('cuda:0')
    V = T.wino_input(x.to(dev).permute(0, 2, 3, 1).contiguous(), N, H, W)
    start = synth.uniform((Co, 3, 3, Ci), 1733, -1.0, 1.0).to(dev)
    dw = start.clone()
    T.wino_wgrad_nhwc(V, dy.to(dev).permute(0, 2, 3, 1).contiguous(), dw, N, H, W, accumulate)
    got = (dw - start if accumulate else dw).permute(0, 3, 1, 2).cpu().numpy()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got, ref.float().numpy(), rtol=1e-5, atol=(3e-5 if accumulate else 1e-5) * scale)


def test_each_transform_kernel_matches_the_numpy_restatement():
    """k_wino_input / k_wino_weight / k_wino_dy / k_wino_output / k_wino_dw one by one against oracle/winograd.py (fp64)."""
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    from oracle import winograd as wg
    N, H, W, Ci, Co = 3, 6, 10, 64, 128
    dev = torch.device('cuda:0')
    x = synth.uniform((N, H, W, Ci), 1741, -1.0, 1.0)
    w = synth.uniform((Co, 3, 3, Ci), 1742, -1.0, 1.0)
    dy = synth.uniform((N, H, W, Co), 1743, -1.0, 1.0)
    Tt = N * (H // 2) * (W // 2)
    V = T.wino_input(x.to(dev), N, H, W).cpu().numpy()
    assert V.shape[1] % 256 == 0 and not V[:, Tt:].any()                       # zero rows up to the padded tile count
    np.testing.assert_allclose(V[:, :Tt], wg.input_transform(x.numpy().astype(np.float64)), rtol=0, atol=1e-6)
    U = T.wino_weight(w.to(dev), Co, Ci).cpu().numpy()
    np.testing.assert_allclose(U, wg.weight_transform(w.numpy()), rtol=0, atol=1e-6)
    lib = _lib.load()
    st = T._stream(dev)
    Ad = torch.empty(16, V.shape[1], Co, device=dev)
    dyg = dy.to(dev)
    _lib.check(lib.t2o_wino_dy_transform(T._ptr(dyg), T._ptr(Ad), N, H, W, Co, st), 't2o_wino_dy_transform')
    assert not Ad[:, Tt:].any()
    np.testing.assert_allclose(Ad[:, :Tt].cpu().numpy(), wg.dy_transform(dy.numpy()), rtol=0, atol=1e-6)
    M = synth.uniform((16, N * (H // 2) * (W // 2), Co), 1744, -1.0, 1.0)
    y = torch.empty(N, H, W, Co, device=dev)
    Mg = M.to(dev)
    _lib.check(lib.t2o_wino_output_transform(T._ptr(Mg), None, T._ptr(y), None, N, H, W, Co, st), 't2o_wino_output_transform')
    np.testing.assert_allclose(y.cpu().numpy(), wg.output_transform(M.numpy().astype(np.float64), N, H, W), rtol=0, atol=4e-6)
    dU = synth.uniform((16, Co, Ci), 1745, -1.0, 1.0)
    dw = torch.empty(Co, 3, 3, Ci, device=dev)
    dUg = dU.to(dev)
    _lib.check(lib.t2o_wino_dw_transform(T._ptr(dUg), T._ptr(dw), Co, Ci, 1, 0, st), 't2o_wino_dw_transform')
    ref = np.einsum('ai,aboc,bj->oijc', wg.G, dU.numpy().astype(np.float64).reshape(4, 4, Co, Ci), wg.G)
    np.testing.assert_allclose(dw.cpu().numpy(), ref, rtol=0, atol=4e-6)


@pytest.mark.parametrize('shape', [(16, 1024, 512, 512), (16, 4096, 256, 256), (3, 5, 64, 32), (2, 300, 128, 96), (1, 257, 192, 64), (4, 8, 64, 160),
                                   (20, 3500, 256, 64), (9, 7500, 512, 96)])
def test_own_batched_gemm_matches_fp64(shape):
    """t2o_gemm_nt_batched (k_gemm_nt): both tile widths, ragged M (rows past M re-read the last row), one and many stages."""
    import t2onet_amd.functional as T
    b, M, N, K = shape
    A = synth.uniform((b, M, K), 1751, -1.0, 1.0)
    B = synth.uniform((b, N, K), 1752, -1.0, 1.0)
    dev = torch.device('cuda:0')
    C = T.gemm_nt_batched(A.to(dev), B.to(dev))
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    scale = float(ref.abs().max())
    np.testing.assert_allclose(C.cpu().numpy(), ref.float().numpy(), rtol=1e-5, atol=1e-5 * scale)
    assert torch.equal(C, T.gemm_nt_batched(A.to(dev), B.to(dev)))


@pytest.mark.parametrize('shape', [(16, 1024, 512, 512), (16, 4096, 256, 256), (2, 256, 128, 384), (16, 256, 128, 128), (1, 768, 256, 128)])
def test_own_row_gemm_matches_fp64(shape):
    """t2o_gemm_tn_batched (k_gemm_tn): C = A^T B over the rows, 1 / 2 / 4 pieces of the row range added in order."""
    import t2onet_amd.functional as T
    b, rows, M, N = shape
    A = synth.uniform((b, rows, M), 1761, -1.0, 1.0)
    B = synth.uniform((b, rows, N), 1762, -1.0, 1.0)
    dev = torch.device('cuda:0')
    C = T.gemm_tn_batched(A.to(dev), B.to(dev))
    assert C.shape[1:] == (b, M, N) and C.shape[0] in (1, 2, 4)
    got = C.double().sum(0).cpu()
    ref = torch.bmm(A.double().transpose(1, 2), B.double())
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * scale)
    assert torch.equal(C, T.gemm_tn_batched(A.to(dev), B.to(dev)))
    # t2o_gemm_tn_batched_ld: the operands as row ranges of larger arenas (different plane strides): bit-identical
    bigA = torch.full((b, rows + 512, M), float('nan'), device=dev)
    bigB = torch.full((b, rows + 256, N), float('nan'), device=dev)
    bigA[:, 256:256 + rows] = A.to(dev)
    bigB[:, :rows] = B.to(dev)
    assert torch.equal(C, T.gemm_tn_batched(bigA[:, 256:256 + rows], bigB[:, :rows]))
    with pytest.raises(ValueError):
        T.gemm_tn_batched(bigA[:, :, :M // 2], bigB[:, :rows])


def test_combined_backward_equals_the_two_separate_pipelines():
    """wino_backward_nhwc (one transform pass over dy for both gradients) == data gradient + weight gradient done separately, bit for bit."""
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = 3, 256, 128, 6, 10
    dev = torch.device('cuda:0')
    x = synth.uniform((N, H, W, Ci), 1771, -1.0, 1.0).to(dev)
    dy = synth.uniform((N, H, W, Co), 1772, -1.0, 1.0).to(dev)
    w = synth.uniform((Co, Ci, 3, 3), 1773, -1.0, 1.0).to(dev).contiguous(memory_format=torch.channels_last)
    ad = synth.uniform((N, H, W, Ci), 1774, -1.0, 1.0).to(dev)
    Ud = T.wino_weight(T.conv_weight_transform(w, 9, True), Ci, Co)
    Vx = T.wino_input(x, N, H, W)
    dx0, _ = T.wino_conv_nhwc(dy, Ud, N, H, W, ad)
    dw0 = torch.zeros(Co, 3, 3, Ci, device=dev)
    T.wino_wgrad_nhwc(Vx, dy, dw0, N, H, W, False)
    dx1 = torch.empty_like(dx0)
    dw1 = torch.zeros_like(dw0)
    T.wino_backward_nhwc(dy, Vx, Ud, dw1, dx1, N, H, W, ad, False)
    assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1)


def test_batched_weight_transforms_equal_the_single_ones():
    """t2o_conv_weight_transform_batch / t2o_wino_weight_transform_batch (every layer of a step in one launch) == the per-layer calls."""
    import ctypes
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    from t2onet_amd.encoder import _batched
    lib = _lib.load()
    dev = torch.device('cuda:0')
    st = T._stream(dev)
    specs = [(64, 64, 9, 1), (128, 64, 9, 0), (128, 64, 1, 0), (256, 256, 9, 1), (512, 256, 9, 0), (64, 128, 9, 1)]
    ws = [synth.uniform((co, taps, ci), 1900 + i, -1.0, 1.0).to(dev) for i, (co, ci, taps, flip) in enumerate(specs)]
    outs = [torch.empty(ci * taps * co, device=dev) for (co, ci, taps, flip) in specs]
    _batched(lib.t2o_conv_weight_transform_batch, 'batch', st, ws, outs,
             [[s[0] for s in specs], [s[1] for s in specs], [s[2] for s in specs], [s[3] for s in specs]])
    for w, o, (co, ci, taps, flip) in zip(ws, outs, specs):
        ref = torch.empty_like(o)
        _lib.check(lib.t2o_conv_weight_transform(T._ptr(w), T._ptr(ref), co, ci, taps, flip, st), 'single')
        assert torch.equal(o, ref)
    banks = [synth.uniform((cn, 3, 3, ck), 1950 + i, -1.0, 1.0).to(dev) for i, (cn, ck) in enumerate([(64, 128), (256, 256), (32, 512)])]
    us = [torch.empty(16, b.shape[0], b.shape[3], device=dev) for b in banks]
    _batched(lib.t2o_wino_weight_transform_batch, 'batch', st, banks, us, [[b.shape[0] for b in banks], [b.shape[3] for b in banks]])
    for b, u in zip(banks, us):
        assert torch.equal(u, T.wino_weight(b, b.shape[0], b.shape[3]))


# (N, Ci, Co, H, W): the 64- and 128-channel stage shapes (small batch), several channel tiles, Ci != Co, one block, wide maps
WF_SHAPES = [(2, 64, 64, 16, 16), (3, 64, 64, 32, 48), (2, 128, 128, 32, 32), (1, 64, 128, 16, 32), (2, 128, 64, 48, 16), (1, 8, 64, 16, 16),
             (5, 64, 64, 64, 64)]


@pytest.mark.parametrize('shape', WF_SHAPES)
@pytest.mark.parametrize('with_addend', [False, True])
def test_on_chip_winograd_matches_conv2d_fp64(shape, with_addend):
    """t2o_wino_fused_conv_nhwc (input transform, 16 products and output transform in one launch, V and M in LDS / registers)
    against fp64 conv2d: Winograd F(2x2,3x3) in fp32 is good to a few 1e-6 of the output scale; the batch-norm statistics
    rows add up to the sums of y and y^2; a repeat is bit-identical."""
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    dev = torch.device('cuda:0')
    x = synth.uniform((N, Ci, H, W), 1801, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 3, 3), 1802, -1.0, 1.0)
    add = synth.uniform((N, Co, H, W), 1803, -1.0, 1.0) if with_addend else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    if with_addend:
        ref = ref + add.double()
    xh = x.to(dev).permute(0, 2, 3, 1).contiguous()
    U = T.wino_weight(w.to(dev).permute(0, 2, 3, 1).contiguous(), Co, Ci)
    Uc = T.wino_u_chunked(U)
    assert torch.equal(Uc.permute(1, 2, 0, 3).reshape(16, Co, Ci), U)
    import ctypes
    from t2onet_amd import _lib
    wl = w.to(dev).permute(0, 2, 3, 1).contiguous()
    Ucb = torch.empty_like(Uc)                               # the batched transform writes the chunk-major layout directly
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ints = lambda v: (ctypes.c_int * len(v))(*v)
    _lib.check(_lib.load().t2o_wino_weight_transform_chunked_batch(arr([wl, wl]), arr([Ucb, torch.empty_like(Uc)]), ints([Co, Co]), ints([Ci, Ci]), 2,
                                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'chunked batch')
    assert torch.equal(Ucb, Uc)
    ah = None if add is None else add.to(dev).permute(0, 2, 3, 1).contiguous()
    y, stats = T.wino_fused_conv_nhwc(xh, Uc, N, H, W, ah, want_stats=True)
    scale = float(ref.abs().max())
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) <= 1e-5 * scale, float((got - ref).abs().max()) / scale
    s = stats.double().sum(0).cpu()
    want = torch.stack([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert float((s - want).abs().max()) <= 1e-5 * float(want.abs().max())
    y2, stats2 = T.wino_fused_conv_nhwc(xh, Uc, N, H, W, ah, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(stats, stats2)


# (n_img, Ci, Co, H, W): the stage shapes at small batch, several (co, ci) tiles, Ci != Co, wide / tall maps, one step per split
WGW_SHAPES = [(2, 64, 64, 16, 16), (3, 64, 64, 32, 48), (2, 128, 128, 32, 32), (1, 64, 128, 16, 32), (2, 128, 64, 48, 16), (5, 64, 64, 64, 64),
              (1, 192, 64, 16, 16), (2, 256, 256, 16, 16)]          # (the last: 16 channel-tile combinations)


@pytest.mark.parametrize('shape', WGW_SHAPES)
@pytest.mark.parametrize('accumulate', [False, True])
def test_on_chip_winograd_weight_gradient_matches_fp64(shape, accumulate):
    """t2o_wino_fused_wgrad_nhwc (B^T d B and A dY A^T formed on chip, dU accumulated in the matrix-core registers over a range
    of the tile index, fixed-order sum of the ranges, G^T dU G) against the fp64 weight gradient of conv2d; a repeat is
    bit-identical."""
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 1831, -1.0, 1.0)
    dy = synth.uniform((N, Co, H, W), 1832, -1.0, 1.0)
    w64 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w64, None, 1, 1) * dy.double()).sum().backward()
    ref = w64.grad
    dev = torch.device('cuda:0')
    xh, dyh = x.to(dev).permute(0, 2, 3, 1).contiguous(), dy.to(dev).permute(0, 2, 3, 1).contiguous()
    start = synth.uniform((Co, 3, 3, Ci), 1833, -1.0, 1.0).to(dev)
    dw = start.clone()
    assert T.wino_fused_wgrad_nhwc(xh, dyh, dw, N, H, W, accumulate)
    got = (dw - start if accumulate else dw).permute(0, 3, 1, 2).cpu().numpy()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got, ref.float().numpy(), rtol=1e-5, atol=(3e-5 if accumulate else 1e-5) * scale)
    dw2 = start.clone()
    assert T.wino_fused_wgrad_nhwc(xh, dyh, dw2, N, H, W, accumulate)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('shape', [(320, 64, 64, 64, 64), (320, 128, 128, 32, 32), (320, 256, 256, 16, 16)])
def test_on_chip_winograd_weight_gradient_at_the_train_step_size(shape):
    """The shapes of one train step (bs = 64, five encoder passes side by side in encoder.WgradArena): against the direct
    weight-gradient kernel (itself held to fp64 in test_gpu_conv.py) over all 320 images, against fp64 on the first 16,
    and bitwise run-to-run reproducible."""
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    N, Ci, Co, H, W = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1841)
    xh = (torch.rand(N, H, W, Ci, generator=g) - 0.5).to(dev)
    dyh = (torch.rand(N, H, W, Co, generator=g) - 0.5).to(dev)
    dw = torch.zeros(Co, 3, 3, Ci, device=dev)
    assert T.wino_fused_wgrad_nhwc(xh, dyh, dw, N, H, W, False)
    dw2 = torch.full_like(dw, 7.0)
    assert T.wino_fused_wgrad_nhwc(xh, dyh, dw2, N, H, W, False)
    assert torch.equal(dw, dw2)
    lib = _lib.load()
    need = lib.t2o_conv3x3_wgrad_workspace_bytes(N, H, W, Ci, Co)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ref = torch.zeros_like(dw)
    _lib.check(lib.t2o_conv3x3_wgrad_acc_nhwc(xh.data_ptr(), dyh.data_ptr(), ref.data_ptr(), ws.data_ptr(), need, N, H, W, Ci, Co, 1, 0,
                                              torch.cuda.current_stream().cuda_stream), 'direct wgrad')
    scale = float(ref.abs().max())
    assert float((dw - ref).abs().max()) <= 2e-5 * scale
    n = 16
    dwn = torch.zeros_like(dw)
    assert T.wino_fused_wgrad_nhwc(xh[:n].contiguous(), dyh[:n].contiguous(), dwn, n, H, W, False)
    w64 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(xh[:n].cpu().permute(0, 3, 1, 2).double(), w64, None, 1, 1) * dyh[:n].cpu().permute(0, 3, 1, 2).double()).sum().backward()
    r64 = w64.grad.permute(0, 2, 3, 1)
    assert float((dwn.cpu().double() - r64).abs().max()) <= 1e-5 * float(r64.abs().max())


def test_on_chip_winograd_weight_gradient_refuses_what_it_does_not_take():
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    assert not lib.t2o_wino_fused_wgrad_supported(2, 24, 16, 64, 64) and not lib.t2o_wino_fused_wgrad_supported(2, 16, 16, 32, 64)
    assert not lib.t2o_wino_fused_wgrad_supported(2, 16, 16, 64, 576) and lib.t2o_wino_fused_wgrad_workspace_bytes(2, 8, 8, 64, 64) == 0
    x = torch.zeros(2, 16, 16, 64, device=dev)
    assert T.wino_fused_wgrad_nhwc(torch.zeros(2, 8, 8, 64, device=dev), torch.zeros(2, 8, 8, 64, device=dev), torch.zeros(64, 3, 3, 64, device=dev), 2, 8, 8, False) is False
    ws = torch.empty(16, dtype=torch.uint8, device=dev)
    assert lib.t2o_wino_fused_wgrad_nhwc(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), ws.data_ptr(), 16, 2, 16, 16, 64, 64, 0, None) != 0
    assert b'workspace' in lib.t2o_last_error()
