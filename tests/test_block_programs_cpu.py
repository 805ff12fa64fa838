"""CPU check of the kernels' per-thread code (t2o_pixel_math.h / t2o_block_programs.h) against
the oracle, through the host emulation harness (tests/host_emul).  The same comparisons run on
the real kernels in tests/test_gpu_operators.py (-m gpu).

Tolerances (fp32): forward 1e-6 abs (same operation order as the oracle; cos differs by an ulp);
gimg 2e-5 abs + 1e-4 rel against the oracle's fp32 autograd THROUGH the HSV round trip, which is
itself noisy, and 2e-6 against its fp64 autograd for the closed forms."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth
from tests import emul

OPT = cpu_ref.default_opt()
OPS = [0, 1, 2, 3, 5, 6, 7]
SHAPES = [(2, 24, 20), (2, 23, 19), (1, 40, 150), (3, 17, 68)]


def oracle_fwd_bwd(op, img, p, mask, gout, dtype=torch.float32):
    x = img.to(dtype).clone().requires_grad_(True)
    pp = p.to(dtype).clone().requires_grad_(True)
    m = None if mask is None else mask.to(dtype)
    if dtype == torch.float64:
        out = operator_apply64(op, x, pp, m)
    else:
        out = cpu_ref.operator_apply(op, x, pp, m, OPT)
    out.backward(gout.to(dtype))
    gp = pp.grad if pp.grad is not None else torch.zeros_like(pp)
    return out.detach(), x.grad, gp


def operator_apply64(op, x, p, m):
    # the oracle formulas are dtype-agnostic except the sharpness kernel tensor
    if op == 6:
        k = torch.tensor([[cpu_ref.SHARP_KERNEL]], dtype=x.dtype)
        d = torch.cat([torch.nn.functional.conv2d(x[:, c:c + 1], k, padding=1) for c in range(3)], 1)
        out = x + p.unsqueeze(-1).unsqueeze(-1) * d
        mm = torch.ones_like(x) if m is None else m
        return torch.clamp(out * mm + x * (1 - mm), 0, 1)
    if op == 1:
        lum = torch.clamp(cpu_ref.rgb2lum(x), 0, 1)
        clum = -torch.cos(np.pi * lum) * 0.5 + 0.5
        out = cpu_ref.lerp(x, x / (lum + 1e-6) * clum, p.unsqueeze(-1).unsqueeze(-1))
        mm = torch.ones_like(x) if m is None else m
        return torch.clamp(out * mm + x * (1 - mm), 0, 1)
    return cpu_ref.operator_apply(op, x, p, m, OPT)


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('op', OPS)
def test_forward_backward_vs_oracle(op, shape):
    B, H, W = shape
    img = synth.images(B, H, W, 61)
    gout = synth.uniform((B, 3, H, W), 62, -1.0, 1.0)
    masks = {'none': None, 'm1': synth.masks(B, 1, H, W, 63), 'm3': synth.masks(B, 3, H, W, 64, soft=False)}
    for si, setting in enumerate(['mid', 'strong', 'neg']):
        p = synth.op_params(op, B, 400 + 10 * op + si, setting)
        for mname, mask in masks.items():
            tag = 'op%d %s %s %s' % (op, shape, setting, mname)
            o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img, p, mask, gout)
            _, gi64, gp64 = oracle_fwd_bwd(op, img, p, mask, gout, torch.float64)
            for iters in (0, 3):
                out, _ = emul.fwd(op, img.numpy(), p.numpy(), None if mask is None else mask.numpy(), iters=iters)
                np.testing.assert_allclose(out, o_ref.numpy(), rtol=0, atol=1e-6, err_msg=tag)
                gi, gp = emul.bwd(op, img.numpy(), p.numpy(), gout.numpy(), None if mask is None else mask.numpy(), iters=iters)
                np.testing.assert_allclose(gi, gi64.numpy(), rtol=1e-5, atol=2e-6, err_msg=tag + ' gimg/f64')
                np.testing.assert_allclose(gi, gi_ref.numpy(), rtol=1e-4, atol=2e-5 if op in (0, 2) else 2e-6, err_msg=tag + ' gimg/f32')
                scale = max(1.0, float(gp64.abs().max()))
                np.testing.assert_allclose(gp, gp64.numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg=tag + ' gparam')


def test_identity_and_dynamic_batch():
    B, H, W = 9, 24, 20
    img = synth.images(B, H, W, 71)
    gout = synth.uniform((B, 3, H, W), 72, -1.0, 1.0)
    ops = [0, 1, 2, 3, 5, 6, 7, -1, 6]
    params = torch.zeros(B, 24)
    for b, op in enumerate(ops):
        if op >= 0:
            params[b, :cpu_ref.OP_NPARAM[op]] = synth.op_params(op, 1, 500 + b, 'mid')[0]
    out, _ = emul.fwd(-2, img.numpy(), params.numpy(), op_id=ops)
    gi, gp = emul.bwd(-2, img.numpy(), params.numpy(), gout.numpy(), op_id=ops)
    for b, op in enumerate(ops):
        if op < 0:
            np.testing.assert_array_equal(out[b], img[b].numpy())
            np.testing.assert_array_equal(gi[b], gout[b].numpy())
            continue
        n = cpu_ref.OP_NPARAM[op]
        o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img[b:b + 1], params[b:b + 1, :n], None, gout[b:b + 1], torch.float64)
        np.testing.assert_allclose(out[b], o_ref[0].numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(gi[b], gi_ref[0].numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(gp[b, :n], gp_ref[0].numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(gp_ref.abs().max())))
        assert np.all(gp[b, n:] == 0)


@pytest.mark.parametrize('op', [0, 1, 2, 3, 5, 6, -1])
def test_fused_l1(op):
    B, H, W = 2, 20, 36
    img = synth.images(B, H, W, 81)
    tgt = synth.images(B, H, W, 82)
    n = cpu_ref.OP_NPARAM[op] if op >= 0 else 1
    p = synth.op_params(max(op, 0), B, 83, 'mid')
    x = img.double().clone().requires_grad_(True)
    pp = p.double().clone().requires_grad_(True)
    out = x if op < 0 else operator_apply64(op, x, pp, None)
    loss = (out - tgt.double()).abs().mean()
    (loss * 3.0).backward()
    o, l = emul.fwd(op, img.numpy(), p.numpy(), target=tgt.numpy())
    assert abs(l - loss.item()) < 1e-6
    gi, gp = emul.bwd(op, img.numpy(), p.numpy(), target=tgt.numpy(), gloss=3.0)
    np.testing.assert_allclose(gi, x.grad.numpy(), rtol=1e-5, atol=1e-8)
    if op >= 0:
        np.testing.assert_allclose(gp, pp.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('ops', [[0, 1, 2, 3, 5, 6], [5, 3, 5, 3, 0, 1, 2, 6], [6, 0, -1, 6, 3], [1], [6], [-1, -1],
                                 [0, 1, 2, 3, 5, 0, 1, 2, 3, 5, 7, 1]])
@pytest.mark.parametrize('shape', [(3, 32, 40), (2, 23, 19)])
def test_fused_sequence_equals_oracle_chain(ops, shape):
    """The fused sequence (pointwise runs in registers, LDS curve lookup) against the oracle's
    operator-by-operator chain: images bit-for-bit close (1e-6), loss 1e-6, gradients vs fp64."""
    B, H, W = shape
    img = synth.images(B, H, W, 31)
    tgt = synth.images(B, H, W, 32)
    params = torch.zeros(len(ops), B, 24)
    ps64 = []
    for k, op in enumerate(ops):
        if op >= 0:
            n = cpu_ref.OP_NPARAM[op]
            params[k, :, :n] = synth.op_params(op, B, 300 + k, 'mid')
            ps64.append(params[k, :, :n].double().clone().requires_grad_(True))
        else:
            ps64.append(None)
    # fp32 oracle chain for the forward, fp64 for the gradients
    cur = img
    for k, op in enumerate(ops):
        if op >= 0:
            cur = cpu_ref.operator_apply(op, cur, params[k, :, :cpu_ref.OP_NPARAM[op]], None, OPT)
    ref_loss = (cur - tgt).abs().mean().item()
    x64 = img.double().clone().requires_grad_(True)
    c64 = x64
    for k, op in enumerate(ops):
        if op >= 0:
            c64 = operator_apply64(op, c64, ps64[k], None)
    ((c64 - tgt.double()).abs().mean() * 2.0).backward()
    out, loss, gimg, gparams = emul.fused(ops, img.numpy(), params.numpy(), tgt.numpy(), gloss=2.0)
    np.testing.assert_allclose(out, cur.numpy(), rtol=0, atol=2e-6)
    assert abs(loss - ref_loss) < 1e-6
    g = x64.grad.numpy()
    np.testing.assert_allclose(gimg, g, rtol=2e-3, atol=2e-3 * np.abs(g).max())
    for k, op in enumerate(ops):
        if op >= 0 and op != 7:
            gk = ps64[k].grad.numpy()
            np.testing.assert_allclose(gparams[k, :, :gk.shape[1]], gk, rtol=2e-3, atol=2e-3 * max(np.abs(gk).max(), 1e-4))
        else:
            assert np.all(gparams[k] == 0)


@pytest.mark.parametrize('ops', [[0, 1, 2, 3, 5, 6], [5, 3, 5, 3, 0, 1, 2, 6], [0, 1, 2, 3, 5], [5, 3, 5, 3, 0, 1, 2]])
@pytest.mark.parametrize('iters', [0, 3])
def test_static_chain_backward_equals_runtime_loop_program(ops, iters):
    """chain_bwd_thread_static (operator list fixed at compile time: unrolled sweeps, parameter sums kept in
    registers across the thread's pixels, one flush per thread) against the run-time-loop program on the same
    inputs: identical image gradients (same per-pixel arithmetic), parameter gradients equal up to summation order.
    Shapes with dead threads in the last workgroup and several pixels per thread."""
    B, H, W = 2, 23, 19
    img = synth.images(B, H, W, 61)
    tgt = synth.images(B, H, W, 62)
    params = torch.zeros(len(ops), B, 24)
    for k, op in enumerate(ops):
        n = cpu_ref.OP_NPARAM[op]
        params[k, :, :n] = synth.op_params(op, B, 400 + k, 'mid')
    o0, l0, g0, p0 = emul.fused(ops, img.numpy(), params.numpy(), tgt.numpy(), gloss=1.5, iters=iters, use_static=0)
    o1, l1, g1, p1 = emul.fused(ops, img.numpy(), params.numpy(), tgt.numpy(), gloss=1.5, iters=iters, use_static=1)
    assert np.array_equal(o0, o1) and l0 == l1
    np.testing.assert_array_equal(g1, g0)
    np.testing.assert_allclose(p1, p0, rtol=2e-5, atol=1e-6 * max(np.abs(p0).max(), 1e-6))


@pytest.mark.parametrize('ops', [[0, 1, 2, 3, 5], [5, 3, 5, 3, 0, 1, 2], [6, 0, 7, 5]])
@pytest.mark.parametrize('iters,use_static', [(0, 0), (3, 0), (0, 1), (3, 1)])
def test_last_chain_backward_also_yields_the_image_and_the_loss(ops, iters, use_static):
    """t2o_fused_sequence_l1_value_grad: the last per-pixel segment's L1 backward recomputes the final pixel, so it also
    stores it (ChainArgs.out) and returns |out - target| for the loss -- the device-side forward launch of that segment is
    dropped.  The image must be the forward program's bit for bit, the loss equal to the forward's up to summation order."""
    B, H, W = 2, 23, 19
    img, tgt = synth.images(B, H, W, 71), synth.images(B, H, W, 72)
    params = torch.zeros(len(ops), B, 24)
    for k, op in enumerate(ops):
        n = cpu_ref.OP_NPARAM[op]
        if n:
            params[k, :, :n] = synth.op_params(op, B, 500 + k, 'mid')
    out, loss, _, _, vout, vloss = emul.fused(ops, img.numpy(), params.numpy(), tgt.numpy(), gloss=1.5, iters=iters,
                                              use_static=use_static, with_value=True)
    np.testing.assert_array_equal(vout, out)
    assert abs(vloss - loss) <= 2e-7 * max(loss, 1e-3)


def edge_image():
    """Exact ties the random suite never hits: black, white, greys, two equal maxima/minima,
    values on curve knots, flat saturated regions (sharpness output exactly 0 / 1)."""
    H, W = 16, 24
    img = synth.images(2, H, W, 91)
    img[:, :, 0:4, 0:8] = 0.0                        # flat black
    img[:, :, 0:4, 8:16] = 1.0                       # flat white
    img[:, :, 4:6, :] = 0.5                          # grey, on a knot
    img[0, :, 6, :] = torch.tensor([0.25, 0.625, 0.875]).view(3, 1)      # knots
    img[:, 1, 7, :] = img[:, 0, 7, :]                # r == g (two equal maxima or minima)
    img[:, 2, 8, :] = img[:, 1, 8, :]                # g == b
    img[1, :, 9, :] = 0.125
    return img


@pytest.mark.parametrize('op', OPS)
def test_edge_cases_ties_follow_pytorch(op):
    """Ties: PyTorch's conventions (clamp inclusive, max/min over channels -> first index,
    elementwise max/min split 1/2) are reproduced; compared with the oracle's fp32 autograd."""
    img = edge_image()
    B, _, H, W = img.shape
    gout = synth.uniform((B, 3, H, W), 92, -1.0, 1.0)
    for setting in ['mid', 'strong']:
        p = synth.op_params(op, B, 600 + op, setting)
        o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img, p, None, gout)
        out, _ = emul.fwd(op, img.numpy(), p.numpy())
        np.testing.assert_allclose(out, o_ref.numpy(), rtol=0, atol=1e-6)
        gi, gp = emul.bwd(op, img.numpy(), p.numpy(), gout.numpy())
        gi_ref = gi_ref.numpy()
        np.testing.assert_allclose(gi, gi_ref, rtol=1e-4, atol=5e-5 if op in (0, 2) else 2e-6)
        np.testing.assert_allclose(gp, gp_ref.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(gp_ref.abs().max())))


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 1, 7), (2, 5, 1), (1, 3, 5), (1, 2, 130)])
def test_tiny_and_ragged_shapes(shape):
    B, H, W = shape
    img = synth.images(B, H, W, 93)
    gout = synth.uniform((B, 3, H, W), 94, -1.0, 1.0)
    for op in OPS:
        p = synth.op_params(op, B, 700 + op, 'mid')
        o_ref, _, _ = oracle_fwd_bwd(op, img, p, None, gout)
        _, gi64, gp64 = oracle_fwd_bwd(op, img, p, None, gout, torch.float64)
        out, _ = emul.fwd(op, img.numpy(), p.numpy())
        np.testing.assert_allclose(out, o_ref.numpy(), rtol=0, atol=1e-6)
        gi, gp = emul.bwd(op, img.numpy(), p.numpy(), gout.numpy())
        np.testing.assert_allclose(gi, gi64.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(gp, gp64.numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(gp64.abs().max())))


@pytest.mark.parametrize('shape', [(1, 3, 48, 40), (2, 3, 33, 70), (1, 1, 5, 7)])
def test_ssim_block_program(shape, golden_dir):
    import os
    B, C, H, W = shape
    a = synth.uniform(shape, 51)
    b = (a + synth.uniform(shape, 52, -0.1, 0.1)).clamp(0, 1)
    ref = cpu_ref.ssim(a, b, size_average=False)
    np.testing.assert_allclose(emul.ssim(a.numpy(), b.numpy()), ref.numpy(), rtol=1e-5, atol=1e-6)
    if shape == (1, 3, 48, 40):            # the golden value computed by the reference's own utils/ssim
        g = np.load(os.path.join(golden_dir, 'ssim.npz'))
        assert abs(float(emul.ssim(a.numpy(), b.numpy())[0]) - float(g['ssim'])) < 1e-5


@pytest.mark.parametrize('shape', [(1, 3, 48, 40), (2, 3, 33, 70), (1, 1, 5, 7), (1, 2, 64, 37)])
def test_ssim_backward_block_program(shape):
    """The five phases of k_ssim_bwd (closed-form gradient of utils/ssim/__init__.py:20-40) thread by thread on the host against
    fp64 autograd of the oracle's SSIM, for both images and a non-uniform output gradient; tiles with every kind of border."""
    B, C, H, W = shape
    a = synth.uniform(shape, 51)
    b = (a + synth.uniform(shape, 52, -0.1, 0.1)).clamp(0, 1)
    gout = synth.uniform((B,), 53, 0.5, 1.5)
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    (cpu_ref.ssim(a64, b64, size_average=False) * gout.double()).sum().backward()
    ga, gb = emul.ssim_bwd(a.numpy(), b.numpy(), gout.numpy())
    for got, ref in ((ga, a64.grad), (gb, b64.grad)):
        scale = float(ref.abs().max())
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, ref.numpy(), rtol=0, atol=2e-5 * scale)


def quantized_image(B=2, H=24, W=32, levels=16, seed=95):
    """8-bit-photo-like image on a coarse lattice: most pixels have two or three equal channels,
    many sit on curve knots or at 0 / 1."""
    return synth.integers((B, 3, H, W), seed, 0, levels).float() / levels


@pytest.mark.parametrize('op', OPS)
def test_quantized_images_ties_everywhere(op):
    img = quantized_image()
    B, _, H, W = img.shape
    gout = synth.uniform((B, 3, H, W), 96, -1.0, 1.0)
    p = synth.op_params(op, B, 800 + op, 'mid')
    o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img, p, None, gout)
    out, _ = emul.fwd(op, img.numpy(), p.numpy())
    np.testing.assert_allclose(out, o_ref.numpy(), rtol=0, atol=1e-6)
    gi, gp = emul.bwd(op, img.numpy(), p.numpy(), gout.numpy())
    np.testing.assert_allclose(gi, gi_ref.numpy(), rtol=1e-4, atol=5e-5 if op in (0, 2) else 2e-6)
    np.testing.assert_allclose(gp, gp_ref.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(gp_ref.abs().max())))


def test_forward_is_bit_identical_to_oracle():
    """The kernels' forward arithmetic reproduces the reference's eager fp32 path to the last bit
    (every operator except contrast, whose cos() differs by an ulp between libms)."""
    img = synth.images(4, 64, 64, 5)
    for op in [0, 2, 3, 5, 6]:
        p = synth.op_params(op, 4, 50 + op, 'mid')
        ref = cpu_ref.operator_apply(op, img, p, None, OPT).numpy()
        out, _ = emul.fwd(op, img.numpy(), p.numpy())
        assert np.array_equal(out, ref), op
    p = synth.op_params(1, 4, 51, 'mid')
    ref = cpu_ref.operator_apply(1, img, p, None, OPT).numpy()
    out, _ = emul.fwd(1, img.numpy(), p.numpy())
    assert np.abs(out - ref).max() <= 1.2e-7 and (out != ref).mean() < 0.02
