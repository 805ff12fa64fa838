"""The transform algebra of t2o_winograd.hip, restated in numpy (oracle/winograd.py), against the direct convolution and
torch's fp64 conv2d / autograd -- no GPU: what the HIP kernels are asked to compute is right before they are run."""
import numpy as np
import pytest
import torch

from oracle import synth, winograd


@pytest.mark.parametrize('shape', [(2, 4, 6, 5, 7), (1, 2, 2, 3, 3), (3, 8, 4, 4, 2)])
def test_winograd_equals_direct_convolution(shape):
    N, H, W, Ci, Co = shape
    x = synth.uniform((N, H, W, Ci), 1801, -1.0, 1.0).numpy()
    w = synth.uniform((Co, 3, 3, Ci), 1802, -1.0, 1.0).numpy()
    y = winograd.conv(x, w)
    ref = winograd.direct_conv(x.astype(np.float64), w)
    np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-12)
    t = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(0, 3, 1, 2), None, 1, 1)
    np.testing.assert_allclose(y, t.permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)


def test_data_gradient_is_the_same_pipeline_on_the_mirrored_transpose():
    N, H, W, Ci, Co = 2, 4, 6, 3, 5
    dy = synth.uniform((N, H, W, Co), 1811, -1.0, 1.0).numpy()
    w = synth.uniform((Co, 3, 3, Ci), 1812, -1.0, 1.0).numpy()
    x = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x, torch.from_numpy(w).double().permute(0, 3, 1, 2), None, 1, 1) * torch.from_numpy(dy).double().permute(0, 3, 1, 2)).sum().backward()
    wt = np.ascontiguousarray(w[:, ::-1, ::-1].transpose(3, 1, 2, 0))          # (Ci,3,3,Co): t2o_conv_weight_transform(flip)
    dx = winograd.conv(dy, wt)
    np.testing.assert_allclose(dx, x.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)


def test_weight_gradient_in_the_transformed_domain():
    N, H, W, Ci, Co = 3, 6, 4, 4, 5
    xn = synth.uniform((N, H, W, Ci), 1821, -1.0, 1.0).numpy()
    dy = synth.uniform((N, H, W, Co), 1822, -1.0, 1.0).numpy()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(torch.from_numpy(xn).double().permute(0, 3, 1, 2), w, None, 1, 1) * torch.from_numpy(dy).double().permute(0, 3, 1, 2)).sum().backward()
    dw = winograd.weight_gradient(xn, dy)
    np.testing.assert_allclose(dw, w.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('shape', [(1, 16, 16, 3, 2), (2, 16, 32, 2, 3), (1, 32, 16, 2, 2)])
def test_weight_gradient_the_way_the_on_chip_kernel_forms_it(shape):
    """t2o_wino_wgrad.hip's formulation restated in numpy (steps of 8 tiles, zero rows, clamped + masked edge columns inside the
    column step's multiply-add, A dY A^T without its sign flips and the six planes' sums negated at the end, split ranges added
    in order) against the plain transformed-domain weight gradient and the direct one: the algebra of the kernel's shortcuts is
    exact."""
    N, H, W, Ci, Co = shape
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (N, H, W, Ci))
    dy = rng.uniform(-1, 1, (N, H, W, Co))
    wg = winograd
    ref = wg.weight_gradient(x, dy)
    for splits in (1, 3):
        got = wg.weight_gradient_onchip_form(x, dy, splits)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    # and the reference form itself is the gradient of the direct convolution (finite difference on one weight)
    w = rng.uniform(-1, 1, (Co, 3, 3, Ci))
    e = np.zeros_like(w)
    e[1, 2, 0, 1] = 1.0
    fd = ((wg.direct_conv(x, w + 1e-6 * e) - wg.direct_conv(x, w - 1e-6 * e)) * dy).sum() / 2e-6
    assert abs(fd - ref[1, 2, 0, 1]) < 1e-6 * max(1.0, abs(fd))
    assert wg.NEGATED_PLANES == (3, 7, 11, 12, 13, 14)
