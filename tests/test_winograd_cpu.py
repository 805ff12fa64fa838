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
