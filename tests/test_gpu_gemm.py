"""t2o_gemm / t2o_colsum (t2o_gemm.hip): the general fp32 matrix-core GEMM behind the request encoder's input projection and
gradients (models/lang_encoder.py:91-102) and the decoder tape's weight gradients (models/action_decoder.py:52-63), against
fp64 -- every operand layout, ragged sizes (11 classes, 300 embedding columns, 812 decoder inputs, request lengths), column
slices of larger matrices, in-place accumulation -- and bitwise repeatable (no split-K, no atomics: the reason it exists)."""
import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# (M, N, K): the train step's own products first, then edge cases (one row, one column, K below / across a chunk, tile tails)
SHAPES = [(1088, 2048, 300), (1088, 2048, 512), (1088, 300, 2048), (1024, 300, 1088), (1024, 256, 1024), (2048, 812, 320),
          (2048, 512, 320), (512, 1024, 320), (11, 512, 320), (512, 512, 64),
          (1, 1, 1), (3, 5, 7), (64, 64, 16), (65, 63, 17), (130, 70, 33), (7, 200, 1000)]


def _operand(rows, cols, seed, pad):
    """a (rows, cols) matrix as a column slice of a (rows, cols + pad) one"""
    full = synth.uniform((rows, cols + pad), seed, -1.0, 1.0).to(DEV)
    return full[:, pad // 2:pad // 2 + cols] if pad else full


@pytest.mark.parametrize('layout', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('shape', SHAPES)
def test_gemm_matches_fp64_in_every_layout(shape, layout):
    import t2onet_amd.functional as T
    M, N, K = shape
    ak, bk = layout
    for pad in (0, 8, 3):                                  # dense; a 16-byte aligned slice; a misaligned one (scalar loads)
        A = _operand(K, M, 11, pad) if ak else _operand(M, K, 11, pad)
        B = _operand(K, N, 12, pad) if bk else _operand(N, K, 12, pad)
        ref = (A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t())
        out = T.gemm(A, B, a_kmajor=ak, b_kmajor=bk)
        tol = 2e-6 * max(1.0, float(ref.abs().max()))      # (K <= 2048 fp32 products, sums of |.| <= K)
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=tol * (K ** 0.5))
        assert torch.equal(out, T.gemm(A, B, a_kmajor=ak, b_kmajor=bk))
        # accumulate into a column block of a wider matrix (attention.linear_out's two halves: decoder_step.py)
        wide = synth.uniform((M, N + 5), 13, -1.0, 1.0).to(DEV)
        before = wide.clone()
        T.gemm(A, B, out=wide[:, 2:2 + N], a_kmajor=ak, b_kmajor=bk, accumulate=True)
        np.testing.assert_allclose(wide[:, 2:2 + N].cpu().numpy(), (before[:, 2:2 + N].double() + ref).cpu().numpy(), rtol=1e-5, atol=tol * (K ** 0.5))
        assert torch.equal(wide[:, :2], before[:, :2]) and torch.equal(wide[:, 2 + N:], before[:, 2 + N:])


@pytest.mark.parametrize('shape', [(320, 2048), (1088, 2048), (320, 11), (1, 1), (5, 70), (1089, 513)])
def test_colsum_matches_fp64(shape):
    import t2onet_amd.functional as T
    R, N = shape
    for pad in (0, 3):
        X = _operand(R, N, 21, pad)
        ref = X.double().sum(0)
        out = T.colsum(X)
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6 * R ** 0.5 * 4)
        assert torch.equal(out, T.colsum(X))
        acc = out.clone()
        T.colsum(X, out=acc, accumulate=True)
        np.testing.assert_allclose(acc.cpu().numpy(), 2 * ref.cpu().numpy(), rtol=1e-5, atol=1e-5 * R ** 0.5)


def test_gemm_rejects_bad_arguments():
    import t2onet_amd.functional as T
    A = torch.zeros(4, 6, device=DEV)
    B = torch.zeros(5, 7, device=DEV)
    with pytest.raises(ValueError):
        T.gemm(A, B)                                        # contraction lengths differ
    with pytest.raises(ValueError):
        T.gemm(A, torch.zeros(5, 6, device=DEV), out=torch.zeros(4, 4, device=DEV))
    with pytest.raises(RuntimeError):
        T.gemm(torch.zeros(4, 6), torch.zeros(5, 6))        # host tensors: no CPU fallback


def test_lstm_layer_and_tape_launch_no_library_gemm():
    """One request-encoder layer forward + backward with every product recorded: only this library's entry points."""
    import t2onet_amd._lib as L
    import t2onet_amd.functional as T
    B_, Lq, E, H = 8, 9, 300, 256
    x = synth.uniform((B_, Lq, E), 31, -1.0, 1.0).to(DEV).requires_grad_(True)
    lengths = torch.tensor([9, 3, 5, 9, 1, 2, 7, 4])
    dirs = [tuple(synth.uniform(s, 40 + i + 10 * d, -0.1, 0.1).to(DEV).requires_grad_(True)
                  for i, s in enumerate(((4 * H, E), (4 * H, H), (4 * H,), (4 * H,)))) for d in range(2)]
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        out, h, c = T.lstm_layer(x, lengths, dirs)
        (out.sum() + h.sum() + c.sum()).backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any('k_gemm_any' in n for n in names), names
    assert not [n for n in names if n.startswith('Cijk_') or 'rocblas' in n.lower()], names


def test_train_steps_launch_no_library_gemm_or_convolution():
    """Both train steps of a model that was simply built and moved to the GPU (no layout call): every kernel is this library's, an
    element-wise / copy / reduction kernel of the framework, or a runtime copy -- no BLAS (Cijk_*, rocblas*), no MIOpen kernel."""
    import re
    from torch.profiler import profile, ProfilerActivity
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    opt = t2onet_amd.default_options()
    torch.manual_seed(3)
    model = Actor(opt).to(DEV).train()
    tr = Trainer(model, opt)
    B, H, W = 8, 128, 128
    x = synth.requests(B, 17, 41).to(DEV)
    img, tgt = synth.images(B, H, W, 42).to(DEV), synth.images(B, H, W, 43).to(DEV)
    y = synth.op_targets(B, 45).to(DEV)
    img_y = synth.uniform((B, 6, 3, H, W), 46).to(DEV)
    gt = synth.uniform((B, 5, 24), 47, -1, 1).to(DEV)
    for _ in range(2):                                      # (first steps: run-time specialisations, allocator warm-up)
        tr.episode_step(x, img, tgt)
        tr.supervised_step(x, y, img, img_y, gt)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr.episode_step(x, img, tgt)
        tr.supervised_step(x, y, img, img_y, gt)
        torch.cuda.synchronize()
    from torch.autograd import DeviceType
    names = sorted({e.key for e in prof.key_averages() if e.device_type == DeviceType.CUDA})      # (kernels, not runtime API calls)
    own = [n for n in names if '(anonymous namespace)::' in n or 't2o::' in n]
    assert len(own) > 40, names
    allowed = re.compile(r'\(anonymous namespace\)::k_|t2o::|at::native::|^void at::|at::cuda::|__amd_rocclr_|^Memcpy|^Memset|hipMemcpy|hipMemset')
    foreign = [n for n in names if not allowed.search(n)]
    assert not foreign, foreign
