"""CPU checks of the actor's host logic (everything that does not need the HIP library) and
of the data-parallel path over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import cpu_ref, synth

OPT = cpu_ref.default_opt(input_dropout_p=0.0, dropout_p=0.0)


@pytest.fixture(scope='module')
def model():
    import t2onet_amd
    from t2onet_amd.actor import Actor
    m = Actor(t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0))
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), seed=7))
    return m.eval()


def test_state_dict_matches_reference_layout(model, golden_dir):
    g = np.load(os.path.join(golden_dir, 'actor.npz'))
    sd = model.state_dict()
    assert list(sd.keys()) == list(g['sd_keys'])
    assert [v.numel() for v in sd.values()] == list(g['sd_numel'])
    assert [n for n, _ in model.named_parameters()] == list(g['param_names'])


def test_encoders_match_reference_outputs(model, golden_dir):
    g = np.load(os.path.join(golden_dir, 'actor.npz'))
    x = synth.requests(4, 17, 41)
    img = synth.images(4, 64, 64, 42)
    with torch.no_grad():
        enc_out, (h, c), _ = model.lang_encoder(x)
        np.testing.assert_allclose(enc_out.numpy(), g['enc_out'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(h.numpy(), g['enc_h'], rtol=1e-6, atol=1e-7)
        enc2, _, _ = model.lang_encoder(x, lengths=(x != 0).sum(1))
        assert torch.equal(enc2, enc_out)
        np.testing.assert_allclose(model.image_features(img).numpy(), g['img_feat_eval'], rtol=1e-5, atol=1e-6)
        hid = model.decoder._init_state((h, c))
        assert hid[0].shape == (2, 4, 512)


def test_select_end_images_matches_reference_loop():
    from t2onet_amd.train import select_end_images
    pred_imgs = synth.uniform((6, 5, 3, 4, 4), 5)
    pred_ops = torch.tensor([[4, 2, 3, 5, 6], [2, 4, 5, 6, 8], [3, 4, 5, 6, 8], [3, 4, 5, 6, 2], [4, 2, 2, 3, 5], [9, 8, 6, 5, 4]])
    assert torch.equal(select_end_images(pred_imgs, pred_ops, 2), cpu_ref.select_end_images(pred_imgs, pred_ops, 2))


def test_op_mask_scatter_equals_python_loop():
    mask = torch.tensor(cpu_ref.OP_MASK).repeat(5, 1)
    ops = torch.tensor([[2], [3], [9], [8], [5]])
    ref = mask.clone()
    for b in range(5):
        ref[b, ops[b, 0]] = 0
    assert torch.equal(mask.scatter_(1, ops, 0.0), ref)


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from t2onet_amd.train import FlatGradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    grads = FlatGradients(net.parameters())
    opt = torch.optim.Adam(grads.params, lr=1e-2)
    data = synth.uniform((8, 8), 3, -1, 1)
    tgt = synth.uniform((8, 4), 4, -1, 1)
    shard = slice(rank * 4, rank * 4 + 4)
    for _ in range(3):
        grads.zero()
        # the last layer is only used on rank 0: its gradient must count as zeros on rank 1
        h = net[2](net[1](net[0](data[shard])))
        y = net[3](h) if rank == 0 else h
        ((y - tgt[shard]) ** 2).mean().backward()
        grads.all_reduce_mean()
        opt.step()
    torch.save([p.detach().clone() for p in net.parameters()], out % rank)
    dist.destroy_process_group()


def test_data_parallel_flat_allreduce_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    out = str(tmp_path / 'rank%d.pt')
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out % 0), torch.load(out % 1)
    for u, v in zip(a, b):
        assert torch.equal(u, v)                     # replicas stay bit-identical
    # single-process reference: mean of the two shard gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    data, tgt = synth.uniform((8, 8), 3, -1, 1), synth.uniform((8, 4), 4, -1, 1)
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        total = 0
        for r in range(2):
            h = net[2](net[1](net[0](data[r * 4:r * 4 + 4])))
            y = net[3](h) if r == 0 else h
            total = total + 0.5 * ((y - tgt[r * 4:r * 4 + 4]) ** 2).mean()
        total.backward()
        opt.step()
    for u, v in zip(a, net.parameters()):
        assert torch.allclose(u, v.detach(), rtol=1e-5, atol=1e-7)


def test_sample_categorical_is_a_categorical_draw():
    """Inverse-CDF sampling (no host sync) draws from the same distribution as Categorical(probs):
    zero-weight entries are never chosen, frequencies follow the (unnormalised) weights."""
    from t2onet_amd.actor import sample_categorical
    torch.manual_seed(3)
    w = torch.tensor([[0.0, 0.0, 2.0, 1.0, 0.0, 1.0, 0.0]]).repeat(40000, 1)
    idx = sample_categorical(w)
    assert idx.shape == (40000, 1) and idx.dtype == torch.long
    counts = torch.bincount(idx.view(-1), minlength=7).float() / 40000
    assert float(counts[[0, 1, 4, 6]].sum()) == 0.0
    np.testing.assert_allclose(counts[[2, 3, 5]].numpy(), [0.5, 0.25, 0.25], atol=0.01)
    # a row whose last entries are zero never falls off the end
    w2 = torch.zeros(1000, 11)
    w2[:, 3] = 1e-3
    assert bool((sample_categorical(w2) == 3).all())


def test_decoder_single_step_equals_nn_lstm():
    """Decoder._rnn_step (fused cell calls on the module's own weights) == nn.LSTM on a length-1 sequence,
    values and gradients."""
    from t2onet_amd.action_decoder import Decoder
    torch.manual_seed(5)
    dec = Decoder(11, 6, 300, 256, 2, 'lstm', bidirectional=True, use_attention=False)
    x = torch.randn(4, 1, 300 + 512, requires_grad=True)
    h0, c0 = torch.randn(2, 4, 512), torch.randn(2, 4, 512)
    out, (h, c) = dec._rnn_step(x, (h0, c0))
    ref_out, (rh, rc) = dec.rnn(x, (h0, c0))
    np.testing.assert_allclose(out.detach().numpy(), ref_out.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(h.detach().numpy(), rh.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(c.detach().numpy(), rc.detach().numpy(), rtol=0, atol=2e-6)
    g1, = torch.autograd.grad(out.sum() + c.sum(), x, retain_graph=True)
    g2, = torch.autograd.grad(ref_out.sum() + rc.sum(), x)
    np.testing.assert_allclose(g1.numpy(), g2.numpy(), rtol=0, atol=5e-6)


def test_flat_gradient_segments_are_aligned():
    """Every parameter's gradient (and, on the GPU, its re-homed storage) starts on a 256-byte boundary of the flat
    buffer: kernels read them with 16-byte accesses and LDS-DMA (head / bn1 tensors used to sit at odd offsets)."""
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import FlatGradients
    model = Actor(t2onet_amd.default_options())
    fg = FlatGradients(model.parameters())
    base = fg.flat.data_ptr()
    assert all((p.grad.data_ptr() - base) % 256 == 0 for p in fg.params)
    assert all(p.grad.shape == p.shape and p.grad.stride() == p.stride() for p in fg.params)
    assert fg.flat.numel() >= sum(p.numel() for p in fg.params) and float(fg.flat.abs().sum()) == 0.0
    # segments do not overlap
    spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + 4 * p.numel()) for p in fg.params)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))


def test_beam_search_rejects_non_l1_requests_before_any_fit():
    """the batched sweep scored one-parameter operators with the L1 kernels whatever dist_type said."""
    import pytest
    from t2onet_amd import planner
    with pytest.raises(NotImplementedError):
        planner.beam_search(None, None, None, None, None, 2, [0, 1], ['brightness', 'contrast'], 1, 1e-3, dist_type='L2')
    with pytest.raises(NotImplementedError):
        planner.beam_search(None, None, None, None, object(), 2, [0, 1], ['brightness', 'contrast'], 1, 1e-3)


class _StubExecutor(torch.nn.Module):
    """Host stand-in for the Executor (the real one is GPU-only): per-operator heads 512 -> 24 like the real parameter
    heads -- a head no local sample selected gets NO gradient on that rank -- and a differentiable per-pixel edit."""

    def __init__(self):
        super().__init__()
        self.heads = torch.nn.ModuleList([torch.nn.Linear(512, 24) for _ in range(8)])

    def execute_per_sample(self, img, exec_op, mask, features=None, specified_param=None):
        B = img.shape[0]
        par = torch.zeros(B, 24)
        for op in range(8):
            sel = (exec_op.view(-1) == op).nonzero().view(-1)
            if len(sel):
                par = par.index_add(0, sel, self.heads[op](features[sel]))
        gain = 1.0 + 0.1 * torch.tanh(par[:, :1]).view(B, 1, 1, 1)
        used = (exec_op.view(-1) >= 0).view(B, 1, 1, 1)
        return torch.where(used, (img * gain).clamp(0, 1), img), par


def _episode_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import t2onet_amd
    import t2onet_amd.functional as T
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    from t2onet_amd.train_cli import sync_batchnorm_buffers
    torch.set_num_threads(2)
    # the product's L1 and attention core are GPU kernels: host stand-ins for exactly these two; everything else is the real code
    T.l1_loss = lambda a, b: (a - b).abs().mean()

    def attention_core(q, ctx):
        a = torch.softmax(torch.bmm(ctx, q.unsqueeze(2)).squeeze(2), dim=1)
        return torch.bmm(a.unsqueeze(1), ctx).squeeze(1), a
    T.attention_core = attention_core
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    torch.manual_seed(0)
    model = Actor(opt)
    model.executor = _StubExecutor()
    model.train()
    for t in list(model.parameters()) + list(model.buffers()):     # identical replicas, as bench.py / train_cli do it
        dist.broadcast(t.data, 0)
    tr = Trainer(model, opt, lr=1e-3)
    B, S = 4, 32
    img = synth.images(B, S, S, 31 + rank)                          # every rank its own shard ...
    tgt = synth.images(B, S, S, 41 + rank)
    x = synth.requests(B, 17, 51 + rank)
    torch.manual_seed(100 + rank)                                   # ... and its own sampling stream: the ranks draw different operators
    drawn = []
    orig = model._execute

    def spy(img_, ops, ctx, mask=None, exec_op=None):
        drawn.append(ops.view(-1).clone())
        return orig(img_, ops, ctx, mask, exec_op)
    model._execute = spy
    head_grad_seen = []
    for _ in range(2):
        tr.episode_step(x, img, tgt, reinforce_sample=1)
        head_grad_seen.append([bool(h.weight.grad.abs().sum() > 0) for h in model.executor.heads])
    sync_batchnorm_buffers(model, world)
    torch.save({'params': [p.detach().clone() for p in model.parameters()],
                'buffers': {k: v.clone() for k, v in model.named_buffers() if 'running' in k},
                'drawn': torch.stack(drawn), 'flat': tr.grads.flat.clone()}, out % rank)
    dist.destroy_process_group()


def test_episode_step_data_parallel_with_rank_local_operators(tmp_path):
    """World size 2 (gloo), the real Trainer.episode_step: the ranks sample DIFFERENT operators, so parameter heads are
    used on one rank only (zeros in the flat all-reduce from the other); afterwards the replicas are bit-identical, the
    all-reduced gradient buffers are bit-identical, and the batch-norm running statistics are the ranks' mean."""
    port = 31500 + os.getpid() % 2000
    out = str(tmp_path / 'ep%d.pt')
    mp.spawn(_episode_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out % 0), torch.load(out % 1)
    assert not torch.equal(a['drawn'], b['drawn'])                  # the premise: different operator sequences
    assert all(torch.equal(u, v) for u, v in zip(a['params'], b['params']))
    assert torch.equal(a['flat'], b['flat']) and float(a['flat'].abs().sum()) > 0
    assert all(torch.equal(a['buffers'][k], b['buffers'][k]) for k in a['buffers'])
    assert all(bool(torch.isfinite(p).all()) for p in a['params'])


def _jit_worker(rank, world, port, cache, out):
    """Two processes prepare the SAME new operator list at once: the hipRTC cache (compile, atomic file write, load) and --
    when a GPU is present -- the autotuner's result file must be safe under that race."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ['T2O_JIT_CACHE'] = cache
    import ctypes
    from t2onet_amd import _lib
    lib = _lib.load()
    ops = (ctypes.c_int * 4)(6, 2, 1, 0)
    rc = lib.t2o_fused_sequence_prepare(ops, 4)
    with open(out % rank, 'w') as f:
        f.write('%d %s' % (rc, lib.t2o_last_error().decode() if rc else ''))


def test_chain_specialisation_cache_is_rank_safe(tmp_path):
    """t2o_fused_sequence_prepare from two ranks at once, same cache directory: both return a status (0 with a GPU and
    hipRTC, a clean error code without a device), neither crashes, and no partial file is left behind."""
    cache = str(tmp_path / 'jit')
    os.makedirs(cache)
    out = str(tmp_path / 'jit%d.txt')
    mp.spawn(_jit_worker, args=(2, 32500 + os.getpid() % 2000, cache, out), nprocs=2, join=True)
    codes = [open(out % r).read().split(' ', 1) for r in range(2)]
    assert codes[0][0] == codes[1][0], codes                        # the same outcome on both ranks
    assert not [f for f in os.listdir(cache) if '.tmp' in f]        # atomic writes: nothing half-written remains


def test_decoder_tape_forms_each_weight_gradient_once_over_all_steps(monkeypatch):
    """decoder_step.DecoderTape host logic: with the (X, dY) pairs of n recorded steps in the tape's arrays, flush() must ADD
    exactly sum_s dY_s^T X_s (and the bias / embedding / batch-norm sums) to every parameter's .grad, skip the output projection
    when no step had a gradient through its scores, zero the slots of a step whose backward never ran, and rewind.  Checked
    against the same sums written out step by step.  (The products themselves are t2o_gemm / t2o_colsum launches -- GPU only,
    tests/test_gpu_gemm.py; here, on host tensors, the two wrappers are stood in for by their definitions.)"""
    from t2onet_amd.action_decoder import Decoder
    from t2onet_amd.decoder_step import DecoderTape
    import t2onet_amd.functional as TF

    def gemm(A, B, out=None, a_kmajor=False, b_kmajor=False, accumulate=False):
        prod = (A.t() if a_kmajor else A) @ (B if b_kmajor else B.t())
        return out.add_(prod) if accumulate else out.copy_(prod)

    def colsum(X, out=None, accumulate=False):
        return out.add_(X.sum(0)) if accumulate else out.copy_(X.sum(0))
    monkeypatch.setattr(TF, 'gemm', gemm)
    monkeypatch.setattr(TF, 'colsum', colsum)
    torch.manual_seed(1)
    B, D, E, V, K, S, n = 3, 64, 8, 11, 32, 4, 3
    dec = Decoder(V, 5, E, D // 2, 2, bidirectional=True, use_attention=True)
    fc, bn = torch.nn.Linear(K, D), torch.nn.BatchNorm1d(D)

    class Holder:
        pass
    model = Holder()
    model.decoder, model.bn1, model.vis_encoder = dec, bn, Holder()
    model.vis_encoder.fc = fc
    tape = DecoderTape(B, D, E, V, K, S, torch.device('cpu'), persistent=True)
    for name, arr in tape.a.items():
        arr.copy_(torch.randn(arr.shape))
    tape.prev_ops.copy_(torch.randint(0, V, tape.prev_ops.shape))
    params = list(dec.parameters()) + list(fc.parameters()) + list(bn.parameters())

    def run(step_done, step_logp, feat_done):
        for p in params:
            p.grad = torch.full_like(p, 0.5)                       # (flush ADDS: a recognisable start value)
        tape.begin()
        tape.n_steps = tape.n_feats = n
        tape.step_done, tape.step_logp, tape.feat_done = set(step_done), set(step_logp), set(feat_done)
        snap = {k: v.clone() for k, v in tape.a.items()}
        tape.flush(model)
        assert tape.n_steps == 0 and tape.n_feats == 0 and not tape.step_done
        return snap

    snap = run({0, 1, 2}, {0, 2}, {0, 1, 2})
    A = snap

    def acc(dy, x, steps):
        return sum(A[dy][s].t() @ A[x][s] for s in steps)
    rnn, lo = dec.rnn, dec.attention.linear_out
    all_s = range(n)
    want = {
        dec.vis_linear.weight: acc('d_vis', 'featc', all_s), dec.vis_linear.bias: sum(A['d_vis'][s].sum(0) for s in all_s),
        rnn.weight_ih_l0: acc('d_gates0', 'step_in', all_s), rnn.weight_hh_l0: acc('d_gates0', 'hp0', all_s),
        rnn.weight_ih_l1: acc('d_gates1', 'h0n', all_s), rnn.weight_hh_l1: acc('d_gates1', 'hp1', all_s),
        rnn.bias_ih_l0: sum(A['d_gates0'][s].sum(0) for s in all_s), rnn.bias_hh_l1: sum(A['d_gates1'][s].sum(0) for s in all_s),
        lo.weight: torch.cat([acc('d_lin', 'mix', all_s), acc('d_lin', 'h1n', all_s)], 1), lo.bias: sum(A['d_lin'][s].sum(0) for s in all_s),
        dec.out_linear.weight: acc('d_logits', 'ctx', (0, 2)), dec.out_linear.bias: sum(A['d_logits'][s].sum(0) for s in (0, 2)),
        fc.weight: acc('d_fc', 'pooledc', all_s), fc.bias: sum(A['d_fc'][s].sum(0) for s in all_s),
        bn.weight: A['d_bn'][:n, 0].sum(0), bn.bias: A['d_bn'][:n, 1].sum(0),
    }
    emb = torch.zeros_like(dec.embedding.weight)
    for s in all_s:
        emb.index_add_(0, tape.prev_ops[s], A['d_step_in'][s][:, :E])
    want[dec.embedding.weight] = emb
    for p, w in want.items():
        np.testing.assert_allclose((p.grad - 0.5).numpy(), w.numpy(), rtol=1e-4, atol=1e-4)
    # step 1's backward never ran, no step had a gradient through its scores, feature head 2 unused
    snap = run({0, 2}, set(), {0, 1})
    A = snap
    np.testing.assert_allclose((rnn.weight_ih_l0.grad - 0.5).numpy(), acc('d_gates0', 'step_in', (0, 2)).numpy(), rtol=1e-4, atol=1e-4)
    assert float((dec.out_linear.weight.grad - 0.5).abs().max()) == 0.0
    np.testing.assert_allclose((fc.weight.grad - 0.5).numpy(), acc('d_fc', 'pooledc', (0, 1)).numpy(), rtol=1e-4, atol=1e-4)
    assert float(tape.a['d_gates0'][1].abs().max()) == 0.0         # the unused slot was cleared, not summed
