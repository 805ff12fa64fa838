#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): bs=64 per GPU, 256x256 fp32, the six-operator executor
sequence [brightness, contrast, saturation, color-curve, tone-curve, sharpness] forward + L1
loss + backward to every operator parameter and to the input image.  Inputs are synthetic
(U[0,1) images, torch.Generator seed 10; parameters as in SURVEY.md 8(d)) and already resident
in HBM when the timed region starts.  A "step" = one such forward+L1+backward over the batch.

The step runs through the C ABI (t2o_sequence_fwd / t2o_sequence_bwd): every intermediate
image is materialised, as Executor.execute returns it, so the HBM traffic is the algorithmic
K*60*P + 12*P bytes of SURVEY.md 8(d).  N > 1: the batch dimension shards (64 images per GPU,
weak scaling), the executor path has no data-path collective; one process per
GPU under torch.distributed/RCCL, timed between barriers, MAX over ranks.

One JSON line on rank 0, with
  roofline      dominant kernel (largest share of step time): algorithmic bytes per launch /
                its mean duration, measured with HIP events on the launch stream in a second,
                per-kernel instrumented pass over the same steps
  cpu_baseline  the oracle (oracle/cpu_ref.py, eager PyTorch restatement of the reference)
                timed on this node's host cores on a bounded sample of the same workload
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OPS = [0, 1, 2, 3, 5, 6]
OP_NAMES = {0: 'brightness', 1: 'contrast', 2: 'saturation', 3: 'color', 5: 'tone', 6: 'sharpness'}
PARAM_RANGES = {0: (1, -0.3, 0.3), 1: (1, -0.3, 0.3), 2: (1, -0.3, 0.3), 3: (24, 0.5, 1.5), 5: (8, 0.5, 1.5), 6: (1, 0.0, 1.0)}
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def make_inputs(B, H, W, device, seed=10):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    tgt = torch.rand(B, 3, H, W, generator=g)
    params = torch.zeros(len(OPS), B, 24)
    for k, op in enumerate(OPS):
        n, lo, hi = PARAM_RANGES[op]
        params[k, :, :n] = torch.rand(B, n, generator=g) * (hi - lo) + lo
    return img.to(device), tgt.to(device), params.to(device)


class SequenceRunner:
    """Preallocated buffers + direct C-ABI calls (no autograd, no allocation in the step)."""

    def __init__(self, B, H, W, device):
        from t2onet_amd import _lib
        self.lib = _lib.load()
        self.check = _lib.check
        self.B, self.H, self.W, self.K = B, H, W, len(OPS)
        self.img, self.tgt, self.params = make_inputs(B, H, W, device)
        self.acts = torch.empty(self.K, B, 3, H, W, device=device)
        self.gbuf = torch.empty(2, B, 3, H, W, device=device)
        self.gimg = torch.empty(B, 3, H, W, device=device)
        self.gparams = torch.zeros(self.K, B, 24, device=device)
        self.loss = torch.zeros((), device=device)
        self.gloss = torch.ones((), device=device)
        self.ws = torch.empty(self.lib.t2o_workspace_bytes(B, H, W), dtype=torch.uint8, device=device)
        self.c_ops = (ctypes.c_int * self.K)(*OPS)

    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def step(self):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        B, H, W = self.B, self.H, self.W
        rc = self.lib.t2o_sequence_fwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                       self._p(self.acts), self._p(self.loss), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_sequence_fwd')
        rc = self.lib.t2o_sequence_bwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                       self._p(self.acts), self._p(self.gloss), self._p(self.gimg), self._p(self.gparams),
                                       self._p(self.gbuf), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_sequence_bwd')

    def profiled_step(self, rec):
        """The same launches, one C call per operator, each bracketed by HIP events recorded on
        the launch stream (torch's current stream)."""
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        B, H, W, K = self.B, self.H, self.W, self.K
        wsn = self.ws.numel()

        def timed(name, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.check(fn(), name)
            e1.record()
            rec.setdefault(name, []).append((e0, e1))

        cur = self.img
        for k, op in enumerate(OPS):
            out, p = self.acts[k], self.params[k]
            if k == K - 1:
                timed('fwd_%s+l1' % OP_NAMES[op], lambda: self.lib.t2o_op_fwd_l1(
                    op, self._p(cur), self._p(p), 24, None, 0, self._p(self.tgt), self._p(out), self._p(self.loss),
                    self._p(self.ws), wsn, B, H, W, st))
            else:
                timed('fwd_%s' % OP_NAMES[op], lambda: self.lib.t2o_op_fwd(
                    op, self._p(cur), self._p(p), 24, None, 0, self._p(out), B, H, W, st))
            cur = out
        gcur = None
        for k in range(K - 1, -1, -1):
            op = OPS[k]
            inp = self.img if k == 0 else self.acts[k - 1]
            gnext = self.gimg if k == 0 else self.gbuf[k & 1]
            p, gp = self.params[k], self.gparams[k]
            if k == K - 1:
                timed('bwd_%s+l1' % OP_NAMES[op], lambda: self.lib.t2o_op_bwd_l1(
                    op, self._p(inp), self._p(p), 24, None, 0, self._p(self.tgt), self._p(self.gloss), self._p(gnext),
                    self._p(gp), 24, self._p(self.ws), wsn, B, H, W, st))
            else:
                timed('bwd_%s' % OP_NAMES[op], lambda: self.lib.t2o_op_bwd(
                    op, self._p(inp), self._p(p), 24, None, 0, self._p(gcur), self._p(gnext), self._p(gp), 24,
                    self._p(self.ws), wsn, B, H, W, st))
            gcur = gnext


class FusedRunner(SequenceRunner):
    """Same workload through t2o_fused_sequence_fwd/bwd: the five per-pixel operators run in
    registers in one kernel pair, sharpness (+L1) in its stencil pair; only the image before the
    sharpness is materialised.  Same loss and gradients (tests/test_gpu_operators.py)."""

    def __init__(self, B, H, W, device):
        super().__init__(B, H, W, device)
        self.nbuf = self.lib.t2o_fused_sequence_buffers(self.c_ops, self.K)
        self.seg = torch.empty(max(self.nbuf, 1), B, 3, H, W, device=device)
        self.out = torch.empty(B, 3, H, W, device=device)
        self.acts = None                               # not needed: free the K materialised images
        self.c_chain = (ctypes.c_int * 5)(*OPS[:5])

    def step(self):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        B, H, W = self.B, self.H, self.W
        rc = self.lib.t2o_fused_sequence_fwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                             self._p(self.out), self._p(self.loss), self._p(self.seg), self._p(self.ws),
                                             self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_fused_sequence_fwd')
        rc = self.lib.t2o_fused_sequence_bwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                             self._p(self.gloss), None, self._p(self.gimg), self._p(self.gparams),
                                             self._p(self.seg), self._p(self.gbuf), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_fused_sequence_bwd')

    def profiled_step(self, rec):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        B, H, W = self.B, self.H, self.W
        wsn = self.ws.numel()

        def timed(name, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.check(fn(), name)
            e1.record()
            rec.setdefault(name, []).append((e0, e1))

        mid, p6, gp6 = self.seg[0], self.params[5], self.gparams[5]
        timed('fwd_chain5', lambda: self.lib.t2o_fused_sequence_fwd(
            self.c_chain, 5, self._p(self.img), self._p(self.params), None, self._p(mid), None, None,
            self._p(self.ws), wsn, B, H, W, st))
        timed('fwd_sharpness+l1', lambda: self.lib.t2o_op_fwd_l1(
            6, self._p(mid), self._p(p6), 24, None, 0, self._p(self.tgt), self._p(self.out), self._p(self.loss),
            self._p(self.ws), wsn, B, H, W, st))
        timed('bwd_sharpness+l1', lambda: self.lib.t2o_op_bwd_l1(
            6, self._p(mid), self._p(p6), 24, None, 0, self._p(self.tgt), self._p(self.gloss), self._p(self.gbuf[0]),
            self._p(gp6), 24, self._p(self.ws), wsn, B, H, W, st))
        timed('bwd_chain5', lambda: self.lib.t2o_fused_sequence_bwd(
            self.c_chain, 5, self._p(self.img), self._p(self.params), None, None, self._p(self.gbuf[0]),
            self._p(self.gimg), self._p(self.gparams), None, None, self._p(self.ws), wsn, B, H, W, st))


def api_path_bench(B, H, W, device, steps, warmup):
    """The same step through the drop-in Python surface: six Executor.execute calls + l1_loss +
    torch autograd backward (ctypes calls into the C ABI from autograd Functions)."""
    import t2onet_amd
    import t2onet_amd.functional as T
    ex = t2onet_amd.Executor(t2onet_amd.default_options()).to(device)
    img, tgt, params = make_inputs(B, H, W, device)
    x = img.clone().requires_grad_(True)
    ps = [params[k, :, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(OPS)]

    def step():
        x.grad = None
        for p in ps:
            p.grad = None
        cur = x
        for op, p in zip(OPS, ps):
            cur, _ = ex.execute(cur, op, None, specified_param=p)
        loss = T.l1_loss(cur, tgt)
        loss.backward()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'what': 'six Executor.execute calls + l1_loss + autograd backward (Python API path)',
            'value': round(B * steps / dt, 1), 'unit': 'images/sec', 'ms_per_step': round(dt / steps * 1e3, 4),
            'loss': float(loss.item())}


def algorithmic_bytes(name, P):
    """operator forward 24 B/pixel, backward 36 B/pixel, +12 B/pixel (target)
    for the forward fused with the L1 loss (its backward reads the target instead of gout).
    A fused launch is credited with the operator applications it performs (5 for chain5)."""
    n = 5 if 'chain5' in name else 1
    if name.startswith('fwd'):
        return (24 * n + (12 if name.endswith('+l1') else 0)) * P
    return 36 * n * P


def hbm_min_bytes(name, P):
    """Bytes the launch must move whatever the fusion: read input (+target/gout), write output."""
    if name.startswith('fwd'):
        return (24 + (12 if name.endswith('+l1') else 0)) * P
    return 36 * P


def pmc_traffic(kernel, P):
    """HBM bytes per launch from the rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    separate passes: tools/gpu_pmc.sh), stored per pixel in profiles/pmc_traffic.json.  bench.py
    cannot run the profiler on itself, so this is the last committed measurement, not a live one."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            per_px = json.load(f)['bytes_per_pixel'].get(kernel)
        return None if per_px is None else int(per_px * P)
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(B_sample, H, W, reps=7):
    """Oracle timed on the host cores: same sequence, fwd + L1 + bwd, on a bounded sample."""
    from oracle import cpu_ref
    img, tgt, params = make_inputs(B_sample, H, W, 'cpu')
    opt = cpu_ref.default_opt()

    def once():
        x = img.clone().requires_grad_(True)
        ps = [params[k, :, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(OPS)]
        out, _ = cpu_ref.run_sequence(x, OPS, ps, opt)
        cpu_ref.l1_loss(out, tgt).backward()

    # eager ATen ops on 50 MB tensors do not scale to every core of a big host (256 threads ran
    # 50x slower than 32 on the first MI355X node): pick the thread count on a short bs=8 probe
    ncpu = os.cpu_count() or 1
    pimg, ptgt, pparams = img[:8], tgt[:8], params[:, :8]

    def probe():
        x = pimg.clone().requires_grad_(True)
        ps = [pparams[k, :, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(OPS)]
        out, _ = cpu_ref.run_sequence(x, OPS, ps, opt)
        cpu_ref.l1_loss(out, ptgt).backward()

    best = None
    for nt in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(nt)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
        elif dt > 1.5 * best[0]:
            break                                   # past the knee: more threads only get slower
    torch.set_num_threads(best[1])
    once()                                          # warm-up at the chosen thread count
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        once()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    res = {'value': B_sample / med, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': 'oracle/cpu_ref.py eager fp32, same 6-op sequence fwd+L1+bwd, bs=%d %dx%d, median of %d reps '
                     '(%.2f s each)' % (B_sample, H, W, reps, med)}
    # parity gate reported with the number: the GPU paths against the oracle on the first 4 images
    try:
        import t2onet_amd
        n = min(4, B_sample)
        x = img[:n].clone().requires_grad_(True)
        ps = [params[k, :n, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(OPS)]
        ref, _ = cpu_ref.run_sequence(x, OPS, ps, opt)
        ref_loss = cpu_ref.l1_loss(ref, tgt[:n])
        ref_loss.backward()
        ex = t2onet_amd.Executor(t2onet_amd.default_options()).cuda()
        xg = img[:n].cuda().requires_grad_(True)
        pg = params[:, :n].cuda().requires_grad_(True)
        loss, out = ex.run_sequence_fused(xg, OPS, pg, tgt[:n].cuda())
        loss.backward()
        gerr = (xg.grad.cpu() - x.grad).abs().max().item() / max(x.grad.abs().max().item(), 1e-30)
        res['parity'] = {'images': n, 'fwd_max_abs_err': (out.detach().cpu() - ref.detach()).abs().max().item(),
                         'loss_abs_dev': abs(loss.item() - ref_loss.item()),
                         'gimg_max_err_rel_to_max': gerr, 'tolerance': 1e-5}
    except Exception as e:                     # noqa: BLE001
        res['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return res


def train_step_bench(device, dist, world, B, H, W, steps, warmup):
    """BASELINE.json configs[2]/[3]: the episode/L1 train step of train_seq2seqL1.py:74-88 (request
    encoder + 5 x (ResNet features + attention decoder step + sampled per-sample operator) +
    END-image select + L1 + backward + one flat gradient all-reduce + Adam), FiveK-shaped synthetic
    batch, random-init weights, fp32."""
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    opt = t2onet_amd.default_options()
    torch.manual_seed(10 + (dist.get_rank() if dist is not None else 0))
    model = Actor(opt).to(device).train()
    # (NHWC: the ResNet alone is 20 % faster, tools/bench_resnet.py, but the whole step measured
    # slower -- 122 vs 94 ms -- so the default stays NCHW; Actor.use_channels_last() switches)
    if dist is not None:                                   # identical replicas
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    # hipGraph capture of the image encoder: on for the single-process run.  With a process group alive the
    # RCCL watchdog thread can touch the device during a (global-mode) capture and invalidate it on one rank
    # only, which would leave the other ranks waiting in the gradient all-reduce -- not worth risking in a
    # benchmark that cannot be rehearsed on this pool; T2O_GRAPH_ENCODER=1 forces it.
    want_graph = os.environ.get('T2O_GRAPH_ENCODER', '1' if dist is None else '0') != '0'
    tr = Trainer(model, opt, graph_encoder=want_graph)
    g = torch.Generator().manual_seed(10)
    img = torch.rand(B, 3, H, W, generator=g).to(device)
    tgt = torch.rand(B, 3, H, W, generator=g).to(device)
    n = torch.randint(1, 16, (B,), generator=g)
    x = torch.zeros(B, 17, dtype=torch.long)
    for b in range(B):
        k = int(n[b])
        x[b, 0] = 1
        x[b, 1:1 + k] = torch.randint(4, 918, (k,), generator=g)
        x[b, 1 + k] = 2
    lengths = (x != 0).sum(1)
    x = x.to(device)
    for _ in range(warmup):
        tr.episode_step(x, img, tgt, lengths=lengths)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.episode_step(x, img, tgt, lengths=lengths)
    t_enq = time.perf_counter() - t0                       # host time to enqueue (includes the step's own host syncs)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {'images_per_sec': round(world * B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 2),
            'host_enqueue_ms_per_step': round(t_enq / steps * 1e3, 2),
            'steps': steps, 'warmup': warmup, 'global_batch': world * B, 'loss': float(loss.item()),
            'workload': 'episode/L1 train step (train_seq2seqL1.py:74-88), bs=%d/GPU %dx%d fp32, sampled ops, '
                        'flat-gradient all-reduce (%d ranks) + Adam' % (B, H, W, world)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=64)
    ap.add_argument('--train-steps', type=int, default=3,
                    help='also time this many full episode/L1 train steps (BASELINE configs[2]/[3], with the flat '
                         'gradient all-reduce when N > 1) -> "train_step"; 0 to skip')
    ap.add_argument('--train-warmup', type=int, default=2)
    ap.add_argument('--train-timeout', type=int, default=420, help='seconds before the train-step leg is abandoned')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist = None
    if world > 1 or 'RANK' in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    B, H, W = args.batch, args.size, args.size
    P = B * H * W

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(run):
        for _ in range(args.warmup):
            run.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        # second pass: the same launches, one HIP-event pair per kernel pair
        rec = {}
        for _ in range(args.steps):
            run.profiled_step(rec)
        torch.cuda.synchronize()
        kernels = {}
        for name, evs in rec.items():
            ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            by = algorithmic_bytes(name, P)
            kernels[name] = {'ms': round(ms, 5), 'algorithmic_MB': round(by / 1e6, 2), 'GBps': round(by / ms / 1e6, 1),
                             'hbm_min_MB': round(hbm_min_bytes(name, P) / 1e6, 2),
                             'hbm_min_GBps': round(hbm_min_bytes(name, P) / ms / 1e6, 1)}
        return el, kernels, float(run.loss.item())

    mat_elapsed, mat_kernels, mat_loss = measure(SequenceRunner(B, H, W, device))
    torch.cuda.empty_cache()
    elapsed, kernels, loss_value = measure(FusedRunner(B, H, W, device))
    dom = max(kernels, key=lambda n: kernels[n]['ms'])
    total_bytes = (6 * 60 + 12) * P                      # K*60*P + 12*P
    sum_ms = sum(k['ms'] for k in kernels.values())

    api = None
    try:
        api = api_path_bench(B, H, W, device, args.steps, args.warmup)
    except Exception as e:                     # noqa: BLE001
        api = {'error': '%s: %s' % (type(e).__name__, e)}
    torch.cuda.empty_cache()

    def emit(train):
        if rank != 0:
            return
        ms_per_step = elapsed / args.steps * 1e3
        n_gpus = world
        line = {
            'metric': 'images/sec (executor step: 6-op sequence fwd + L1 + bwd, bs=64/GPU, 256x256 fp32)',
            'value': round(n_gpus * B * args.steps / elapsed, 1),
            'unit': 'images/sec',
            'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[1]: bs=%d/GPU %dx%d fp32, executor ops %s forward + L1 + '
                                   'backward to all parameters and the image, via t2o_fused_sequence_fwd/bwd (C ABI): '
                                   'per-pixel operators fused in registers, sharpness+L1 stencil kernels' % (B, H, W, OPS),
                       'global_batch': n_gpus * B, 'parallelism': 'batch shards, no collective on the executor path'},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': kernels[dom]['GBps'], 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(kernels[dom]['GBps'] / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(dom, P),
                         'algorithmic_bytes_per_launch': algorithmic_bytes(dom, P),
                         'avg_launch_ms': kernels[dom]['ms'],
                         'note': 'algorithmic bytes = SURVEY 8(d) per-operator figure x operator applications in the '
                                 'launch (materialised accounting); a fused launch moves only hbm_min bytes, so frac can '
                                 'exceed what HBM alone allows: fused_min_* give the fraction on the bytes it must move',
                         'fused_min_bytes_per_launch': hbm_min_bytes(dom, P),
                         'fused_min_achieved': kernels[dom]['hbm_min_GBps'],
                         'fused_min_frac': round(kernels[dom]['hbm_min_GBps'] / HBM_PEAK_GBS, 4)},
            'step_roofline': {'algorithmic_GB_per_step': round(total_bytes / 1e9, 4),
                              'achieved_GBps_whole_step': round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              'frac_of_peak': round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              'sum_kernel_ms': round(sum_ms, 4)},
            'kernels': kernels,
            'loss': loss_value,
            'materialised_path': {
                'what': 'same step through t2o_sequence_fwd/bwd: one kernel pair per operator, all 6 intermediates in '
                        'HBM (what 6 Executor.execute calls + autograd do); HBM traffic = the algorithmic 1.56 GB',
                'value': round(n_gpus * B * args.steps / mat_elapsed, 1), 'unit': 'images/sec',
                'ms_per_step': round(mat_elapsed / args.steps * 1e3, 4),
                'achieved_GBps_whole_step': round(total_bytes / (mat_elapsed / args.steps) / 1e9, 1),
                'frac_of_peak': round(total_bytes / (mat_elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                'loss': mat_loss, 'kernels': mat_kernels},
        }
        if api is not None:
            line['executor_api_path'] = api
        if train is not None:
            line['train_step'] = train
        if not args.no_cpu_baseline and n_gpus == 1:
            line['cpu_baseline'] = cpu_baseline(args.cpu_sample, H, W)
        print(json.dumps(line))

    train = None
    if args.train_steps > 0:
        # secondary measurement: it must never take the headline line down.  Exceptions are reported in the
        # line; a hang (e.g. one rank failing inside a collective) is cut by an alarm that still prints the line.
        # (a timer THREAD: a signal handler would not run while the main thread sits in a blocking device call)
        import threading

        def on_timeout():
            emit({'error': 'train step did not finish within %d s' % args.train_timeout})
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(args.train_timeout, on_timeout)
        watchdog.daemon = True
        watchdog.start()
        try:
            train = train_step_bench(device, dist, world, B, H, W, args.train_steps, args.train_warmup)
        except Exception as e:                 # noqa: BLE001
            train = {'error': '%s: %s' % (type(e).__name__, e)}
        watchdog.cancel()
    emit(train)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
