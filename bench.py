#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (`value`): images/sec of the episode/L1 TRAIN STEP (BASELINE.json metric; configs[2] at
N=1, configs[3] at N=8): request encoder + 5 x (ResNet image features + attention decoder step +
sampled per-sample operator) + END-image select + L1 + backward + ONE flat fp32 gradient
all-reduce (RCCL) + Adam, bs=64 per GPU, 256x256 fp32, FiveK-shaped synthetic batch, random-init
weights (experiments/t2onet/train_seq2seqL1.py:74-88).  W untimed warm-up steps, then exactly K
timed steps between barrier + synchronize on both sides, MAX over ranks.

N > 1: when RANK is not in the environment this process is only a LAUNCHER -- before touching a
GPU it checks that N devices exist (exit 2 otherwise, never a silent 1-GPU run), starts N child
ranks of this script (one per GPU, rendezvous on 127.0.0.1) and relays rank 0's JSON line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is a rank directly.

The same line carries the op-pipeline half of the metric ("GB/s vs HBM roofline"):
  executor.cfg2_bs64     BASELINE configs[1]: 6-op executor sequence fwd + L1 + bwd, bs=64 256x256
  executor.bs256         the same at bs=256 (the north-star >= 60 % target size)
  executor.cfg5_16x512   configs[4] per-GPU shape: 16 x 512x512, ops [5,3,5,3,0,1,2,6]
each through the fused C-ABI path and the fully materialised one, with per-kernel HIP-event
timings.  `roofline` = the dominant hand-written kernel of the cfg2 fused step (HBM);
`train_roofline` = the train step against the fp32 matrix peak (4.34 TFLOP per bs=64 step).
`cpu_baseline` = the oracle (oracle/cpu_ref.py) on this node's host cores for configs[0], [1]
and (bounded sample) [2], N=1 only.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2_OPS = [0, 1, 2, 3, 5, 6]
CFG5_OPS = [5, 3, 5, 3, 0, 1, 2, 6]
CFG2_GENERIC_OPS = [2, 0, 3, 1, 5, 6]        # configs[1]'s six operators in another order: no benchmark-specific code path
OP_NAMES = {0: 'brightness', 1: 'contrast', 2: 'saturation', 3: 'color', 5: 'tone', 6: 'sharpness'}
PARAM_RANGES = {0: (1, -0.3, 0.3), 1: (1, -0.3, 0.3), 2: (1, -0.3, 0.3), 3: (24, 0.5, 1.5), 5: (8, 0.5, 1.5), 6: (1, 0.0, 1.0)}
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MATRIX_PEAK_TF = 157.3     # same guide: v_mfma_f32_* = the fp32 vector rate
TRAIN_FLOP_PER_IMAGE = 67.8e9   # 5 x ResNet fwd+bwd at 256x256 (scales with H*W)


# ---------------------------------------------------------------------------------------------
# launcher (no GPU call in this process)
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv):
    """Start args.gpus ranks of this script and relay rank 0's stdout.  Exit code: 2 if the node has
    fewer devices than asked for, else the worst child's."""
    import torch
    n = args.gpus
    if not args.selftest:
        have = torch.cuda.device_count()            # does not initialise the GPU runtime on this image
        if have < n:
            sys.stderr.write('bench.py: --gpus %d but only %d device(s) visible; refusing to run on fewer\n' % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out,
                                      start_new_session=True))
    deadline = time.time() + args.launch_timeout
    line, rc = None, 0
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        # poll: a rank that dies takes the others into a collective that never completes -- stop them at once instead
        # of waiting for the communicator's own timeout
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                rc = max(abs(c) for c in codes)
                break
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = abs(bad[0][1])
                sys.stderr.write('bench.py: rank %d exited with code %d; stopping the other ranks\n' % bad[0])
                break
            if time.time() > deadline:
                rc = 124
                sys.stderr.write('bench.py: ranks did not finish within %d s; stopping them\n' % args.launch_timeout)
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)             # the exact process groups started above
                except OSError:
                    pass
    reader.join(timeout=10)
    for ln in (out0[0] if out0 else b'').decode('utf-8', 'replace').splitlines():
        if ln.startswith('{'):
            line = ln
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------
# executor legs (C ABI, preallocated buffers, no autograd)
# ---------------------------------------------------------------------------------------------
def make_inputs(ops, B, H, W, device, seed=10):
    import torch
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    tgt = torch.rand(B, 3, H, W, generator=g)
    params = torch.zeros(len(ops), B, 24)
    for k, op in enumerate(ops):
        n, lo, hi = PARAM_RANGES[op]
        params[k, :, :n] = torch.rand(B, n, generator=g) * (hi - lo) + lo
    return img.to(device), tgt.to(device), params.to(device)


class SequenceRunner:
    """t2o_sequence_fwd/bwd: one kernel pair per operator, every intermediate in HBM (what K
    Executor.execute calls + autograd do); HBM traffic = the algorithmic K*60*P + 12*P bytes."""

    def __init__(self, ops, B, H, W, device):
        import torch
        from t2onet_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self.check = _lib.check
        self.ops = list(ops)
        self.B, self.H, self.W, self.K = B, H, W, len(ops)
        self.img, self.tgt, self.params = make_inputs(ops, B, H, W, device)
        self.acts = torch.empty(self.K, B, 3, H, W, device=device)
        self.gbuf = torch.empty(2, B, 3, H, W, device=device)
        self.gimg = torch.empty(B, 3, H, W, device=device)
        self.gparams = torch.zeros(self.K, B, 24, device=device)
        self.loss = torch.zeros((), device=device)
        self.gloss = torch.ones((), device=device)
        self.ws = torch.empty(self.lib.t2o_workspace_bytes(B, H, W), dtype=torch.uint8, device=device)
        self.c_ops = (ctypes.c_int * self.K)(*ops)

    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _timed(self, rec, name, fn):
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        self.check(fn(), name)
        e1.record()
        rec.setdefault(name, []).append((e0, e1))

    def step(self):
        st, B, H, W = self._stream(), self.B, self.H, self.W
        rc = self.lib.t2o_sequence_fwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                       self._p(self.acts), self._p(self.loss), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_sequence_fwd')
        rc = self.lib.t2o_sequence_bwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                       self._p(self.acts), self._p(self.gloss), self._p(self.gimg), self._p(self.gparams),
                                       self._p(self.gbuf), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_sequence_bwd')

    def profiled_step(self, rec):
        """The same launches, one C call per operator, each bracketed by HIP events recorded on the launch
        stream (torch's current stream).  Repeated operators are numbered (tone, tone#2)."""
        st, B, H, W, K = self._stream(), self.B, self.H, self.W, self.K
        wsn = self.ws.numel()
        seen, names = {}, []
        for op in self.ops:
            seen[op] = seen.get(op, 0) + 1
            names.append(OP_NAMES[op] + ('' if seen[op] == 1 else '#%d' % seen[op]))
        cur = self.img
        for k, op in enumerate(self.ops):
            out, p = self.acts[k], self.params[k]
            if k == K - 1:
                self._timed(rec, 'fwd_%s+l1' % names[k], lambda: self.lib.t2o_op_fwd_l1(
                    op, self._p(cur), self._p(p), 24, None, 0, self._p(self.tgt), self._p(out), self._p(self.loss),
                    self._p(self.ws), wsn, B, H, W, st))
            else:
                self._timed(rec, 'fwd_%s' % names[k], lambda: self.lib.t2o_op_fwd(
                    op, self._p(cur), self._p(p), 24, None, 0, self._p(out), B, H, W, st))
            cur = out
        gcur = None
        for k in range(K - 1, -1, -1):
            op = self.ops[k]
            inp = self.img if k == 0 else self.acts[k - 1]
            gnext = self.gimg if k == 0 else self.gbuf[k & 1]
            p, gp = self.params[k], self.gparams[k]
            if k == K - 1:
                self._timed(rec, 'bwd_%s+l1' % names[k], lambda: self.lib.t2o_op_bwd_l1(
                    op, self._p(inp), self._p(p), 24, None, 0, self._p(self.tgt), self._p(self.gloss), self._p(gnext),
                    self._p(gp), 24, self._p(self.ws), wsn, B, H, W, st))
            else:
                self._timed(rec, 'bwd_%s' % names[k], lambda: self.lib.t2o_op_bwd(
                    op, self._p(inp), self._p(p), 24, None, 0, self._p(gcur), self._p(gnext), self._p(gp), 24,
                    self._p(self.ws), wsn, B, H, W, st))
            gcur = gnext


class FusedRunner(SequenceRunner):
    """Same workload through t2o_fused_sequence_fwd/bwd: the leading per-pixel operators run in registers in
    one kernel pair, the final sharpness (+L1) in its stencil pair; only the image before the sharpness is
    materialised.  Same loss and gradients (tests/test_gpu_operators.py)."""

    def __init__(self, ops, B, H, W, device):
        import torch
        super().__init__(ops, B, H, W, device)
        assert ops[-1] == 6 and 6 not in ops[:-1] and len(ops) - 1 <= 8, 'bench sequences end with one sharpness'
        rc = self.lib.t2o_fused_sequence_prepare(self.c_ops, self.K)      # run-time specialisation of lists without an ahead-of-time kernel
        self.specialised = rc == 0
        self.nbuf = self.lib.t2o_fused_sequence_buffers(self.c_ops, self.K)
        self.seg = torch.empty(max(self.nbuf, 1), B, 3, H, W, device=device)
        self.out = torch.empty(B, 3, H, W, device=device)
        self.acts = None                               # not needed: free the K materialised images
        self.nc = self.K - 1
        self.c_chain = (ctypes.c_int * self.nc)(*ops[:-1])

    def step(self):
        st, B, H, W = self._stream(), self.B, self.H, self.W
        rc = self.lib.t2o_fused_sequence_fwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                             self._p(self.out), self._p(self.loss), self._p(self.seg), self._p(self.ws),
                                             self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_fused_sequence_fwd')
        rc = self.lib.t2o_fused_sequence_bwd(self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt),
                                             self._p(self.gloss), None, self._p(self.gimg), self._p(self.gparams),
                                             self._p(self.seg), self._p(self.gbuf), self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_fused_sequence_bwd')

    def profiled_step(self, rec):
        st, B, H, W, nc = self._stream(), self.B, self.H, self.W, self.nc
        wsn = self.ws.numel()
        mid, p6, gp6 = self.seg[0], self.params[nc], self.gparams[nc]
        self._timed(rec, 'fwd_chain%d' % nc, lambda: self.lib.t2o_fused_sequence_fwd(
            self.c_chain, nc, self._p(self.img), self._p(self.params), None, self._p(mid), None, None,
            self._p(self.ws), wsn, B, H, W, st))
        self._timed(rec, 'fwd_sharpness+l1', lambda: self.lib.t2o_op_fwd_l1(
            6, self._p(mid), self._p(p6), 24, None, 0, self._p(self.tgt), self._p(self.out), self._p(self.loss),
            self._p(self.ws), wsn, B, H, W, st))
        self._timed(rec, 'bwd_sharpness+l1', lambda: self.lib.t2o_op_bwd_l1(
            6, self._p(mid), self._p(p6), 24, None, 0, self._p(self.tgt), self._p(self.gloss), self._p(self.gbuf[0]),
            self._p(gp6), 24, self._p(self.ws), wsn, B, H, W, st))
        self._timed(rec, 'bwd_chain%d' % nc, lambda: self.lib.t2o_fused_sequence_bwd(
            self.c_chain, nc, self._p(self.img), self._p(self.params), None, None, self._p(self.gbuf[0]),
            self._p(self.gimg), self._p(self.gparams), None, None, self._p(self.ws), wsn, B, H, W, st))


class ValueGradRunner(FusedRunner):
    """Same workload, same outputs (loss, image gradient, parameter gradients) through ONE call,
    t2o_fused_sequence_l1_value_grad: the last segment's forward is not launched -- the sharpness backward forms the
    final pixel anyway and also emits the loss.  3 launches per step instead of 4 (tests/test_gpu_operators.py:
    gradients bit-identical to the two-call path)."""

    def step(self):
        st, B, H, W = self._stream(), self.B, self.H, self.W
        rc = self.lib.t2o_fused_sequence_l1_value_grad(
            self.c_ops, self.K, self._p(self.img), self._p(self.params), self._p(self.tgt), self._p(self.gloss), None,
            self._p(self.loss), self._p(self.gimg), self._p(self.gparams), self._p(self.seg), self._p(self.gbuf),
            self._p(self.ws), self.ws.numel(), B, H, W, st)
        self.check(rc, 't2o_fused_sequence_l1_value_grad')

    def profiled_step(self, rec):
        st, B, H, W, nc = self._stream(), self.B, self.H, self.W, self.nc
        wsn = self.ws.numel()
        mid, p6, gp6 = self.seg[0], self.params[nc], self.gparams[nc]
        c6 = (ctypes.c_int * 1)(6)
        self._timed(rec, 'fwd_chain%d' % nc, lambda: self.lib.t2o_fused_sequence_fwd(
            self.c_chain, nc, self._p(self.img), self._p(self.params), None, self._p(mid), None, None,
            self._p(self.ws), wsn, B, H, W, st))
        self._timed(rec, 'bwd_sharpness+l1+loss', lambda: self.lib.t2o_fused_sequence_l1_value_grad(
            c6, 1, self._p(mid), self._p(p6), self._p(self.tgt), self._p(self.gloss), None, self._p(self.loss),
            self._p(self.gbuf[0]), self._p(gp6), None, None, self._p(self.ws), wsn, B, H, W, st))
        self._timed(rec, 'bwd_chain%d' % nc, lambda: self.lib.t2o_fused_sequence_bwd(
            self.c_chain, nc, self._p(self.img), self._p(self.params), None, None, self._p(self.gbuf[0]),
            self._p(self.gimg), self._p(self.gparams), None, None, self._p(self.ws), wsn, B, H, W, st))


def _chain_len(name):
    i = name.find('chain')
    return int(name[i + 5:]) if i >= 0 else 1


def algorithmic_bytes(name, P):
    """operator forward 24 B/pixel, backward 36 B/pixel, +12 B/pixel (target) for the forward
    fused with the L1 loss (its backward reads the target instead of gout).  A fused launch is credited with
    the operator applications it performs (n for chain<n>)."""
    n = _chain_len(name)
    if name.startswith('fwd'):
        return (24 * n + (12 if name.endswith('+l1') else 0)) * P
    return 36 * n * P


def hbm_min_bytes(name, P):
    """Bytes the launch must move whatever the fusion: read input (+target/gout), write output."""
    if name.startswith('fwd'):
        return (24 + (12 if name.endswith('+l1') else 0)) * P
    return 36 * P


def pmc_traffic(kernel, P):
    """HBM bytes per launch from the rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    passes: tools/gpu_pmc.sh), stored per pixel in profiles/pmc_traffic.json together with the digest of the
    library they were measured on.  bench.py cannot run the profiler on itself; the figure is reported only
    when that digest is the digest of the library being benchmarked, else null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        from t2onet_amd import build
        with open(path) as f:
            d = json.load(f)
        if d.get('lib_digest') != build.source_digest():
            return None
        per_px = d['bytes_per_pixel'].get(kernel)
        return None if per_px is None else int(per_px * P)
    except (OSError, ValueError, KeyError):
        return None


def pmc_launch_traffic(kernel):
    """HBM bytes per launch of a kernel measured at bench.py's own shapes (profiles/pmc_traffic.json 'bytes_per_launch'),
    under the same digest rule as pmc_traffic()."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        from t2onet_amd import build
        with open(path) as f:
            d = json.load(f)
        if d.get('lib_digest') != build.source_digest():
            return None
        per = d.get('bytes_per_launch', {})
        v = per.get(kernel)
        if v is None:            # the profiler prints defaulted template arguments too: k_conv3x3_fwd<2, 1> is "<2, 1, false>" there
            v = next((b for k, b in sorted(per.items()) if k.startswith(kernel[:-1] + ',') and 'true' not in k), None)
        return None if v is None else int(v)
    except (OSError, ValueError, KeyError):
        return None


def executor_leg(ctx, ops, B, H, W, steps, warmup, with_api=False):
    """Fused + materialised timing of one executor configuration on this rank's GPU; throughput aggregated
    over ranks (batch shards, no collective: SURVEY 8(e))."""
    import torch
    dist, world, device = ctx['dist'], ctx['world'], ctx['device']
    P, K = B * H * W, len(ops)
    total_bytes = (K * 60 + 12) * P                     # K*60*P + 12*P

    def measure(run):
        for _ in range(warmup):
            run.step()
        ctx['barrier']()
        t0 = time.perf_counter()
        for _ in range(steps):
            run.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        el = ctx['max_over_ranks'](time.perf_counter() - t0)
        rec = {}
        for _ in range(min(steps, 50)):
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

    fused_min_bytes = 36 * P                            # read image + target, write the image gradient, nothing else

    def summary(el, kernels, loss):
        ms = el / steps * 1e3
        return {'value': round(world * B * steps / el, 1), 'unit': 'images/sec', 'ms_per_step': round(ms, 4),
                'achieved_GBps_whole_step': round(total_bytes / (ms * 1e-3) / 1e9, 1),
                'frac_of_peak': round(total_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # the whole sequence against the FUSED minimum (36 B/pixel for forward + L1 + backward, no intermediate
                # ever written): the physical figure beside the credited one above -- it cannot exceed 1
                'fused_min_MB': round(fused_min_bytes / 1e6, 2),
                'fused_min_frac': round(fused_min_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'sum_kernel_ms': round(sum(k['ms'] for k in kernels.values()), 4), 'loss': loss, 'kernels': kernels}

    mat = summary(*measure(SequenceRunner(ops, B, H, W, device)))
    torch.cuda.empty_cache()
    fused_runner = FusedRunner(ops, B, H, W, device)
    fus = summary(*measure(fused_runner))
    fus['compile_time_specialised'] = bool(fused_runner.specialised)
    del fused_runner
    torch.cuda.empty_cache()
    vg_runner = ValueGradRunner(ops, B, H, W, device)
    vg = summary(*measure(vg_runner))
    del vg_runner
    torch.cuda.empty_cache()
    res = {'workload': 'bs=%d/GPU %dx%d fp32, executor ops %s forward + L1 + backward to all parameters and the image'
                       % (B, H, W, list(ops)),
           'steps': steps, 'warmup': warmup, 'algorithmic_GB_per_step': round(total_bytes / 1e9, 4),
           'fused': fus, 'value_grad': vg, 'materialised': mat}
    res['value_grad']['what'] = ('t2o_fused_sequence_l1_value_grad: loss + all gradients in one call, no forward launch for the '
                                 'last segment (its backward also emits the loss): 3 launches instead of 4, same gradients bit for bit')
    res['fused']['what'] = ('t2o_fused_sequence_fwd/bwd: per-pixel operators fused in registers, sharpness+L1 stencil '
                            'kernels; frac_of_peak uses the materialised algorithmic bytes (SURVEY 8(d)), i.e. it '
                            'includes fusion credit')
    res['materialised']['what'] = 't2o_sequence_fwd/bwd: one kernel pair per operator, every intermediate in HBM'
    if with_api:
        try:
            res['api_path'] = api_path_bench(ops, B, H, W, device, steps, warmup)
        except Exception as e:                     # noqa: BLE001
            res['api_path'] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return res


def api_path_bench(ops, B, H, W, device, steps, warmup):
    """The same step through the drop-in Python surface: K Executor.execute calls + l1_loss + torch autograd
    backward (ctypes calls into the C ABI from autograd Functions)."""
    import torch
    import t2onet_amd
    import t2onet_amd.functional as T
    ex = t2onet_amd.Executor(t2onet_amd.default_options()).to(device)
    img, tgt, params = make_inputs(ops, B, H, W, device)
    x = img.clone().requires_grad_(True)
    ps = [params[k, :, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(ops)]

    def step():
        x.grad = None
        for p in ps:
            p.grad = None
        cur = x
        for op, p in zip(ops, ps):
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
    return {'what': '%d Executor.execute calls + l1_loss + autograd backward (Python API path)' % len(ops),
            'value': round(B * steps / dt, 1), 'unit': 'images/sec', 'ms_per_step': round(dt / steps * 1e3, 4),
            'loss': float(loss.item())}


# ---------------------------------------------------------------------------------------------
# CPU baselines: the oracle on this node's host cores (rank 0, N = 1 only)
# ---------------------------------------------------------------------------------------------
def _median_time(fn, reps, budget_s):
    fn()                                                # warm-up
    ts = []
    t_all = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(ts) >= 1:
            break
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def _pick_threads(probe=None):
    """Eager ATen ops on 50 MB tensors do not scale to every core of a big host (256 threads ran 50x slower than 32 on the
    first MI355X node, and a timing probe chose 16 in one run and 32 in the next): a FIXED count, min(32, logical CPUs)."""
    import torch
    nt = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nt)
    return nt


def _host_info():
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    phys = None
    try:
        out = subprocess.run(['lscpu', '-p=CORE,SOCKET'], capture_output=True, text=True).stdout
        phys = len({ln for ln in out.splitlines() if ln and not ln.startswith('#')})
    except OSError:
        pass
    return {'cpu_model': model, 'logical_cpus': os.cpu_count(), 'physical_cores': phys}


def _cpu_train_step(B3, H, W):
    """The oracle's episode/L1 train step (train_seq2seqL1.py:74-88) on B3 synthetic images: a closure that runs one step."""
    import torch
    from oracle import cpu_ref, synth
    opt = cpu_ref.default_opt()
    sd = cpu_ref.make_leaf_params(synth.fill_state_dict(cpu_ref.actor_state_skeleton(opt)))
    leaves = [v for v in sd.values() if v.requires_grad]
    adam = torch.optim.Adam(leaves, lr=1e-3)
    ximg, xtgt = synth.images(B3, H, W, 21), synth.images(B3, H, W, 22)
    req = synth.requests(B3, 17, 23)
    gen = torch.Generator().manual_seed(10)

    def train_once():
        adam.zero_grad(set_to_none=True)
        r = cpu_ref.episode_forward(sd, req, ximg, opt, reinforce_sample=1, training=True, generator=gen)
        pred = cpu_ref.select_end_images(r['pred_imgs'], r['pred_ops'], opt.end_id)
        cpu_ref.l1_loss(pred, xtgt).backward()
        adam.step()
    return train_once


def cpu_probe_main(spec):
    """--cpu-probe THREADS:IMAGES:H:W (child process of cpu_baselines): one warm-up + one timed oracle train step, one JSON line."""
    import torch
    threads, images, H, W = (int(v) for v in spec.split(':'))
    torch.set_num_threads(threads)
    fn = _cpu_train_step(images, H, W)
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print(json.dumps({'threads': threads, 'images': images, 'seconds': round(dt, 3), 'value': round(images / dt, 3)}), flush=True)
    return 0


def _cpu_probe_child(threads, images, H, W, limit_s):
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-probe', '%d:%d:%d:%d' % (threads, images, H, W)],
                           capture_output=True, text=True, timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
        return json.loads(p.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': '%d threads: not finished within %d s' % (threads, limit_s)}
    except Exception as e:                     # noqa: BLE001
        return {'value': None, 'error': '%s: %s' % (type(e).__name__, e)}


def cpu_baselines(H, W, sample_cfg2=64, sample_cfg3=64, with_gpu_parity=True):
    """BASELINE.md section 3 through oracle/cpu_ref.py -- the eager-PyTorch restatement of the reference, validated against
    it by the committed goldens -- on this node's host cores.  The object's own value is configs[2], the headline's
    workload (the episode/L1 train step at its full batch); `configs_0` (one image, brightness->contrast->saturation) and
    `configs_1` (6-op sequence fwd + L1 + bwd, bs=64) are sub-objects.  `cores` = `threads` = torch threads used (fixed at
    min(32, logical CPUs): all hardware threads are far slower on eager 50 MB ops), `physical_cores` = what the node has."""
    import torch
    from oracle import cpu_ref, synth
    opt = cpu_ref.default_opt()
    host = _host_info()

    def seq_once(ops, img, tgt, params):
        x = img.clone().requires_grad_(True)
        ps = [params[k, :, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(ops)]
        out, _ = cpu_ref.run_sequence(x, ops, ps, opt)
        cpu_ref.l1_loss(out, tgt).backward()

    img, tgt, params = make_inputs(CFG2_OPS, sample_cfg2, H, W, 'cpu')
    threads = _pick_threads()
    med2, n2 = _median_time(lambda: seq_once(CFG2_OPS, img, tgt, params), 7, 25.0)
    cfg1 = {'value': round(sample_cfg2 / med2, 2), 'unit': 'images/sec', 'cores': threads, 'kind': 'port', 'reps': n2,
            'config': 'configs[1]',
            'sample': 'oracle/cpu_ref.py eager fp32, ops %s fwd+L1+bwd, bs=%d %dx%d, median of %d reps (%.2f s each), '
                      '%d torch threads (fixed)' % (CFG2_OPS, sample_cfg2, H, W, n2, med2, threads)}
    # configs[0]: single 256x256 image, 3-op sequence, forward + L1 + backward
    ops1 = [0, 1, 2]
    i1, t1, p1 = make_inputs(ops1, 1, H, W, 'cpu')
    torch.set_num_threads(min(threads, 8))
    med1, n1 = _median_time(lambda: seq_once(ops1, i1, t1, p1), 15, 5.0)
    cfg0 = {'value': round(1.0 / med1, 2), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port', 'reps': n1,
            'config': 'configs[0]', 'sample': 'one %dx%d image, ops %s fwd+L1+bwd, median of %d reps (%.4f s each)'
                                              % (H, W, ops1, n1, med1)}
    torch.set_num_threads(threads)
    # configs[2] -- the HEADLINE's own baseline: the episode/L1 train step (fwd + END select + L1 + bwd + Adam) on a bounded
    # sample of the batch (default 32 of the 64 images: ~6 s per step on the MI355X host), 1 warm-up + exactly 3 reps
    res = {'value': None, 'unit': 'images/sec', 'cores': threads, 'threads': threads, 'physical_cores': host['physical_cores'],
           'logical_cpus': host['logical_cpus'], 'kind': 'port', 'config': 'configs[2]', 'sample': None, 'host': host}
    try:
        B3 = sample_cfg3
        train_once = _cpu_train_step(B3, H, W)
        med3, n3 = _median_time(train_once, 3, 1e9)
        res['value'] = round(B3 / med3, 3)
        res['reps'] = n3
        res['sample'] = ('oracle/cpu_ref.py episode_forward (training mode, sampled ops) + END select + L1 + backward + Adam, '
                         'bs=%d %dx%d fp32, median of %d reps after 1 warm-up (%.2f s each), %d torch threads (fixed) of '
                         '%s physical cores' % (B3, H, W, n3, med3, threads, host['physical_cores']))
    except Exception as e:                     # noqa: BLE001
        res['error'] = '%s: %s' % (type(e).__name__, e)
    # BASELINE.md section 3 says torch.set_num_threads(os.cpu_count()): that figure too, beside the fixed-thread one --
    # the same step on a small sample in CHILD processes with a time limit (all hardware threads have been 50x slower than 32 on
    # eager ops of this size: the probe must not be able to hold the benchmark up)
    try:
        n_probe = 8
        allc = _cpu_probe_child(os.cpu_count() or 1, n_probe, H, W, 30)
        fixed = _cpu_probe_child(threads, n_probe, H, W, 30)
        res['all_cores'] = {'threads': os.cpu_count(), 'images': n_probe, 'value': allc.get('value'), 'unit': 'images/sec',
                            'same_sample_at_fixed_threads': fixed.get('value'), 'fixed_threads': threads,
                            'sample': 'the configs[2] step on %d images, 1 warm-up + 1 repetition, in a child process (30 s limit)' % n_probe}
        for r_ in (allc, fixed):
            if 'error' in r_:
                res['all_cores']['error'] = r_['error']
    except Exception as e:                     # noqa: BLE001
        res['all_cores'] = {'error': '%s: %s' % (type(e).__name__, e)}
    res['configs_0'] = cfg0
    res['configs_1'] = cfg1
    if with_gpu_parity:
        # parity gate reported with the numbers: the GPU path against the oracle on 4 images
        try:
            import t2onet_amd
            n = min(4, sample_cfg2)
            x = img[:n].clone().requires_grad_(True)
            ps = [params[k, :n, :PARAM_RANGES[op][0]].clone().requires_grad_(True) for k, op in enumerate(CFG2_OPS)]
            ref, _ = cpu_ref.run_sequence(x, CFG2_OPS, ps, opt)
            ref_loss = cpu_ref.l1_loss(ref, tgt[:n])
            ref_loss.backward()
            ex = t2onet_amd.Executor(t2onet_amd.default_options()).cuda()
            xg = img[:n].cuda().requires_grad_(True)
            pg = params[:, :n].cuda().requires_grad_(True)
            loss, out = ex.run_sequence_fused(xg, CFG2_OPS, pg, tgt[:n].cuda())
            loss.backward()
            gerr = (xg.grad.cpu() - x.grad).abs().max().item() / max(x.grad.abs().max().item(), 1e-30)
            res['parity'] = {'images': n, 'fwd_max_abs_err': (out.detach().cpu() - ref.detach()).abs().max().item(),
                             'loss_abs_dev': abs(loss.item() - ref_loss.item()),
                             'gimg_max_err_rel_to_max': gerr, 'tolerance': 1e-5}
        except Exception as e:                     # noqa: BLE001
            res['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return res


# ---------------------------------------------------------------------------------------------
# train step (the headline)
# ---------------------------------------------------------------------------------------------
def synthetic_requests(B, g):
    import torch
    n = torch.randint(1, 16, (B,), generator=g)
    x = torch.zeros(B, 17, dtype=torch.long)
    for b in range(B):
        k = int(n[b])
        x[b, 0] = 1
        x[b, 1:1 + k] = torch.randint(4, 918, (k,), generator=g)
        x[b, 1 + k] = 2
    return x


def train_step_bench(ctx, B, H, W, steps, warmup, with_extras=True):
    """BASELINE.json configs[2]/[3]: the episode/L1 train step of train_seq2seqL1.py:74-88, FiveK-shaped
    synthetic batch, random-init weights, fp32; exactly `steps` timed steps."""
    import torch
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dist, world, device = ctx['dist'], ctx['world'], ctx['device']
    opt = t2onet_amd.default_options()
    torch.manual_seed(10 + ctx['rank'])
    model = Actor(opt).to(device).train()
    if os.environ.get('T2O_NHWC', '1') != '0':
        model.use_channels_last()
    if dist is not None:                                   # identical replicas
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    want_graph = os.environ.get('T2O_GRAPH_ENCODER', '0') != '0'
    want_step_graph = os.environ.get('T2O_GRAPH_STEP', '0') != '0'
    tr = Trainer(model, opt, graph_encoder=want_graph, graph_step=want_step_graph)
    g = torch.Generator().manual_seed(10 + ctx['rank'])
    img = torch.rand(B, 3, H, W, generator=g).to(device)
    tgt = torch.rand(B, 3, H, W, generator=g).to(device)
    x = synthetic_requests(B, g)
    lengths = (x != 0).sum(1)
    x = x.to(device)
    # (with --warmup 0 one untimed set-up step still runs: run-time kernel specialisation and the library GEMM selection happen
    # on the first step and are initialisation, not steady-state work)
    for _ in range(max(warmup, 1)):
        tr.episode_step(x, img, tgt, lengths=lengths)
    ctx['barrier']()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.episode_step(x, img, tgt, lengths=lengths)
    t_enq = time.perf_counter() - t0                       # host time to enqueue (includes the step's own host syncs)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    dt = ctx['max_over_ranks'](dt_local)
    spread = {'min': round(dt_local / steps * 1e3, 3), 'max': round(dt_local / steps * 1e3, 3)}
    allreduce_ms = None
    if dist is not None:
        t = torch.tensor([dt_local, -dt_local], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        spread = {'min': round(-float(t[1]) / steps * 1e3, 3), 'max': round(float(t[0]) / steps * 1e3, 3)}
        # the step's one collective on its own: HIP events around the flat-buffer all-reduce (RCCL) on this rank
        evs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.barrier()
            e0.record()
            tr.grads.all_reduce_mean()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        allreduce_ms = round(ctx['max_over_ranks'](sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]), 4)
    # the other half of the reference's alternation (train_seq2seqL1.py:51-65, teacher forced: six encoder passes) and the
    # alternating pair, AFTER the timed headline region: extra keys, same model and batch shapes
    extra = {}
    try:
        if not with_extras:
            raise StopIteration
        ops_t = torch.stack([torch.randperm(6, generator=g)[:5] for _ in range(B)])
        y = torch.cat([torch.full((B, 1), 1), torch.tensor([3, 4, 5, 6, 8, 9])[ops_t], torch.full((B, 1), 2)], 1).to(device)
        img_y = torch.rand(B, 6, 3, H, W, generator=g).to(device)
        gt = torch.rand(B, 5, 24, generator=g) * 2 - 1
        npar = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
        for b_ in range(B):
            for k_ in range(5):
                gt[b_, k_, npar[int(y[b_, k_ + 1])]:] = 0
        gt = gt.to(device)
        n_sup = max(2, min(steps, 8))

        def timed(fn, n, warm):
            for _ in range(warm):
                fn()
            ctx['barrier']()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return ctx['max_over_ranks'](time.perf_counter() - t1) / n * 1e3
        sup = timed(lambda: tr.supervised_step(x, y, img, img_y, gt, lengths=lengths), n_sup, 2)
        pair = timed(lambda: (tr.supervised_step(x, y, img, img_y, gt, lengths=lengths),
                              tr.episode_step(x, img, img_y[:, -1], lengths=lengths)), max(3, n_sup // 2), 2)   # (2 warm-up pairs: the
                                                                                   # caching allocator settles on the alternation's pattern)
        extra = {'supervised_step': {'ms_per_step': round(sup, 3), 'images_per_sec': round(world * B / sup * 1e3, 1), 'steps': n_sup,
                                     'what': 'teacher-forced step (train_seq2seqL1.py:51-65): START + 5 operators + END, NLL + MSE, '
                                             'six encoder passes'},
                 'alternating_pair': {'ms_per_pair': round(pair, 3), 'images_per_sec': round(2 * world * B / pair * 1e3, 1),
                                      'what': 'one supervised + one episode step, the reference\'s iteration parity'}}
    except StopIteration:
        pass
    except Exception as e:                 # noqa: BLE001
        extra = {'supervised_step': {'error': '%s: %s' % (type(e).__name__, e)}}
    flop = TRAIN_FLOP_PER_IMAGE * (H * W) / (256.0 * 256.0) * B
    tf = flop / (dt / steps) / 1e12                        # per GPU
    # what the matrix cores really execute: derived from the plan's OWN per-layer, per-direction kernel choice (encoder.TrunkPlan.
    # flop_table: the function the trunk's forward / backward schedule asks too) -- Winograd F(2x2,3x3) families do 16 of the direct
    # convolution's 36 multiplies; x 5 encoder passes of the episode step
    executed, families = flop, None
    if getattr(model.vis_encoder, 'trunk_plan', None) is not None:
        table = model.vis_encoder.trunk_plan().flop_table(B, H, W)
        passes = 5
        trunk_algo = passes * sum(r[3] for r in table)
        executed = flop - trunk_algo + passes * sum(r[4] for r in table)     # (SURVEY's figure also holds the small dense layers)
        families = {}
        for _, _, fam, algo, ex in table:
            f = families.setdefault(fam, [0.0, 0.0])
            f[0] += passes * algo
            f[1] += passes * ex
        families = {k: {'algorithmic_TFLOP': round(v[0] / 1e12, 4), 'executed_TFLOP': round(v[1] / 1e12, 4)} for k, v in families.items()}
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())   # (a step that produced inf/nan gradients is no measurement)
    if not finite:
        raise RuntimeError('train leg: non-finite parameters after %d steps' % (warmup + steps))
    return {'images_per_sec': round(world * B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 3),
            'host_enqueue_ms_per_step': round(t_enq / steps * 1e3, 2),
            'steps': steps, 'warmup': warmup, 'global_batch': world * B, 'loss': float(loss.item()), 'parameters_finite': finite,
            'encoder_hipgraphs': bool(tr.graph_encoder and '_graphed_encoders' in model.__dict__),
            'step_hipgraphs': len(tr._step_graphs) if tr.graph_step else 0,
            'own_communicator': bool(tr.grads.__dict__.get('_comm') is not None),      # T2O_OWN_COMM=1: the C ABI's t2o_allreduce_mean
            'ms_per_step_over_ranks': spread, 'allreduce_ms': allreduce_ms, 'allreduce_bytes': tr.grads.flat.numel() * 4,
            'roofline': {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': FP32_MATRIX_PEAK_TF, 'unit': 'TFLOP/s',
                         'frac': round(tf / FP32_MATRIX_PEAK_TF, 4), 'flop_per_step_per_gpu': flop,
                         'executed_flop_per_step_per_gpu': executed,
                         'executed_TFLOPs': round(executed / (dt / steps) / 1e12, 2),
                         'executed_frac': round(executed / (dt / steps) / 1e12 / FP32_MATRIX_PEAK_TF, 4),
                         'by_kernel_family': families,
                         'note': 'whole step against the dense fp32 matrix peak: 5 x ResNet-18 forward+backward = '
                                 '67.8 GFLOP/image at 256x256 (SURVEY 8(d)), the ALGORITHMIC count of the direct convolutions; '
                                 'per GPU.  executed_*: the same with every layer and direction priced by the kernel family the '
                                 'trunk plan runs it on (TrunkPlan.flop_table; Winograd F(2x2,3x3): 16 of 36 multiplies) -- the '
                                 'matrix pipe\'s real load'},
            'workload': 'episode/L1 train step (train_seq2seqL1.py:74-88), bs=%d/GPU %dx%d fp32, sampled ops, '
                        'flat-gradient all-reduce (%d ranks) + Adam' % (B, H, W, world), **extra}


def conv_kernel_table(B, H, W, device, reps=40):
    """HIP-event timings (on the launch stream) of the train step's dominant kernel, k_conv3x3_fwd<2,1>: the forward
    and the data gradient of the stride-1 3x3 convolutions of the encoder's 64- and 128-channel stages
    (models/actor_resnet.py:27-44) at this batch / image size -- 60 launches per train step, each 2 * 9 * C * C * pixels
    FLOP (19.33 GFLOP at bs=64 256x256 in every stage).  The 256- and 512-channel stages run Winograd F(2x2,3x3) in the
    train step (t2o_winograd.hip transform kernels around the 16 GEMMs of k_gemm_nt / k_gemm_tn, 8.6 GFLOP); their direct kernels and the
    Winograd pipeline are both listed, the latter with the DIRECT convolution's FLOP as the numerator ("effective")."""
    import torch
    from t2onet_amd import _lib
    import t2onet_amd.functional as T
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = {}
    g = torch.Generator().manual_seed(3)
    for C, div in ((64, 4), (128, 8), (256, 16), (512, 32)):
        h, w = H // div, W // div
        x = torch.rand(B, h, w, C, generator=g).to(device) - 0.5
        wt = (torch.rand(C, 3, 3, C, generator=g).to(device) - 0.5) * 0.05
        y = torch.empty_like(x)
        ws = T._conv_workspace(device, 64 << 10)
        flop = 2.0 * 9 * C * C * B * h * w
        for name, fn in (('fwd', lambda: lib.t2o_conv3x3_fwd_nhwc(x.data_ptr(), wt.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(),
                                                                 B, h, w, C, C, st)),
                         ('dgrad', lambda: lib.t2o_conv3x3_dgrad_pre_nhwc(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                                                         ws.numel(), B, h, w, C, C, st))):
            for _ in range(3):
                _lib.check(fn(), 'conv table')
            evs = []
            group = 10                                      # back-to-back launches between two events: the per-launch figure is
            for _ in range(max(reps // group, 2)):          # the kernel's duration + the dispatch gap, not event / launch latency
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(group):
                    _lib.check(fn(), 'conv table')
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in evs) / (len(evs) * group)
            rows['%s_c%d_%dx%d' % (name, C, h, w)] = {'ms': round(ms, 5), 'GFLOP': round(flop / 1e9, 3), 'TFLOPs': round(flop / ms / 1e9, 2),
                                                   'kernel': 'k_conv3x3_fwd<2,1>' if C < 512 else 'k_conv3x3_fwd<1,1>'}
    for C, div in ((256, 16), (512, 32)):
        h, w = H // div, W // div
        x = torch.rand(B, h, w, C, generator=g).to(device) - 0.5
        wt = (torch.rand(C, 3, 3, C, generator=g).to(device) - 0.5) * 0.05
        U = T.wino_weight(wt, C, C)
        y = torch.empty_like(x)
        dw = torch.zeros_like(wt)
        V = T.wino_input(x, B, h, w)
        flop = 2.0 * 9 * C * C * B * h * w
        for name, fn in (('wino_fwd', lambda: T.wino_conv_nhwc(x, U, B, h, w, None, True, out=y)),
                         ('wino_wgrad', lambda: T.wino_wgrad_nhwc(V, x, dw, B, h, w, True))):
            for _ in range(3):
                fn()
            evs = []
            for _ in range(max(reps // 10, 2)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in evs) / (len(evs) * 10)
            rows['%s_c%d_%dx%d' % (name, C, h, w)] = {
                'ms': round(ms, 5), 'GFLOP': round(flop / 1e9, 3), 'TFLOPs': round(flop / ms / 1e9, 2), 'effective': True,
                'kernel': ('k_wino_input + k_gemm_nt (16 GEMMs) + k_wino_output<stats>' if name == 'wino_fwd'
                           else 'k_wino_dy + k_gemm_tn (16 GEMMs over the tiles) + k_wino_dw (V kept from the forward)')}
    # the kernel that runs these layers in the train step since round 4: Winograd F(2x2,3x3) with V and M kept on chip
    # (4 + 3 + 3 stride-1 layers of the 64- / 128- / 256-channel stages x forward + data gradient x 5 encoder passes = up to 100 launches
    # per step over its epilogue variants)
    for C, div in ((64, 4), (128, 8), (256, 16)):
        h, w = H // div, W // div
        if not lib.t2o_wino_fused_supported(B, h, w, C, C):
            continue
        x = torch.rand(B, h, w, C, generator=g).to(device) - 0.5
        wt = (torch.rand(C, 3, 3, C, generator=g).to(device) - 0.5) * 0.05
        Uc = T.wino_u_chunked(T.wino_weight(wt, C, C))
        add = torch.rand(B, h, w, C, generator=g).to(device)
        y = torch.empty_like(x)
        flop = 2.0 * 9 * C * C * B * h * w
        for name, fn in (('onchip_wino_fwd', lambda: T.wino_fused_conv_nhwc(x, Uc, B, h, w, None, True, out=y)),
                         ('onchip_wino_dgrad', lambda: T.wino_fused_conv_nhwc(x, Uc, B, h, w, None, False, out=y)),
                         ('onchip_wino_dgrad_addend', lambda: T.wino_fused_conv_nhwc(x, Uc, B, h, w, add, False, out=y))):
            for _ in range(3):
                fn()
            evs = []
            for _ in range(max(reps // 10, 2)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in evs) / (len(evs) * 10)
            rows['%s_c%d_%dx%d' % (name, C, h, w)] = {
                'ms': round(ms, 5), 'GFLOP': round(flop / 1e9, 3), 'TFLOPs': round(flop / ms / 1e9, 2), 'effective': True,
                'executed_GFLOP': round(flop * 16.0 / 36.0 / 1e9, 3), 'executed_TFLOPs': round(flop * 16.0 / 36.0 / ms / 1e9, 2),
                'kernel': 'k_wino_fused'}
    # the weight gradient of the same layers: Winograd domain, both transforms on chip (round 5), over the five encoder passes of a
    # train step side by side (encoder.WgradArena) -- against the direct kernel it replaced, same operands
    P5 = 5
    for C, div in ((64, 4), (128, 8), (256, 16)):
        h, w = H // div, W // div
        if not lib.t2o_wino_fused_wgrad_supported(P5 * B, h, w, C, C):
            continue
        x = torch.rand(P5 * B, h, w, C, generator=g).to(device) - 0.5
        dy = torch.rand(P5 * B, h, w, C, generator=g).to(device) - 0.5
        dw = torch.zeros(C, 3, 3, C, device=device)
        flop = 2.0 * 9 * C * C * P5 * B * h * w
        need = lib.t2o_conv3x3_wgrad_workspace_bytes(P5 * B, h, w, C, C)
        wsd = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
        cands = [('onchip_wino_wgrad', lambda: T.wino_fused_wgrad_nhwc(x, dy, dw, P5 * B, h, w, True), 'k_wino_wgrad + k_wgw_reduce + k_wino_dw', 16.0 / 36.0)]
        if need > 0:
            cands.append(('direct_wgrad', lambda: _lib.check(lib.t2o_conv3x3_wgrad_acc_nhwc(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsd.data_ptr(), need,
                                                                                           P5 * B, h, w, C, C, 1, 1, st), 'direct wgrad'),
                          'k_conv3x3_wgrad + k_conv_wgrad_reduce', 1.0))
        for name, fn, kern, frac_exec in cands:
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            rows['%s_c%d_%dx%d_x%dpasses' % (name, C, h, w, P5)] = {
                'ms': round(ms, 5), 'GFLOP': round(flop / 1e9, 3), 'TFLOPs': round(flop / ms / 1e9, 2), 'effective': frac_exec < 1.0,
                'executed_GFLOP': round(flop * frac_exec / 1e9, 3), 'executed_TFLOPs': round(flop * frac_exec / ms / 1e9, 2), 'kernel': kern}
        del x, dy, dw, wsd
    dom = [v for k, v in rows.items() if v['kernel'] == 'k_wino_fused' and '_addend' not in k]
    if not dom:                                             # (an image size the on-chip kernel does not take: the direct kernel runs)
        dom = [v for k, v in rows.items() if v['kernel'] == 'k_conv3x3_fwd<2,1>' and ('_c64_' in k or '_c128_' in k)]
    kernel = dom[0]['kernel']
    avg_ms = sum(v['ms'] for v in dom) / len(dom)
    flop = dom[0]['GFLOP'] * 1e9
    executed = dom[0].get('executed_GFLOP', dom[0]['GFLOP']) * 1e9
    tf = executed / avg_ms / 1e9
    return rows, {'bound': 'mfma', 'kernel': kernel, 'of': 'train step: encoder 3x3 stride-1 convolutions, forward + data gradient, 64/128/256-channel stages',
                  'achieved': round(tf, 2), 'peak': FP32_MATRIX_PEAK_TF, 'unit': 'TFLOP/s', 'frac': round(tf / FP32_MATRIX_PEAK_TF, 4),
                  'traffic': None, 'executed_flop_per_launch': executed, 'algorithmic_flop_per_launch': flop,
                  'credited_achieved': round(flop / avg_ms / 1e9, 2), 'credited_frac': round(flop / avg_ms / 1e9 / FP32_MATRIX_PEAK_TF, 4),
                  'avg_launch_ms': round(avg_ms, 5), 'launches_per_step': 90 if len(dom) == 6 else None,
                  'note': 'dominant kernel of the train step.  k_wino_fused: Winograd F(2x2,3x3) in one launch, 16 of the direct '
                          'convolution\'s 36 multiplies.  frac = the MFMA work the launch EXECUTES (2 x 16/36 x 9 x C^2 x pixels) / time / '
                          'the v_mfma_f32_32x32x2_f32 dense peak -- a physical fraction; credited_* use the direct convolution\'s '
                          'algorithmic FLOP (2 x 9 x C^2 x pixels, SURVEY 8(d)) and may exceed 1.  Duration = mean of HIP-event timed '
                          'launches over the (stage, direction) shapes it runs with, equally often'}


# ---------------------------------------------------------------------------------------------
# the printed line: compact (the driver parses the LAST stdout line); everything else goes to bench_detail.json
# ---------------------------------------------------------------------------------------------
LINE_LIMIT = 6144                       # bytes; round 4's 24 KB line was not parsed by the driver
REQUIRED_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=160):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + '...'


def compact_line(detail, detail_path=None):
    """The one JSON line bench.py prints: the contract's keys, the three roofline objects, the CPU baseline and a few
    numbers per leg -- no per-kernel tables, no prose.  `detail` is the full record (written to bench_detail.json)."""
    out = _pick(detail, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                         'vs_baseline', 'dtype', 'data'))
    out['metric'] = _short(out.get('metric'), 200)
    cfg = detail.get('config', {})
    out['config'] = {k: _short(v, 200) for k, v in cfg.items()}
    if 'error' in detail:
        out['error'] = _short(detail['error'], 300)
    rf = detail.get('roofline') or {}
    out['roofline'] = rf if 'error' in rf else _pick(rf, (
        'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'launches_per_step',
        'executed_flop_per_launch', 'credited_frac'))
    if 'of' in rf:
        out['roofline']['of'] = _short(rf['of'], 120)
    er = detail.get('executor_roofline')
    if er:
        out['executor_roofline'] = _pick(er, ('bound', 'kernel', 'of', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                              'moved_bytes_per_launch', 'avg_launch_ms', 'credited_frac'))
    tr = detail.get('train_roofline')
    if tr:
        out['train_roofline'] = _pick(tr, ('bound', 'achieved', 'peak', 'unit', 'frac', 'executed_TFLOPs', 'executed_frac'))
    ex = {}
    for name, leg in (detail.get('executor') or {}).items():
        if not isinstance(leg, dict):
            ex[name] = _short(leg, 200)
            continue
        row = {}
        for kind in ('fused', 'value_grad', 'materialised'):
            if kind in leg:
                row[kind] = _pick(leg[kind], ('value', 'ms_per_step', 'frac_of_peak', 'fused_min_frac'))
        if 'api_path' in leg:
            row['api_path_ms'] = leg['api_path'].get('ms_per_step', leg['api_path'].get('error'))
        ex[name] = row
    if ex:
        out['executor'] = ex
    ts = detail.get('train_step')
    if ts:
        t = _pick(ts, ('host_enqueue_ms_per_step', 'global_batch', 'loss', 'step_hipgraphs', 'ms_per_step_over_ranks',
                       'allreduce_ms', 'allreduce_bytes', 'own_communicator'))
        if isinstance(ts.get('supervised_step'), dict):
            t['supervised_ms'] = ts['supervised_step'].get('ms_per_step', _short(ts['supervised_step'].get('error'), 120))
        if isinstance(ts.get('alternating_pair'), dict):
            t['alternating_pair_ms'] = ts['alternating_pair'].get('ms_per_pair')
        out['train_step'] = t
    t128 = detail.get('train_step_128')
    if t128:
        out['train_step_128'] = _pick(t128, ('images_per_sec', 'ms_per_step', 'host_enqueue_ms_per_step', 'vs_256', 'executed_frac', 'error'))
    cb = detail.get('cpu_baseline')
    if cb is not None:
        c = _pick(cb, ('value', 'unit', 'cores', 'threads', 'kind', 'config', 'physical_cores', 'logical_cpus', 'reps', 'gpu_over_cpu', 'error'))
        if isinstance(cb.get('all_cores'), dict):
            c['all_cores'] = _pick(cb['all_cores'], ('threads', 'value', 'same_sample_at_fixed_threads', 'images', 'error'))
        c['sample'] = _short(cb.get('sample'), 240)
        if isinstance(cb.get('host'), dict):
            c['cpu_model'] = cb['host'].get('cpu_model')
        for sub in ('configs_0', 'configs_1'):
            if isinstance(cb.get(sub), dict):
                c[sub] = _pick(cb[sub], ('value', 'unit', 'cores', 'reps'))
        out['cpu_baseline'] = c
        if isinstance(cb.get('parity'), dict):
            out['parity'] = {k: (float('%.3g' % v) if isinstance(v, float) else _short(v, 120)) for k, v in cb['parity'].items()}
    if detail_path:
        out['detail'] = detail_path
    return out


def emit_line(detail, extra_keys=()):
    """Write the full record next to bench.py (and under gpurun_out/ when that exists; T2O_BENCH_DETAIL_DIR overrides both),
    print the compact line LAST."""
    rel = None
    override = os.environ.get('T2O_BENCH_DETAIL_DIR')
    for d in ((override,) if override else (ROOT, os.path.join(ROOT, 'gpurun_out'))):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, 'bench_detail.json'), 'w') as f:
                json.dump(detail, f, indent=1)
            rel = rel or 'bench_detail.json'
        except OSError:
            pass
    c = compact_line(detail, rel)
    c.update({k: detail[k] for k in extra_keys if k in detail})
    s = json.dumps(c, separators=(',', ':'))
    if len(s) > LINE_LIMIT:               # never a line the driver cannot parse: drop the optional blocks, largest first
        for k in ('executor', 'train_step', 'parity', 'train_step_128', 'executor_roofline', 'train_roofline'):
            c.pop(k, None)
            s = json.dumps(c, separators=(',', ':'))
            if len(s) <= LINE_LIMIT:
                break
    sys.stdout.flush()
    sys.stderr.flush()
    print(s, flush=True)


# ---------------------------------------------------------------------------------------------
# selftest worker: the launcher / rendezvous / timing / aggregation path on CPU with gloo
# ---------------------------------------------------------------------------------------------
def selftest_worker(args):
    """No GPU, no product path: a stand-in step (a small CPU all-reduce) run through exactly the rank set-up,
    barrier-bracketed timing, MAX-over-ranks and rank-0 JSON code the real benchmark uses.  tests/ drive it at
    world_size 2 (gloo) to prove `bench.py --gpus N` starts N ranks."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE %d != --gpus %d\n' % (world, args.gpus))
        return 2
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    if os.environ.get('T2O_SELFTEST_FAIL_RANK') == str(rank):          # (tests: a rank that dies after the rendezvous)
        os._exit(7)
    buf = torch.ones(1 << 16) * (rank + 1)

    def step():
        if world > 1:
            dist.all_reduce(buf)
            buf.div_(world)
        else:
            buf.mul_(1.0)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    spread, allreduce_ms = {'min': round(dt / args.steps * 1e3, 4), 'max': round(dt / args.steps * 1e3, 4)}, None
    if world > 1:
        t = torch.tensor([dt, -dt], dtype=torch.float64)    # (the N > 1 keys of the real line, formed the same way: MAX of +-dt)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        spread = {'min': round(-float(t[1]) / args.steps * 1e3, 4), 'max': round(float(t[0]) / args.steps * 1e3, 4)}
        dt = float(t[0])
        ts = []
        for _ in range(5):
            dist.barrier()
            a0 = time.perf_counter()
            dist.all_reduce(buf)
            ts.append(time.perf_counter() - a0)
            buf.div_(world)
        t = torch.tensor([sorted(ts)[len(ts) // 2]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_ms = round(float(t.item()) * 1e3, 4)
        seen = torch.zeros(world)
        seen[rank] = os.getpid()
        dist.all_reduce(seen)
        pids = [int(v) for v in seen.tolist()]
    else:
        pids = [os.getpid()]
    if rank == 0:
        # the printed line goes through the real emission path (emit_line) with a record shaped -- and sized -- like a real run's:
        # bulky per-kernel tables and prose that must end up in bench_detail.json, not on stdout
        kernels = {'bwd_kernel_%02d' % i: {'ms': 0.03 + i * 1e-3, 'algorithmic_MB': 150.99, 'GBps': 4400.0 + i, 'hbm_min_MB': 150.99,
                                           'hbm_min_GBps': 4400.0 + i} for i in range(12)}
        leg = {k: {'value': 3.0e5, 'unit': 'images/sec', 'ms_per_step': 0.2, 'frac_of_peak': 1.0, 'fused_min_frac': 0.1, 'kernels': kernels,
                   'what': 'selftest stand-in ' * 10} for k in ('fused', 'value_grad', 'materialised')}
        emit_line({'metric': 'selftest steps/sec (launcher path only, not a benchmark)', 'selftest': True,
                   'value': round(args.steps / dt, 1), 'unit': 'steps/sec', 'n_gpus': world, 'steps': args.steps,
                   'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
                   'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                   'config': {'workload': 'selftest', 'world_size': world, 'backend': 'gloo', 'rank_pids': pids},
                   'roofline': {'bound': 'mfma', 'kernel': 'selftest', 'achieved': 0.0, 'peak': FP32_MATRIX_PEAK_TF, 'unit': 'TFLOP/s',
                                'frac': 0.0, 'traffic': None, 'note': 'selftest stand-in ' * 40},
                   'cpu_baseline': {'value': 0.0, 'unit': 'steps/sec', 'cores': 1, 'kind': 'port', 'sample': 'selftest stand-in ' * 40},
                   'executor': {n: dict(leg) for n in ('cfg2_bs64', 'bs256', 'cfg5_16x512', 'cfg2_generic')},
                   'conv_kernels': {'row_%02d' % i: {'ms': 0.1, 'GFLOP': 19.3, 'TFLOPs': 100.0, 'kernel': 'selftest'} for i in range(24)},
                   'train_step': {'host_enqueue_ms_per_step': 0.0, 'global_batch': world, 'ms_per_step_over_ranks': spread,
                                  'allreduce_ms': allreduce_ms, 'allreduce_bytes': buf.numel() * 4},
                   'mean_after_allreduce': float(buf[0])}, extra_keys=('selftest', 'mean_after_allreduce'))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
def worker(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE %d != --gpus %d (start it as `python bench.py --gpus N` or under '
                         'torch.distributed.run with --nproc-per-node N)\n' % (world, args.gpus))
        return 2
    if not torch.cuda.is_available():
        sys.stderr.write('bench.py needs a GPU (the product path has no CPU fallback)\n')
        return 2
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist = None
    backend_info = {'world_size': world, 'backend': None}
    if world > 1 or 'RANK' in os.environ:          # launched by this script's launcher or torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        backend_info['backend'] = 'nccl (RCCL)'
        try:
            backend_info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                      # noqa: BLE001
            backend_info['rccl_version'] = None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(el):
        if dist is None:
            return el
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx = {'dist': dist, 'world': world, 'rank': rank, 'device': device, 'barrier': barrier, 'max_over_ranks': max_over_ranks}
    B, H, W = args.batch, args.size, args.size
    P = B * H * W
    line = {
        'metric': 'images/sec (train step: episode forward + END select + L1 + backward + gradient all-reduce + Adam, '
                  'bs=%d/GPU, %dx%d fp32)' % (B, H, W),
        'value': None, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': dict({'workload': 'BASELINE.json configs[%d]: full T2ONet episode/L1 train step (train_seq2seqL1.py:74-88), '
                                    'bs=%d per GPU, %dx%d fp32, FiveK-shaped synthetic batch, random-init weights'
                                    % (2 if world == 1 else 3, B, H, W),
                        'global_batch': world * B,
                        'parallelism': 'dp%d: batch shards, one flat fp32 gradient all-reduce (22.2 M floats) per step' % world},
                       **backend_info),
    }
    executor = {}
    try:
        executor['cfg2_bs64'] = executor_leg(ctx, CFG2_OPS, B, H, W, args.exec_steps, args.exec_warmup, with_api=(world == 1))
        if not args.quick:
            executor['bs256'] = executor_leg(ctx, CFG2_OPS, 4 * B, H, W, max(args.exec_steps // 4, 10), max(args.exec_warmup // 2, 3))
            executor['cfg5_16x512'] = executor_leg(ctx, CFG5_OPS, max(B // 4, 1), 2 * H, 2 * W, args.exec_steps, args.exec_warmup)
        executor['cfg2_generic'] = executor_leg(ctx, CFG2_GENERIC_OPS, B, H, W, args.exec_steps, args.exec_warmup)
        fk = executor['cfg2_bs64']['fused']['kernels']
        dom = max(fk, key=lambda n: fk[n]['ms'])
        line['executor_roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'of': 'executor.cfg2_bs64.fused', 'achieved': fk[dom]['hbm_min_GBps'], 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(fk[dom]['hbm_min_GBps'] / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(dom, P),
            'moved_bytes_per_launch': hbm_min_bytes(dom, P), 'avg_launch_ms': fk[dom]['ms'],
            'credited_bytes_per_launch': algorithmic_bytes(dom, P), 'credited_achieved': fk[dom]['GBps'],
            'credited_frac': round(fk[dom]['GBps'] / HBM_PEAK_GBS, 4),
            'note': 'dominant hand-written kernel of the op pipeline.  frac = bytes the launch MOVES (read image + gradient, '
                    'write gradient: 36 B/pixel whatever the fusion) / time / 8 TB/s -- a physical fraction, <= 1.  credited_* '
                    'uses SURVEY 8(d)\'s materialised accounting (36 B/pixel per operator application the fused launch '
                    'performs) and may exceed 1.  traffic = PMC bytes when profiles/pmc_traffic.json was measured on this '
                    'very library, else null'}
    except Exception as e:                     # noqa: BLE001
        executor['error'] = '%s: %s' % (type(e).__name__, e)
    line['executor'] = executor
    try:                                   # the train step's dominant kernel on its own, HIP-event timed: the line's `roofline`
        line['conv_kernels'], line['roofline'] = conv_kernel_table(B, H, W, device)
        line['roofline']['traffic'] = pmc_launch_traffic('k_wino_fused<0>' if line['roofline'].get('kernel') == 'k_wino_fused' else 'k_conv3x3_fwd<2, 1>')
    except Exception as e:                 # noqa: BLE001
        line['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    rc = [0]
    if args.no_train:
        line['error'] = '--no-train: executor legs only, no headline value'
        if rank == 0:
            emit_line(line)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    def emit():
        if rank == 0:
            emit_line(line)

    # the train leg is the headline.  A hang in it (one rank failing inside a collective) must still leave a line
    # with everything measured so far: a timer THREAD prints it and exits non-zero (a signal handler would not run
    # while the main thread sits in a blocking device call).  Nothing GPU-side is touched from the timer thread.
    def on_timeout():
        line['error'] = 'train step did not finish within %d s' % args.train_timeout
        emit()
        os._exit(3)
    watchdog = threading.Timer(args.train_timeout, on_timeout)
    watchdog.daemon = True
    watchdog.start()
    try:
        train = train_step_bench(ctx, B, H, W, args.steps, args.warmup)
        line['value'] = train['images_per_sec']
        line['ms_per_step'] = train['ms_per_step']
        line['train_step'] = train
        line['train_roofline'] = train['roofline']
    except Exception as e:                 # noqa: BLE001
        import traceback
        traceback.print_exc()
        line['error'] = 'train step: %s: %s' % (type(e).__name__, e)
        rc[0] = 1
    watchdog.cancel()
    # EXTRA key (the headline stays configs[2]): the same step at the size the reference itself trains at -- 128 x 128 crops,
    # bs = 64 (README.md:91, datasets/FiveKdataset.py:25,68).  A quarter of the pixels: the encoder's maps are 32 x 32 ... 4 x 4.
    if rc[0] == 0 and not args.quick and not args.no_128 and (H, W) == (256, 256) and world == 1:
        try:
            torch.cuda.empty_cache()
            t128 = train_step_bench(ctx, B, 128, 128, args.steps, max(args.warmup, 3), with_extras=False)
            line['train_step_128'] = dict(_pick(t128, ('images_per_sec', 'ms_per_step', 'host_enqueue_ms_per_step', 'steps', 'warmup', 'loss',
                                                       'workload')),
                                          vs_256=round(t128['images_per_sec'] / train['images_per_sec'], 3),
                                          executed_TFLOPs=t128['roofline']['executed_TFLOPs'], executed_frac=t128['roofline']['executed_frac'],
                                          by_kernel_family=t128['roofline']['by_kernel_family'])
        except Exception as e:             # noqa: BLE001
            line['train_step_128'] = {'error': '%s: %s' % (type(e).__name__, e)}
    # the CPU baseline LAST (reported, never the target): run before the train leg, its eager CPU autograd over 64 images left
    # the process in a state in which the supervised / episode alternation read 100-113 ms per pair instead of 83 (the headline
    # and each step on its own were unaffected; not the intra-op thread count -- pinned to 1, the same)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line['cpu_baseline'] = cpu_baselines(H, W, args.cpu_sample, args.cpu_train_sample)
            if line.get('value') and line['cpu_baseline'].get('value'):     # (vs_baseline stays null: BASELINE.md publishes no number)
                line['cpu_baseline']['gpu_over_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
        except Exception as e:             # noqa: BLE001
            line['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20, help='timed train steps')
    ap.add_argument('--warmup', type=int, default=5, help='untimed train steps (the first also captures the encoder hipGraphs)')
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--exec-steps', type=int, default=200, help='timed steps of each executor leg')
    ap.add_argument('--exec-warmup', type=int, default=20)
    ap.add_argument('--quick', action='store_true', help='skip the bs=256 and cfg5 executor legs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-128', action='store_true', help='skip the extra train_step_128 leg (profiling passes: its shorter launches of the same kernels would mix into the per-kernel averages)')
    ap.add_argument('--no-train', action='store_true',
                    help='executor legs only (profiling passes; the line then has no headline value)')
    ap.add_argument('--cpu-sample', type=int, default=64, help='images of the configs[1] CPU baseline')
    ap.add_argument('--cpu-train-sample', type=int, default=64, help='images of the configs[2] CPU baseline (the headline\'s own batch; 1 warm-up + 3 reps, ~35 s on the MI355X host)')
    ap.add_argument('--train-timeout', type=int, default=600, help='seconds before the train-step leg is abandoned')
    ap.add_argument('--launch-timeout', type=int, default=1500, help='launcher: seconds before the ranks are stopped')
    ap.add_argument('--cpu-probe', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--selftest', action='store_true',
                    help='CPU/gloo stand-in step through the launcher, rendezvous and timing code (tests only)')
    args = ap.parse_args()
    if args.cpu_probe:
        return cpu_probe_main(args.cpu_probe)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'RANK' not in os.environ:
        return launch(args, sys.argv[1:])
    if args.selftest:
        return selftest_worker(args)
    return worker(args)


if __name__ == '__main__':
    sys.exit(main())
