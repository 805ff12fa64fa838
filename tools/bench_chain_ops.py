#!/usr/bin/env python3
"""Time the fused chain kernels for arbitrary operator lists (cost of each operator inside a chain).

    python tools/bench_chain_ops.py [B H W] [--quant]     (--quant: 8-bit images, so channels tie)

Prints, per operator list, the HIP-event time of one t2o_fused_sequence_fwd and one
t2o_fused_sequence_bwd call (gout path, no L1) -- differences between lists give the incremental
cost of an operator in the chain, which is VALU-bound.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from t2onet_amd import _lib  # noqa: E402

LISTS = [[0], [1], [2], [3], [5], [0, 0], [1, 1], [2, 2], [3, 3], [5, 5], [0, 1, 2, 3, 5], [5, 3, 2, 1, 0],
         [0, 1, 2], [3, 5], [0, 0, 0, 0, 0, 0, 0, 0], [5, 5, 5, 5, 5, 5, 5, 5],
         [5, 3, 5, 3, 0, 1, 2]]          # the per-pixel run of BASELINE configs[4]


def main():
    quant = '--quant' in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    B, H, W = (int(v) for v in argv[:3]) if len(argv) >= 3 else (64, 256, 256)
    dev = torch.device('cuda:0')
    lib = _lib.load()
    img, tgt, _ = bench.make_inputs(B, H, W, dev)
    if quant:
        img = torch.round(img * 255.0) / 255.0
    ws = torch.empty(lib.t2o_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    out, gimg, gout = torch.empty_like(img), torch.empty_like(img), torch.randn_like(img)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for ops in LISTS:
        K = len(ops)
        params = torch.zeros(K, B, 24, device=dev)
        for k, op in enumerate(ops):
            n, lo, hi = bench.PARAM_RANGES[op]
            params[k, :, :n] = torch.rand(B, n, device=dev) * (hi - lo) + lo
        gparams = torch.zeros(K, B, 24, device=dev)
        c_ops = (ctypes.c_int * K)(*ops)

        def fwd():
            return lib.t2o_fused_sequence_fwd(c_ops, K, P(img), P(params), None, P(out), None, None, P(ws), ws.numel(), B, H, W, st)

        def bwd():
            return lib.t2o_fused_sequence_bwd(c_ops, K, P(img), P(params), None, None, P(gout), P(gimg), P(gparams),
                                              None, None, P(ws), ws.numel(), B, H, W, st)
        res = []
        for fn in (fwd, bwd):
            for _ in range(3):
                assert fn() == 0, lib.t2o_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f'{str(ops):34s} fwd {res[0]:7.1f} us   bwd(+finalize) {res[1]:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
