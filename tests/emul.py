"""ctypes front end of tests/host_emul/emul.cpp (host emulation of the HIP block programs).
TEST HARNESS ONLY: lets the CPU suite check the kernels' per-thread code against the oracle."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        out = os.path.join(ROOT, 'tests', '_build')
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, 'libt2o_emul.so')
        src = os.path.join(ROOT, 'tests', 'host_emul', 'emul.cpp')
        deps = [src] + [os.path.join(ROOT, 't2onet_amd', 'csrc', f) for f in ('t2o_pixel_math.h', 't2o_block_programs.h')]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared', '-o', so, src])
        _lib = ctypes.CDLL(so)
    return _lib


def _p(a, t=ctypes.c_float):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


def _f(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def fwd(op, img, param, mask=None, op_id=None, target=None, iters=0):
    img, param, mask, target = _f(img), _f(param), _f(mask), _f(target)
    B, _, H, W = img.shape
    out = np.empty_like(img)
    loss = np.zeros(1, np.float32)
    ids = None if op_id is None else np.ascontiguousarray(np.asarray(op_id, dtype=np.int32))
    rc = lib().emul_fwd(op, _p(ids, ctypes.c_int), _p(img), _p(param), 0 if param is None else param.shape[1],
                        _p(mask), 0 if mask is None else mask.shape[1], _p(target), _p(out), _p(loss), B, H, W, iters)
    assert rc == 0
    return out, float(loss[0])


def bwd(op, img, param, gout=None, mask=None, op_id=None, target=None, gloss=1.0, iters=0):
    img, param, mask, target, gout = _f(img), _f(param), _f(mask), _f(target), _f(gout)
    B, _, H, W = img.shape
    gimg = np.empty_like(img)
    gparam = np.zeros_like(param)
    gl = np.array([gloss], np.float32)
    ids = None if op_id is None else np.ascontiguousarray(np.asarray(op_id, dtype=np.int32))
    rc = lib().emul_bwd(op, _p(ids, ctypes.c_int), _p(img), _p(param), param.shape[1], _p(mask),
                        0 if mask is None else mask.shape[1], _p(gout), _p(target), _p(gl), _p(gimg), _p(gparam),
                        gparam.shape[1], B, H, W, iters)
    assert rc == 0
    return gimg, gparam


def fused(ops, img, params, target, gloss=1.0, iters=0, use_static=0, with_value=False):
    """Fused sequence forward + backward.  params (K,B,24).  Returns out, loss, gimg, gparams."""
    img, params, target = _f(img), _f(params), _f(target)
    B, _, H, W = img.shape
    K = len(ops)
    c_ops = (ctypes.c_int * max(K, 1))(*ops)
    nbuf = lib().emul_fused_buffers(c_ops, K)
    assert nbuf >= 0
    seg = np.zeros((max(nbuf, 1),) + img.shape, np.float32)
    gbuf = np.zeros((2,) + img.shape, np.float32)
    out = np.empty_like(img)
    loss = np.zeros(1, np.float32)
    assert lib().emul_fused_fwd(c_ops, K, _p(img), _p(params), _p(target), _p(out), _p(loss), _p(seg), B, H, W, iters) == 0
    gimg = np.empty_like(img)
    gparams = np.zeros_like(params)
    gl = np.array([gloss], np.float32)
    vout, vloss = np.full_like(img, np.nan), np.full(1, np.nan, np.float32)
    assert lib().emul_fused_bwd(c_ops, K, _p(img), _p(params), _p(target), _p(gl), _p(gimg), _p(gparams), _p(seg),
                                _p(gbuf), B, H, W, iters, use_static, _p(vout), _p(vloss)) == 0
    if with_value:                       # what the last segment's backward ALSO produced (value-and-gradient entry point);
        return out, float(loss[0]), gimg, gparams, vout, float(vloss[0])     # NaN when the last segment is a sharpness
    return out, float(loss[0]), gimg, gparams


def ssim(a, b):
    a, b = _f(a), _f(b)
    B, C, H, W = a.shape
    out = np.zeros(B, np.float32)
    assert lib().emul_ssim(_p(a), _p(b), _p(out), B, C, H, W) == 0
    return out


def ssim_bwd(a, b, gout):
    """Both image gradients of the per-sample SSIM means for the output gradient gout (B)."""
    a, b, gout = _f(a), _f(b), _f(gout)
    B, C, H, W = a.shape
    ga, gb = np.full_like(a, np.nan), np.full_like(b, np.nan)
    assert lib().emul_ssim_bwd(_p(a), _p(b), _p(gout), _p(ga), _p(gb), B, C, H, W) == 0
    return ga, gb
