"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/t2onet_hip.h declares, validates arguments without touching a device, and the Python
surface mirrors the reference's Executor/Operator API and refuses CPU tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from t2onet_amd import build, _lib
    build.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from t2onet_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 't2onet_hip.h')).read()
    declared = set(re.findall(r'\b(t2o_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_argument_validation_without_a_device(lib):
    assert lib.t2o_abi_version() == 4
    assert [lib.t2o_op_num_params(i) for i in range(-1, 9)] == [-1, 1, 1, 1, 24, 1, 8, 1, 1, -1]
    assert lib.t2o_workspace_bytes(0, 4, 4) == 0 and lib.t2o_workspace_bytes(64, 256, 256) > 0
    # null image / unsupported operator / bad mask are rejected before any launch
    assert lib.t2o_op_fwd(0, None, None, 1, None, 0, None, 1, 4, 4, None) == 1
    assert b'null' in lib.t2o_last_error()
    buf = torch.zeros(64)
    p = buf.data_ptr()
    assert lib.t2o_op_fwd(4, p, p, 1, None, 0, p, 1, 4, 4, None) == 2          # inpaint
    assert lib.t2o_op_fwd(9, p, p, 1, None, 0, p, 1, 4, 4, None) == 2
    assert lib.t2o_op_fwd(0, p, p, 1, p, 2, p, 1, 4, 4, None) == 1             # mask_ch must be 1 or 3
    assert lib.t2o_op_bwd(0, p, p, 1, None, 0, p, p, p, 1, None, 0, 1, 4, 4, None) == 3   # no workspace
    assert lib.t2o_attn_fwd(p, p, p, p, 1, 65, 64, None) == 1
    # fused batch norm: null tensors, one running statistic without the other, missing workspace / ReLU mask source
    assert lib.t2o_bn_workspace_bytes(64, 64) > 0
    assert lib.t2o_bn_relu_fwd(None, None, p, p, p, p, p, p, p, 0.1, 1e-5, p, 1 << 20, 2, 4, 16, None) == 1
    assert b'bn_relu_fwd' in lib.t2o_last_error()
    assert lib.t2o_bn_relu_fwd(p, None, p, p, p, None, p, p, p, 0.1, 1e-5, p, 1 << 20, 2, 4, 16, None) == 1
    assert lib.t2o_bn_relu_fwd(p, None, p, p, p, p, p, p, p, 0.1, 1e-5, None, 0, 2, 4, 16, None) == 3
    assert lib.t2o_bn_relu_bwd(p, None, p, p, p, p, p, p, None, p, p, 1, p, 1 << 20, 2, 4, 16, None) == 1   # has_res needs y
    assert lib.t2o_bn_relu_bwd(p, p, p, p, p, p, p, p, None, p, p, 0, None, 0, 2, 4, 16, None) == 3


def test_python_surface_mirrors_reference():
    import t2onet_amd
    from oracle import cpu_ref
    ex = t2onet_amd.Executor(t2onet_amd.default_options())
    assert ex.name_list == cpu_ref.OP_NAMES
    assert [ex.get_param_num(i) for i in range(8)] == cpu_ref.OP_NPARAM
    for i in range(8):
        assert tuple(ex.get_param_bnd(i)) == tuple(cpu_ref.param_range(i, cpu_ref.default_opt()))
    sk = cpu_ref.actor_state_skeleton()
    want = [k[len('executor.'):] for k in sk if k.startswith('executor.')]
    assert list(ex.state_dict().keys()) == want
    assert all(tuple(v.shape) == tuple(sk['executor.' + k].shape) for k, v in ex.state_dict().items())
    # the parameter heads are plain torch and agree with the oracle on the CPU
    from oracle import synth
    ex.load_state_dict(synth.fill_state_dict(ex.state_dict(), seed=3))
    sd = {'executor.' + k: v for k, v in ex.state_dict().items()}
    f = synth.uniform((3, 512), 13, -1, 1)
    for i in [0, 1, 2, 3, 5, 6, 7]:
        assert torch.allclose(ex.ops[i].extract_parameters(f), cpu_ref.param_head(sd, i, f, cpu_ref.default_opt()), atol=1e-6)
    # identity path and the no-CPU-fallback rule
    img = torch.rand(2, 3, 8, 8)
    out, par = ex.execute(img, -1, None)
    assert out is img and par.shape == (2, 24)
    with pytest.raises(RuntimeError, match='no CPU'):
        ex.execute(img, 0, None, specified_param=torch.zeros(2, 1))


def test_stale_library_is_refused_not_loaded(lib, tmp_path):
    """A library built from other sources (digest compiled in) must not load silently: with no compiler
    reachable the loader raises; it never binds the new signatures onto the old binary."""
    import subprocess
    import sys
    from t2onet_amd import build
    assert build.library_digest() == build.source_digest() == lib.t2o_source_digest().decode()
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['HIPCC'] = '/nonexistent/hipcc'\n"
        "from t2onet_amd import build, _lib\n"
        "build.source_digest = lambda: 'f' * 64\n"          # as if a checkout had changed the sources
        "try:\n"
        "    _lib.load()\n"
        "except RuntimeError as e:\n"
        "    assert 'other sources' in str(e), e\n"
        "    print('refused')\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert r.stdout.strip() == 'refused', r.stdout + r.stderr
    # a failing compiler is an error, not a fallback to whatever binary is lying around
    code2 = code.replace("'/nonexistent/hipcc'", "'/bin/false'").replace("except RuntimeError as e:\n    assert 'other sources' in str(e), e",
                                                                        "except RuntimeError as e:\n    assert 'hipcc failed' in str(e), e")
    r = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True)
    assert r.stdout.strip() == 'refused', r.stdout + r.stderr
