"""Static check of the weight-gradient kernels' machine code (no GPU needed: hipcc cross-compiles gfx950).

k_conv3x3_wgrad reads its MFMA fragments with inline-asm ds_read_b32 (immediate offsets, no address arithmetic), so
the compiler does not know that those registers are filled later, when the LDS returns the data.  If it ever places
a register copy (or any other vector instruction) on such a register between the ds_read and the wave's next
`s_waitcnt lgkmcnt(0)`, the copy picks up stale data -- silently, and only sometimes (the forward kernel had exactly
this bug while it used the same asm reads across its loop's back edge).  This test fails the build that does it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernel_bodies(asm):
    """name -> instruction lines of every k_conv3x3_wgrad instantiation"""
    out, name, body = {}, None, []
    for line in asm.split('\n'):
        m = re.match(r'^(_ZN\S*k_conv3x3_wgrad\S*):', line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith('.end_amdhsa_kernel') or t.startswith('.section'):
            out[name] = body
            name = None
            continue
        if t and not t.startswith((';', '.')):
            body.append(t)
    return out


def _early_uses(body):
    pending, bad = set(), []
    for t in body:
        m = re.match(r'ds_read_b32 v(\d+),', t)
        if m:
            pending.add(int(m.group(1)))
            continue
        if t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t:
            pending.clear()
            continue
        if pending and t.startswith('v_'):
            regs = {int(a) for a in re.findall(r'\bv(\d+)\b', t)}
            for lo, hi in re.findall(r'v\[(\d+):(\d+)\]', t):
                regs |= set(range(int(lo), int(hi) + 1))
            if regs & pending:
                bad.append(t)
    return bad


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not available')
def test_no_vector_instruction_touches_an_asm_ds_read_result_before_the_wait(tmp_path):
    src = os.path.join(ROOT, 't2onet_amd', 'csrc', 't2o_conv.hip')
    asm = tmp_path / 'conv.s'
    subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                    '-S', '--cuda-device-only', '-o', str(asm), src], check=True, capture_output=True, timeout=600)
    bodies = _kernel_bodies(asm.read_text())
    assert len(bodies) >= 4, sorted(bodies)
    for name, body in bodies.items():
        assert sum('v_mfma_f32_32x32x2_f32' in t for t in body) >= 96, name
        assert any(t.startswith('ds_read_b32') for t in body), name
        bad = _early_uses(body)
        assert not bad, '%s: %d vector instructions use an asm ds_read result before lgkmcnt(0), e.g. %s' % (name, len(bad), bad[:3])
