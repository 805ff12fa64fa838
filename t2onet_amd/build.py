"""Build libt2onet_hip.so for gfx950 in-tree (t2onet_amd/lib/), so the binary travels
with the repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.

    python -m t2onet_amd.build [--force] [--report]

The sha256 of the sources + flags is compiled INTO the library (t2o_source_digest()); nothing
beside the binary records what it was built from, so a checkout that changes the sources can
never be mistaken for up to date.  The link goes to a temporary file that is renamed over the
target under a file lock: concurrent ranks never see a half-written library.
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libt2onet_hip.so')
SOURCES = ['t2o_kernels.hip', 't2o_norm.hip', 't2o_conv.hip', 't2o_conv1x1.hip', 't2o_rnn.hip', 't2o_heads.hip', 't2o_optim.hip']
HEADERS = ['t2o_pixel_math.h', 't2o_block_programs.h', os.path.join(ROOT, 'include', 't2onet_hip.h')]
# -ffp-contract=off: one rounding per arithmetic step, like the reference's eager fp32 ops
FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared',
         '-I' + os.path.join(ROOT, 'include')]


def hipcc_path():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def source_digest():
    """sha256 over the flags (paths made repo-relative: the tree moves between machines) and every source."""
    h = hashlib.sha256(' '.join(f.replace(ROOT, '.') for f in FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(f if os.path.isabs(f) else os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def library_digest(path=LIB):
    """The digest compiled into an existing library file, or None (missing / predates the digest).  Read from
    the file's bytes, not through dlopen: a library loaded here would shadow the rebuilt one in this process."""
    tag = b't2o-src-digest:'
    try:
        with open(path, 'rb') as f:
            data = f.read()
    except OSError:
        return None
    i = data.find(tag)
    if i < 0:
        return None
    dig = data[i + len(tag):i + len(tag) + 64]
    return dig.decode() if len(dig) == 64 and all(c in b'0123456789abcdef' for c in dig) else None


def build(force=False, report=False):
    """Compile if the library is missing or was built from other sources.  Raises on any compiler error."""
    os.makedirs(LIBDIR, exist_ok=True)
    dig = source_digest()
    if not force and not report and library_digest() == dig:
        return LIB
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not report and library_digest() == dig:         # another process built it meanwhile
            return LIB
        tmp = '%s.tmp.%d' % (LIB, os.getpid())
        cmd = [hipcc_path()] + FLAGS + ['-DT2O_SRC_DIGEST="%s"' % dig, '-o', tmp] + [os.path.join(CSRC, s) for s in SOURCES]
        if report:
            cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            if os.path.exists(tmp):
                os.unlink(tmp)
            raise RuntimeError('hipcc failed building libt2onet_hip.so')
        os.replace(tmp, LIB)
    if report:
        _print_report(r.stderr)
    return LIB


def _print_report(text):
    name, row = None, {}
    for line in text.splitlines():
        if 'Function Name:' in line:
            name = line.split('Function Name:')[1].split('[')[0].strip()
            row = {}
        for key in ('VGPRs:', 'TotalSGPRs:', 'ScratchSize [bytes/lane]:', 'Occupancy [waves/SIMD]:', 'LDS Size [bytes/block]:'):
            if key in line and 'AGPR' not in line:
                row[key] = line.split(key)[1].split('[')[0].strip()
                if key.startswith('LDS'):
                    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                    dem = dem.replace('(anonymous namespace)::', '').split('(')[0]
                    print('%-48s vgpr %3s sgpr %3s scratch %s occ %s lds %s' % (
                        dem, row.get('VGPRs:'), row.get('TotalSGPRs:'), row.get('ScratchSize [bytes/lane]:'),
                        row.get('Occupancy [waves/SIMD]:'), row.get('LDS Size [bytes/block]:')))


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, report='--report' in sys.argv))
