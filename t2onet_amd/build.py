"""Build libt2onet_hip.so for gfx950 in-tree (t2onet_amd/lib/), so the binary travels
with the repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.

    python -m t2onet_amd.build [--force] [--report]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libt2onet_hip.so')
SOURCES = ['t2o_kernels.hip', 't2o_norm.hip']
HEADERS = ['t2o_pixel_math.h', 't2o_block_programs.h', os.path.join(ROOT, 'include', 't2onet_hip.h')]
# -ffp-contract=off: one rounding per arithmetic step, like the reference's eager fp32 ops
FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared',
         '-I' + os.path.join(ROOT, 'include')]


def _digest():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(f if os.path.isabs(f) else os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, report=False):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + '.sha256'
    dig = _digest()
    if not force and not report and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if report:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError('hipcc failed building libt2onet_hip.so')
    if report:
        _print_report(r.stderr)
    open(stamp, 'w').write(dig)
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
