"""Per-kernel means of the rocprofv3 --pmc CSVs written by tools/gpu_pmc.sh, plus pmc_traffic.json: HBM bytes per
pixel of the fused step's kernels (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests as 64 B;
WRITE_SIZE exact), stamped with the digest of the library that was measured."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
P = float(sys.argv[2]) if len(sys.argv) > 2 else 64 * 256 * 256.0
acc = defaultdict(lambda: defaultdict(list))
full = defaultdict(lambda: defaultdict(list))      # keyed by the full kernel name incl. template arguments
for f in glob.glob(os.path.join(out, '*', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
        if not name.startswith('k_'):
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        long_name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        full[long_name[:long_name.index('(')] if '(' in long_name else long_name][r['Counter_Name']].append(float(r['Counter_Value']))
traffic = {}
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print(k)
    for c in sorted(d):
        print('    %-24s %14.1f' % (c, d[c]))
    if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
        fe, wr = d.get('FETCH_SIZE', 0) * 1024 * 2, d.get('WRITE_SIZE', 0) * 1024
        print('    => HBM traffic (corrected) read %.1f MB + write %.1f MB = %.1f MB' % (fe / 1e6, wr / 1e6, (fe + wr) / 1e6))
        traffic[k] = (fe + wr) / P
# bench.py's kernel names for the cfg2 fused step
alias = {'fwd_chain5': [k for k in traffic if k.startswith('k_chain_fwd')],
         'bwd_chain5': [k for k in traffic if k.startswith('k_chain_bwd')],
         'fwd_sharpness+l1': [k for k in traffic if k.startswith('k_sharp_fwd')],
         'bwd_sharpness+l1': [k for k in traffic if k.startswith('k_sharp_bwd')]}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2onet_amd import build  # noqa: E402
js = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/gpu_pmc.sh); FETCH_SIZE doubled per '
                'MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); bs=64 256x256',
      'lib_digest': build.source_digest(),
      'bytes_per_pixel': {a: round(max(traffic[k] for k in ks), 3) for a, ks in alias.items() if ks},
      'kernels': {a: ks for a, ks in alias.items() if ks}}
per_launch = {}
for k, d in full.items():
    if k.startswith('k_conv3x3') or k.startswith('k_wino_fused') or k.startswith('k_wino_wgrad'):
        fe = sum(d.get('FETCH_SIZE', [0])) / max(len(d.get('FETCH_SIZE', [0])), 1) * 1024 * 2
        wr = sum(d.get('WRITE_SIZE', [0])) / max(len(d.get('WRITE_SIZE', [0])), 1) * 1024
        per_launch[k] = round(fe + wr)
js['bytes_per_launch'] = per_launch        # mean over bench.py's conv_kernel_table launches (HIP-event timed there)
json.dump(js, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
