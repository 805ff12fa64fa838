"""Per-kernel means of the rocprofv3 --pmc CSVs written by tools/gpu_pmc.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, '*', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
        if not name.startswith('k_'):
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    rows.append((k, d))
    dur = [float(x) for x in acc[k].get('_dur', [])]
    print(k)
    for c in sorted(d):
        print('    %-24s %14.1f' % (c, d[c]))
    if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
        # guide: FETCH_SIZE is in KiB and under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE exact
        fe, wr = d.get('FETCH_SIZE', 0) * 1024 * 2, d.get('WRITE_SIZE', 0) * 1024
        print('    => HBM traffic (corrected) read %.1f MB + write %.1f MB = %.1f MB' % (fe / 1e6, wr / 1e6, (fe + wr) / 1e6))
