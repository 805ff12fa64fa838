"""Kernel-trace timeline digest: rocprofv3 --kernel-trace CSV -> busy / idle time of the device over the last steps.
usage: python tools/trace_gaps.py <kernel_trace.csv> [n_last_kernels_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0
cur_end = rows[0][0]
gaps = defaultdict(lambda: [0, 0])
for s, e, name in rows:
    if s > cur_end:
        g = s - cur_end
        key = name.split('(')[0][-60:]
        gaps[key][0] += g
        gaps[key][1] += 1
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
span = t1 - t0
print('kernels %d span %.2f ms busy %.2f ms idle %.2f ms (%.1f %%)' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
print('idle time by the kernel that FOLLOWS the gap:')
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print('  %8.1f us %5d gaps  avg %6.2f us  %s' % (g / 1e3, n, g / 1e3 / n, k))
