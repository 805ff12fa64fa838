"""Launches that cannot fill the chip: rocprofv3 --kernel-trace CSV -> per kernel name and grid size the workgroup count, the
average duration and the time per step, for launches with fewer than 256 workgroups (one per CU) and non-trivial duration.
usage: python tools/trace_underfilled.py <kernel_trace.csv> <steps traced> [min us]"""
import csv
import sys
from collections import defaultdict

steps = float(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
only = sys.argv[4] if len(sys.argv) > 4 else None        # name filter: then every grid size is listed
agg = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) * max(int(r.get('Grid_Size_Z', 1) or 1), 1)
        wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1) * max(int(r.get('Workgroup_Size_Z', 1) or 1), 1)
        wgs = gx // max(wx, 1)
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        a = agg[(name, wgs, wx)]
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
rows = []
for (name, wgs, wx), (n, t) in agg.items():
    avg = t / n / 1e3
    if (only in name and avg >= thr) if only else (wgs < 512 and avg >= thr):
        rows.append((t / steps / 1e3, n / steps, avg, wgs, wx, name))
rows.sort(reverse=True)
print('launches with < 512 workgroups and >= %.0f us: us/step, launches/step, avg us, workgroups, threads, kernel' % thr)
if only:
    print('%s: %.1f us/step in %.1f launches/step' % (only, sum(r[0] for r in rows), sum(r[1] for r in rows)))
for r in rows[:60]:
    print('  %8.1f  %5.1f  %7.1f  %5d  %4d  %s' % r)
