"""The individual idle gaps of the device in the LAST traced step: rocprofv3 --kernel-trace CSV -> every gap above a threshold
with the kernels on both sides and its position in the step.  usage: python tools/trace_biggaps.py <kernel_trace.csv> <kernels per step> [min us]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
per = int(sys.argv[2])
thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 8e3
rows = rows[-2 * per:-per] if len(rows) >= 2 * per else rows


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:58]


cur_end = rows[0][1]
t0 = rows[0][0]
tot = small = 0
out = []
for i, (s, e, n) in enumerate(rows[1:], 1):
    if s > cur_end:
        g = s - cur_end
        tot += g
        if g >= thr:
            out.append((i, (s - t0) / 1e6, g / 1e3, short(rows[i - 1][2]), short(n)))
        else:
            small += g
    cur_end = max(cur_end, e)
print('one step: %d kernels, span %.2f ms, idle %.2f ms of which gaps under %.0f us: %.2f ms' % (len(rows), (cur_end - t0) / 1e6, tot / 1e6, thr / 1e3, small / 1e6))
for i, at, g, a, b in out:
    print('  #%04d at %6.2f ms  gap %6.1f us   %-58s -> %s' % (i, at, g, a, b))
