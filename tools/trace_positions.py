"""Where in the step is a kernel slow?  rocprofv3 --kernel-trace CSV -> for every launch position of the named kernel within a
step (launches per step given), the average / min / max duration over the traced steps, the kernel that ran before it, and the
time another kernel was executing concurrently (side stream).
usage: python tools/trace_positions.py <kernel_trace.csv> <name substring> <launches per step> [steps to use]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
want, per = sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
idx = [i for i, r in enumerate(rows) if want in r[2]]
idx = idx[-per * steps:]
pos = {}
for n, i in enumerate(idx):
    s, e, _ = rows[i]
    # concurrent time: any other kernel whose interval intersects
    conc = 0
    j = i - 1
    while j >= 0 and rows[j][0] > s - 2_000_000:
        if rows[j][1] > s:
            conc += min(rows[j][1], e) - s
        j -= 1
    j = i + 1
    while j < len(rows) and rows[j][0] < e:
        conc += min(rows[j][1], e) - rows[j][0]
        j += 1
    prev = rows[i - 1][2].split('(')[0][-40:] if i else ''
    pos.setdefault(n % per, []).append((e - s, conc, prev))
print('%s: %d launches/step over %d steps' % (want, per, len(idx) // per))
tot = 0.0
for p in sorted(pos):
    d = [x[0] for x in pos[p]]
    c = [x[1] for x in pos[p]]
    tot += sum(d) / len(d)
    print('  #%02d avg %7.1f us  min %7.1f  max %7.1f   concurrent %6.1f us   after %s' % (p, sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(c) / len(c) / 1e3, pos[p][0][2]))
print('sum of averages %.1f us per step' % (tot / 1e3))
