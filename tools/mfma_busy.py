"""Matrix-pipe occupancy per kernel from two rocprofv3 passes over the same command (tools/gpu_conv_profile.sh):
a --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU, SQ_INSTS_SALU) and a
--kernel-trace --stats pass (durations without counter overhead).
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x 256 CUs / 8 XCDs ...) is calibrated on the bare MFMA
loop of tools/diag/mfma_clock.hip (by construction ~100 % busy), printed first."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        name = name.split('(')[0][:64]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob(os.path.join(out, 'stats*', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0][:64]
        dur[name] = (float(r['AverageNs']), int(r['Calls']))
rows = []
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
        continue
    rows.append((k, m))
cal = [m for k, m in rows if k.startswith('k_mfma_loop')]
# SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the bare loop = the value that means "every SIMD's pipe busy every cycle"
full = cal[0]['SQ_VALU_MFMA_BUSY_CYCLES'] / cal[0]['GRBM_GUI_ACTIVE'] if cal else None
print('calibration: bare MFMA loop SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE =', full)
for k, m in sorted(rows, key=lambda r: r[0]):
    ratio = m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['GRBM_GUI_ACTIVE']
    d = dur.get(k, (0.0, 0))
    print('%-64s dur %7.1f us x%-4d  MFMA busy %5.3f of the bare loop   VALU %9.0f  SALU %9.0f wave-instructions' % (
        k, d[0] / 1e3, d[1], ratio / full if full else float('nan'), m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_SALU', 0)))
