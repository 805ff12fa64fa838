"""Ordered kernel list of ONE train step from a rocprofv3 kernel trace (csv): python tools/step_sequence.py trace.csv [step_index]
Own convolution / batch-norm kernels of the encoder are collapsed into one line per run."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if 'k_adam' in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 2
lo, hi = adam[k] + 1, adam[k + 1] + 1
enc = re.compile(r'k_conv|k_bn_nhwc|k_sc_|k_stem|k_wino|Cijk_Alik_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT256|MT64x64x128|MT128x128x64|k_conv_wgrad_reduce|k_conv_flip')
run = 0; run_t = 0
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:110]
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if enc.search(n):
        run += 1; run_t += d; continue
    if run:
        print('      ... %d encoder kernels, %.0f us' % (run, run_t)); run = 0; run_t = 0
    print('%9.1f %6.1f us  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, short(n)))
if run: print('      ... %d encoder kernels, %.0f us' % (run, run_t))
