"""Print a compact table from one bench.py JSON line (stdin or file)."""
import json
import sys

d = json.loads((open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()).strip().splitlines()[-1])
print('value %.0f img/s  ms/step %.4f  dominant %s frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']))
for k, v in d['kernels'].items():
    print('   %-20s %8.2f us  %7.0f GB/s alg  %7.0f GB/s min' % (k, v['ms'] * 1e3, v['GBps'], v.get('hbm_min_GBps', 0)))
m = d.get('materialised_path')
if m:
    print(' materialised: %.0f img/s  ms/step %.4f  frac %.3f' % (m['value'], m['ms_per_step'], m['frac_of_peak']))
    for k, v in m['kernels'].items():
        print('   %-20s %8.2f us  %7.0f GB/s' % (k, v['ms'] * 1e3, v['GBps']))
if 'cpu_baseline' in d:
    print(' cpu:', d['cpu_baseline'])
if 'train_step' in d:
    print(' train:', d['train_step'])
if 'executor_api_path' in d:
    print(' api path:', d['executor_api_path'])
