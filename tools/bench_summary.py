"""Print a compact table from one bench.py JSON line (stdin or file) [+ the bench_detail.json it names as 2nd argument]."""
import json
import sys

d = json.loads((open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()).strip().splitlines()[-1])
if len(sys.argv) > 2:
    try:
        full = json.load(open(sys.argv[2]))
        full.update({k: v for k, v in d.items() if k not in full})
        d = full
    except (OSError, ValueError):
        pass
print('HEADLINE train step: %s img/s  %s ms/step  (n_gpus %s)  %s' % (d.get('value'), d.get('ms_per_step'), d.get('n_gpus'), d.get('error', '')))
t = d.get('train_step')
if t:
    print('   host enqueue %.1f ms, graphs %s, mfma frac %.3f (%.1f TF/s), loss %.5f' % (
        t['host_enqueue_ms_per_step'], t.get('encoder_hipgraphs'), (t.get('roofline') or d.get('train_roofline'))['frac'], (t.get('roofline') or d.get('train_roofline'))['achieved'], t['loss']))
r = d.get('roofline')
if r and 'kernel' in r:
    print('roofline: %s %.1f %s frac %.3f (%.4f ms/launch) traffic %s' % (r['kernel'], r['achieved'], r['unit'], r['frac'], r['avg_launch_ms'], r['traffic']))
elif r:
    print('roofline:', r)
for k, v in (d.get('conv_kernels') or {}).items():
    print('      %-40s %8.2f us  %6.1f TF/s  %s' % (k, v['ms'] * 1e3, v['TFLOPs'], v['kernel']))
r = d.get('executor_roofline')
if r:
    print('executor_roofline: %s %.0f GB/s moved, frac %.3f (credited %.3f) traffic %s' % (r['kernel'], r['achieved'], r['frac'], r['credited_frac'], r['traffic']))
for name, leg in d.get('executor', {}).items():
    if not isinstance(leg, dict):
        print('executor', name, leg)
        continue
    for path in ('value_grad', 'fused', 'materialised'):
        m = leg.get(path)
        if m is None:
            continue
        print('%-12s %-12s %9.0f img/s  %8.4f ms/step  frac %.3f' % (name, path, m['value'], m['ms_per_step'], m['frac_of_peak']) + ('  fused-min frac %.3f' % m['fused_min_frac'] if 'fused_min_frac' in m else ''))
        for k, v in m.get('kernels', {}).items():
            print('      %-22s %8.2f us  %7.0f GB/s alg  %7.0f GB/s moved' % (k, v['ms'] * 1e3, v['GBps'], v.get('hbm_min_GBps', 0)))
    if 'api_path' in leg:
        print('   api path:', leg['api_path'])
c = d.get('cpu_baseline')
if c:
    print('cpu configs[2] (headline): %s img/s on %s threads of %s physical cores; configs[1] %s; configs[0] %s; host %s'
          % (c.get('value'), c.get('threads', c.get('cores')), c.get('physical_cores'), c.get('configs_1', {}).get('value'),
             c.get('configs_0', {}).get('value'), c.get('host')))
    print('   parity:', c.get('parity'))
