"""Does the train step run into the board's power / clock management?  Runs tools/step_only.py in a child process (eager,
then as one whole-step hipGraph) and samples the GPU's shader clock and power draw from sysfs / rocm-smi meanwhile:
python tools/clock_watch.py [steps]"""
import glob, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else '600'


def sample():
    """Counters of the BUSIEST card (the box shows every GPU of the node; the leased one is the one drawing power)."""
    best = {}
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        out = {}
        for hw in glob.glob(os.path.join(card, 'hwmon/hwmon*')):
            for name in ('power1_average', 'power1_input', 'freq1_input', 'power1_cap'):
                try:
                    with open(os.path.join(hw, name)) as f:
                        out[name] = int(f.read().strip())
                except (OSError, ValueError):
                    pass
        try:
            cur = [ln for ln in open(os.path.join(card, 'pp_dpm_sclk')).read().splitlines() if ln.rstrip().endswith('*')]
            if cur:
                out['sclk'] = cur[0]
        except OSError:
            pass
        if out.get('power1_input', out.get('power1_average', 0)) > best.get('power1_input', best.get('power1_average', -1)):
            best = out
    out = best
    if not out:
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=10)
            card = next(iter(json.loads(r.stdout).values()))
            out = {k: v for k, v in card.items() if 'sclk' in k.lower() or 'power' in k.lower()}
        except Exception as e:                     # noqa: BLE001
            out = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


for mode, label in (('0', 'eager'), ('1', 'whole-step hipGraph')):
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'step_only.py'), steps, mode, '0'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    rows = []
    while p.poll() is None:                        # (the first ~10 s are start-up: the last two thirds of the samples are reported)
        rows.append(sample())
        time.sleep(0.25)
    rows = rows[len(rows) // 3:]
    out = p.communicate()[0]
    print('==', label, out.strip().splitlines()[0] if out.strip() else '')
    keys = sorted({k for r in rows for k in r})
    for k in keys:
        vals = [r[k] for r in rows if k in r]
        nums = [v for v in vals if isinstance(v, (int, float))]
        if nums:
            print('   %-16s n=%d min %s mean %.1f max %s' % (k, len(nums), min(nums), sum(nums) / len(nums), max(nums)))
        else:
            print('   %-16s %s' % (k, sorted(set(map(str, vals)))[:6]))
