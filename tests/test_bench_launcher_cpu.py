"""bench.py --gpus N must start N ranks itself (or refuse): driven here with the gloo stand-in step
(`--selftest`), world size 2, through the same launcher / rendezvous / MAX-over-ranks / rank-0 JSON code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(argv, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=e, timeout=300)


def test_launcher_starts_two_ranks_and_relays_one_line():
    r = _run(['--gpus', '2', '--selftest', '--steps', '4', '--warmup', '1'])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['world_size'] == 2 and d['config']['backend'] == 'gloo'
    pids = d['config']['rank_pids']
    assert len(set(pids)) == 2 and 0 not in pids            # two distinct processes took part in the collective
    assert d['steps'] == 4 and d['warmup'] == 1 and d['ms_per_step'] > 0
    assert abs(d['mean_after_allreduce'] - 1.5) < 1e-6      # mean of ranks' (rank + 1): the all-reduce really ran
    # the N > 1 keys a scaling run is checked by: the collective's own time and size, the spread over the ranks
    ts = d['train_step']
    assert ts['allreduce_ms'] > 0 and ts['allreduce_bytes'] == 4 << 16
    assert 0 < ts['ms_per_step_over_ranks']['min'] <= ts['ms_per_step_over_ranks']['max']


def test_refuses_to_run_on_fewer_devices_than_asked():
    r = _run(['--gpus', '2', '--steps', '1'])                # no GPU in the build container
    assert r.returncode == 2 and 'refusing' in r.stderr and '{' not in r.stdout


def test_rank_refuses_a_world_size_that_is_not_gpus():
    r = _run(['--gpus', '4', '--selftest'], env={'RANK': '0', 'WORLD_SIZE': '1'})
    assert r.returncode == 2 and 'WORLD_SIZE 1 != --gpus 4' in r.stderr


def test_under_torch_distributed_run():
    """The driver's own launch form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ..."""
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29731', BENCH, '--gpus', '2', '--selftest',
                        '--steps', '3', '--warmup', '1'], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


def test_a_dying_rank_stops_the_job_quickly():
    """If one rank exits with an error the launcher must not wait for the others to time out in a collective."""
    import time
    t0 = time.time()
    r = _run(['--gpus', '2', '--selftest', '--steps', '3', '--warmup', '1'], env={'T2O_SELFTEST_FAIL_RANK': '1'})
    assert r.returncode != 0 and 'exited with code' in r.stderr
    assert time.time() - t0 < 60


REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def test_last_stdout_line_is_small_and_carries_the_contract_keys(tmp_path):
    """the driver parses the LAST stdout line; round 4's was 24 KB and was not parsed.  The selftest sends a
    record as bulky as a real run's through the real emission path: the line stays small, the bulk lands in the side file."""
    r = _run(['--gpus', '1', '--selftest', '--steps', '2', '--warmup', '1'], env={'T2O_BENCH_DETAIL_DIR': str(tmp_path)})
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 6144
    d = json.loads(last)
    for k in REQUIRED:
        assert k in d, k
    assert d['roofline']['bound'] == 'mfma' and 'note' not in d['roofline'] and 'kernels' not in d['executor']['bs256']['fused']
    full = json.load(open(os.path.join(str(tmp_path), 'bench_detail.json')))
    assert len(json.dumps(full)) > 20000 and 'conv_kernels' in full and 'kernels' in full['executor']['bs256']['fused']
    assert d['detail'] == 'bench_detail.json'


def test_compact_line_of_a_real_record():
    """Round 4's own 24 KB record (profiles/r04_bench.json) through compact_line: < 6 KB, every contract key, the three
    roofline objects and the CPU baseline with its sub-configs."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')))
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full, 'bench_detail.json')
    s = json.dumps(c, separators=(',', ':'))
    assert len(s) < bench.LINE_LIMIT
    for k in REQUIRED + ('executor_roofline', 'train_roofline', 'parity'):
        assert k in c, k
    assert c['value'] == full['value'] and c['ms_per_step'] == full['ms_per_step']
    assert c['roofline']['frac'] == full['roofline']['frac'] and c['cpu_baseline']['value'] == full['cpu_baseline']['value']
    assert c['cpu_baseline']['configs_1']['value'] == full['cpu_baseline']['configs_1']['value']
