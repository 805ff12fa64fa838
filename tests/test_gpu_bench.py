"""bench.py on the GPU box, as one rank of a data-parallel job (N = 8 readiness without a node): started the way
torch.distributed.run starts a rank -- RANK / WORLD_SIZE / MASTER_* in the environment -- with T2O_OWN_COMM=1, so that the step's
one collective is the C ABI's t2o_allreduce_mean (include/t2onet_hip.h) on a communicator built through
t2o_comm_*, INSIDE the timed region; the line must carry the keys a scaling run is checked by."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_one_rank_job_with_the_c_abi_allreduce_on_the_timed_path(tmp_path):
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               T2O_OWN_COMM='1', T2O_BENCH_DETAIL_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--quick',
                        '--no-cpu-baseline', '--exec-steps', '5', '--exec-warmup', '2'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['config']['backend'].startswith('nccl')
    ts = d['train_step']
    assert ts['own_communicator'] is True                   # the C ABI's all-reduce ran in every timed step
    assert ts['allreduce_ms'] is not None and ts['allreduce_ms'] > 0 and ts['allreduce_bytes'] == 22165917 * 4 + (ts['allreduce_bytes'] - 22165917 * 4)
    assert ts['allreduce_bytes'] >= 22165917 * 4            # the flat buffer: every parameter on a 256-byte boundary
    assert 0 < ts['ms_per_step_over_ranks']['min'] <= ts['ms_per_step_over_ranks']['max']
    with open(os.path.join(str(tmp_path), 'bench_detail.json')) as f:
        detail = json.load(f)
    assert detail['train_step']['parameters_finite'] is True
