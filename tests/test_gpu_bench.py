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


_TWO_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
rank = int(os.environ['RANK'])
torch.cuda.set_device(rank)
dev = torch.device('cuda', rank)
dist.init_process_group('nccl', rank=rank, world_size=2, device_id=dev)
from t2onet_amd.train import Communicator
comm = Communicator(dev)                                     # id by an object broadcast on the group, t2o_comm_init_rank on both
t = torch.full((1 << 20,), float(rank + 1), device=dev)
comm.all_reduce_(t)                                          # 1 + 2
ok = bool((t == 3.0).all())
comm.all_reduce_(t, mean=True)                               # (3 + 3) / 2
ok = ok and bool((t == 3.0).all())
u = torch.full((5,), float(rank), device=dev)
dist.all_reduce(u)                                           # torch's own communicator beside it still works
ok = ok and bool((u == 1.0).all())
comm.close()
dist.destroy_process_group()
print('RANK%%d %%s' %% (rank, 'OK' if ok else 'WRONG'), flush=True)
'''


def test_two_ranks_through_the_c_abi_communicator(tmp_path):
    """t2o_comm_unique_id / t2o_comm_init_rank / t2o_allreduce(_mean) with TWO ranks over RCCL, next to torch's own communicator
    (the N > 1 path of Communicator had only ever run with one rank).  Needs two GPUs: skipped on the 1-GPU pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (the pool leases one)')
    script = tmp_path / 'worker.py'
    script.write_text(_TWO_RANK_WORKER % {'root': ROOT})
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'RANK0 OK' in outs[0] and 'RANK1 OK' in outs[1], outs
