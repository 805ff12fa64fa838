"""Data-step row: the planner-record logic against the reference's own FiveKAct.get_act /
analyze_traj (tests/golden/data.npz, produced by tools/gen_golden.py on synthetic records)."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records():
    spec = importlib.util.spec_from_file_location('gen_golden', os.path.join(ROOT, 'tools', 'gen_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.synthetic_records()


def test_action_records_match_reference(golden_dir):
    from t2onet_amd import data
    g = np.load(os.path.join(golden_dir, 'data.npz'))
    kept = set()
    for i, rec in enumerate(_records()):
        ops, params, n = data.parse_action_record(rec)
        np.testing.assert_array_equal(ops, g['ops%d' % i])
        np.testing.assert_array_equal(params, g['params%d' % i])
        dists = [rec['init distance']] + [v[2] for v in rec['operation sequence'][0]]
        assert data.analyze_traj(dists) == int(g['trunc%d' % i])
        kept.add(n)
    assert len(kept) > 1                       # the fixtures exercise different truncation lengths


def test_synthetic_dataset_shapes():
    from t2onet_amd import data
    ds = data.SyntheticFiveK(n=4, size=32)
    img_x, imgs, x, ops, params, req = ds[1]
    assert img_x.shape == (3, 32, 32) and imgs.shape == (6, 3, 32, 32) and x.shape == (17,)
    assert ops.shape == (7,) and ops[0] == 1 and ops[-1] == 2 and params.shape == (5, 24)
    a, b = ds[1], ds[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    batch = next(iter(loader))
    assert batch[1].shape == (2, 6, 3, 32, 32)
