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


def test_resize_restates_cv2_inter_linear_properties(tmp_path):
    """data.resize_linear_u8 = OpenCV's 8-bit INTER_LINEAR as published (cv2 itself is absent: unpinned against it).
    Held to the algorithm's own properties and to float bilinear without antialiasing within one grey level, on a
    JPEG written and decoded here."""
    import io
    import torch
    from PIL import Image
    from t2onet_amd import data
    rng = np.random.default_rng(3)
    base = (rng.random((37, 53, 3)) * 255).astype(np.uint8)
    smooth = np.asarray(Image.fromarray(base).resize((212, 148), Image.BICUBIC))       # a natural-ish image
    buf = io.BytesIO()
    Image.fromarray(smooth).save(buf, format='JPEG', quality=92)
    path = tmp_path / 'x.jpg'
    path.write_bytes(buf.getvalue())
    img = np.asarray(Image.open(str(path)).convert('RGB'))
    H, W = img.shape[:2]
    assert np.array_equal(data.resize_linear_u8(img, H, W), img)                       # same size: untouched
    const = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(data.resize_linear_u8(const, 13, 17) == 77) and np.all(data.resize_linear_u8(const, 41, 64) == 77)
    half = data.resize_linear_u8(img, H // 2, W // 2)                                  # exact 2x: rounded 2x2 block mean
    s = img.astype(np.int32)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    for oh, ow in ((128, 128), (64, 100), (200, 300), (31, 17)):
        got = data.resize_linear_u8(img, oh, ow).astype(np.float32)
        t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, size=(oh, ow), mode='bilinear', align_corners=False, antialias=False)[0]
        assert np.abs(got - ref.permute(1, 2, 0).numpy()).max() <= 1.0 + 1e-3, (oh, ow)
    x = data.load_image(str(path), 128)
    assert tuple(x.shape) == (3, 128, 128) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    np.testing.assert_array_equal((x * 255).round().numpy().astype(np.uint8), data.resize_linear_u8(img, 128, 128).transpose(2, 0, 1))
    y = data.load_image_short_side(str(path), 100)
    assert min(y.shape[1:]) == 100 and tuple(y.shape[1:]) == (100, int(np.round(W * 100 / H)))


def test_fivek_datasets_on_a_generated_tree(tmp_path):
    """FiveKAct.__getitem__ / FiveK.__getitem__ (datasets/FiveKdataset.py:42-52, :118-135) on a tree written here in the
    reference's layout: JPEG decode + resize, planner records, the validation split at full resolution (short side 600)."""
    from t2onet_amd import data
    from tests import fivek_tree
    img_dir, anno_dir, act_dir, _ = fivek_tree.write_tree(str(tmp_path), n_train=4, n_val=2)
    ds = data.FiveKAct(img_dir, anno_dir, act_dir, 'train', 1, 64)
    assert len(ds) == 4
    for i in range(4):
        img_x, imgs, x, ops, params, req = ds[i]
        assert img_x.shape == (3, 64, 64) and imgs.shape == (6, 3, 64, 64) and x.shape == (17,) and req == 'make it %d' % i
        n = int((ops > 2).sum())
        assert ops[0] == 1 and ops[n + 1] == 2 and (ops[n + 2:] == 0).all() and 1 <= n <= 5
        assert float(imgs[n:5].abs().sum()) == 0.0 and float(imgs[5].sum()) > 0          # unused steps stay zero, the target is last
        assert float(np.abs(params[n:]).sum()) == 0.0 and np.abs(params[:n]).max() <= 5.0
        import json
        rec = json.load(open(os.path.join(act_dir, 'train%d' % i, '%05d.json' % i)))
        np.testing.assert_array_equal(ops, data.parse_action_record(rec)[0])
        np.testing.assert_array_equal(imgs[0].numpy(), data.load_image(os.path.join(act_dir, 'train%d' % i, 'edit0.jpg'), 64).numpy())
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))
    assert batch[1].shape == (2, 6, 3, 64, 64) and batch[3].shape == (2, 7) and batch[4].shape == (2, 5, 24)
    val = data.FiveK(img_dir, anno_dir, 'val', 1)
    img_x, img_y, x, req = val[1]
    assert min(img_x.shape[1:]) == 600 and img_x.shape == img_y.shape and x.shape == (17,)
    assert img_x.shape[1:] == (600, 900)                                                  # 96 x 144 source
    tr = data.FiveK(img_dir, anno_dir, 'train', 1, train_img_size=32)
    assert tr[0][0].shape == (3, 32, 32)
