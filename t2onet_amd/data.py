"""The data step in front of the train loop: turning a planned-action
record into the tensors `train_seq2seqL1.py` consumes, as datasets/FiveKdataset.py:54-135 does.

Image decoding is kept behind `load_image` (PIL, bilinear resize to a square like
utils/visual_utils.py:6-14; the reference uses cv2, absent here, so pixel-exact parity of the
decode/resize is NOT claimed).  The record logic -- trajectory truncation, operator ids,
curve-parameter normalisation, outlier handling -- is restated exactly and tested against the
reference's own functions on synthetic records (tests/golden/data.npz).
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

ACTIONS = ['brightness', 'contrast', 'saturation', 'color', 'inpaint', 'tone', 'sharpness', 'white']
ACT2PN = {'brightness': 1, 'contrast': 1, 'saturation': 1, 'color': 24, 'inpaint': 0, 'tone': 8, 'sharpness': 1, 'white': 0}
OP_MAX_LEN = 5


def analyze_traj(seq):
    """Number of planned steps to keep: stop at the first step whose distance drop is not more
    than 1 % of the initial distance (FiveKdataset.py:54-64); at least 1."""
    seq = np.array(seq)
    over_shot = (seq[:-1] - seq[1:]) / seq[0]
    stops = np.where(~(over_shot > 0.01))[0]
    trunc_len = int(stops[0]) if len(stops) else len(over_shot)
    return max(trunc_len, 1)


def parse_action_record(record, op_max_len=OP_MAX_LEN):
    """record = the planner's JSON ({'init distance': d0, 'operation sequence': [[(name, params, dist), ...], ...]}).
    Returns (op_seq (op_max_len+2,) int64 = [START, ids..., END, 0...], params (op_max_len,24) float32, n_kept)
    exactly as FiveKAct.get_act (FiveKdataset.py:86-116): curve parameters divided by their max
    magnitude, one-parameter values beyond +-5 replaced by 0."""
    seq = record['operation sequence'][0]
    dists = [record['init distance']] + [v[2] for v in seq]
    trunc_len = min(analyze_traj(dists), op_max_len)
    seq = seq[:trunc_len]
    params = np.zeros((op_max_len, 24), dtype=np.float32)
    op_seq = np.zeros(op_max_len + 2, dtype=np.int64)
    for i, act in enumerate(seq):
        name, values = act[0], np.array(act[1], dtype=np.float64)
        op_seq[i + 1] = ACTIONS.index(name) + 3
        n = ACT2PN[name]
        if name in ('color', 'tone'):
            params[i, :n] = values / np.abs(values).max()
        elif abs(values[0]) > 5:
            params[i, :n] = 0.0
        else:
            params[i, :n] = values
    op_seq[0] = 1
    op_seq[len(seq) + 1] = 2
    return op_seq, params, trunc_len


def load_image(path, size=None):
    """RGB float tensor (3,H,W) in [0,1]; `size` -> square resize (training), None -> as is."""
    from PIL import Image
    img = Image.open(path).convert('RGB')
    if size is not None:
        img = img.resize((size, size), Image.BILINEAR)
    return torch.from_numpy(np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0)


class FiveKAct(Dataset):
    """(img_x, img_ys (6,3,S,S), req_idx, ops (7,), params (5,24), req) per item, like
    datasets/FiveKdataset.py:67-135.  Directory layout as the reference's:
    anno_dir/{phase}_sess_{session}.json, act_dir/{phase}{i}/{i:05d}.json + edit{k}.jpg."""

    def __init__(self, img_dir, anno_dir, act_dir, phase='train', session=1, train_img_size=128):
        self.img_dir, self.act_dir, self.phase, self.size = img_dir, act_dir, phase, train_img_size
        with open(os.path.join(anno_dir, '{}_sess_{}.json'.format(phase, session))) as f:
            self.data = json.load(f)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        dic = self.data[item]
        item_dir = os.path.join(self.act_dir, '{}{}'.format(self.phase, item))
        with open(os.path.join(item_dir, '{:05d}.json'.format(item))) as f:
            ops, params, n = parse_action_record(json.load(f))
        imgs = torch.zeros(OP_MAX_LEN + 1, 3, self.size, self.size)
        for k in range(n):
            imgs[k] = load_image(os.path.join(item_dir, 'edit{}.jpg'.format(k)), self.size)
        imgs[OP_MAX_LEN] = load_image(os.path.join(self.img_dir, dic['output']), self.size)
        img_x = load_image(os.path.join(self.img_dir, dic['input']), self.size)
        return img_x, imgs, np.array(dic['request_idx']), ops, params, dic['request']


class SyntheticFiveK(Dataset):
    """FiveK-shaped random items: what bench.py and the tests train on."""

    def __init__(self, n=256, size=128, seed=10, vocab=918, req_len=17):
        self.n, self.size, self.seed, self.vocab, self.req_len = n, size, seed, vocab, req_len

    def __len__(self):
        return self.n

    def __getitem__(self, item):
        g = torch.Generator().manual_seed(self.seed * 100003 + item)
        S = self.size
        img_x = torch.rand(3, S, S, generator=g)
        imgs = torch.rand(OP_MAX_LEN + 1, 3, S, S, generator=g)
        k = int(torch.randint(1, self.req_len - 1, (1,), generator=g))
        x = torch.zeros(self.req_len, dtype=torch.long)
        x[0] = 1
        x[1:1 + k] = torch.randint(4, self.vocab, (k,), generator=g)
        x[1 + k] = 2
        pool = torch.tensor([3, 4, 5, 6, 8, 9])[torch.randperm(6, generator=g)[:OP_MAX_LEN]]
        ops = torch.cat([torch.tensor([1]), pool, torch.tensor([2])])
        params = torch.zeros(OP_MAX_LEN, 24)
        npar = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
        for i, o in enumerate(pool.tolist()):
            params[i, :npar[o]] = torch.rand(npar[o], generator=g) * 2 - 1
        return img_x, imgs, x, ops, params, 'synthetic request'
