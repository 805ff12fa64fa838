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


def resize_linear_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) for a uint8 (H,W,C) image with the default interpolation -- what the
    reference's loaders call (utils/visual_utils.py:9,24,42).  cv2 is absent from this image, so this restates
    OpenCV's published INTER_LINEAR algorithm for 8-bit images (imgproc/resize.cpp): half-pixel centres, NO
    antialiasing when shrinking, 11-bit fixed-point coefficients (cvRound), horizontal pass in int32, vertical pass
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; and OpenCV's special case: an exact 2x shrink in both
    directions is the rounded mean of each 2x2 block (INTER_LINEAR is replaced by the fast INTER_AREA path there).
    Parity with cv2 itself is unpinned (nothing to run it against here); tests/test_data_cpu.py holds it to the
    algorithm's own properties and to float bilinear within one grey level."""
    img = np.ascontiguousarray(img)
    H, W = img.shape[:2]
    if (out_h, out_w) == (H, W):
        return img.copy()
    if H == 2 * out_h and W == 2 * out_w:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(n_src, n_dst):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        f = f - i0
        lo = i0 < 0
        f[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_src - 1
        f[hi], i0[hi] = 0.0, n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        c1 = np.rint(f * 2048.0).astype(np.int32)                  # cvRound: round half to even
        c0 = np.rint((1.0 - f) * 2048.0).astype(np.int32)
        return i0, i1, c0, c1
    x0, x1, a0, a1 = taps(W, out_w)
    y0, y1, b0, b1 = taps(H, out_h)
    src = img.astype(np.int32)
    shape = (1, out_w) + (1,) * (img.ndim - 2)
    rows = src[:, x0] * a0.reshape(shape) + src[:, x1] * a1.reshape(shape)          # (H, out_w, C) int32
    S0, S1 = rows[y0], rows[y1]
    vb = (out_h, 1) + (1,) * (img.ndim - 2)
    out = (((b0.reshape(vb) * (S0 >> 4)) >> 16) + ((b1.reshape(vb) * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def load_image(path, size=None):
    """RGB float tensor (3,H,W) in [0,1] as the reference's loaders produce it (utils/visual_utils.py:6-31:
    cv2.imread -> cv2.resize -> BGR->RGB -> /255).  `size`: None (as is), an int (square, training:
    load_train_img) or (h, w).  Decoding is PIL's (libjpeg, like cv2's); the resize is resize_linear_u8."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)
    if size is not None:
        h, w = (size, size) if isinstance(size, int) else size
        img = resize_linear_u8(img, h, w)
    return torch.from_numpy(img.astype(np.float32).transpose(2, 0, 1) / 255.0)


def load_image_short_side(path, short_size=600):
    """utils/visual_utils.py:34-47 (full-resolution inference): the short side scaled to `short_size`."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)
    h, w = img.shape[:2]
    ratio = short_size / min(h, w)
    img = resize_linear_u8(img, int(np.round(h * ratio)), int(np.round(w * ratio)))
    return torch.from_numpy(img.astype(np.float32).transpose(2, 0, 1) / 255.0)


def parse_sent(desc):
    """Request text -> tokens as utils/text_utils.py:9-26 cleans them: whitespace split, lower case, punctuation removed
    from every token, one-letter tokens and tokens with non-letters dropped."""
    import string
    table = str.maketrans('', '', string.punctuation)
    words = [w.lower().translate(table) for w in desc.split()]
    return [w for w in words if len(w) > 1 and w.isalpha()]


def txt2idx(sent, vocab2id, max_len):
    """utils/text_utils.py:42-67: (1, max_len) token ids = START (1), the request's tokens (unknown words -> 3) cut to
    max_len - 2, END (2) right behind them, zero padding."""
    body = max_len - 2
    ids = [vocab2id.get(tok, 3) for tok in parse_sent(sent)][:body]
    row = [1] + ids + [2] + [0] * (body - len(ids))
    return torch.tensor(row, dtype=torch.long).unsqueeze(0)


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


class FiveK(Dataset):
    """(img_x, img_y, req_idx, req) per item with NO planned actions, like datasets/FiveKdataset.py:24-52: the split the
    reference validates and tests on (train_seq2seqL1.py:155-156, batch_size=1).  phase 'train': square training size;
    any other phase: full resolution with the short side scaled to 600 pixels (load_infer_img_short_size_bounded)."""

    def __init__(self, img_dir, anno_dir, phase='val', session=1, train_img_size=128, short_size=600):
        self.img_dir, self.phase, self.size, self.short_size = img_dir, phase, train_img_size, short_size
        with open(os.path.join(anno_dir, '{}_sess_{}.json'.format(phase, session))) as f:
            self.data = json.load(f)

    def __len__(self):
        return len(self.data)

    def _load(self, name):
        path = os.path.join(self.img_dir, name)
        return load_image(path, self.size) if self.phase == 'train' else load_image_short_side(path, self.short_size)

    def __getitem__(self, item):
        dic = self.data[item]
        return self._load(dic['input']), self._load(dic['output']), np.array(dic['request_idx']), dic['request']


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
