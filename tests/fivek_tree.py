"""A tiny on-disk tree in the reference's FiveK layout (datasets/FiveKdataset.py:24-135), written by the tests:

    img_dir/{k}_in.jpg, {k}_out.jpg                      input / expert-retouched pairs (JPEG)
    anno_dir/{phase}_sess_1.json                         [{'input', 'output', 'request', 'request_idx'}, ...]
    act_dir/{phase}{i}/{i:05d}.json + edit{k}.jpg        the planner's record and intermediate images (train only)
"""
import json
import os

import numpy as np


def _jpeg(path, h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    small = (rng.random((h // 8 + 2, w // 8 + 2, 3)) * 255).astype(np.uint8)
    Image.fromarray(small).resize((w, h), Image.BICUBIC).save(path, format='JPEG', quality=90)


def write_tree(root, n_train=8, n_val=2, vocab=918):
    img_dir, anno_dir, act_dir = (os.path.join(root, d) for d in ('images', 'annotations', 'actions'))
    for d in (img_dir, anno_dir, act_dir):
        os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(0)
    names = ['brightness', 'contrast', 'saturation', 'color', 'tone', 'sharpness']
    npar = {'brightness': 1, 'contrast': 1, 'saturation': 1, 'color': 24, 'tone': 8, 'sharpness': 1}
    for phase, n in (('train', n_train), ('val', n_val)):
        items = []
        for i in range(n):
            h, w = (96, 144) if i % 2 else (120, 80)               # portrait and landscape, not multiples of the train size
            fin, fout = '%s%d_in.jpg' % (phase, i), '%s%d_out.jpg' % (phase, i)
            _jpeg(os.path.join(img_dir, fin), h, w, 100 + i)
            _jpeg(os.path.join(img_dir, fout), h, w, 200 + i)
            k = int(rng.integers(2, 14))
            idx = [1] + [int(v) for v in rng.integers(4, vocab, k)] + [2]
            idx += [0] * (17 - len(idx))
            items.append({'input': fin, 'output': fout, 'request': 'make it %d' % i, 'request_idx': idx})
            if phase == 'train':
                d = os.path.join(act_dir, 'train%d' % i)
                os.makedirs(d, exist_ok=True)
                steps = int(rng.integers(1, 6))
                order = [names[j] for j in rng.permutation(6)[:steps]]
                dist, seq = 0.30, []
                for s, name in enumerate(order):
                    dist *= 0.6                                    # every step improves by more than 1 %: nothing truncated
                    vals = [float(v) for v in (rng.random(npar[name]) * (1.0 if npar[name] == 1 else 0.5) + (0.0 if npar[name] == 1 else 0.75))]
                    seq.append([name, vals, dist])
                    _jpeg(os.path.join(d, 'edit%d.jpg' % s), h, w, 300 + 10 * i + s)
                with open(os.path.join(d, '%05d.json' % i), 'w') as f:
                    json.dump({'init distance': 0.30, 'operation sequence': [seq]}, f)
        with open(os.path.join(anno_dir, '%s_sess_1.json' % phase), 'w') as f:
            json.dump(items, f)
    glove = (np.random.default_rng(1).random((vocab - 4, 300)).astype(np.float32) - 0.5)
    np.save(os.path.join(root, 'glove.npy'), glove)
    return img_dir, anno_dir, act_dir, os.path.join(root, 'glove.npy')
