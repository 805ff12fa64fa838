#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python tools/gen_golden.py            # writes tests/golden/

/root/reference does not exist on the GPU box, and its Python cannot travel;
this script imports it here, feeds it the formula-generated inputs/weights of
oracle/synth.py and stores only the reference's OUTPUTS (data, no source).

Import shims (modules the reference imports at module scope but which are
absent from this image; none of them is on the numeric path except kornia):
  cv2, h5py                         import-only here (h5py: GloVe table -> formula)
  pyutils.edgeconnect.src.*         InpaintOperator.__init__ builds it unconditionally
                                    (operators.py:631-649); git submodule, empty dir
  kornia.rgb_to_hsv / hsv_to_rgb    -> oracle/hsv_spec.py (the HSV spec this build owns;
                                    kornia is unpinned and not vendored)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
from oracle import hsv_spec, synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def install_shims():
    cv2 = types.ModuleType('cv2')
    sys.modules['cv2'] = cv2

    h5py = types.ModuleType('h5py')

    class _DS:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, _):
            return self.a

    class File(dict):
        def __init__(self, path, mode='r'):
            rows = 918 - 4 if 'FiveK' in path else 2268 - 4
            super().__init__(glove=_DS(synth.uniform((rows, 300), 99, -0.5, 0.5).numpy()))
    h5py.File = File
    sys.modules['h5py'] = h5py

    kornia = types.ModuleType('kornia')
    kornia.rgb_to_hsv = hsv_spec.rgb_to_hsv
    kornia.hsv_to_rgb = hsv_spec.hsv_to_rgb
    sys.modules['kornia'] = kornia

    for name in ['pyutils', 'pyutils.edgeconnect', 'pyutils.edgeconnect.src',
                 'pyutils.edgeconnect.src.config', 'pyutils.edgeconnect.src.edge_connect']:
        sys.modules[name] = types.ModuleType(name)

    class Config:
        def __init__(self, path):
            self.path = path

    class EdgeConnect:
        def __init__(self, config):
            pass

        def load(self):
            pass
    sys.modules['pyutils.edgeconnect.src.config'].Config = Config
    sys.modules['pyutils.edgeconnect.src.edge_connect'].EdgeConnect = EdgeConnect


def reference_opt():
    from options.seq2seqGAN_train_options import TrainOptions
    opt = TrainOptions().parser.parse_args([])
    opt.vocab_dir = os.path.join(REF, 'data/language')
    return opt


def enter_workdir():
    """InpaintOperator copies pyutils/edgeconnect/config.yml.example relative to cwd."""
    d = tempfile.mkdtemp(prefix='t2o_gold_')
    os.makedirs(os.path.join(d, 'pyutils/edgeconnect/checkpoints/places2'))
    open(os.path.join(d, 'pyutils/edgeconnect/config.yml.example'), 'w').write('MODE: 2\n')
    os.chdir(d)


# --------------------------------------------------------------------------
def gen_operators(opt):
    from executors.executor import Executor
    torch.manual_seed(0)
    ex = Executor(opt)
    ex.load_state_dict(synth.fill_state_dict(ex.state_dict(), seed=3))
    g = {}
    g['name_list'] = np.array(ex.name_list)
    g['param_num'] = np.array([ex.get_param_num(i) for i in range(8)])
    g['param_bnd'] = np.array([ex.get_param_bnd(i) for i in range(8)], dtype=np.float64)

    B, H, W = 2, 24, 20
    img = synth.images(B, H, W, 11)
    gout = synth.uniform((B, 3, H, W), 12, -1.0, 1.0)
    feats = synth.uniform((B, 512), 13, -1.0, 1.0)
    mask1 = synth.masks(B, 1, H, W, 14)
    mask3 = synth.masks(B, 3, H, W, 15, soft=False)
    for op in [0, 1, 2, 3, 5, 6, 7]:
        for si, setting in enumerate(['mid', 'strong', 'neg']):
            for mname, mask in [('none', None), ('m1', mask1), ('m3', mask3)]:
                if mname == 'm3' and setting != 'mid':
                    continue
                x = img.clone().requires_grad_(True)
                p = synth.op_params(op, B, 100 + 10 * op + si, setting).requires_grad_(True)
                out, par = ex.execute(x, op, mask, specified_param=p)
                out.backward(gout)
                key = 'op%d_%s_%s' % (op, setting, mname)
                g[key + '_out'] = out.detach().numpy()
                g[key + '_gimg'] = x.grad.numpy()
                g[key + '_gparam'] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        # learned-parameter path: features -> fc1 -> lrelu -> fc2 -> regressor
        f = feats.clone().requires_grad_(True)
        out, par = ex.execute(img, op, None, features=f)
        g['op%d_feat_out' % op] = out.detach().numpy()
        g['op%d_feat_param' % op] = par.detach().numpy()
        if out.requires_grad:          # WhiteOperator's output does not depend on its parameter
            out.backward(gout)
            g['op%d_feat_gfeat' % op] = f.grad.numpy()
    out, par = ex.execute(img, -1, None, features=feats)
    g['identity_same_object'] = np.array(out is img)
    g['identity_param'] = par.numpy()

    # BASELINE config 1: single 256x256 image, brightness -> contrast -> saturation
    x = synth.images(1, 256, 256, 21)
    cur = x
    for k, op in enumerate([0, 1, 2]):
        cur, _ = ex.execute(cur, op, None, specified_param=synth.op_params(op, 1, 200 + k, 'mid'))
    g['cfg1_out_crop'] = cur[:, :, 100:132, 60:92].numpy()
    g['cfg1_out_sum'] = np.array(cur.double().sum().item())

    # 6-op chain [0,1,2,3,5,6] + L1 + backward (BASELINE config 2 at a small size)
    B2, H2, W2 = 3, 32, 40
    x = synth.images(B2, H2, W2, 31).requires_grad_(True)
    tgt = synth.images(B2, H2, W2, 32)
    ps = [synth.op_params(op, B2, 300 + k, 'mid').requires_grad_(True)
          for k, op in enumerate([0, 1, 2, 3, 5, 6])]
    cur = x
    for op, p in zip([0, 1, 2, 3, 5, 6], ps):
        cur, _ = ex.execute(cur, op, None, specified_param=p)
    loss = torch.abs(cur - tgt).mean()
    loss.backward()
    g['chain6_out'] = cur.detach().numpy()
    g['chain6_loss'] = np.array(loss.item())
    g['chain6_gimg'] = x.grad.numpy()
    for k, p in enumerate(ps):
        g['chain6_gparam%d' % k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'operators.npz'), **g)
    print('operators.npz: %d arrays' % len(g))


# --------------------------------------------------------------------------
def gen_actor(opt):
    from models.actor import Actor
    import torch.nn.functional as F
    torch.manual_seed(0)
    opt.input_dropout_p = 0.0
    opt.dropout_p = 0.0
    model = Actor(opt)
    keys = list(model.state_dict().keys())
    shapes = [tuple(v.shape) for v in model.state_dict().values()]
    model.load_state_dict(synth.fill_state_dict(model.state_dict(), seed=7))
    g = {'sd_keys': np.array(keys), 'sd_numel': np.array([int(np.prod(s)) for s in shapes])}

    B, H, W, L = 4, 64, 64, opt.encoder_max_len
    x = synth.requests(B, L, 41)
    img = synth.images(B, H, W, 42)
    tgt = synth.images(B, H, W, 43)

    # record per-step logprobs / attention from the decoder
    rec = {'logp': [], 'attn': []}
    orig_step = model.decoder.forward_step

    def spy(*a, **k):
        r = orig_step(*a, **k)
        rec['logp'].append(r[0].detach().clone()), rec['attn'].append(r[2].detach().clone())
        return r
    model.decoder.forward_step = spy

    # pieces: language encoder, image features, one attention call (eval mode)
    model.eval()
    with torch.no_grad():
        enc_out, enc_hid, _ = model.lang_encoder(x)
        g['enc_out'] = enc_out.numpy()
        g['enc_h'] = enc_hid[0].numpy()
        g['enc_c'] = enc_hid[1].numpy()
        feat = F.relu(model.bn1(model.vis_encoder(img)))
        g['img_feat_eval'] = feat.numpy()
        q = synth.uniform((B, 1, 512), 44, -1, 1)
        ao, aw = model.decoder.attention(q, enc_out)
        g['attn_out'] = ao.numpy()
        g['attn_w'] = aw.numpy()

    for mode in ['eval', 'train']:
        model.train(mode == 'train')
        rec['logp'].clear(), rec['attn'].clear()
        model.zero_grad()
        state, pred_imgs, pred_ops, pred_params = model.episode_forward(x, img, None, reinforce_sample=0)
        # END-column select + L1 (train_seq2seqL1.py:78-85)
        picked = []
        for b in range(B):
            idxs = (pred_ops[b] == opt.end_id).nonzero()
            col = idxs[0][0] if len(idxs) > 0 else pred_imgs.shape[1] - 1
            picked.append(pred_imgs[b, col])
        loss = torch.abs(torch.stack(picked) - tgt).mean()
        loss.backward()
        p = 'ep_%s_' % mode
        g[p + 'pred_ops'] = pred_ops.numpy()
        g[p + 'pred_params'] = torch.stack(pred_params, 0).detach().numpy()
        g[p + 'logprobs'] = torch.cat(rec['logp'], 1).numpy()
        g[p + 'attn'] = torch.cat(rec['attn'], 1).numpy()
        g[p + 'imgs_crop'] = pred_imgs[:, :, :, 8:24, 8:24].detach().numpy()
        g[p + 'imgs_mean'] = pred_imgs.detach().double().mean((2, 3, 4)).numpy()
        g[p + 'loss'] = np.array(loss.item())
        g[p + 'grad_norm'] = np.array([(0.0 if q_.grad is None else q_.grad.double().norm().item())
                                       for _, q_ in model.named_parameters()])
        g[p + 'grad_none'] = np.array([q_.grad is None for _, q_ in model.named_parameters()])
        if mode == 'train':
            g['bn1_running_mean_after'] = model.bn1.running_mean.numpy().copy()
            model.load_state_dict(synth.fill_state_dict(model.state_dict(), seed=7))
    g['param_names'] = np.array([n for n, _ in model.named_parameters()])

    # teacher-forced step (train_seq2seqL1.py:51-61)
    y = synth.op_targets(B, 45)
    img_y = synth.uniform((B, 6, 3, H, W), 46)
    gt_params = synth.uniform((B, 5, 24), 47, -1, 1)
    nparam = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
    for b in range(B):
        for k in range(5):
            gt_params[b, k, nparam[int(y[b, k + 1])]:] = 0
    for mode in ['eval', 'train']:
        model.train(mode == 'train')
        rec['logp'].clear(), rec['attn'].clear()
        model.zero_grad()
        pred_imgs, pred_params, pred_logprobs = model.supervised_forward(x, y, img, img_y, gt_params, mask=None)
        step = (y != opt.null_id).sum(1).max().item()
        op_loss = torch.nn.NLLLoss()(pred_logprobs.view(-1, 11), y[:, 1:step].contiguous().view(-1))
        param_loss = torch.nn.MSELoss(reduction='sum')(pred_params, gt_params[:, :step - 2]) / \
            ((gt_params[:, :step - 2] != 0).sum())
        (op_loss + param_loss).backward()
        p = 'sup_%s_' % mode
        g[p + 'pred_params'] = pred_params.detach().numpy()
        g[p + 'logprobs'] = pred_logprobs.detach().numpy()
        g[p + 'imgs_crop'] = pred_imgs[:, :, :, 8:24, 8:24].detach().numpy()
        g[p + 'op_loss'] = np.array(op_loss.item())
        g[p + 'param_loss'] = np.array(param_loss.item())
        g[p + 'grad_norm'] = np.array([(0.0 if q_.grad is None else q_.grad.double().norm().item())
                                       for _, q_ in model.named_parameters()])
        if mode == 'train':
            model.load_state_dict(synth.fill_state_dict(model.state_dict(), seed=7))
    np.savez_compressed(os.path.join(OUT, 'actor.npz'), **g)
    print('actor.npz: %d arrays' % len(g))


def gen_ssim():
    from utils.ssim import ssim as ref_ssim
    a = synth.images(1, 48, 40, 51)
    b = (a + synth.uniform((1, 3, 48, 40), 52, -0.1, 0.1)).clamp(0, 1)
    np.savez_compressed(os.path.join(OUT, 'ssim.npz'), ssim=np.array(ref_ssim(a, b).item()))
    print('ssim.npz')


def synthetic_records():
    """Planner records covering truncation, curve normalisation and the |p| > 5 outlier rule."""
    recs = []
    for seed in range(8):
        n = 2 + seed % 5
        names = ['brightness', 'tone', 'contrast', 'color', 'saturation', 'sharpness'][seed % 3:][:n]
        d = 0.2
        seq = []
        for i, name in enumerate(names):
            drop = [0.05, 0.03, 0.0015, 0.02, 0.0, 0.01][(i + seed) % 6]
            d = d - drop
            k = {'color': 24, 'tone': 8}.get(name, 1)
            vals = (synth.uniform((k,), 500 + 10 * seed + i, -2.0, 2.0) * (4.0 if (seed + i) % 4 == 0 else 1.0)).tolist()
            seq.append([name, vals, d])
        recs.append({'init distance': 0.2, 'operation sequence': [seq]})
    return recs


def gen_data():
    """FiveKAct.get_act's record logic (datasets/FiveKdataset.py:86-116) on synthetic records; the
    reference method is called with its image loader stubbed out (no image files here)."""
    import datasets.FiveKdataset as ds
    import json
    ds.load_train_img = lambda path, size: torch.zeros(3, size, size)
    obj = ds.FiveKAct.__new__(ds.FiveKAct)
    obj.op_max_len, obj.train_img_size, obj.phase = 5, 8, 'train'
    obj.actions = ['brightness', 'contrast', 'saturation', 'color', 'inpaint', 'tone', 'sharpness', 'white']
    obj.act2pn = {'brightness': 1, 'contrast': 1, 'saturation': 1, 'color': 24, 'inpaint': 0, 'tone': 8, 'sharpness': 1, 'white': 0}
    d = tempfile.mkdtemp(prefix='t2o_act_')
    obj.act_dir = d
    g = {}
    for i, rec in enumerate(synthetic_records()):
        os.makedirs(os.path.join(d, 'train%d' % i))
        json.dump(rec, open(os.path.join(d, 'train%d' % i, '%05d.json' % i), 'w'))
        ops, params, imgs = obj.get_act(i)
        g['ops%d' % i], g['params%d' % i] = np.asarray(ops), np.asarray(params)
        g['trunc%d' % i] = np.array(ds.analyze_traj([rec['init distance']] + [v[2] for v in rec['operation sequence'][0]]))
    np.savez_compressed(os.path.join(OUT, 'data.npz'), **g)
    print('data.npz: %d arrays' % len(g))


if __name__ == '__main__':
    assert os.path.isdir(REF), 'run in the build container (needs /root/reference)'
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    sys.path.insert(0, REF)
    enter_workdir()
    torch.set_num_threads(4)
    opt = reference_opt()
    if len(sys.argv) > 1 and sys.argv[1] == 'data':
        gen_data()
        sys.exit(0)
    gen_operators(opt)
    gen_ssim()
    gen_actor(opt)
    gen_data()
