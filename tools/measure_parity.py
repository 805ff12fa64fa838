"""Measured distances behind the tolerances of tests/test_gpu_actor.py and tests/test_gpu_operators.py (the
tests should be as tight as the implementation is).  Prints, per check, the largest deviation from the committed goldens /
the oracle so that a tolerance can be set at ~3x the measured figure.  python tools/measure_parity.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref, synth                                   # noqa: E402
import t2onet_amd                                                    # noqa: E402
from tests.test_gpu_actor import make_model, B, H, W, L              # noqa: E402

dev = torch.device('cuda:0')
gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'actor.npz'))


def dev_abs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def dev_mixed(a, b, rtol):
    """smallest atol for which |a - b| <= atol + rtol |b| holds everywhere"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) - rtol * np.abs(b)).max())


for mode in ('eval', 'train'):
    model, opt = make_model(dev)
    model.train(mode == 'train')
    p = 'ep_%s_' % mode
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    tgt = synth.images(B, H, W, 43).to(dev)
    state, pred_imgs, pred_ops, pred_params = model.episode_forward(x, img, None, reinforce_sample=0)
    print(p, 'ops equal', bool((pred_ops.cpu().numpy() == gold[p + 'pred_ops']).all()))
    pp = torch.stack(pred_params, 0).detach().cpu().numpy()
    print(p, 'pred_params abs', dev_abs(pp, gold[p + 'pred_params']), 'atol@rtol1e-4', dev_mixed(pp, gold[p + 'pred_params'], 1e-4))
    print(p, 'imgs_crop abs', dev_abs(pred_imgs[:, :, :, 8:24, 8:24].detach().cpu().numpy(), gold[p + 'imgs_crop']))
    print(p, 'imgs_mean abs', dev_abs(pred_imgs.detach().double().mean((2, 3, 4)).cpu().numpy(), gold[p + 'imgs_mean']))
    from t2onet_amd.train import select_end_images
    import t2onet_amd.functional as T
    loss = T.l1_loss(select_end_images(pred_imgs, pred_ops, opt.end_id), tgt)
    print(p, 'loss abs', abs(loss.item() - float(gold[p + 'loss'])))
    if mode == 'train':
        loss.backward()
        names = list(gold['param_names'])
        params = dict(model.named_parameters())
        gn = np.array([0.0 if params[n].grad is None else params[n].grad.double().norm().item() for n in names])
        ref = gold[p + 'grad_norm']
        big = ref > 1e-3 * ref.max()
        print(p, 'grad_norm max rel', float((np.abs(gn[big] - ref[big]) / ref[big]).max()), 'of', int(big.sum()))
        rel = np.abs(gn - ref) / np.maximum(ref, 1e-30)
        order = [i for i in np.argsort(-rel) if big[i]][:6]
        print(p, 'largest gradient-norm deviations:', ', '.join('%s %.3g' % (names[i], rel[i]) for i in order))
        print(p, 'gradient norms off by > 1e-3: %d, > 1e-4: %d, of %d' % (int((rel[big] > 1e-3).sum()), int((rel[big] > 1e-4).sum()), int(big.sum())))
        import hashlib
        flat = torch.cat([params[n].grad.reshape(-1) for n in names if params[n].grad is not None and 'embedding' not in n])
        print(p, 'sha256 of all non-embedding gradients (box independence):', hashlib.sha256(flat.cpu().numpy().tobytes()).hexdigest()[:16])
    if mode == 'train':
        # Which side is right?  The oracle in fp64 on the host (the same network, exact to ~1e-15): its gradient norms against (a)
        # this library's fp32 kernels and (b) the reference's own fp32 run (the golden).  If both fp32 executions sit at comparable
        # distances from the fp64 values, their distance from EACH OTHER is what fp32 does to this 4-image fixture, not an error of
        # either.
        def oracle_norms(dt):
            sd = {k: (v.detach().cpu().to(dt) if torch.is_floating_point(v) else v.detach().cpu()) for k, v in model.state_dict().items()}
            sd = cpu_ref.make_leaf_params(sd)
            r = cpu_ref.episode_forward(sd, x.cpu(), img.cpu().to(dt), opt, reinforce_sample=0, training=True)
            same = bool((r['pred_ops'].numpy() == gold[p + 'pred_ops']).all())
            cpu_ref.l1_loss(cpu_ref.select_end_images(r['pred_imgs'], r['pred_ops'], opt.end_id), tgt.cpu().to(dt)).backward()
            return same, np.array([0.0 if sd[n].grad is None else sd[n].grad.double().norm().item() for n in names])
        same64, g64 = oracle_norms(torch.float64)
        same32, g32 = oracle_norms(torch.float32)

        def dist(v):
            d = np.abs(v - g64) / np.maximum(g64, 1e-30)
            return 'max %.3g median %.3g' % (d[big].max(), np.median(d[big]))
        print(p, 'operators of the fp64 / fp32 oracle equal the golden\'s: %s / %s' % (same64, same32))
        print(p, 'gradient norms against the fp64 oracle -- this library (GPU, fp32): %s; the reference\'s fp32 run (golden): %s; '
                 'the oracle itself in fp32 on this host: %s' % (dist(gn), dist(ref), dist(g32)))
    # supervised
    model, opt = make_model(dev)
    model.train(mode == 'train')
    p = 'sup_%s_' % mode
    y = synth.op_targets(B, 45)
    img_y = synth.uniform((B, 6, 3, H, W), 46).to(dev)
    gt_params = synth.uniform((B, 5, 24), 47, -1, 1)
    nparam = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
    for b in range(B):
        for k in range(5):
            gt_params[b, k, nparam[int(y[b, k + 1])]:] = 0
    y, gt_params = y.to(dev), gt_params.to(dev)
    pred_imgs, pred_params, logp = model.supervised_forward(x, y, img, img_y, gt_params, None)
    print(p, 'pred_params abs', dev_abs(pred_params.detach().cpu().numpy(), gold[p + 'pred_params']))
    print(p, 'logprobs abs', dev_abs(logp.detach().cpu().numpy(), gold[p + 'logprobs']))
    print(p, 'imgs_crop abs', dev_abs(pred_imgs[:, :, :, 8:24, 8:24].detach().cpu().numpy(), gold[p + 'imgs_crop']))
    op_loss, param_loss = cpu_ref.supervised_loss(pred_params, logp, y, gt_params, opt)
    print(p, 'op_loss abs', abs(op_loss.item() - float(gold[p + 'op_loss'])), 'param_loss abs', abs(param_loss.item() - float(gold[p + 'param_loss'])))
    if mode == 'train':
        (op_loss + param_loss).backward()
        names = list(gold['param_names'])
        params = dict(model.named_parameters())
        gn = np.array([0.0 if params[n].grad is None else params[n].grad.double().norm().item() for n in names])
        ref = gold[p + 'grad_norm']
        big = ref > 1e-3 * ref.max()
        print(p, 'grad_norm max rel', float((np.abs(gn[big] - ref[big]) / ref[big]).max()), 'of', int(big.sum()))

# ---- operators: forward against the reference's goldens and the oracle on ragged sizes
ex = t2onet_amd.Executor(t2onet_amd.default_options())
ex.load_state_dict(synth.fill_state_dict(ex.state_dict(), seed=3))
ex = ex.to(dev)
og = np.load(os.path.join(ROOT, 'tests', 'golden', 'operators.npz'))
OPT = cpu_ref.default_opt()
worst = {}
for op in [0, 1, 2, 3, 5, 6, 7]:
    Bq, Hq, Wq = 2, 24, 20
    img = synth.images(Bq, Hq, Wq, 11)
    masks = {'none': None, 'm1': synth.masks(Bq, 1, Hq, Wq, 14), 'm3': synth.masks(Bq, 3, Hq, Wq, 15, soft=False)}
    for si, setting in enumerate(['mid', 'strong', 'neg']):
        for mname, mask in masks.items():
            key = 'op%d_%s_%s' % (op, setting, mname)
            if key + '_out' not in og:
                continue
            prm = synth.op_params(op, Bq, 100 + 10 * op + si, setting).to(dev)
            out, _ = ex.execute(img.to(dev), op, None if mask is None else mask.to(dev), specified_param=prm)
            worst[('golden', op)] = max(worst.get(('golden', op), 0.0), dev_mixed(out.cpu().numpy(), og[key + '_out'], 1e-5))
    for shape in [(2, 23, 19), (1, 40, 150), (3, 17, 68), (2, 128, 128), (1, 397, 600)]:
        Bq, Hq, Wq = shape
        img = synth.images(Bq, Hq, Wq, 61)
        for mask in (None, synth.masks(Bq, 1, Hq, Wq, 63)):
            prm = synth.op_params(op, Bq, 400 + op, 'mid')
            ref = cpu_ref.operator_apply(op, img, prm, mask, OPT)
            out, _ = ex.execute(img.to(dev), op, None if mask is None else mask.to(dev), specified_param=prm.to(dev))
            worst[('ragged', op)] = max(worst.get(('ragged', op), 0.0), dev_mixed(out.cpu().numpy(), ref.numpy(), 1e-5))
for k in sorted(worst):
    print('operator fwd', k, 'needed atol at rtol 1e-5: %.3g' % worst[k])

# ---- the learned-parameter path (Operator.extract_parameters, models/operators.py:73-88): head -> operator
for op in [0, 1, 2, 3, 5, 6, 7]:
    Bq, Hq, Wq = 2, 24, 20
    img = synth.images(Bq, Hq, Wq, 11).to(dev)
    f = synth.uniform((Bq, 512), 13, -1.0, 1.0).to(dev)
    out, par = ex.execute(img, op, None, features=f)
    print('learned op %d: param abs %.3g, needed atol at rtol 1e-5: param %.3g out %.3g' % (
        op, dev_abs(par.detach().cpu().numpy(), og['op%d_feat_param' % op]),
        dev_mixed(par.detach().cpu().numpy(), og['op%d_feat_param' % op], 1e-5),
        dev_mixed(out.detach().cpu().numpy(), og['op%d_feat_out' % op], 1e-5)))
