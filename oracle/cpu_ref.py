"""cpu_ref: eager-PyTorch CPU restatement of the T2ONet hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to /root/reference).  The code is
written functionally (state in plain dicts of tensors keyed like the
reference's state_dict) so that the product package, which mirrors the
reference's class surface, shares no code with it.

Operator index order (executors/executor.py:30):
    0 brightness  1 contrast  2 saturation  3 color curve ("hue")
    4 inpaint (unsupported here: needs the EdgeConnect submodule, empty in the
      checkout)  5 tone curve  6 sharpness  7 white ("color_bg")
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import hsv_spec

OP_NAMES = ['brightness', 'contrast', 'saturation', 'hue', 'inpaint_obj',
            'tone', 'sharpness', 'color_bg']                # operators.py short_name fields
OP_ATTRS = ['brightness_op', 'contrast_op', 'saturation_op', 'color_op',
            'inpaint_op', 'tone_op', 'sharpness_op', 'white_op']   # executor.py:22-30
OP_NPARAM = [1, 1, 1, 24, 1, 8, 1, 1]
PARAM_PAD = 24                                              # executor.py:46, actor.py:166


def default_opt(**over):
    """Hot-path flags and defaults: options/seq2seqGAN_base_options.py:55-90,
    options/seq2seqGAN_train_options.py:35-58."""
    o = dict(hidden_size=256, word_vec_dim=300, n_layers=2, bidirectional=1,
             use_attention=1, decoder_max_len=5, encoder_max_len=17,
             operator_fc_dim=512, discrete_param=0, discrete_step=10, curve_steps=8,
             brightness_range=2, sharpness_range=1.5, exposure_range=3.5,
             saturation_range=(-0.2, 0.8), tone_curve_range=(0.5, 2),
             color_curve_range=(0.90, 1.10), input_dropout_p=0.2, dropout_p=0.2,
             variable_lengths=1, fix_input_embedding=1, start_id=1, end_id=2,
             null_id=0, explore_prob=0.05, learning_rate=1e-3, param_noise_factor=0.6,
             dataset='FiveK', session=1, vocab_dir='data/language', manual_seed=10,
             input_vocab_size=918, output_vocab_size=11)
    o.update(over)
    return SimpleNamespace(**o)


# ---------------------------------------------------------------------------
# utils/operator_utils.py
# ---------------------------------------------------------------------------
def lerp(a, b, l):                                  # operator_utils.py:5-6
    return (1 - l) * a + l * b


def rgb2lum(image):                                 # operator_utils.py:9-11
    lum = 0.27 * image[:, 0, :, :] + 0.67 * image[:, 1, :, :] + 0.06 * image[:, 2, :, :]
    return lum[:, None, :, :]


def tanh_range(l, r, initial=None):                 # operator_utils.py:21-34
    def act(x):
        bias = 0
        if initial is not None:
            z = 2 * (initial - l) / (r - l) - 1
            bias = 0.5 * math.log((1 + z) / (1 - z))
        return (torch.tanh(x + bias) * 0.5 + 0.5) * (r - l) + l
    return act


# ---------------------------------------------------------------------------
# models/operators.py : parameter regressors, ranges, process()
# ---------------------------------------------------------------------------
def param_range(op_ind, opt):
    """(ub, lb, initial) per get_param_range: operators.py:288,250,484,617,
    :677,586,363,519."""
    if op_ind == 0:
        return opt.brightness_range, -opt.brightness_range, 0
    if op_ind == 1:
        return 1, -1, 0
    if op_ind == 2:
        return opt.saturation_range[1], opt.saturation_range[0], 0
    if op_ind == 3:
        ub, lb = opt.color_curve_range[1], opt.color_curve_range[0]
        return ub, lb, (ub + lb) / 2
    if op_ind == 4:
        return 0, 0, 0
    if op_ind == 5:
        ub, lb = opt.tone_curve_range[1], opt.tone_curve_range[0]
        return ub, lb, (ub + lb) / 2
    if op_ind == 6:
        return opt.sharpness_range, 0, opt.sharpness_range / 2
    if op_ind == 7:
        return 1, 0, 0.5
    raise IndexError(op_ind)


def regress(op_ind, f, opt):
    """op_param_regressor of each registered operator."""
    if op_ind == 0:                                 # operators.py:266-269
        return tanh_range(-opt.brightness_range, opt.brightness_range, initial=0)(f)
    if op_ind == 1:                                 # :231-232
        return torch.tanh(f)
    if op_ind == 2:                                 # :461-465
        return torch.tanh(F.relu(f)) * opt.saturation_range[1] + \
            torch.tanh(F.relu(-f)) * opt.saturation_range[0]
    if op_ind in (3, 5):                            # :602-603, :566-567 (identity)
        return f
    if op_ind == 6:                                 # :340-343
        return torch.sigmoid(f) * opt.sharpness_range
    if op_ind == 7:                                 # :501-502
        return torch.sigmoid(f)
    raise NotImplementedError('operator %d' % op_ind)


SHARP_KERNEL = [[0., -1., 0.], [-1., 4., -1.], [0., -1., 0.]]       # operators.py:338


def process(op_ind, img, param, opt):
    """process() of each registered operator (the per-pixel math)."""
    p4 = param.unsqueeze(-1).unsqueeze(-1)
    if op_ind == 0:                                 # BrightnessOperator.process :277-283
        hsv = hsv_spec.rgb_to_hsv(img)
        h, s, v = torch.chunk(hsv, chunks=3, dim=1)
        v_out = (v * (1 + p4)).clamp(0, 1)
        return hsv_spec.hsv_to_rgb(torch.cat([h, s, v_out], dim=1))
    if op_ind == 1:                                 # ContrastOperator.process :240-245
        lum = torch.min(torch.max(rgb2lum(img), torch.tensor(0.0)), torch.tensor(1.0))
        clum = -torch.cos(math.pi * lum) * 0.5 + 0.5
        cimg = img / (lum + 1e-6) * clum
        return lerp(img, cimg, p4)
    if op_ind == 2:                                 # SaturationOperator.process :473-479
        hsv = hsv_spec.rgb_to_hsv(img)
        h, s, v = torch.chunk(hsv, chunks=3, dim=1)
        s_out = (s * (1 + p4)).clamp(0, 1)
        return hsv_spec.hsv_to_rgb(torch.cat([h, s_out, v], dim=1))
    if op_ind == 3:                                 # ColorOperator.process :607-616
        n = opt.curve_steps
        curve = param.view(-1, 3, n, 1, 1)
        csum = curve.sum(2) + 1e-10
        total = torch.zeros_like(img)
        for i in range(n):
            total = total + torch.clamp(img - 1.0 * i / n, 0, 1.0 / n) * curve[:, :, i, :, :]
        total = total * (n / csum)
        return total
    if op_ind == 5:                                 # ToneOperator.process :571-585
        n = opt.curve_steps
        curve = param.view(-1, 1, n, 1, 1)
        csum = curve.sum(2) + 1e-10
        total = torch.zeros_like(img)
        for i in range(n):
            total = total + torch.clamp(img - 1.0 * i / n, 0, 1.0 / n) * curve[:, :, i, :, :]
        return total * n / csum
    if op_ind == 6:                                 # SharpnessOperator.process :351-358
        k = torch.tensor([[SHARP_KERNEL]], dtype=img.dtype)
        r, g, b = img.split([1, 1, 1], 1)
        delta = torch.cat((F.conv2d(r, k, padding=1), F.conv2d(g, k, padding=1),
                           F.conv2d(b, k, padding=1)), 1)
        return img + p4 * delta
    if op_ind == 7:                                 # WhiteOperator.process :509-511
        return torch.ones_like(img)
    raise NotImplementedError('operator %d' % op_ind)


def operator_apply(op_ind, img, param, mask, opt):
    """Operator.execute after the parameter is known: operators.py:123-131."""
    if mask is None:
        mask = torch.ones_like(img)
    out = process(op_ind, img, param, opt)
    out = out * mask + img * (1 - mask)
    return torch.clamp(out, 0, 1)


def param_head(sd, op_ind, features, opt, prefix='executor.'):
    """extract_parameters: fc1 -> LeakyReLU(0.01) -> fc2 -> regressor
    (operators.py:73-88, setup :43-55)."""
    k = prefix + OP_ATTRS[op_ind]
    x = F.linear(features, sd[k + '.fc1.weight'], sd[k + '.fc1.bias'])
    x = F.leaky_relu(x, 0.01)
    x = F.linear(x, sd[k + '.fc2.weight'], sd[k + '.fc2.bias'])
    return regress(op_ind, x, opt)


def executor_execute(sd, img, op_ind, mask, opt, features=None, specified_param=None,
                     prefix='executor.'):
    """Executor.execute: executors/executor.py:33-55 -> (out, param)."""
    if op_ind < 0:
        return img, torch.zeros(img.shape[0], PARAM_PAD, dtype=torch.float)
    assert (features is None) ^ (specified_param is None)      # operators.py:113
    if features is not None:
        param = param_head(sd, op_ind, features, opt, prefix)
    else:
        param = specified_param
    return operator_apply(op_ind, img, param, mask, opt), param


def run_sequence(img, ops, params, opt, masks=None):
    """Apply a known (op, param) list through Executor.execute with
    specified_param, the way the planner drives it (utils/beam_search.py:79)."""
    outs = []
    x = img
    for k, (op, p) in enumerate(zip(ops, params)):
        m = None if masks is None else masks[k]
        x, _ = executor_execute(None, x, op, m, opt, specified_param=p)
        outs.append(x)
    return x, outs


def l1_loss(pred, target):
    """experiments/t2onet/train_seq2seqL1.py:85."""
    return torch.abs(pred - target).mean()


def select_end_images(pred_imgs, pred_ops, end_id):
    """Column of the first END token, else the last: train_seq2seqL1.py:78-84."""
    bs, max_len = pred_ops.shape
    picked = []
    for b in range(bs):
        idxs = (pred_ops[b] == end_id).nonzero()
        col = idxs[0][0] if len(idxs) > 0 else max_len - 1
        picked.append(pred_imgs[b, col])
    return torch.stack(picked)


# ---------------------------------------------------------------------------
# utils/ssim/__init__.py (evaluation metric)
# ---------------------------------------------------------------------------
def _gauss_window(size, sigma):                     # ssim/__init__.py:7-17
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2))
                      for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)


def ssim(img1, img2, window_size=11, size_average=True):
    """_ssim: utils/ssim/__init__.py:20-40 (11x11, sigma 1.5, C1=0.01^2, C2=0.03^2)."""
    ch = img1.shape[1]
    w = _gauss_window(window_size, 1.5).expand(ch, 1, window_size, window_size).contiguous().to(img1.dtype)     # (fp64 runs: the
    pad = window_size // 2                                                                                     # reference's fp32 window values, exactly)
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


# ---------------------------------------------------------------------------
# models/attention.py, models/action_decoder.py
# ---------------------------------------------------------------------------
def attention(sd, output, context, prefix='decoder.attention.'):
    """Attention.forward: attention.py:17-44.  output (B,1,d), context (B,L,d).
    The softmax runs over every encoder position including zero-padded rows
    (no padding mask) -- reproduced deliberately."""
    B, _, d = output.shape
    L = context.size(1)
    attn = torch.bmm(output, context.transpose(1, 2))
    attn = torch.softmax(attn.view(-1, L), dim=1).view(B, -1, L)
    mix = torch.bmm(attn, context)
    comb = torch.cat((mix, output), dim=2)
    out = torch.tanh(F.linear(comb.view(-1, 2 * d), sd[prefix + 'linear_out.weight'],
                              sd[prefix + 'linear_out.bias']).view(B, -1, d))
    return out, attn


def _lstm_weights(sd, prefix, n_layers, bidirectional):
    ws = []
    for l in range(n_layers):
        for suf in ([''] + (['_reverse'] if bidirectional else [])):
            ws += [sd['%sweight_ih_l%d%s' % (prefix, l, suf)], sd['%sweight_hh_l%d%s' % (prefix, l, suf)],
                   sd['%sbias_ih_l%d%s' % (prefix, l, suf)], sd['%sbias_hh_l%d%s' % (prefix, l, suf)]]
    return ws


def decoder_step(sd, input_var, hidden, enc_out, img_feat, opt, prefix='decoder.'):
    """Decoder.forward_step: action_decoder.py:38-64 (dropouts are 0 there)."""
    B = input_var.size(0)
    d = 2 * opt.hidden_size if opt.bidirectional else opt.hidden_size
    vis = F.relu(F.linear(img_feat, sd[prefix + 'vis_linear.weight'], sd[prefix + 'vis_linear.bias']))
    emb = F.embedding(input_var, sd[prefix + 'embedding.weight'])
    emb = torch.cat((emb, vis.view(B, 1, -1)), 2)
    ws = _lstm_weights(sd, prefix + 'rnn.', opt.n_layers, False)
    ctx, h, c = torch._VF.lstm(emb, hidden, ws, True, opt.n_layers, 0.0, False, False, True)
    ctx, attn = attention(sd, ctx, enc_out, prefix + 'attention.')
    logits = F.linear(ctx.contiguous().view(-1, d), sd[prefix + 'out_linear.weight'],
                      sd[prefix + 'out_linear.bias'])
    logp = F.log_softmax(logits.view(B, 1, -1), -1)
    return logp, (h, c), attn, ctx.squeeze(1)


def init_decoder_state(enc_hidden):
    """Decoder._init_state/_cat_directions: action_decoder.py:66-78."""
    return tuple(torch.cat([h[0:h.size(0):2], h[1:h.size(0):2]], 2) for h in enc_hidden)


# ---------------------------------------------------------------------------
# models/lang_encoder.py
# ---------------------------------------------------------------------------
def lang_encoder(sd, tokens, opt, training=False, prefix='lang_encoder.'):
    """RNNEncoder.forward: lang_encoder.py:70-113 with Embedding.forward :22-31.
    Dropout only when training=True (the parity fixtures use eval mode).
    lengths go to pack_padded_sequence on the CPU (the reference passes a device
    tensor, which modern torch rejects on GPU: SURVEY.md section 7)."""
    lengths = (tokens != opt.null_id).sum(1)
    sorted_len, sort_ix = lengths.sort(descending=True)
    recover = sort_ix.argsort()
    tokens = tokens[:, :int(lengths.max())][sort_ix]
    W = sd[prefix + 'embedding.weight']
    if opt.fix_input_embedding:
        emb = F.embedding(tokens, W * sd[prefix + 'embedding.mask_spec']) + \
            F.embedding(tokens, W.detach() * sd[prefix + 'embedding.mask_word'])
    else:
        emb = F.embedding(tokens, W)
    emb = F.dropout(emb, opt.input_dropout_p, training)
    packed = torch.nn.utils.rnn.pack_padded_sequence(emb, sorted_len.cpu(), batch_first=True)
    ws = _lstm_weights(sd, prefix + 'rnn.', opt.n_layers, bool(opt.bidirectional))
    B = tokens.size(0)
    nd = 2 if opt.bidirectional else 1
    zeros = torch.zeros(opt.n_layers * nd, B, opt.hidden_size, dtype=emb.dtype, device=emb.device)     # (fp64 runs of the oracle: tools/measure_parity.py)
    out, h, c = torch._VF.lstm(packed.data, packed.batch_sizes, (zeros, zeros), ws, True,
                               opt.n_layers, opt.dropout_p, training, bool(opt.bidirectional))
    out = torch.nn.utils.rnn.PackedSequence(out, packed.batch_sizes)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
    return out[recover], (h[:, recover, :], c[:, recover, :])


# ---------------------------------------------------------------------------
# models/actor_resnet.py
# ---------------------------------------------------------------------------
def _bn(sd, x, k, training):
    return F.batch_norm(x, sd[k + '.running_mean'], sd[k + '.running_var'], sd[k + '.weight'],
                        sd[k + '.bias'], training, 0.1, 1e-5)


def _basic_block(sd, x, k, stride, training):       # actor_resnet.py:21-44
    out = F.relu(_bn(sd, F.conv2d(x, sd[k + '.conv1.weight'], stride=stride, padding=1), k + '.bn1', training))
    out = _bn(sd, F.conv2d(out, sd[k + '.conv2.weight'], padding=1), k + '.bn2', training)
    if (k + '.shortcut.0.weight') in sd:
        sc = _bn(sd, F.conv2d(x, sd[k + '.shortcut.0.weight'], stride=stride), k + '.shortcut.1', training)
    else:
        sc = x
    return F.relu(out + sc)


def resnet18(sd, x, training=False, prefix='vis_encoder.'):
    """ResNet.forward: actor_resnet.py:98-107 (3x3 stride-2 stem, no maxpool,
    every stage stride 2, global mean, fc)."""
    x = F.relu(_bn(sd, F.conv2d(x, sd[prefix + 'conv1.weight'], stride=2, padding=1), prefix + 'bn1', training))
    for li in range(1, 5):
        for bi in range(2):
            x = _basic_block(sd, x, '%slayer%d.%d' % (prefix, li, bi), 2 if bi == 0 else 1, training)
    x = x.mean((2, 3)).view(x.size(0), -1)
    return F.linear(x, sd[prefix + 'fc.weight'], sd[prefix + 'fc.bias'])


def image_features(sd, img, training=False):
    """actor.py:142-143 / :215-216: relu(bn1(vis_encoder(img)))."""
    return F.relu(_bn(sd, resnet18(sd, img, training), 'bn1', training))


# ---------------------------------------------------------------------------
# models/actor.py
# ---------------------------------------------------------------------------
OP_MASK = [0., 0., 1., 1., 1., 1., 1., 0., 1., 1., 0.]      # actor.py:211


def _execute_by_group(sd, img_x, ops, context, opt):
    """divide_op_group + per-group execute + regroup: actor.py:100-114, :244-259.
    Written per sample group but WITHOUT the gather/scatter copies: results are
    written back by index, which is the same permutation."""
    out = torch.empty_like(img_x)
    params = torch.zeros(img_x.size(0), PARAM_PAD)
    outs, pars, inds_all = [], [], []
    for op in torch.unique(ops).tolist():
        inds = torch.nonzero(ops == op).squeeze(1)
        o, p = executor_execute(sd, img_x.index_select(0, inds), op - 3, None, opt,
                                features=context.index_select(0, inds))
        p = torch.cat([p, torch.zeros(len(inds), PARAM_PAD - p.shape[-1])], 1)
        outs.append(o), pars.append(p), inds_all.append(inds)
    inv = torch.argsort(torch.cat(inds_all))
    return torch.cat(outs).index_select(0, inv), torch.cat(pars).index_select(0, inv)


def episode_forward(sd, x, img_x, opt, reinforce_sample=0, training=False, generator=None):
    """Actor.episode_forward: actor.py:184-284 (mask_dict=None path)."""
    B = x.shape[0]
    enc_out, enc_hidden = lang_encoder(sd, x, opt, training)
    hidden = init_decoder_state(enc_hidden)
    op_mask = torch.tensor(OP_MASK).repeat(B, 1)
    pred_op = torch.full((B, 1), opt.start_id, dtype=torch.long)
    pred_ops, pred_params, pred_imgs, logps, attns = [], [], [], [], []
    for _ in range(opt.decoder_max_len):
        feat = image_features(sd, img_x, training)
        logp, hidden, attn, context = decoder_step(sd, pred_op, hidden, enc_out, feat, opt)
        probs = torch.exp(logp).squeeze(1)
        probs = probs * (1 - opt.explore_prob) + opt.explore_prob
        probs = probs * op_mask
        probs = probs / (probs.sum(1, keepdim=True) + 1e-30)
        if reinforce_sample:
            pred_op = torch.multinomial(probs, 1, generator=generator).view(B, -1)
        else:
            pred_op = probs.topk(1)[1].view(B, -1)
        for b in range(B):
            op_mask[b, pred_op[b, 0]] = 0
        img_x, par = _execute_by_group(sd, img_x, pred_op.view(-1), context, opt)
        pred_imgs.append(img_x), pred_params.append(par), pred_ops.append(pred_op.squeeze(-1))
        logps.append(logp), attns.append(attn)
    return dict(pred_ops=torch.stack(pred_ops, 1), pred_imgs=torch.stack(pred_imgs, 1),
                pred_params=pred_params, logprobs=torch.cat(logps, 1), attns=torch.cat(attns, 1))


def supervised_forward(sd, x, y, img_x, img_y, opt, training=False):
    """Actor.supervised_forward: actor.py:116-181 (mask=None)."""
    enc_out, enc_hidden = lang_encoder(sd, x, opt, training)
    hidden = init_decoder_state(enc_hidden)
    step = int((y != opt.null_id).sum(1).max())
    ops = y[:, 0].unsqueeze(-1)
    pred_imgs, pred_params, logps = [], [], []
    for i in range(1, step):
        feat = image_features(sd, img_x, training)
        logp, hidden, _, context = decoder_step(sd, ops, hidden, enc_out, feat, opt)
        logps.append(logp)
        ops = y[:, i].unsqueeze(-1)
        if i == step - 1:
            break
        out, par = _execute_by_group(sd, img_x, ops.view(-1), context, opt)
        pred_imgs.append(out), pred_params.append(par)
        img_x = img_y[:, i - 1]
    return torch.stack(pred_imgs, 1), torch.stack(pred_params, 1), torch.cat(logps, 1)


def supervised_loss(pred_params, pred_logprobs, y, gt_params, opt):
    """train_seq2seqL1.py:52-61: NLL (mean, no ignore_index) + MSE(sum)/nnz."""
    step = int((y != opt.null_id).sum(1).max())
    target = y[:, 1:step].contiguous().view(-1)
    op_loss = F.nll_loss(pred_logprobs.reshape(-1, pred_logprobs.shape[-1]), target)
    gt = gt_params[:, :step - 2]
    param_loss = F.mse_loss(pred_params, gt, reduction='sum') / ((gt != 0).sum())
    return op_loss, param_loss


# ---------------------------------------------------------------------------
# state_dict skeleton (key names, order and shapes of Actor(opt).state_dict();
# 199 tensors) -- lets tests build formula weights without the
# reference or the product package.
# ---------------------------------------------------------------------------
def actor_state_skeleton(opt=None):
    opt = opt or default_opt()
    sd = {}

    def bn(k, c):
        sd[k + '.weight'] = torch.ones(c)
        sd[k + '.bias'] = torch.zeros(c)
        sd[k + '.running_mean'] = torch.zeros(c)
        sd[k + '.running_var'] = torch.ones(c)
        sd[k + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)

    sd['vis_encoder.conv1.weight'] = torch.zeros(64, 3, 3, 3)
    bn('vis_encoder.bn1', 64)
    cin = 64
    for li, c in enumerate([64, 128, 256, 512], 1):
        for bi in range(2):
            k = 'vis_encoder.layer%d.%d' % (li, bi)
            sd[k + '.conv1.weight'] = torch.zeros(c, cin, 3, 3)
            bn(k + '.bn1', c)
            sd[k + '.conv2.weight'] = torch.zeros(c, c, 3, 3)
            bn(k + '.bn2', c)
            if bi == 0:
                sd[k + '.shortcut.0.weight'] = torch.zeros(c, cin, 1, 1)
                bn(k + '.shortcut.1', c)
            cin = c
    sd['vis_encoder.fc.weight'] = torch.zeros(512, 512)
    sd['vis_encoder.fc.bias'] = torch.zeros(512)
    V, E, Hd = opt.input_vocab_size, opt.word_vec_dim, opt.hidden_size
    sd['lang_encoder.embedding.weight'] = torch.zeros(V, E)
    sd['lang_encoder.embedding.mask_spec'] = torch.cat([torch.ones(4, E), torch.zeros(V - 4, E)])
    sd['lang_encoder.embedding.mask_word'] = 1 - sd['lang_encoder.embedding.mask_spec']

    def lstm(prefix, nin, nh, layers, bidir):
        for l in range(layers):
            i = nin if l == 0 else nh * (2 if bidir else 1)
            for suf in [''] + (['_reverse'] if bidir else []):
                sd['%sweight_ih_l%d%s' % (prefix, l, suf)] = torch.zeros(4 * nh, i)
                sd['%sweight_hh_l%d%s' % (prefix, l, suf)] = torch.zeros(4 * nh, nh)
                sd['%sbias_ih_l%d%s' % (prefix, l, suf)] = torch.zeros(4 * nh)
                sd['%sbias_hh_l%d%s' % (prefix, l, suf)] = torch.zeros(4 * nh)
    lstm('lang_encoder.rnn.', E, Hd, opt.n_layers, True)
    D = 2 * Hd
    sd['decoder.embedding.weight'] = torch.zeros(opt.output_vocab_size, E)
    lstm('decoder.rnn.', E + D, D, opt.n_layers, False)
    sd['decoder.out_linear.weight'] = torch.zeros(opt.output_vocab_size, D)
    sd['decoder.out_linear.bias'] = torch.zeros(opt.output_vocab_size)
    sd['decoder.vis_linear.weight'] = torch.zeros(D, D)
    sd['decoder.vis_linear.bias'] = torch.zeros(D)
    sd['decoder.attention.linear_out.weight'] = torch.zeros(D, 2 * D)
    sd['decoder.attention.linear_out.bias'] = torch.zeros(D)
    nout = dict(zip(OP_ATTRS, OP_NPARAM))
    for name in ['brightness_op', 'sharpness_op', 'color_op', 'contrast_op', 'inpaint_op',
                 'white_op', 'saturation_op', 'tone_op']:                 # executor.py:22-29
        sd['executor.%s.fc1.weight' % name] = torch.zeros(opt.operator_fc_dim, D)
        sd['executor.%s.fc1.bias' % name] = torch.zeros(opt.operator_fc_dim)
        sd['executor.%s.fc2.weight' % name] = torch.zeros(nout[name], opt.operator_fc_dim)
        sd['executor.%s.fc2.bias' % name] = torch.zeros(nout[name])
    bn('bn1', 512)
    return sd


NON_PARAM_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked', 'mask_spec', 'mask_word')


def make_leaf_params(sd):
    """Clone a state dict so that every trainable tensor is a grad-requiring leaf."""
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if torch.is_floating_point(v) and not k.endswith(NON_PARAM_SUFFIXES):
            v.requires_grad_(True)
        out[k] = v
    return out
