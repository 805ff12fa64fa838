"""Actor with the reference's surface (models/actor.py:36-364): ResNet image encoder + request
encoder + one-step attention decoder + Executor.

Kept: constructor `Actor(opt)`, attributes `vis_encoder / lang_encoder / decoder / executor / bn1`
(199-tensor state_dict, same keys), `supervised_forward`, `episode_forward`, `forward`,
`get_entropy_penalty`, `divide_op_group`, return structures.

Redesigned for the GPU (results unchanged):
  * the per-step "unique ops -> per-group index_select -> execute -> cat -> index_select back"
    loop (actor.py:100-114, :157-172, :244-259) is ONE Executor.execute_per_sample launch with a
    per-sample operator id; no image gather/scatter, no `.item()` host sync per group;
  * the per-sample python loop that clears op_mask (actor.py:235-236) is one scatter_;
  * the attention core is a HIP kernel (attention.py).
"""
import json
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as T
from .action_decoder import Decoder
from .actor_resnet import ResNet
from .executor import Executor, PARAM_PAD
from .lang_encoder import RNNEncoder

_FUSED_FEATURE = True         # relu(bn1(fc(pooled))) as one launch (decoder_step.image_feature); module switch for the tests
_OVERLAP_LANG = True          # request encoder on a side stream (Actor._encode_request); module switch for A/B runs
_SIDE_STREAMS = {}

# operators the FiveK path may choose: END + brightness/contrast/saturation/color/tone/sharpness
# (inpaint_obj = 7 and color_bg = 10 are local edits, blocked: actor.py:211)
OP_MASK = [0., 0., 1., 1., 1., 1., 1., 0., 1., 1., 0.]
N_SPEC_TOKEN = 4


def _vocab_sizes(opt):
    """Vocabulary sizes from the reference's JSON files when present (utils/text_utils.py:28-38),
    else from opt.input_vocab_size / opt.output_vocab_size (FiveK: 918 / 11)."""
    vdir = getattr(opt, 'vocab_dir', None)
    try:
        with open(os.path.join(vdir, '%s_vocabs_sess_%s.json' % (opt.dataset, opt.session))) as f:
            n_in = len(json.load(f))
        with open(os.path.join(vdir, '%s_operator_vocabs_sess_%s.json' % (opt.dataset, opt.session))) as f:
            n_out = len(json.load(f))
        return n_in, n_out
    except (OSError, TypeError, AttributeError):
        return opt.input_vocab_size, opt.output_vocab_size


def sample_categorical(probs):
    """One draw per row from unnormalised non-negative weights (B,K) -> (B,1) int64: what
    Categorical(probs).sample() returns (actor.py:229), by inverse CDF.  torch.multinomial (and
    Categorical's argument validation) check their input with a host synchronisation per call; this does not."""
    cdf = probs.cumsum(1)
    u = torch.rand(probs.shape[0], 1, device=probs.device, dtype=probs.dtype) * cdf[:, -1:]
    idx = (cdf <= u).sum(1, keepdim=True)                    # first index whose cumulative weight exceeds u
    return torch.where(idx >= probs.shape[1], probs.argmax(1, keepdim=True), idx)


class Actor(nn.Module):
    def __init__(self, opt, word2vec=None):
        super().__init__()
        self.opt = opt
        n_in, n_out = _vocab_sizes(opt)
        self.vis_encoder = ResNet(3, 18, 512)
        self.lang_encoder = RNNEncoder(n_in, opt.word_vec_dim, opt.hidden_size, N_SPEC_TOKEN,
                                       bidirectional=bool(opt.bidirectional), input_dropout_p=opt.input_dropout_p,
                                       dropout_p=opt.dropout_p, n_layers=opt.n_layers,
                                       variable_lengths=opt.variable_lengths, word2vec=word2vec,
                                       fix_embedding=opt.fix_input_embedding, pad_id=opt.null_id)
        self.decoder = Decoder(n_out, opt.decoder_max_len, opt.word_vec_dim, opt.hidden_size, opt.n_layers,
                               bidirectional=bool(opt.bidirectional), use_attention=opt.use_attention)
        self.variable_lengths = opt.variable_lengths
        self.null_id, self.start_id, self.end_id = opt.null_id, opt.start_id, opt.end_id
        self.executor = Executor(opt)
        self.bn1 = nn.BatchNorm1d(512)
        # OP_MASK on the device (not in the state_dict): building it per call is a blocking host-to-device copy
        self.register_buffer('_op_mask_row', torch.tensor(OP_MASK, dtype=torch.float).view(1, -1), persistent=False)
        # The image encoder is channels-last from the start (round 6): that is the layout this library's convolution kernels
        # take, so a model that is simply built and moved to the GPU trains on them -- in the default NCHW layout every
        # convolution was a framework (MIOpen) call, whose algorithm choice differs from machine to machine (the episode step's
        # gradient norms against the reference moved 9e-5 .. 1.3e-2 with the box: tests/test_gpu_actor.py).  state_dict shapes
        # and values are unaffected (a memory format, not a shape); use_channels_last(False) restores NCHW.
        self.use_channels_last(True)

    # ------------------------------------------------------------------ helpers
    def use_channels_last(self, on=True):
        """The image encoder's parameters and activations channels-last end to end (the default since round 6): the layout
        of this library's convolution and batch-norm kernels (encoder.py).  on=False: NCHW -- the convolutions are then
        framework calls (an A/B reference; not a path the product needs).  state_dict shapes are unchanged."""
        self._nhwc = bool(on)
        self.vis_encoder.to(memory_format=torch.channels_last if on else torch.contiguous_format)
        return self

    def image_features(self, img, call=None):
        """relu(bn1(vis_encoder(img))) (actor.py:142-143, :215-216).  `call` = index of this encoder call
        inside the step (0, 1, ...): selects the captured hipGraph of that call when graph_image_encoder()
        has been run for this image shape."""
        graphed = self.__dict__.get('_graphed_encoders')
        if graphed is not None and self.training and graphed.usable(img, call):
            return F.relu(self.bn1(graphed(img, call)))
        if img.is_cuda and img.dtype == torch.float32 and _FUSED_FEATURE:
            from . import decoder_step as DS
            pooled = self.vis_encoder.pooled_features(img)
            if DS.feature_supported(pooled, self.vis_encoder.fc, self.bn1):
                return DS.image_feature(pooled, self.vis_encoder.fc, self.bn1, self.__dict__.get('_tape'))     # fc + bn1 + ReLU: one launch
            return F.relu(self.bn1(T.linear_acc(pooled, self.vis_encoder.fc.weight, self.vis_encoder.fc.bias)))
        return F.relu(self.bn1(self.vis_encoder(img)))

    def graph_image_encoder(self, sample_img, calls):
        """Capture the image encoder's forward and backward as hipGraphs, one pair per encoder call of a step
        (t2onet_amd/graphs.py): ~350 kernel launches per encoder forward+backward become 2 graph launches and the
        parameter gradients are accumulated into the existing .grad tensors inside the backward graph.  Training
        mode, fixed (B,3,H,W); anything else runs eagerly.  Needs persistent .grad tensors (Trainer's flat
        buffer).  Not a submodule: state_dict and parameters() are unchanged."""
        from .graphs import GraphedEncoder
        self.vis_encoder.train()
        self.__dict__['_graphed_encoders'] = GraphedEncoder(self.vis_encoder, sample_img, calls)
        return self

    def get_gt_mask(self, img, mask_dict, op):
        """Per-sample local-edit masks (actor.py:78-98): mask_dict[i] maps an operator id (as a
        string) to a list whose first entry is that sample's mask array; samples without an entry
        for the chosen operator get an all-ones mask (global edit).  op: (B,1) host array.
        Returns (B,3,H,W) on the image's device."""
        masks = []
        for i in range(len(mask_dict)):
            entry = mask_dict[i].get(str(op[i][0]))
            mask = None
            if entry is not None:
                try:
                    mask = torch.as_tensor(entry[0], dtype=img.dtype).to(img.device).expand_as(img[i])
                except (RuntimeError, TypeError, IndexError):
                    mask = None
            masks.append(torch.ones_like(img[i]) if mask is None else mask)
        return torch.stack(masks, dim=0)

    def divide_op_group(self, ops):
        """Kept for API compatibility (actor.py:100-114); the forward passes do not use it."""
        unqs = torch.unique(ops)
        group_inds = [torch.nonzero(ops == u).squeeze(1) for u in unqs]
        return unqs, group_inds, torch.argsort(torch.cat(group_inds))

    def _encode_request(self, x, lengths, img_x, want_feat=True, longest=None):
        """Request encoder + decoder initial state, and the image features of call 0.

        On a GPU in training mode the request encoder (a 17-step BiLSTM: ~1 ms forward and ~2 ms backward of
        small, latency-bound library kernels) runs on a SIDE stream while the first image-encoder pass -- which
        does not depend on it -- runs on the caller's stream; autograd runs each node's backward on its forward's
        stream, so the recurrent backward likewise overlaps the last image-encoder backward."""
        if not (img_x.is_cuda and self.training and torch.is_grad_enabled() and _OVERLAP_LANG):
            enc_out, enc_hidden, _ = self.lang_encoder(x, lengths, longest)
            return enc_out, self.decoder._init_state(enc_hidden), (self.image_features(img_x, 0) if want_feat else None)
        dev = img_x.device
        side = _SIDE_STREAMS.get(dev)                          # one per device for the process (module state, not the
        if side is None:                                       # model's: copy.deepcopy(model) must not meet a stream)
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)                                 # x (and the parameters' last update) come from the caller's stream
        with torch.cuda.stream(side):
            enc_out, enc_hidden, _ = self.lang_encoder(x, lengths, longest)
            hidden = self.decoder._init_state(enc_hidden)
        feat = self.image_features(img_x, 0) if want_feat else None    # enqueued before the caller's stream waits for the side stream
        main.wait_stream(side)
        for t in (enc_out,) + tuple(hidden):
            t.record_stream(main)                              # allocated on the side stream, consumed on the caller's
        return enc_out, hidden, feat

    def _execute(self, img, ops_vocab, context, mask=None, exec_op=None):
        """ops_vocab (B,) operator-vocabulary ids; executor index = id - 3, negative -> identity (exec_op: that index
        when the caller has it already, int32)."""
        return self.executor.execute_per_sample(img, ops_vocab.view(-1) - 3 if exec_op is None else exec_op, mask, features=context)

    # ------------------------------------------------------------------ teacher forcing
    def supervised_forward(self, x, y, img_x, img_y, gt_params, mask, lengths=None):
        """actor.py:116-181.  Returns (pred_imgs (B,step-2,3,H,W), pred_params (B,step-2,24),
        pred_logprobs (B,step-1,n_cls)).

        mask: None (every caller of the reference) or the documented (B, step-2, 1|3, H, W) local-edit masks,
        one per executed step: step i blends with mask[:, i-1].  The reference unpacks a 5-D mask the same way
        (actor.py:136-137) but then hands the WHOLE 5-D tensor to every operator, which only broadcasts for one
        sample and one masked step; in that case both give the same images (golden `sup_mask_*`).  Anything
        that is not 5-D fails the reference's unpack and is rejected here too."""
        if mask is not None:
            if mask.dim() != 5 or mask.shape[0] != x.shape[0] or mask.shape[2] not in (1, 3):
                raise ValueError('mask must be (bs, n_steps, 1|3, h, w), got %s' % (tuple(mask.shape),))
            mask = mask.to(img_x.device)
        step = int((y != self.null_id).sum(1).max())
        enc_out, hidden, feat0 = self._encode_request(x, lengths, img_x, step > 1)
        if mask is not None and mask.shape[1] < step - 2:
            raise ValueError('mask covers %d steps, the operator sequence has %d' % (mask.shape[1], step - 2))
        pred_params, pred_imgs, logprobs = [], [], []
        ops = y[:, 0].unsqueeze(-1)
        for i in range(1, step):
            feat = feat0 if i == 1 else self.image_features(img_x, i - 1)
            logp, hidden, _, context = self.decoder.forward_step(ops, hidden, enc_out, feat)
            logprobs.append(logp)
            ops = y[:, i].unsqueeze(-1)
            if i == step - 1:
                break
            out, par = self._execute(img_x, ops, context, None if mask is None else mask[:, i - 1])
            pred_imgs.append(out)
            pred_params.append(par)
            img_x = img_y[:, i - 1]                        # next input = planned ground-truth intermediate
        return torch.stack(pred_imgs, 1), torch.stack(pred_params, 1), torch.cat(logprobs, 1)

    # ------------------------------------------------------------------ free running
    def episode_forward(self, x, img_x, mask_dict, reinforce_sample=1, lengths=None, longest=None, stack=True):
        """actor.py:184-284.  Returns (state, pred_imgs (B,T,3,H,W), pred_ops (B,T), pred_params list of T (B,24)).
        lengths / longest: see RNNEncoder.forward (host-side lengths save a synchronisation; with `longest` given no
        host value depends on device data and the whole call can be captured in a hipGraph); stack: episode_decode."""
        enc_out, hidden, feat0 = self._encode_request(x, lengths, img_x, self.opt.decoder_max_len > 0, longest)
        return self.episode_decode(x, img_x, enc_out, hidden, mask_dict, reinforce_sample, feat0, stack)

    def episode_decode(self, x, img_x, enc_out, hidden, mask_dict=None, reinforce_sample=1, feat0=None, stack=True):
        """The free-running decode of episode_forward from an encoded request (enc_out (B,L,d), hidden = decoder initial
        state): everything after the request encoder (actor.py:213-284).  No host synchronisation when mask_dict is None,
        so a whole train step from here on can be captured in one hipGraph (graphs.GraphedEpisodeStep).
        stack=False: pred_imgs is returned as the list of T images (no (B,T,3,H,W) copy; `state` is then None)."""
        B = img_x.shape[0]
        dev = img_x.device
        hiddens = [tuple(h.detach() for h in hidden)] if stack else None
        if isinstance(hidden, tuple) and torch.is_tensor(hidden[0]) and hidden[0].dim() == 3 and isinstance(self.decoder.rnn, nn.LSTM):
            hidden = (list(hidden[0].unbind(0)), list(hidden[1].unbind(0)))      # per-layer tensors through the steps (Decoder._rnn_step)
        op_mask = self._op_mask_row.repeat(B, 1)                # device-resident: no host-to-device copy (a sync)
        pred_op = torch.full((B, 1), self.start_id, dtype=torch.long, device=dev)
        pred_ops, pred_params, pred_imgs, pred_masks = [], [], [], []
        fused_choice = img_x.is_cuda and img_x.dtype == torch.float32 and op_mask.shape[1] <= 32
        # the Categorical draws' uniform numbers for ALL steps in one launch
        draws = torch.rand(self.opt.decoder_max_len, B, device=dev, dtype=torch.float32) if (fused_choice and reinforce_sample) else None
        for call in range(self.opt.decoder_max_len):
            feat = feat0 if (call == 0 and feat0 is not None) else self.image_features(img_x, call)
            logp, hidden, _, context = self.decoder.forward_step(pred_op, hidden, enc_out, feat)
            if stack:
                hiddens.append(tuple(torch.stack([t.detach() for t in h], 0) if isinstance(h, list) else h.detach() for h in hidden))
            exec_op = None
            if fused_choice:
                # exp, exploration floor, op-mask, renormalisation, draw / arg-max and the op-mask update: ONE launch
                pred_op, exec_op = T.choose_op(logp, op_mask, self.opt.explore_prob, bool(reinforce_sample),
                                               None if draws is None else draws[call])
            else:
                probs = torch.exp(logp).squeeze(1)
                probs = probs * (1 - self.opt.explore_prob) + self.opt.explore_prob
                probs = probs * op_mask
                probs = probs / (probs.sum(1, keepdim=True) + 1e-30)
                if reinforce_sample:
                    pred_op = sample_categorical(probs)
                else:
                    pred_op = probs.topk(1)[1].view(B, -1)
                op_mask.scatter_(1, pred_op, 0.0)          # an operator is used at most once
            pred_mask = None
            if mask_dict is not None:                      # local edits (GIER): one host sync per step, as the reference
                pred_mask = self.get_gt_mask(img_x, mask_dict, pred_op.detach().cpu().numpy())
                pred_masks.append(pred_mask)
            img_x, par = self._execute(img_x, pred_op, context, pred_mask, exec_op)
            pred_imgs.append(img_x)
            pred_params.append(par)
            pred_ops.append(pred_op.squeeze(-1))
        if not pred_ops:
            return {}, img_x.unsqueeze(1), [], pred_params
        pred_ops = torch.stack(pred_ops, 1)
        if not stack:
            return None, pred_imgs, pred_ops, pred_params
        pred_imgs = torch.stack(pred_imgs, 1)
        state = {'reqs': x, 'imgs': pred_imgs.detach(), 'ops': pred_ops, 'param': pred_params,
                 'hidden': hiddens, 'masks': torch.stack(pred_masks, 1) if pred_masks else None}
        return state, pred_imgs, pred_ops, pred_params

    # ------------------------------------------------------------------ single RL step (no caller in the reference)
    def forward(self, x, img_x, hidden, op, mask_dict=None, lengths=None):
        """actor.py:286-354."""
        op = op.view(-1, 1)
        B = img_x.shape[0]
        with torch.no_grad():
            enc_out, _, _ = self.lang_encoder(x, lengths)
        op_mask = self._op_mask_row.repeat(B, 1)
        logp, dec_hidden, _, context = self.decoder.forward_step(op, hidden, enc_out, self.image_features(img_x))
        entropy_penalty = self.get_entropy_penalty(logp)
        probs = torch.exp(logp).squeeze(1) * (1 - self.opt.explore_prob) + self.opt.explore_prob
        probs = probs * op_mask
        probs = probs / (probs.sum(1, keepdim=True) + 1e-30)
        pred_op = sample_categorical(probs)
        pred_mask = self.get_gt_mask(img_x, mask_dict, pred_op.detach().cpu().numpy()) if mask_dict is not None else None
        pred_img, _ = self._execute(img_x, pred_op, context, pred_mask)
        _, _, _, next_context = self.decoder.forward_step(pred_op, dec_hidden, enc_out, self.image_features(pred_img))
        return pred_img, logp, entropy_penalty, context, next_context

    def get_entropy_penalty(self, logprobs):
        probs = torch.exp(logprobs)
        entropy = -(probs * logprobs).sum(1, keepdim=True)
        return math.log(logprobs.shape[-1]) - entropy
