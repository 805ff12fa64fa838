"""Executor with the reference's surface (executors/executor.py:14-63) on the HIP library.

`execute(img, op_ind, mask, features=None, specified_param=None, has_noise=False) -> (out, param)`
behaves as the reference's: one operator over the (sub)batch, `op_ind < 0` returns the image
itself and zeros(bs,24).  Added for the actor and the planner, which the reference drives
through per-group Python loops:

  execute_per_sample(img, op_ids, mask, features)  one launch for a batch whose samples use
                                                   different operators (actor.py:244-259)
  run_sequence(img, ops, params, target)           a known operator list + L1 on the last
                                                   output (beam_search.py:79 style use)
"""
import torch
import torch.nn as nn

from . import functional as T
from .operators import (BrightnessOperator, ColorOperator, ContrastOperator, InpaintOperator,
                        SaturationOperator, SharpnessOperator, ToneOperator, WhiteOperator)

PARAM_PAD = T.PARAM_PAD


class Executor(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        # registration order fixes the state_dict key order (executor.py:22-29)
        self.brightness_op = BrightnessOperator(opt)
        self.sharpness_op = SharpnessOperator(opt)
        self.color_op = ColorOperator(opt)
        self.contrast_op = ContrastOperator(opt)
        self.inpaint_op = InpaintOperator(opt)
        self.white_op = WhiteOperator(opt)
        self.saturation_op = SaturationOperator(opt)
        self.tone_op = ToneOperator(opt)
        # index order (executor.py:30)
        self.ops = [self.brightness_op, self.contrast_op, self.saturation_op, self.color_op,
                    self.inpaint_op, self.tone_op, self.sharpness_op, self.white_op]
        self.name_list = [op.short_name for op in self.ops]

    def execute(self, img, op_ind, mask, features=None, specified_param=None, has_noise=False):
        if op_ind < 0:                                            # executor.py:44-46
            return img, torch.zeros(img.shape[0], PARAM_PAD, dtype=torch.float).to(img.device)
        Op = self.ops[op_ind]                                     # IndexError for op_ind > 7, as the reference
        if specified_param is not None:
            out = Op.execute(img, mask=mask, features=None, specified_param=specified_param, has_noise=has_noise)
        else:
            out = Op.execute(img, mask=mask, features=features, has_noise=has_noise)
        return out, Op.param

    def get_param_bnd(self, op_ind):
        return self.ops[op_ind].get_param_range()

    def get_param_num(self, op_ind):
        return self.ops[op_ind].num_op_param

    # ------------------------------------------------------------------ batched extensions
    def predict_params(self, op_ids, features):
        """(B,24) zero-padded parameters: row b from the head of operator op_ids[b] (t2o_param_heads_fwd: one
        workgroup per sample evaluates that sample's own head).  No host sync, no regrouping of the batch; heads
        that no sample selected get an all-zero gradient.  Identity / inpaint rows are zeros."""
        if features.is_cuda and features.shape[1] == 512 and self.opt.operator_fc_dim == 512:
            heads = {k: (Op.fc1.weight, Op.fc1.bias, Op.fc2.weight, Op.fc2.bias) for k, Op in enumerate(self.ops) if k != 4}
            lo, hi = self.opt.saturation_range
            return T.param_heads(features, op_ids.to(torch.int32), heads,
                                 (self.opt.brightness_range, lo, hi, self.opt.sharpness_range),
                                 into_grad=self.__dict__.get('heads_grad_in_place', False))
        return self.predict_params_gemm(op_ids, features)

    def predict_params_gemm(self, op_ids, features):
        """The same through library GEMMs: every head on the whole batch, rows picked with a gather (other feature
        widths; also the comparison the tests hold the fused kernels to)."""
        B = features.shape[0]
        table = features.new_zeros(len(self.ops) + 1, B, PARAM_PAD)        # last slot: identity / unsupported
        for k, Op in enumerate(self.ops):
            if k == 4:
                continue
            p = Op.extract_parameters(features)
            table[k, :, :p.shape[1]] = p
        idx = torch.where((op_ids < 0) | (op_ids == 4), torch.full_like(op_ids, len(self.ops)), op_ids)
        return table.gather(0, idx.long().view(1, B, 1).expand(1, B, PARAM_PAD)).squeeze(0)

    def execute_per_sample(self, img, op_ids, mask, features=None, specified_param=None):
        """op_ids: (B,) integer tensor on the GPU of executor indices (-1 = identity).
        Returns (out (B,3,H,W), param (B,24)) like the per-group loop + regroup of actor.py:244-259."""
        assert (features is None) ^ (specified_param is None)
        op_ids = op_ids.to(torch.int32)
        param = self.predict_params(op_ids, features) if features is not None else specified_param
        return T.apply_per_sample(op_ids, img, param, mask), param

    @staticmethod
    def _pad_params(img, params):
        if torch.is_tensor(params):
            return params
        B = img.shape[0]
        rows = [torch.cat([p, p.new_zeros(B, PARAM_PAD - p.shape[1])], 1) if p.shape[1] < PARAM_PAD else p for p in params]
        return torch.stack(rows, 0)

    def run_sequence(self, img, ops, params, target):
        """ops: python ints; params: list of (B,n_k) or a (K,B,24) tensor.
        Returns (loss = mean |out_K - target|, acts (K,B,3,H,W)): every intermediate image is
        materialised, exactly what K calls of execute() would return."""
        return T.sequence_l1(img, ops, self._pad_params(img, params), target)

    def run_sequence_fused(self, img, ops, params, target):
        """Same sequence, same loss and gradients, but runs of per-pixel operators execute in
        registers (one read of the image, one write): returns (loss, out (B,3,H,W)) -- for callers
        that only need the end result, like the planner's candidate evaluation."""
        return T.fused_sequence_l1(img, ops, self._pad_params(img, params), target)

    def value_and_grad(self, img, ops, params, target, gloss=None, want_image=False, want_image_grad=True):
        """loss = mean |sequence(img) - target| and its gradients in ONE call, outside autograd -- one iteration of a
        parameter fit (utils/beam_search.py:65-91) or of an L1 train step's executor part without a separate forward:
        the last segment's backward kernels also produce the loss (T.fused_sequence_l1_value_grad).  Returns
        (loss, gimg or None, gparams (K,B,24), out or None): gradients and image bit-identical to run_sequence_fused +
        backward, the loss equal up to summation order."""
        return T.fused_sequence_l1_value_grad(img, ops, self._pad_params(img, params), target, gloss=gloss,
                                              want_image=want_image, want_image_grad=want_image_grad)
