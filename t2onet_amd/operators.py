"""Edit operators with the reference's Python surface (models/operators.py) on HIP kernels.

Kept from the reference, name for name: `Operator.execute(img, mask=None, features=None,
specified_param=None, has_noise=False) -> out`, the `.param` / `.mask` side effects, `fc1`,
`lrelu`, `fc2` (state_dict keys), `op_param_regressor`, `get_param_range`, `get_param_noise`,
`short_name`, `num_op_param`.  Different by design: `process` + mask blend + clamp
(operators.py:128-130) is ONE fused kernel launch (forward) and one (backward); there is no
eager CPU path -- tensors must live on the GPU.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as T

# executor index of each operator (executors/executor.py:30)
BRIGHTNESS, CONTRAST, SATURATION, COLOR, INPAINT, TONE, SHARPNESS, WHITE = range(8)


def tanh_range(lo, hi, initial=None):
    """utils/operator_utils.py:21-34."""
    bias = 0.0
    if initial is not None:
        z = 2 * (initial - lo) / (hi - lo) - 1
        bias = 0.5 * math.log((1 + z) / (1 - z))
    return lambda x: (torch.tanh(x + bias) * 0.5 + 0.5) * (hi - lo) + lo


class Operator(nn.Module):
    """Predict the operator parameter from a feature vector and apply the operator."""
    op_index = None

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.is_discrete = cfg.discrete_param
        if self.is_discrete:
            raise NotImplementedError('discrete_param=1 is not on the FiveK hot path (SURVEY.md section 5)')
        self.channels = 2 * cfg.hidden_size
        self.num_op_param = None
        self.short_name = None
        self.param = None
        self.mask = None

    def setup(self):                                   # operators.py:43-55
        self.fc1 = nn.Linear(self.channels, self.cfg.operator_fc_dim)
        self.lrelu = nn.LeakyReLU(inplace=True)
        self.fc2 = nn.Linear(self.cfg.operator_fc_dim, self.get_num_op_param())
        self.ub, self.lb, self.initial = self.get_param_range()

    def get_short_name(self):
        return self.short_name

    def get_num_op_param(self):
        assert self.num_op_param is not None, 'Must specify the number of parameter'
        return self.num_op_param

    def get_param_noise(self, bs):                     # operators.py:57-60
        noise = torch.randn(bs, self.num_op_param)
        return (F.relu(noise) * (self.ub - self.initial) + F.relu(-noise) * (self.initial - self.lb)) / 3 \
            * self.cfg.param_noise_factor

    def extract_parameters(self, features):            # operators.py:73-88
        return self.op_param_regressor(self.fc2(self.lrelu(self.fc1(features))))

    def op_param_regressor(self, features):
        raise NotImplementedError

    def get_param_range(self):
        raise NotImplementedError

    def execute(self, img, mask=None, features=None, specified_param=None, has_noise=False):
        assert (features is None) ^ (specified_param is None)          # operators.py:113
        param = self.extract_parameters(features) if features is not None else specified_param
        if has_noise:                                                    # operators.py:118-121
            param = torch.clamp(param + self.get_param_noise(img.shape[0]).to(img.device), self.lb, self.ub)
        # the reference's side effects (operators.py:122,125); plain attributes, so skip nn.Module.__setattr__
        self.__dict__['param'] = param
        self.__dict__['mask'] = mask
        return T.operator_apply(self.op_index, img, param, mask)


class BrightnessOperator(Operator):                    # operators.py:259-295
    op_index = BRIGHTNESS

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'brightness', 1
        self.setup()

    def op_param_regressor(self, features):
        return tanh_range(-self.cfg.brightness_range, self.cfg.brightness_range, initial=0)(features)

    def get_param_range(self):
        return self.cfg.brightness_range, -self.cfg.brightness_range, 0


class ContrastOperator(Operator):                      # operators.py:224-257
    op_index = CONTRAST

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'contrast', 1
        self.setup()

    def op_param_regressor(self, features):
        return torch.tanh(features)

    def get_param_range(self):
        return 1, -1, 0


class SaturationOperator(Operator):                    # operators.py:454-491
    op_index = SATURATION

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'saturation', 1
        self.setup()

    def op_param_regressor(self, features):
        lo, hi = self.cfg.saturation_range
        return torch.tanh(F.relu(features)) * hi + torch.tanh(F.relu(-features)) * lo

    def get_param_range(self):
        return self.cfg.saturation_range[1], self.cfg.saturation_range[0], 0


class ColorOperator(Operator):                         # operators.py:593-622 ("hue": 3 x 8-knot curves)
    op_index = COLOR

    def __init__(self, cfg):
        super().__init__(cfg)
        assert cfg.curve_steps == 8, 'kernels are built for curve_steps=8'
        self.curve_steps = cfg.curve_steps
        self.short_name, self.num_op_param = 'hue', 3 * cfg.curve_steps
        self.setup()

    def op_param_regressor(self, features):
        return features

    def get_param_range(self):
        lo, hi = self.cfg.color_curve_range
        return hi, lo, (hi + lo) / 2


class ToneOperator(Operator):                          # operators.py:557-591 (one 8-knot curve)
    op_index = TONE

    def __init__(self, cfg):
        super().__init__(cfg)
        assert cfg.curve_steps == 8, 'kernels are built for curve_steps=8'
        self.curve_steps = cfg.curve_steps
        self.short_name, self.num_op_param = 'tone', cfg.curve_steps
        self.setup()

    def op_param_regressor(self, features):
        return features

    def get_param_range(self):
        lo, hi = self.cfg.tone_curve_range
        return hi, lo, (hi + lo) / 2


class SharpnessOperator(Operator):                     # operators.py:332-370
    op_index = SHARPNESS

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'sharpness', 1
        self.setup()

    def op_param_regressor(self, features):
        return torch.sigmoid(features) * self.cfg.sharpness_range

    def get_param_range(self):
        return self.cfg.sharpness_range, 0, self.cfg.sharpness_range / 2


class WhiteOperator(Operator):                         # operators.py:494-525 ("color_bg")
    op_index = WHITE

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'color_bg', 1
        self.setup()

    def op_param_regressor(self, features):
        return torch.sigmoid(features)

    def get_param_range(self):
        return 1, 0, 0.5


class InpaintOperator(Operator):                       # operators.py:625-682
    """Registered for state_dict compatibility (fc1/fc2) and index order only: the operator
    itself is the EdgeConnect inpainting network, a git submodule that is empty in the
    reference checkout, and it is masked off on the FiveK path (actor.py:211)."""
    op_index = INPAINT

    def __init__(self, cfg):
        super().__init__(cfg)
        self.short_name, self.num_op_param = 'inpaint_obj', 1
        self.setup()

    def op_param_regressor(self, features):
        return torch.zeros(features.shape[0], self.num_op_param, device=features.device, requires_grad=True)

    def get_param_range(self):
        return 0, 0, 0

    def execute(self, img, mask=None, features=None, specified_param=None, has_noise=False):
        raise RuntimeError('inpaint_obj needs the EdgeConnect network (pyutils/edgeconnect submodule); '
                           'it is outside the per-pixel executor path')
