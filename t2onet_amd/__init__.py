"""t2onet_amd: MI355X-native executor/actor hot path of T2ONet.

    from t2onet_amd import Executor, default_options

The compute path is libt2onet_hip.so (C ABI in include/t2onet_hip.h, hand-written gfx950
kernels).  There is no CPU fallback: the library must be built (`python -m t2onet_amd.build`)
and tensors must live on the GPU.
"""
from .options import default_options  # noqa: F401
from .executor import Executor  # noqa: F401
from . import functional  # noqa: F401
