"""Per-launch time of the parameter-head kernels (t2o_param_heads_fwd/_bwd) at the train step's shape (B=64).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; prints event-timed fwd / bwd totals itself."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import t2onet_amd  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
ex = t2onet_amd.Executor(t2onet_amd.default_options()).to(dev)
B = 64
feats = (torch.rand(B, 512, device=dev) * 2 - 1).requires_grad_(True)
op_ids = torch.tensor([0, 1, 2, 3, 5, 6, 7, -1] * 8, dtype=torch.int32, device=dev)
gout = torch.rand(B, 24, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(30):
    ev[0].record()
    p = ex.predict_params(op_ids, feats)
    ev[1].record()
    p.backward(gout)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 10:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print('heads B=64: forward %.1f us, backward %.1f us (event-timed, incl. launch gaps)' % (tf / 20 * 1e3, tb / 20 * 1e3))
