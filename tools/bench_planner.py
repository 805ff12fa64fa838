"""Time one planner parameter fit per operator: the reference's procedure (scipy Nelder-Mead, one
executor call + .item() per evaluation -- here already on the HIP kernels) vs the GPU-native
'sweep' optimiser, and one full beam search."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd  # noqa: E402
from t2onet_amd import planner  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ex = t2onet_amd.Executor(t2onet_amd.default_options()).cuda()
g = torch.Generator().manual_seed(3)
img = torch.rand(1, 3, S, S, generator=g).cuda()
truth = {0: torch.tensor([[0.3]]), 1: torch.tensor([[-0.25]]), 2: torch.tensor([[0.4]]),
         5: torch.tensor([[0.7, 0.9, 1.1, 1.3, 1.2, 1.0, 0.9, 0.8]])}
for op, p in truth.items():
    tgt, _ = ex.execute(img, op, None, specified_param=p.cuda())
    for optm in ['Nelder-Mead', 'sweep']:
        planner.get_param(img, tgt, None, op, ex, None, 'L1', optm)          # warm up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q, _ = planner.get_param(img, tgt, None, op, ex, None, 'L1', optm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out, _ = ex.execute(img, op, None, specified_param=q)
        print('size %d op %d %-12s %8.2f ms  residual L1 %.2e' % (S, op, optm, dt * 1e3, planner.get_dist(out, tgt).item()), flush=True)
mid, _ = ex.execute(img, 0, None, specified_param=torch.tensor([[0.25]]).cuda())
tgt2, _ = ex.execute(mid, 1, None, specified_param=torch.tensor([[0.3]]).cuda())
names = ['brightness', 'contrast', 'saturation', 'color', 'inpaint', 'tone', 'sharpness', 'white']
for optm in ['Nelder-Mead', 'sweep']:
    planner.beam_search(img, tgt2, None, ex, None, 3, [0, 1, 2], names, 3, 1e-3, 'L1', optm)      # warm up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    actions, _ = planner.beam_search(img, tgt2, None, ex, None, 3, [0, 1, 2], names, 3, 1e-3, 'L1', optm)
    torch.cuda.synchronize()
    print('beam search (3 ops, beam 3) %-12s %8.1f ms  best dist %.2e' % (optm, (time.perf_counter() - t0) * 1e3, actions[0][-1][2]))
