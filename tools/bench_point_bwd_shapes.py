"""Why are the materialised point-operator backwards slower at 16 x 512^2 than at 64 x 256^2 (the same pixel count)?
rocprofv3 --kernel-trace --stats -- python tools/bench_point_bwd_shapes.py A|B   (A = 64x256x256, B = 16x512x512)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd.functional as T
dev = torch.device('cuda:0')
B, H = (64, 256) if sys.argv[1] == 'A' else (16, 512)
for op in (0, 1, 2):
    img = torch.rand(B, 3, H, H, device=dev, requires_grad=True)
    par = (torch.rand(B, 1, device=dev) * 0.4 - 0.2).requires_grad_(True)
    for _ in range(20):
        out = T.operator_apply(op, img, par)
        out.backward(torch.ones_like(out) * 1e-3)
torch.cuda.synchronize()
