"""One 128x128 train step pair (the reference's training size) and one full-resolution (600x900 and 397x600) inference
episode, for rocprofv3 --kernel-trace --stats: the kernel list must hold no library (MIOpen / CK) convolution kernel.
    rocprofv3 --kernel-trace --stats ... -- python tools/conv_kernel_audit.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
from oracle import synth

dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()
torch.manual_seed(1)
model = Actor(opt).to(dev).train()
model.use_channels_last()
tr = Trainer(model, opt)
B, S = 8, 128
x = synth.requests(B, 17, 1).to(dev)
lengths = (x != 0).sum(1).cpu()
img = synth.images(B, S, S, 2).to(dev)
img_y = synth.uniform((B, 6, 3, S, S), 3).to(dev)
y = synth.op_targets(B, 4).to(dev)
gt = synth.uniform((B, 5, 24), 5, -1, 1).to(dev)
for _ in range(2):
    tr.supervised_step(x, y, img, img_y, gt, lengths)
    tr.episode_step(x, img, img_y[:, -1], lengths=lengths)
model.eval()
with torch.no_grad():
    for hw in ((600, 900), (397, 600)):
        im = synth.images(1, hw[0], hw[1], 6).to(dev)
        _, imgs, ops, _ = model.episode_forward(x[:1], im, None, reinforce_sample=0)
torch.cuda.synchronize()
print('audit run done', tuple(imgs.shape))
