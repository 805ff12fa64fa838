import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import synth
from tests.test_gpu_actor_extra import make_model2, PICK_SLICES, L
from t2onet_amd.train import select_end_images
import t2onet_amd.functional as T
g=np.load('tests/golden/extra2.npz')
dev=torch.device('cuda:0')
B2,S=8,128
x=synth.requests(B2,L,141).to(dev); img=synth.images(B2,S,S,142).to(dev); tgt=synth.images(B2,S,S,143).to(dev)
for nhwc in (False, True):
    model,opt=make_model2(dev)
    if nhwc: model.use_channels_last()
    model.train()
    _,pred_imgs,pred_ops,pp=model.episode_forward(x,img,None,reinforce_sample=0)
    loss=T.l1_loss(select_end_images(pred_imgs,pred_ops,opt.end_id),tgt); loss.backward()
    print('nhwc',nhwc,'loss',loss.item(), float(g['ep12864_loss']), 'ops eq', np.array_equal(pred_ops.cpu().numpy(), g['ep128_ops']))
    pr=torch.stack(pp,0).detach().cpu().numpy()
    print('  params err vs 64: %.2e  ref32 vs 64: %.2e' % (np.abs(pr-g['ep12864_params']).max(), np.abs(g['ep128_params']-g['ep12864_params']).max()))
    named=dict(model.named_parameters())
    for n in g['grad_picks']:
        n=str(n); r64=g['ep12864_grad:'+n]; r32=g['ep128_grad:'+n].astype(np.float64)
        gr=named[n].grad; gr=torch.zeros_like(named[n]) if gr is None else gr
        if n in PICK_SLICES: gr=gr[PICK_SLICES[n]]
        got=gr.detach().cpu().numpy().astype(np.float64)
        nb=np.linalg.norm(r64)
        if nb>0: print('  %-45s gpu %.1e  ref32 %.1e' % (n, np.linalg.norm(got-r64)/nb, np.linalg.norm(r32-r64)/nb))
