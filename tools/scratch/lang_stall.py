import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
import bench
dev = torch.device('cuda:0')
for dp in (0.2, 0.0):
    opt = t2onet_amd.default_options(); opt.dropout_p = dp; opt.input_dropout_p = dp
    torch.manual_seed(10)
    model = Actor(opt).to(dev).train()
    g = torch.Generator().manual_seed(10)
    x = bench.synthetic_requests(64, g); lengths = (x != 0).sum(1); x = x.to(dev)
    big = torch.randn(8192, 8192, device=dev)
    for it in range(3):
        enc_out, hid, _ = model.lang_encoder(x, lengths); (enc_out.sum() + hid[0].sum()).backward()
    torch.cuda.synchronize()
    for it in range(2):
        for _ in range(6): big @ big          # ~40 ms of queued GPU work
        t0 = time.perf_counter()
        enc_out, hid, _ = model.lang_encoder(x, lengths)
        t1 = time.perf_counter()
        loss = enc_out.sum() + hid[0].sum()
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print('dropout %.1f: host fwd %.2f ms, bwd %.2f ms, drain %.2f ms' % (dp, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
