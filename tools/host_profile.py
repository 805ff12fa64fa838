"""Where the host time of the episode train step goes: (1) the step at a tiny image size (GPU work ~ nothing,
so ms/step ~ pure host cost), (2) torch.profiler's per-operator CPU self time for a few steps at full size.

    python tools/host_profile.py [nhwc]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd  # noqa: E402
from t2onet_amd.actor import Actor  # noqa: E402
from t2onet_amd.train import Trainer  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda:0')
nhwc = 'nhwc' in sys.argv[1:]


def make(B, S, graph):
    opt = t2onet_amd.default_options()
    torch.manual_seed(10)
    model = Actor(opt).to(dev).train()
    if nhwc:
        model.use_channels_last()
    tr = Trainer(model, opt, graph_encoder=graph)
    g = torch.Generator().manual_seed(10)
    img = torch.rand(B, 3, S, S, generator=g).to(dev)
    tgt = torch.rand(B, 3, S, S, generator=g).to(dev)
    x = bench.synthetic_requests(B, g)
    lengths = (x != 0).sum(1)
    return tr, x.to(dev), img, tgt, lengths


def timeit(tr, x, img, tgt, lengths, n=10, w=4):
    for _ in range(w):
        tr.episode_step(x, img, tgt, lengths=lengths)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.episode_step(x, img, tgt, lengths=lengths)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


for S, graph in ((32, True), (32, False), (256, True)):
    args = make(64, S, graph)
    enq, tot = timeit(*args)
    print('size %3d graphs %-5s  enqueue %.2f ms/step  total %.2f ms/step' % (S, graph, enq, tot), flush=True)
    if S == 256:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                args[0].episode_step(*args[1:4], lengths=args[4])
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=60))
    del args
    torch.cuda.empty_cache()
