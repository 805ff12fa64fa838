"""Soak run of the episode train step (configs[1]: bs 64, 256x256): python tools/soak.py [steps=600] [graph_step 0|1] [window=100] [batches=3] [graph_encoder 0|1]
(T2O_SOAK_ALTERNATE=1: the reference's alternation instead -- teacher-forced and episode steps in turn, train_seq2seqL1.py:51-92.)

Per window of steps: ms/step, the caching allocator's allocated / reserved bytes and the device's free memory (hipMemGetInfo).
The run FAILS (exit 1) when device memory in use keeps growing after the first window, when a window is more than 5 % slower
than the first, or when the loss or a parameter stops being finite -- what a leak in the tape / arena / graph bookkeeping, a
clock throttle or a numeric blow-up would look like in a long training job."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
graph_step = (sys.argv[2] != '0') if len(sys.argv) > 2 else False
window = int(sys.argv[3]) if len(sys.argv) > 3 else 100
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 3
graph_encoder = (sys.argv[5] != '0') if len(sys.argv) > 5 else False    # bench.py's default: everything eager

dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()
torch.manual_seed(10)
model = Actor(opt).to(dev).train()
model.use_channels_last()
tr = Trainer(model, opt, graph_encoder=graph_encoder, graph_step=graph_step)
g = torch.Generator().manual_seed(10)
B, H, W = 64, 256, 256
batches = []
for _ in range(nb):                                            # different batches in turn (shapes equal, request lengths not)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
    x = bench.synthetic_requests(B, g)
    batches.append((x.to(dev), img, tgt, (x != 0).sum(1)))
alternate = os.environ.get('T2O_SOAK_ALTERNATE', '0') != '0'
sup = []
if alternate:                                                  # teacher-forced inputs per batch, as bench.py builds them
    npar = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
    for _ in range(nb):
        ops_t = torch.stack([torch.randperm(6, generator=g)[:5] for _ in range(B)])
        y = torch.cat([torch.full((B, 1), 1), torch.tensor([3, 4, 5, 6, 8, 9])[ops_t], torch.full((B, 1), 2)], 1)
        img_y = torch.rand(B, 6, 3, H, W, generator=g).to(dev)
        gt = torch.rand(B, 5, 24, generator=g) * 2 - 1
        for b_ in range(B):
            for k_ in range(5):
                gt[b_, k_, npar[int(y[b_, k_ + 1])]:] = 0
        sup.append((y.to(dev), img_y, gt.to(dev)))


def one_step(i):
    x, img, tgt, lengths = batches[i % nb]
    if alternate and i % 2 == 0:
        y, img_y, gt = sup[i % nb]
        op_loss, param_loss = tr.supervised_step(x, y, img, img_y, gt, lengths=lengths)
        return op_loss + param_loss
    return tr.episode_step(x, img, tgt, lengths=lengths)


for i in range(6):
    one_step(i)
torch.cuda.synchronize()

rows, ok = [], True
done = 0
while done < steps:
    n = min(window, steps - done)
    t0 = time.perf_counter()
    for i in range(n):
        out = one_step(done + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    done += n
    free, total = torch.cuda.mem_get_info(dev)
    loss = out['loss'] if isinstance(out, dict) and 'loss' in out else out[0] if isinstance(out, (tuple, list)) else out
    loss = float(loss) if not isinstance(loss, float) else loss
    finite = bool(torch.isfinite(torch.tensor(loss))) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    rows.append((done, ms, torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev), total - free, loss, finite))
    print('steps %5d  %.2f ms/step  allocated %.3f GB  reserved %.3f GB  device in use %.3f GB  loss %.6f  finite %s'
          % (done, ms, rows[-1][2] / 1e9, rows[-1][3] / 1e9, rows[-1][4] / 1e9, loss, finite), flush=True)

first = rows[0]
for r in rows[1:]:
    if r[4] > first[4] + (64 << 20):
        ok = False; print('FAIL: device memory in use grew from %.3f to %.3f GB at step %d' % (first[4] / 1e9, r[4] / 1e9, r[0]))
        break
for r in rows[1:]:
    if r[1] > first[1] * 1.05:
        ok = False; print('FAIL: %.2f ms/step at step %d against %.2f in the first window' % (r[1], r[0], first[1]))
        break
if not all(r[6] for r in rows):
    ok = False; print('FAIL: non-finite loss or parameter')
print('soak %s: %d steps, graph_step=%s graph_encoder=%s, %.2f -> %.2f ms/step, device in use %.3f -> %.3f GB'
      % ('ok' if ok else 'FAILED', steps, graph_step, graph_encoder, rows[0][1], rows[-1][1], rows[0][4] / 1e9, rows[-1][4] / 1e9))
sys.exit(0 if ok else 1)
