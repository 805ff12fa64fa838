"""Training driver with the reference's loop structure (experiments/t2onet/train_seq2seqL1.py:22-176):
alternating supervised / episode steps, running-mean timers ('fs time', 'L1 time'), and every
`checkpoint_every` iterations: evaluation on the validation split (evaluate.test), a `model.pth` checkpoint with
the reference's state_dict layout, and `checkpoint_best` when the validation L1 improved (:103-131).
The request encoder's word rows come from the GloVe table (--word2vec, the reference's
{dataset}_vocabs_glove_feat_{session}.h5 or the same matrix as .npy); without one the loop refuses to freeze random
word vectors (fix_input_embedding) -- synthetic runs train the whole table instead.

    python -m t2onet_amd.train_cli --synthetic --batch_size 64 --num_iters 100
    python -m torch.distributed.run --nproc-per-node 8 -m t2onet_amd.train_cli --synthetic ...   # data parallel

Real data: --img_dir/--anno_dir/--act_dir with the reference's FiveK layout (datasets/FiveKdataset.py).
"""
import argparse
import json
import os
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler

from . import default_options
from .actor import Actor
from . import evaluate
from .data import FiveK, FiveKAct, SyntheticFiveK
from .train import Trainer


def load_word2vec(path):
    """(vocab - 4, 300) float32 GloVe rows: .npy, or the reference's .h5 (dataset 'glove', utils/text_utils.py:66-73)."""
    import numpy as np
    if path.endswith('.npy'):
        return torch.from_numpy(np.load(path).astype('float32'))
    import h5py                                     # not in the build image; present where the reference's data lives
    with h5py.File(path, 'r') as f:
        return torch.from_numpy(f['glove'][()].astype('float32'))


class _EvalView(torch.utils.data.Dataset):
    """(img_x, img_y, x, req) items for evaluate.test from a FiveKAct-style dataset (its last image is the target)."""

    def __init__(self, base):
        self.base = base

    def __len__(self):
        return len(self.base)

    def __getitem__(self, i):
        img_x, imgs, x, _, _, req = self.base[i]
        return img_x, imgs[-1], torch.as_tensor(x), req


def sync_batchnorm_buffers(model, world):
    """Data-parallel ranks keep their own running statistics during training (as the reference would at this batch
    size); average them before they are evaluated / saved so that the checkpoint does not depend on which rank writes it."""
    if world <= 1:
        return
    for name, buf in model.named_buffers():
        if name.endswith(('running_mean', 'running_var')):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf.div_(world)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--img_dir', default='data/FiveK/images')
    ap.add_argument('--anno_dir', default='data/FiveK/annotations')
    ap.add_argument('--act_dir', default='output/actions_set_1')
    ap.add_argument('--run_dir', default='output/FiveK_trial_1')
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--img_size', type=int, default=128)
    ap.add_argument('--num_iters', type=int, default=10000)
    ap.add_argument('--learning_rate', type=float, default=1e-3)
    ap.add_argument('--print_every', type=int, default=100)
    ap.add_argument('--checkpoint_every', type=int, default=1000)
    ap.add_argument('--num_workers', type=int, default=1)
    ap.add_argument('--manual_seed', type=int, default=10)
    ap.add_argument('--word2vec', default=None, help='GloVe rows of the request vocabulary (.h5 of the reference, or .npy)')
    ap.add_argument('--val_items', type=int, default=64, help='synthetic runs: size of the validation split')
    ap.add_argument('--eager', action='store_true', help='no channels-last encoder / hipGraphs (debugging)')
    args = ap.parse_args(argv)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    torch.manual_seed(args.manual_seed)                      # identical initial weights on every rank

    word2vec = load_word2vec(args.word2vec) if args.word2vec else None
    if word2vec is None and not args.synthetic:
        raise SystemExit('train_cli: --word2vec is required with real data: the reference freezes the GloVe word rows '
                         '(fix_input_embedding=1, lang_encoder.py:22-31); freezing random rows would train nothing there')
    opt = default_options(batch_size=args.batch_size, learning_rate=args.learning_rate, print_every=args.print_every,
                          fix_input_embedding=1 if word2vec is not None else 0)
    model = Actor(opt, word2vec=word2vec).to(device).train()
    if not args.eager:
        model.use_channels_last()
    trainer = Trainer(model, opt, graph_encoder=not args.eager)
    torch.manual_seed(args.manual_seed + 1000 * rank)        # independent sampling / dropout streams per rank

    dataset = SyntheticFiveK(n=args.batch_size * 64, size=args.img_size) if args.synthetic else \
        FiveKAct(args.img_dir, args.anno_dir, args.act_dir, 'train', 1, args.img_size)
    sampler = DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                        num_workers=args.num_workers, drop_last=True)
    if args.synthetic:
        val_loader = DataLoader(_EvalView(SyntheticFiveK(n=args.val_items, size=args.img_size, seed=args.manual_seed + 1)),
                                batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers)
    else:
        # the reference validates on FiveK(..., 'val'): no planned actions needed, full resolution (short side 600), one
        # image per batch (train_seq2seqL1.py:155-156) -- the validation L1 and checkpoint_best follow that
        val_loader = DataLoader(FiveK(args.img_dir, args.anno_dir, 'val', 1), batch_size=1, shuffle=False, num_workers=1)
    ckpt_dir = os.path.join(args.run_dir, 'seq2seqL1_model')
    stats = {'train_iter': [], 'val_dist': [], 'best_val_dist': float('inf'), 'best_iter': 0}
    itr, epoch = 0, 0
    avg = dict(op=0.0, param=0.0, l1=0.0, fs_t=0.0, l1_t=0.0)
    while itr < args.num_iters:
        epoch += 1
        if sampler is not None:
            sampler.set_epoch(epoch)
        for img_x, img_y, x, y, gt_params, _ in loader:
            itr += 1
            tik = time.time()
            lengths = (x != opt.null_id).sum(1)
            x, y, img_x, img_y, gt_params = (t.to(device, non_blocking=True) for t in (x, y, img_x, img_y, gt_params))
            if itr % 2 == 1:
                op_loss, param_loss = trainer.supervised_step(x, y, img_x, img_y, gt_params, lengths)
                k = 1.0 / (itr // 2 + 1)
                if itr % args.print_every in (0, 1):                 # .item() only when printing
                    avg['op'], avg['param'] = op_loss.item(), param_loss.item()
                avg['fs_t'] += (time.time() - tik - avg['fs_t']) * k
            else:
                l1 = trainer.episode_step(x, img_x, img_y[:, -1], lengths=lengths)
                k = 1.0 / (itr // 2)
                if itr % args.print_every == 0:
                    avg['l1'] = l1.item()
                avg['l1_t'] += (time.time() - tik - avg['l1_t']) * k
            if rank == 0 and itr % args.print_every == 0:
                print('iter {:6d} / {}, epoch {:2d}, op loss {:.2f}, param loss {:.2f}, L1 loss {:.2f}, fs time {:.3f}, '
                      'L1 time {:.3f}'.format(itr, args.num_iters, epoch, avg['op'], avg['param'], avg['l1'],
                                              avg['fs_t'], avg['l1_t']), flush=True)
            if itr % args.checkpoint_every == 0 or itr >= args.num_iters:
                sync_batchnorm_buffers(model, world)                  # every rank takes part in the collective
                if rank == 0:
                    init_val_dist, val_dist = evaluate.test(model, val_loader, opt, device=device, verbose=False)
                    model.train()
                    print('validation L1 dist {:.4f} (init {:.4f})'.format(val_dist, init_val_dist), flush=True)
                    stats['val_dist'].append(val_dist)
                    stats['train_iter'].append(itr)
                    d = os.path.join(ckpt_dir, 'checkpoint_iter{:08d}'.format(itr))
                    os.makedirs(d, exist_ok=True)
                    torch.save(model.state_dict(), os.path.join(d, 'model.pth'))
                    with open(os.path.join(d, 'checkpoint_iter{:08d}.json'.format(itr)), 'w') as f:
                        json.dump(stats, f)
                    if val_dist < stats['best_val_dist']:
                        stats['best_val_dist'], stats['best_iter'] = val_dist, itr
                        best = os.path.join(ckpt_dir, 'checkpoint_best')
                        os.makedirs(best, exist_ok=True)
                        torch.save(model.state_dict(), os.path.join(best, 'model.pth'))
                        with open(os.path.join(best, 'checkpoint_best.json'), 'w') as f:
                            json.dump(stats, f)
                if world > 1:
                    dist.barrier()
            if itr >= args.num_iters:
                break
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    avg['stats'] = stats
    return avg


if __name__ == '__main__':
    main()
