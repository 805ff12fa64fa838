"""Training driver with the reference's loop structure (experiments/t2onet/train_seq2seqL1.py:22-176):
alternating supervised / episode steps, running-mean timers ('fs time', 'L1 time'), periodic
evaluation and `model.pth` checkpoints with the reference's state_dict layout.

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
from .data import FiveKAct, SyntheticFiveK
from .train import Trainer


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

    opt = default_options(batch_size=args.batch_size, learning_rate=args.learning_rate, print_every=args.print_every)
    model = Actor(opt).to(device).train()
    trainer = Trainer(model, opt)
    torch.manual_seed(args.manual_seed + 1000 * rank)        # independent sampling / dropout streams per rank

    dataset = SyntheticFiveK(n=args.batch_size * 64, size=args.img_size) if args.synthetic else \
        FiveKAct(args.img_dir, args.anno_dir, args.act_dir, 'train', 1, args.img_size)
    sampler = DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                        num_workers=args.num_workers, drop_last=True)
    ckpt_dir = os.path.join(args.run_dir, 'seq2seqL1_model')
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
            if rank == 0 and (itr % args.checkpoint_every == 0 or itr >= args.num_iters):
                d = os.path.join(ckpt_dir, 'checkpoint_iter{:08d}'.format(itr))
                os.makedirs(d, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(d, 'model.pth'))
                with open(os.path.join(d, 'checkpoint_iter{:08d}.json'.format(itr)), 'w') as f:
                    json.dump({'train_iter': itr, 'avg': avg}, f)
            if itr >= args.num_iters:
                break
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return avg


if __name__ == '__main__':
    main()
