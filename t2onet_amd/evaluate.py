"""Evaluation loops of experiments/t2onet/test_seq2seqL1.py (test :28-95, test_variance :99-142) and the L1 / SSIM parts of
utils/eval.py:13-60 (FID needs torchvision's InceptionV3: out of scope).
Everything runs on the GPU: argmax episode, END-image select, L1 and SSIM through the HIP kernels."""
import time

import torch

from . import functional as T
from .train import select_end_images


class ImageEvaluator(object):
    """Running means of input/output L1 and SSIM against the ground truth (utils/eval.py:13-60)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.itr = 0
        self.avg_out_L1 = self.avg_in_L1 = 0.0
        self.avg_out_SSIM = self.avg_in_SSIM = 0.0

    def update(self, input, output, gt):
        self.itr += 1
        k = 1.0 / self.itr
        self.avg_in_L1 = self.avg_in_L1 * (1 - k) + T.l1_loss(input, gt).item() * k
        self.avg_out_L1 = self.avg_out_L1 * (1 - k) + T.l1_loss(output, gt).item() * k
        self.avg_in_SSIM = self.avg_in_SSIM * (1 - k) + T.ssim(input, gt).item() * k
        self.avg_out_SSIM = self.avg_out_SSIM * (1 - k) + T.ssim(output, gt).item() * k

    def eval(self):
        print('input L1 dist {:.4f}, output L1 dist {:.4f}'.format(self.avg_in_L1, self.avg_out_L1))
        print('input SSIM {:.4f}, output SSIM {:.4f}'.format(self.avg_in_SSIM, self.avg_out_SSIM))
        return dict(in_L1=self.avg_in_L1, out_L1=self.avg_out_L1, in_SSIM=self.avg_in_SSIM, out_SSIM=self.avg_out_SSIM)


def test(model, loader, opt, is_test=False, device=None, verbose=True):
    """loader yields (img_x, img_y, x, req) like datasets/FiveKdataset.py:FiveK.
    Returns (avg_init_dist, avg_dist) = running means of mean|img_x - img_y| and mean|pred - img_y|."""
    model.eval()
    device = device or next(model.parameters()).device
    single = model.module if hasattr(model, 'module') else model
    evaluator = ImageEvaluator() if is_test else None
    itr, avg_time, avg_dist, avg_init_dist = 0, 0.0, 0.0, 0.0
    for data in loader:
        itr += 1
        tik = time.time()
        img_x, img_y, x = data[0], data[1], data[2]
        lengths = (x != opt.null_id).sum(1)                      # on the host, before the copy
        x, img_x, img_y = x.to(device), img_x.to(device), img_y.to(device)
        with torch.no_grad():
            _, pred_imgs, pred_ops, _ = single.episode_forward(x, img_x, None, reinforce_sample=False, lengths=lengths)
            pred_img = select_end_images(pred_imgs, pred_ops, opt.end_id)
            init_dist = T.l1_loss(img_x, img_y).item()
            dist = T.l1_loss(pred_img, img_y).item()
        avg_time += (time.time() - tik - avg_time) / itr
        avg_init_dist += (init_dist - avg_init_dist) / itr
        avg_dist += (dist - avg_dist) / itr
        if evaluator is not None:
            evaluator.update(img_x, pred_img, img_y)
        if verbose and itr % max(1, getattr(opt, 'print_every', 100)) == 0:
            print('iter {:6d}, init dist {:.2f},  L1 dist {:.2f} time {:.2f}'.format(itr, init_dist, dist, avg_time))
    if evaluator is not None:
        evaluator.eval()
    if verbose:
        print('inference init L1 dist {:.4f}; L1 dist {:.4f}'.format(avg_init_dist, avg_dist))
    return avg_init_dist, avg_dist


def test_variance(model, loader, opt, requests, vocab2id=None, device=None, verbose=True):
    """experiments/t2onet/test_seq2seqL1.py:99-142: how much the edit depends on the wording -- for every batch of the
    loader the arg-max episode is run once per test request (the SAME request for every image of the batch), the END
    images of all requests are concatenated and the unbiased variance over that axis is averaged over pixels; returns
    the running mean over batches.

    requests: the reference imports its list from `core.utils.eval.test_txts`, a module that is not part of the
    repository, so the caller supplies it: strings (tokenised with `vocab2id` like utils/text_utils.py:42-67) or
    (1, encoder_max_len) / (encoder_max_len,) token-id tensors.  As in the reference the request row is (1, L) while the
    images are (bs, ...): it is broadcast over the batch here (the reference's encoder does that only for bs = 1)."""
    from .data import txt2idx
    model.eval()
    device = device or next(model.parameters()).device
    single = model.module if hasattr(model, 'module') else model
    rows = []
    for r in requests:
        if isinstance(r, str):
            if vocab2id is None:
                raise ValueError('test_variance: text requests need vocab2id')
            r = txt2idx(r, vocab2id, opt.encoder_max_len)
        rows.append(torch.as_tensor(r, dtype=torch.long).view(1, -1))
    if len(rows) < 2:
        raise ValueError('test_variance: the variance over fewer than two requests is undefined')
    itr, avg_var, avg_time = 0, 0.0, 0.0
    for data in loader:
        itr += 1
        tik = time.time()
        img_x = data[0].to(device)
        ends = []
        for row in rows:
            x = row.expand(img_x.shape[0], -1).contiguous()
            lengths = (x != opt.null_id).sum(1)
            with torch.no_grad():
                _, pred_imgs, pred_ops, _ = single.episode_forward(x.to(device), img_x, None, reinforce_sample=False, lengths=lengths)
                ends.append(select_end_images(pred_imgs, pred_ops, opt.end_id))
        var = torch.var(torch.cat(ends), dim=0).mean().item()
        avg_var += (var - avg_var) / itr
        avg_time += (time.time() - tik - avg_time) / itr
        if verbose and itr % max(1, getattr(opt, 'print_every', 100)) == 0:
            print('iter {:6d}, var {:.6f}, time {:.2f}'.format(itr, avg_var, avg_time))
    if verbose:
        print('avg var: {:.6f}'.format(avg_var))
    return avg_var
