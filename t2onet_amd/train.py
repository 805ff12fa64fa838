"""The two alternating train steps of experiments/t2onet/train_seq2seqL1.py (:51-65 teacher
forced, :74-88 episode + L1), plus single-node data parallelism the reference does not have:
one process per GPU, the batch dimension sharded, ONE all-reduce of a flat fp32 gradient
buffer per step (RCCL over xGMI when the process group backend is "nccl").

Gradient semantics under data parallelism: every parameter owns a slice
of one pre-zeroed flat buffer, so a head that no local sample selected contributes zeros to the
all-reduce and Adam sees a zero gradient -- the behaviour of the reference under the
zero-filling `zero_grad()` of the torch version it was written for.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import functional as T


def first_end_step(pred_ops, end_id):
    """Step of each sample's first END token, else the last step (train_seq2seqL1.py:78-84), without the per-sample
    nonzero() host syncs."""
    B, Tn = pred_ops.shape
    is_end = pred_ops == end_id
    return torch.where(is_end.any(1), is_end.int().argmax(1), torch.full((B,), Tn - 1, device=pred_ops.device))


def select_end_images(pred_imgs, pred_ops, end_id):
    """Image at the first END token, else the last one; pred_imgs (B,T,3,H,W)."""
    first = first_end_step(pred_ops, end_id)
    return pred_imgs[torch.arange(pred_ops.shape[0], device=pred_ops.device), first]


def end_l1_loss(pred_imgs, pred_ops, end_id, target):
    """The episode loss (train_seq2seqL1.py:78-85) from the LIST of step images: on a GPU one kernel reads each sample's END
    image where it lies (functional.end_select_l1); elsewhere the stacked gather + L1."""
    if target.is_cuda and target.dtype == torch.float32 and 1 <= len(pred_imgs) <= 8:
        return T.end_select_l1(pred_imgs, first_end_step(pred_ops, end_id), target)
    return T.l1_loss(select_end_images(torch.stack(pred_imgs, 1), pred_ops, end_id), target)


class FlatGradients:
    """All trainable parameters' .grad as views of one flat fp32 buffer."""

    ALIGN = 64          # floats: every segment starts on a 256-byte boundary (kernels read parameters and gradients
                        # with 16-byte accesses and LDS-DMA straight from this storage)

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        for p, off in zip(self.params, self.offsets):
            # same strides as the parameter (channels-last convolution weights stay channels-last): optimiser and
            # gradient accumulation then run their dense fast paths; the flat all-reduce does not care about layout
            p.grad = _view_like(self.flat[off:off + p.numel()], p)
        if self.flat.is_cuda:
            # explicit opt-in: the backward kernels add into these buffers (functional.enable_grad_accumulation)
            T.enable_grad_accumulation(self.params)

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        """One collective for the whole model (88.7 MB for the actor): torch.distributed's (RCCL under backend "nccl", gloo in
        the CPU tests), or -- `use_own_communicator()` / T2O_OWN_COMM=1 -- the C ABI's t2o_allreduce_mean on a communicator this
        object owns, launched on torch's current stream like every other kernel of the step."""
        comm = self.__dict__.get('_comm')
        if comm is None and os.environ.get('T2O_OWN_COMM', '0') != '0' and self.flat.is_cuda and dist.is_available() and dist.is_initialized():
            comm = self.use_own_communicator()
        if comm is not None:
            from . import _lib
            rc = _lib.load().t2o_allreduce_mean(self.flat.data_ptr(), self.flat.numel(), comm.handle, T._stream(self.flat.device))
            _lib.check(rc, 't2o_allreduce_mean')
            return
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def use_own_communicator(self, nranks=None, rank=None, exchange=None):
        """Build an RCCL communicator through the C ABI (t2o_comm_unique_id / t2o_comm_init_rank) and use it for the gradient
        all-reduce.  Rank 0's 128-byte id reaches the others through `exchange(id_bytes_or_None) -> id_bytes` (default: a
        torch.distributed object broadcast on whatever process group is up).  Collective: every rank must call it."""
        old = self.__dict__.pop('_comm', None)
        if old is not None:
            old.close()
        comm = Communicator(self.flat.device, nranks, rank, exchange)
        self._comm = comm
        return comm

    def close(self):
        """Destroy the communicator this object built (use_own_communicator / T2O_OWN_COMM), if any."""
        comm = self.__dict__.pop('_comm', None)
        if comm is not None:
            comm.close()


class Communicator:
    """An ncclComm_t owned through the C ABI (t2o_comm_*): one per process, on the process's GPU."""

    def __init__(self, device, nranks=None, rank=None, exchange=None):
        import ctypes
        from . import _lib
        lib = _lib.load()
        if not lib.t2o_comm_available():
            lib.t2o_comm_unique_id(None)                      # (sets the error text)
            raise RuntimeError('t2o_comm: ' + lib.t2o_last_error().decode('utf-8', 'replace'))
        if (nranks is None) != (rank is None):
            raise ValueError('Communicator: give nranks and rank together (or neither: the process group\'s, else one rank)')
        if nranks is None:
            up = dist.is_available() and dist.is_initialized()
            nranks, rank = (dist.get_world_size(), dist.get_rank()) if up else (1, 0)
        if not (0 <= int(rank) < int(nranks)):
            raise ValueError('Communicator: rank %s outside 0..%s' % (rank, int(nranks) - 1))
        ident = None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(lib.t2o_comm_unique_id(buf), 't2o_comm_unique_id')
            ident = buf.raw
        if exchange is not None:
            ident = exchange(ident)
        elif nranks > 1:
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            ident = box[0]
        self.nranks, self.rank = int(nranks), int(rank)
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.t2o_comm_init_rank(ctypes.byref(handle), self.nranks, ident, self.rank), 't2o_comm_init_rank')
        self.handle = handle

    def all_reduce_(self, t, mean=False):
        """In-place sum (mean) of a dense fp32 GPU tensor over the communicator, on torch's current stream."""
        from . import _lib
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError('Communicator.all_reduce_: a dense fp32 GPU tensor')
        lib = _lib.load()
        fn = lib.t2o_allreduce_mean if mean else lib.t2o_allreduce
        _lib.check(fn(t.data_ptr(), t.numel(), self.handle, T._stream(t.device)), 't2o_allreduce')
        return t

    def close(self):
        if getattr(self, 'handle', None):
            from . import _lib
            _lib.load().t2o_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:                      # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass


def _view_like(seg, p):
    """The flat segment with the parameter's own shape and strides (channels-last weights stay channels-last)."""
    dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
    return seg.as_strided(p.size(), p.stride()) if dense else seg.view_as(p)


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) -- what train_seq2seqL1.py:169 constructs -- over flat
    buffers: the parameters are re-homed into one flat fp32 buffer (each keeps its shape and strides), the
    gradients already live in FlatGradients' buffer, the two moments are flat too, and a step is ONE streaming
    kernel (t2o_adam_step) instead of a multi-tensor pass over 199 tensors.  GPU only; hyper-parameters as
    attributes (`lr` may be changed between steps)."""

    def __init__(self, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.grads = grads
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        dev = grads.flat.device
        self.flat_param = torch.zeros_like(grads.flat)       # (same segment layout as the gradients: padding stays zero)
        with torch.no_grad():
            for p, off in zip(grads.params, grads.offsets):
                seg = self.flat_param[off:off + p.numel()]
                view = _view_like(seg, p)
                view.copy_(p.data)
                p.data = view                               # the module's parameter now IS a slice of the flat buffer
        self.exp_avg = torch.zeros_like(grads.flat)
        self.exp_avg_sq = torch.zeros_like(grads.flat)
        self._ptrs = [p.data_ptr() for p in grads.params]
        assert dev.type == 'cuda'

    def check_homes(self):
        """Every parameter must still live in the flat buffer: module.to(...), use_channels_last() after the Trainer was
        built re-allocate parameters (only the 4-D ones for a memory-format change), and the update would then go to
        an orphaned copy."""
        for p, ptr in zip(self.grads.params, self._ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError('FlatAdam: a parameter was re-allocated after the optimiser was built (module.to(...), '
                                   'use_channels_last() ...): build the Trainer afterwards')

    def step(self):
        from . import _lib, functional as T
        self.check_homes()
        self.step_count += 1
        rc = _lib.load().t2o_adam_step(self.flat_param.data_ptr(), self.grads.flat.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.flat_param.numel(), self.lr, self.betas[0],
                                       self.betas[1], self.eps, self.step_count, T._stream(self.flat_param.device))
        _lib.check(rc, 't2o_adam_step')

    LAYOUT = 2          # 2: every parameter's segment starts on a 64-float boundary (FlatGradients.ALIGN); 1: packed back to back

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'lr': self.lr,
                'betas': self.betas, 'eps': self.eps, 'layout': self.LAYOUT, 'offsets': list(self.grads.offsets),
                'numels': [p.numel() for p in self.grads.params]}

    def load_state_dict(self, sd):
        """Moments saved by this class under any segment layout: the state carries its offsets (layout >= 2); a state
        without them is the packed layout of earlier versions and is converted segment by segment."""
        numels = [p.numel() for p in self.grads.params]
        if 'offsets' in sd:
            src_off, src_n = list(sd['offsets']), list(sd['numels'])
        else:                                              # layout 1: no padding between the segments
            src_n, src_off, n = numels, [], 0
            for k in numels:
                src_off.append(n)
                n += k
        if src_n != numels:
            raise ValueError('FlatAdam.load_state_dict: the saved state belongs to other parameters (%d tensors / %d elements, this '
                             'model has %d / %d)' % (len(src_n), sum(src_n), len(numels), sum(numels)))
        need = src_off[-1] + src_n[-1] if src_n else 0
        for name in ('exp_avg', 'exp_avg_sq'):
            if sd[name].numel() < need:
                raise ValueError('FlatAdam.load_state_dict: %s holds %d elements, its layout needs %d' % (name, sd[name].numel(), need))
        self.step_count = int(sd['step'])
        for name, dst in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
            src = sd[name].reshape(-1)
            if src_off == list(self.grads.offsets) and src.numel() == dst.numel():
                dst.copy_(src)
            else:
                dst.zero_()
                for so, do, k in zip(src_off, self.grads.offsets, numels):
                    dst[do:do + k].copy_(src[so:so + k])
        self.lr, self.betas, self.eps = float(sd['lr']), tuple(sd['betas']), float(sd['eps'])


class Trainer:
    """Drives the reference's alternation: odd iterations supervised, even iterations episode/L1."""

    def __init__(self, model, opt, lr=None, graph_encoder=False, graph_step=False):
        """graph_encoder: capture the image encoder's forward/backward as hipGraphs on the first step (fixed
        batch and image size from then on; other shapes run eagerly) -- see Actor.graph_image_encoder.
        graph_step: capture the WHOLE episode step behind the request encoder (all encoder passes, decoder steps,
        sampling, operators, L1, backward) as one hipGraph per (image shape, request length) --
        graphs.GraphedEpisodeStep; takes precedence over graph_encoder for the episode step."""
        self.model, self.opt = model, opt
        self.grads = FlatGradients(model.parameters())
        lr = lr if lr is not None else opt.learning_rate
        if self.grads.flat.is_cuda:
            self.optimizer = FlatAdam(self.grads, lr=lr)
        else:                                              # (host tensors: the multi-process CPU tests)
            self.optimizer = torch.optim.Adam(self.grads.params, lr=lr)
        self.itr = 0
        self.graph_encoder = graph_encoder
        self.graph_step = graph_step
        self._step_graphs = {}
        self.max_step_graphs = 4                            # distinct (shape, request length) keys kept; others run eagerly
        # transformed convolution weights of the encoder trunk live for a whole optimiser step (encoder.TrunkPlan)
        enc = getattr(model, 'vis_encoder', None)
        self._trunk = enc.trunk_plan() if (enc is not None and self.grads.flat.is_cuda and hasattr(enc, 'trunk_plan')) else None
        if self._trunk is not None:
            self._trunk.persistent_wt = True
        # the parameter heads add their gradients into the flat buffer inside their backward kernel (this trainer
        # zeroes it before every backward): 28 tensors x 5 decoder steps of autograd accumulation launches less
        executor = getattr(model, 'executor', None)
        if executor is not None and self.grads.flat.is_cuda:
            executor.__dict__['heads_grad_in_place'] = True
        # the decoder steps and feature heads of a train step record into ONE persistent tape (decoder_step.DecoderTape,
        # allocated for the first batch); their weight gradients are one product per weight over all steps, formed by
        # _flush_tape() after the backward pass
        self._tape_owner = model if (self.grads.flat.is_cuda and hasattr(model, 'decoder') and hasattr(model, 'bn1')) else None

    def close(self):
        """Release what the Trainer owns beyond tensors: the C-ABI communicator of the gradient all-reduce, if one was built."""
        self.grads.close()

    def _tape(self, B):
        """The persistent tape for batches of B rows (re-allocated when B changes; None for models without a decoder),
        INSTALLED on the model for the duration of one train step: _release() takes it off again, so that a grad-enabled
        forward + backward outside the Trainer (a custom loss, a gradient check, a stock optimiser) gets private one-slot
        tapes and ordinary autograd gradients."""
        model = self._tape_owner
        if model is None:
            return None
        tape = self.__dict__.get('_tape_obj')
        if tape is None or tape.B != B:
            from .decoder_step import DecoderTape
            dec = model.decoder
            tape = self._tape_obj = DecoderTape(B, dec.hidden_size, dec.word_vec_dim, dec.output_size, model.vis_encoder.fc.in_features,
                                                self.opt.decoder_max_len + 1, self.grads.flat.device, persistent=True)
        model.__dict__['_tape'] = model.decoder.__dict__['_tape'] = tape
        tape.begin()
        return tape

    def _flush_tape(self):
        model = self._tape_owner
        tape = model.__dict__.get('_tape') if model is not None else None
        if tape is not None:
            tape.flush(model)
        arena = self._trunk.__dict__.get('arena') if self._trunk is not None else None
        if arena is not None:
            arena.flush(self._trunk)

    def _release(self):
        """End of a train step (also after an exception inside it): tape and arena off the model."""
        model = self._tape_owner
        if model is not None:
            model.__dict__.pop('_tape', None)
            model.decoder.__dict__.pop('_tape', None)
        if self._trunk is not None:
            self._trunk.__dict__['arena'] = None

    MAX_ARENAS = 2      # batch shapes whose arenas are kept (a loader's full and last batch); least recently used goes first

    def _arena(self, img, passes):
        """The encoder's per-layer (x, dy) / (V, A dY A^T) arenas for this batch shape (encoder.WgradArena): the
        convolutions' weight gradients are then ONE launch per layer and train step (formed by _flush_tape) instead of one
        per encoder pass.  Not with per-pass encoder graphs (their backward graphs accumulate inside the graph).  ONE arena
        per (N, H, W), sized for the most passes a step can make (decoder_max_len + 1: the teacher-forced step with a full
        operator sequence) and used by steps that make fewer -- not one per pass count met (up to decoder_max_len
        arenas of P x 1.1 GB each at bs = 64, 256 x 256); at most MAX_ARENAS shapes are kept."""
        if self._trunk is None or self.graph_encoder or not img.is_cuda:
            return
        key = (img.shape[0], img.shape[2], img.shape[3])
        arenas = self.__dict__.setdefault('_arenas', {})
        arena = arenas.pop(key, None)                         # (re-inserted below: most recently used last)
        if passes <= 0 or not self._trunk.supported(img):
            # nothing to defer for this call.  NOT remembered (a cached None would have kept every later step of this
            # shape off the arena); an arena built earlier for the shape stays
            if arena is not None:
                arenas[key] = arena
            self._trunk.__dict__['arena'] = None
            return
        if arena is not None and arena.P < passes:
            arena = None                                      # grown on demand: rebuilt for the larger pass count
        if arena is None:
            from .encoder import WgradArena
            while len(arenas) >= self.MAX_ARENAS:
                arenas.pop(next(iter(arenas)))
            # sized for the passes ASKED for (the episode step: decoder_max_len; the teacher-forced step one more) -- training
            # that only ever runs episode steps no longer pays for a sixth pass (~20 % of a multi-GB arena at bs = 64); the
            # reference's alternation grows it once, on the first teacher-forced step.  (Arenas a captured step graph pinned
            # -- graphs.py `_keep` -- live as long as the graph does, beyond MAX_ARENAS: one per captured shape.)
            arena = WgradArena(self._trunk, key[0], key[1], key[2], passes, img.device)
        arenas[key] = arena
        self._trunk.__dict__['arena'] = arena
        arena.begin()

    def _maybe_graph(self, img):
        if self.graph_encoder and img.is_cuda and '_graphed_encoders' not in self.model.__dict__:
            self.model.train()
            try:
                # one slot per encoder call of either step: decoder_max_len for the episode, one more for the
                # teacher-forced step (its y holds decoder_max_len operators + END) -- no step mixes replayed and eager calls
                self.model.graph_image_encoder(img, self.opt.decoder_max_len + 1)
            except Exception as e:                     # noqa: BLE001 -- an optimisation only: run eagerly instead
                import warnings
                warnings.warn('image-encoder graph capture failed (%s: %s); continuing without it' % (type(e).__name__, e))
                self.graph_encoder = False
            # (the warm-up iterations neither touch .grad nor leave anything in the batch-norm running statistics)

    def _finish(self, loss):
        self.grads.zero()
        loss.backward()
        self._flush_tape()
        self._update()

    def _update(self):
        self.grads.all_reduce_mean()
        self.optimizer.step()
        if self._trunk is not None:
            self._trunk.weights_changed()

    def _request_length(self, x, lengths):
        """Length the request encoder truncates this batch to (lang_encoder.py:70-113: the batch maximum of the non-pad
        counts); from the host-side `lengths` when the caller has them -- no device synchronisation."""
        if not self.model.variable_lengths:
            return int(x.shape[1])
        if lengths is not None and not (torch.is_tensor(lengths) and lengths.is_cuda):
            return int(max(lengths)) if not torch.is_tensor(lengths) else int(lengths.max())
        return int((x != self.opt.null_id).sum(1).max())

    def _graphed_episode_step(self, x, img_x, target, reinforce_sample, lengths):
        """The step through graphs.GraphedEpisodeStep, or None when this batch cannot take it (capture failed, too many
        distinct shapes): the caller then runs the eager step."""
        model = self.model
        if not (img_x.is_cuda and model.training and self.opt.decoder_max_len > 0):
            return None
        if lengths is None:
            lengths = (x != self.opt.null_id).sum(1)
        L = self._request_length(x, lengths)
        key = (tuple(img_x.shape), tuple(x.shape), L, bool(reinforce_sample))     # (x: the loader's padded request width is a captured shape too)
        sg = self._step_graphs.get(key)
        if sg is None:
            if len(self._step_graphs) >= self.max_step_graphs:
                return None
            from .graphs import GraphedEpisodeStep
            try:
                sg = GraphedEpisodeStep(self, x, lengths, L, img_x, target, reinforce_sample)
            except Exception as e:                         # noqa: BLE001 -- an optimisation only: run eagerly instead
                import warnings
                warnings.warn('episode-step graph capture failed (%s: %s); continuing without it' % (type(e).__name__, e))
                self.graph_step = False
                return None
            self._step_graphs[key] = sg
            if self._trunk is not None:
                self._trunk.weights_changed()              # (the capture's transformed weights belong to the graph's pool)
        loss = sg.run(x, img_x, target)
        self._update()
        return loss

    def supervised_step(self, x, y, img_x, img_y, gt_params, lengths=None):
        """train_seq2seqL1.py:51-65: NLL (mean, no ignore_index) + MSE(sum)/count_nonzero."""
        step = int((y != self.opt.null_id).sum(1).max())
        self._maybe_graph(img_x)
        try:
            self._tape(img_x.shape[0])
            self._arena(img_x, step - 1)
            _, pred_params, logp = self.model.supervised_forward(x, y, img_x, img_y, gt_params, None, lengths)
            target = y[:, 1:step].contiguous().view(-1)
            op_loss = F.nll_loss(logp.reshape(-1, logp.shape[-1]), target)
            gt = gt_params[:, :step - 2]
            param_loss = F.mse_loss(pred_params, gt, reduction='sum') / ((gt != 0).sum())
            self._finish(op_loss + param_loss)
        finally:
            self._release()
        return op_loss.detach(), param_loss.detach()

    def episode_step(self, x, img_x, target, reinforce_sample=1, lengths=None):
        """train_seq2seqL1.py:74-88: free-running episode, L1 between the END image and the target."""
        if self.graph_step:
            loss = self._graphed_episode_step(x, img_x, target, reinforce_sample, lengths)
            if loss is not None:
                return loss
        self._maybe_graph(img_x)
        try:
            self._tape(img_x.shape[0])
            self._arena(img_x, self.opt.decoder_max_len)
            _, pred_imgs, pred_ops, _ = self.model.episode_forward(x, img_x, None, reinforce_sample, lengths, stack=False)
            loss = end_l1_loss(pred_imgs, pred_ops, self.opt.end_id, target)
            self._finish(loss)
        finally:
            self._release()
        return loss.detach()

    def step(self, batch):
        """batch = (img_x, img_y (B,6,3,H,W), x, y, gt_params) as the reference's loader yields."""
        self.itr += 1
        img_x, img_y, x, y, gt_params = batch
        if self.itr % 2 == 1:
            return self.supervised_step(x, y, img_x, img_y, gt_params)
        return self.episode_step(x, img_x, img_y[:, -1])
