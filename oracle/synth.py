"""Deterministic, formula-generated test inputs (TEST INFRASTRUCTURE ONLY).

Golden fixtures store only OUTPUTS; the inputs and weights that produced them
are regenerated from these integer-hash formulas, identically in the build
container (tools/gen_golden.py) and on the GPU box (tests/).  Pure numpy
uint64 arithmetic: no libm, no torch RNG, so values are bit-reproducible.
"""
import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    # splitmix64 finaliser
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed, lo=0.0, hi=1.0):
    """float32 tensor, values strictly inside (lo, hi), on a 2^-24 lattice
    offset by half a step (so no exact 0, 1 or other grid ties)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
        bits = _mix(_mix(idx)) >> np.uint64(40)                 # 24 bits
    u = (bits.astype(np.float64) + 0.5) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def integers(shape, seed, lo, hi):
    """int64 tensor in [lo, hi]."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
        bits = _mix(_mix(idx)) >> np.uint64(33)
    return torch.from_numpy((lo + (bits % np.uint64(hi - lo + 1)).astype(np.int64)).reshape(shape))


def images(B, H, W, seed):
    """(B,3,H,W) RGB in (0,1); tie-free with probability 1 (no r==g==b, no 0/1)."""
    return uniform((B, 3, H, W), seed)


def op_params(op_ind, B, seed, setting='mid'):
    """Per-operator parameter tensors (B, n).  'mid' = the executor-benchmark
    ranges of SURVEY.md 8(d); 'strong' drives the clamps; 'neg' the other sign."""
    n = [1, 1, 1, 24, 1, 8, 1, 1][op_ind]
    if op_ind in (3, 5):
        rng = {'mid': (0.5, 1.5), 'strong': (0.1, 3.0), 'neg': (0.9, 1.1)}[setting]
    elif op_ind == 6:
        rng = {'mid': (0.0, 1.0), 'strong': (1.0, 1.5), 'neg': (0.0, 0.2)}[setting]
    elif op_ind == 0:
        rng = {'mid': (-0.3, 0.3), 'strong': (0.5, 1.9), 'neg': (-1.5, -0.2)}[setting]
    elif op_ind == 2:
        rng = {'mid': (-0.3, 0.3), 'strong': (0.5, 3.0), 'neg': (-1.4, -0.1)}[setting]
    else:
        rng = {'mid': (-0.3, 0.3), 'strong': (0.5, 0.99), 'neg': (-0.99, -0.3)}[setting]
    return uniform((B, n), seed, *rng)


def masks(B, C, H, W, seed, soft=True):
    m = uniform((B, C, H, W), seed)
    return m if soft else (m > 0.5).float()


def requests(B, L, seed, vocab=918):
    """(B,L) int64 requests: [START, t_1..t_n, END, 0...] with n in 1..L-2,
    tokens in 4..vocab-1."""
    n = integers((B,), seed, 1, L - 2)
    toks = integers((B, L), seed + 1, 4, vocab - 1)
    x = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        k = int(n[b])
        x[b, 0] = 1
        x[b, 1:1 + k] = toks[b, :k]
        x[b, 1 + k] = 2
    return x


def op_targets(B, seed, n_ops=5):
    """(B,7) teacher sequences [START, n_ops distinct ops of {3,4,5,6,8,9}, END]."""
    pool = np.array([3, 4, 5, 6, 8, 9])
    keys = uniform((B, 6), seed).numpy()
    y = torch.zeros(B, n_ops + 2, dtype=torch.long)
    for b in range(B):
        y[b, 0] = 1
        y[b, 1:1 + n_ops] = torch.from_numpy(pool[np.argsort(keys[b])[:n_ops]])
        y[b, 1 + n_ops] = 2
    return y


def fill_state_dict(sd, seed=7):
    """Overwrite every tensor of a state_dict IN KEY ORDER with formula values
    (so no weight file is stored).  Scales keep activations O(1)."""
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        s = seed * 1000 + i
        if not torch.is_floating_point(v):
            out[k] = torch.zeros_like(v)
            continue
        if k.endswith('mask_spec') or k.endswith('mask_word'):
            out[k] = v.clone()
        elif k.endswith('running_var'):
            out[k] = uniform(tuple(v.shape), s, 0.5, 1.5)
        elif k.endswith('running_mean'):
            out[k] = uniform(tuple(v.shape), s, -0.1, 0.1)
        elif ('bn' in k.split('.')[-2] or 'shortcut.1' in k) and k.endswith('weight'):
            out[k] = uniform(tuple(v.shape), s, 0.5, 1.5)
        elif k.endswith('bias'):
            out[k] = uniform(tuple(v.shape), s, -0.05, 0.05)
        else:
            fan_in = int(np.prod(v.shape[1:])) if v.dim() > 1 else int(v.shape[0])
            a = (3.0 / max(fan_in, 1)) ** 0.5
            if 'embedding' in k:
                a = 0.5
            if k.endswith('out_linear.weight'):
                a *= 24.0          # decisive operator logits: argmax margins far above fp32 noise
            out[k] = uniform(tuple(v.shape), s, -a, a)
    return out
