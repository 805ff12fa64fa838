"""RGB <-> HSV specification owned by this build (TEST INFRASTRUCTURE ONLY).

The reference calls ``kornia.rgb_to_hsv`` / ``kornia.hsv_to_rgb``
(models/operators.py:278,282 brightness; :474,478 saturation).  kornia is a
third-party dependency listed unpinned in requirements.txt:5 and is absent from
/root/reference and from this image, so its arithmetic cannot be imported.  This
file restates kornia's published algorithm (the "gather" formulation used from
kornia 0.5 on) with the choices SURVEY.md section 8(c) fixes:

  * s = delta / (v + 1e-6)
  * hue returned in radians, h in [0, 2*pi]
  * hsv_to_rgb selects the sector by gather (no masked assignment), so a channel
    equal to exactly 1.0 has no value-collision behaviour

Inputs are (..., 3, H, W) float tensors.  All arithmetic is plain eager torch so
that, on CPU, every step rounds once in fp32 exactly as the reference would.
The HIP kernels follow this operation order step for step.
"""
import math

import torch

HSV_EPS = 1e-6
TWO_PI = 2.0 * math.pi


def rgb_to_hsv(image: torch.Tensor) -> torch.Tensor:
    maxc, arg = image.max(-3)
    minc = image.min(-3)[0]
    delta = maxc - minc
    v = maxc
    s = delta / (v + HSV_EPS)
    # avoid 0/0 in the hue only; s above already used the true delta
    dsafe = torch.where(delta == 0, torch.ones_like(delta), delta)
    r, g, b = image.unbind(-3)
    rc = maxc - r
    gc = maxc - g
    bc = maxc - b
    h_r = bc - gc
    h_g = (rc - bc) + 2.0 * dsafe
    h_b = (gc - rc) + 4.0 * dsafe
    h = torch.stack((h_r, h_g, h_b), dim=-3) / dsafe.unsqueeze(-3)
    h = torch.gather(h, -3, arg.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = TWO_PI * h
    return torch.stack((h, s, v), dim=-3)


def hsv_to_rgb(image: torch.Tensor) -> torch.Tensor:
    h = image[..., 0, :, :] / TWO_PI
    s = image[..., 1, :, :]
    v = image[..., 2, :, :]
    h6 = h * 6.0
    hi = torch.floor(h6) % 6
    f = (h6 % 6) - hi
    p = v * (1.0 - s)
    q = v * (1.0 - f * s)
    t = v * (1.0 - (1.0 - f) * s)
    hi = hi.long()
    idx = torch.stack((hi, hi + 6, hi + 12), dim=-3)
    #            sector: 0  1  2  3  4  5
    table = torch.stack((v, q, p, p, t, v,      # R
                         t, v, v, q, p, p,      # G
                         p, p, t, v, v, q),     # B
                        dim=-3)
    return torch.gather(table, -3, idx)
