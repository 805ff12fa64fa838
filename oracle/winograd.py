"""Winograd F(2x2, 3x3) restated in numpy (TEST INFRASTRUCTURE ONLY: tests/ use it as the checker of the transform algebra
that t2onet_amd/csrc/t2o_winograd.hip implements; nothing in the product imports it).

The convolution is the reference encoder's 3x3, stride 1, padding 1 layer (models/actor_resnet.py:27-44:
nn.Conv2d(planes, planes, 3, 1, 1, bias=False)); fp64 throughout, NHWC like the kernels:
    V[xi][t][ci]  = (B^T d B)[xi]      d = the 4 x 4 input patch of output tile t (rows 2th-1 .. 2th+2, zero padded)
    U[xi][co][ci] = (G g G^T)[xi]
    M[xi]         = V[xi] @ U[xi]^T
    y tile        = A^T M A
and for the weight gradient  dU[xi] = (A dY A^T)[xi]^T @ V[xi],  dg = G^T dU G.
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def input_transform(x):
    """x (N,H,W,C) -> V (16, T, C), T = N * H/2 * W/2, tile order (n, th, tw)."""
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    TH, TW = H // 2, W // 2
    d = np.empty((N, TH, TW, 4, 4, C))
    for i in range(4):
        for j in range(4):
            d[:, :, :, i, j] = xp[:, i:i + 2 * TH:2, j:j + 2 * TW:2]
    v = np.einsum('ai,nhwijc,bj->abnhwc', BT, d, BT)
    return v.reshape(16, N * TH * TW, C)


def weight_transform(w):
    """w (Co,3,3,Ci) -> U (16, Co, Ci)."""
    return np.einsum('ai,oijc,bj->aboc', G, w.astype(np.float64), G).reshape(16, w.shape[0], w.shape[3])


def output_transform(M, N, H, W):
    """M (16, T, Co) -> y (N,H,W,Co)."""
    TH, TW = H // 2, W // 2
    Co = M.shape[2]
    m = M.reshape(4, 4, N, TH, TW, Co)
    yt = np.einsum('ia,abnhwc,jb->nhiwjc', AT, m, AT)           # (N, TH, 2, TW, 2, Co)
    return yt.reshape(N, H, W, Co)


def conv(x, w):
    """y = conv2d(x, w, stride 1, padding 1), NHWC / (Co,3,3,Ci), through the transforms."""
    N, H, W, _ = x.shape
    V, U = input_transform(x.astype(np.float64)), weight_transform(w)
    M = np.einsum('xtc,xoc->xto', V, U)
    return output_transform(M, N, H, W)


def dy_transform(dy):
    """dy (N,H,W,Co) -> Ad (16, T, Co) = A dY A^T of the 2 x 2 tiles."""
    N, H, W, C = dy.shape
    TH, TW = H // 2, W // 2
    t = dy.astype(np.float64).reshape(N, TH, 2, TW, 2, C)
    a = np.einsum('ia,nhiwjc,jb->abnhwc', AT, t, AT)
    return a.reshape(16, N * TH * TW, C)


def weight_gradient(x, dy):
    """dw (Co,3,3,Ci) of sum(conv(x, w) * dy)."""
    V, Ad = input_transform(x.astype(np.float64)), dy_transform(dy)
    dU = np.einsum('xto,xtc->xoc', Ad, V).reshape(4, 4, dy.shape[3], x.shape[3])
    return np.einsum('ai,aboc,bj->oijc', G, dU, G)


def direct_conv(x, w):
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((N, H, W, w.shape[0]))
    for kh in range(3):
        for kw in range(3):
            y += np.einsum('nhwc,oc->nhwo', xp[:, kh:kh + H, kw:kw + W], w[:, kh, kw].astype(np.float64))
    return y
