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


# planes of A dY A^T that carry a minus sign (k_wino_dy: -r1 of rows 0..2; row 3 = -d1 except its own -r1): k_wino_wgrad accumulates
# them un-negated and k_wgw_reduce flips the sign of their sums
NEGATED_PLANES = (3, 7, 11, 12, 13, 14)


def weight_gradient_onchip_form(x, dy, splits=3):
    """dw (Co,3,3,Ci) the way t2onet_amd/csrc/t2o_wino_wgrad.hip forms it (fp64; the structure, not the rounding):
      * the tile index is walked in STEPS of 8 tiles -- (image, tile row ty, segment sx of 8 tiles = 16 pixels) -- cut into `splits`
        contiguous ranges whose partial sums are added in order;
      * patch rows outside the image are zero rows; patch COLUMNS outside it (column 0 of the first tile at the left edge, column 3
        of the last tile at the right edge) are read from the neighbouring valid column and enter the column step multiplied by 0:
        tv[i][0] = tr[i][0] * mL - tr[i][2], tv[i][3] = tr[i][1] - tr[i][3] * mR;
      * A dY A^T is formed WITHOUT its sign flips -- rows (d0, d0 + d1, d0 - d1, d1), per row (r0, r0 + r1, r0 - r1, r1) -- and the
        sums of the planes in NEGATED_PLANES change sign at the end;
      * G^T dU G closes.
    x (N,H,W,Ci), dy (N,H,W,Co), H and W multiples of 16."""
    x, dy = x.astype(np.float64), dy.astype(np.float64)
    N, H, W, Ci = x.shape
    Co = dy.shape[3]
    assert H % 16 == 0 and W % 16 == 0
    TH, SEG = H // 2, W // 16
    steps = [(n, ty, sx) for n in range(N) for ty in range(TH) for sx in range(SEG)]
    parts = []
    for sp in range(splits):
        dU = np.zeros((16, Co, Ci))
        for (n, ty, sx) in steps[len(steps) * sp // splits:len(steps) * (sp + 1) // splits]:
            for tx in range(8):
                w0 = 16 * sx + 2 * tx - 1                      # column of patch column 0
                d = np.zeros((4, 4, Ci))
                edge_l, edge_r = (sx == 0 and tx == 0), (sx == SEG - 1 and tx == 7)
                for i in range(4):
                    h = 2 * ty - 1 + i
                    if not 0 <= h < H:
                        continue                                # (the zero block)
                    for j in range(4):
                        w = w0 + j
                        if j == 0 and edge_l:
                            w = w0 + 1                          # clamped: a valid neighbour, masked below
                        if j == 3 and edge_r:
                            w = w0 + 2
                        d[i, j] = x[n, h, w]
                m_l, m_r = (0.0 if edge_l else 1.0), (0.0 if edge_r else 1.0)
                tr = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])            # rows of B^T d: tr[r][j]
                tv = np.empty((4, 4, Ci))
                for i in range(4):
                    tv[i, 0] = tr[i, 0] * m_l - tr[i, 2]
                    tv[i, 1] = tr[i, 1] + tr[i, 2]
                    tv[i, 2] = tr[i, 2] - tr[i, 1]
                    tv[i, 3] = tr[i, 1] - tr[i, 3] * m_r
                g = dy[n, 2 * ty:2 * ty + 2, 16 * sx + 2 * tx:16 * sx + 2 * tx + 2]            # (2, 2, Co)
                rows = [g[0], g[0] + g[1], g[0] - g[1], g[1]]                                  # each (2, Co): columns 0, 1
                av = np.empty((4, 4, Co))
                for i in range(4):
                    av[i, 0], av[i, 1], av[i, 2], av[i, 3] = rows[i][0], rows[i][0] + rows[i][1], rows[i][0] - rows[i][1], rows[i][1]
                dU += np.einsum('xo,xc->xoc', av.reshape(16, Co), tv.reshape(16, Ci))
        parts.append(dU)
    total = parts[0].copy()
    for p in parts[1:]:
        total += p
    for xi in NEGATED_PLANES:
        total[xi] = -total[xi]
    return np.einsum('ai,aboc,bj->oijc', G, total.reshape(4, 4, Co, Ci), G)
