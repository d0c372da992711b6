"""CPU restatement (numpy, fp64 by default) of the Winograd F(m x m, 3x3) passes, m = 2 or 4, that the HIP path uses for
the wide conv layers.

TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the product path).  It pins, stage by stage, what the entries
wesup_winograd_input_transform / _outgrad_transform / _pack_weight / _output_transform / _filter_grad and the batched
products between them compute; the end result is Conv2d(k=3, pad=1) and its autograd (the reference's
torchvision VGG16 convs, models/wesup.py:199,279, models/base.py:207), which tests/test_winograd_oracle_cpu.py checks
against torch on the CPU.

Per m x m output tile:  Y = A^T [ (G g G^T) o (B^T d B) ] A  with d the (m+2) x (m+2) input patch whose first row / column
is one pixel above / left of the tile (pad 1).  The matrices are the published minimal-filtering ones (Lavin & Gray,
"Fast Algorithms for Convolutional Neural Networks", 2016: F(2x2,3x3) and F(4x4,3x3) with interpolation points
0, +-1, (+-2), inf).

Layouts follow the HIP side: activations NHWC (B,H,W,C); a transformed tensor is (n^2, tiles, C) with n = m + 2,
position p = n*xi + nu and tiles in (image, tile row, tile column) order, tiles = B * ceil(H/m) * ceil(W/m); filters
(Co,Ci,3,3).
"""
import numpy as np

_BT = {
    2: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64),
    4: np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                 [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64),
}
_G = {
    2: np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64),
    4: np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                 [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64),
}
_AT = {
    2: np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64),
    4: np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64),
}
BT, G, AT = _BT[2], _G[2], _AT[2]           # the F(2x2,3x3) matrices under their round-2 names


def tiles(B, H, W, m=2):
    return B * ((H + m - 1) // m) * ((W + m - 1) // m)


def _split(x, m, Th, Tw, n, step, lead):
    """(B, lead + step*Th + ..., ..., C) padded tensor -> (B,Th,Tw,n,n,C) patches at stride ``step``."""
    B, _, _, C = x.shape
    d = np.empty((B, Th, Tw, n, n, C), dtype=x.dtype)
    for r in range(n):
        for c in range(n):
            d[:, :, :, r, c] = x[:, r:r + step * Th:step, c:c + step * Tw:step]
    return d


def input_transform(x, relu=False, dtype=np.float64, m=2):
    """x (B,H,W,C) -> V (n^2, tiles, C):  B^T d B of every n x n patch (rows m*i-1 .. m*i+m, zero outside the image)."""
    x = np.asarray(x, dtype=dtype)
    if relu:
        x = np.maximum(x, 0)
    B, H, W, C = x.shape
    n = m + 2
    Th, Tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((B, m * Th + 2, m * Tw + 2, C), dtype=dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    d = _split(xp, m, Th, Tw, n, m, 1)
    bt = _BT[m].astype(dtype)
    v = np.einsum('ar,bijrcC,dc->bijadC', bt, d, bt)
    return v.reshape(B * Th * Tw, n * n, C).transpose(1, 0, 2).copy()


def outgrad_transform(dy, dtype=np.float64, m=2):
    """dy (B,H,W,C) -> dM (n^2, tiles, C):  A dY A^T of every m x m output tile (zero outside the image)."""
    dy = np.asarray(dy, dtype=dtype)
    B, H, W, C = dy.shape
    n = m + 2
    Th, Tw = (H + m - 1) // m, (W + m - 1) // m
    yp = np.zeros((B, m * Th, m * Tw, C), dtype=dtype)
    yp[:, :H, :W] = dy
    q = _split(yp, m, Th, Tw, m, m, 0)
    a = _AT[m].T.astype(dtype)
    mm = np.einsum('ar,bijrcC,dc->bijadC', a, q, a)
    return mm.reshape(B * Th * Tw, n * n, C).transpose(1, 0, 2).copy()


def pack_weight(w, dtype=np.float64, m=2):
    """w (Co,Ci,3,3) -> (u_fwd (n^2,Co,Ci) = G g G^T, u_dgrad (n^2,Ci,Co) = the same of the 180-degree rotated filter)."""
    w = np.asarray(w, dtype=dtype)
    g = _G[m].astype(dtype)
    n = m + 2
    Co, Ci = w.shape[:2]
    uf = np.einsum('ar,oirc,dc->adoi', g, w, g).reshape(n * n, Co, Ci)
    ud = np.einsum('ar,oirc,dc->adio', g, w[:, :, ::-1, ::-1], g).reshape(n * n, Ci, Co)
    return uf, ud


def products_nt(V, U):
    """M_p = V_p . U_p^T for every position: (P,tiles,K) x (P,N,K) -> (P,tiles,N)."""
    return np.einsum('ptk,pnk->ptn', V, U)


def output_transform(M, B, H, W, bias=None, dtype=np.float64, m=2):
    """M (n^2, tiles, C) -> y (B,H,W,C) = A^T M A per tile (+ bias), outputs outside the image dropped."""
    M = np.asarray(M, dtype=dtype)
    C = M.shape[2]
    n = m + 2
    Th, Tw = (H + m - 1) // m, (W + m - 1) // m
    mm = M.transpose(1, 0, 2).reshape(B, Th, Tw, n, n, C)
    at = _AT[m].astype(dtype)
    y = np.einsum('ar,bijrcC,dc->bijadC', at, mm, at)                   # (B,Th,Tw,m,m,C)
    y = y.transpose(0, 1, 3, 2, 4, 5).reshape(B, m * Th, m * Tw, C)[:, :H, :W]
    return y + (0 if bias is None else np.asarray(bias, dtype=dtype))


def filter_grad(dU, dtype=np.float64, m=2):
    """dU (n^2,Co,Ci) -> dw (Co,Ci,3,3) = G^T dU G."""
    dU = np.asarray(dU, dtype=dtype)
    Co, Ci = dU.shape[1:]
    n = m + 2
    g = _G[m].astype(dtype)
    return np.einsum('ra,rcoi,cb->oiab', g, dU.reshape(n, n, Co, Ci), g)


def bias_position(m=2):
    """The position of dM whose column sum over the tiles is the bias gradient: A dY A^T at (1,1) is the plain sum of the
    tile's gradients for both m (row 1 of A^T's transpose is all ones over the tile)."""
    return (m + 2) + 1


def conv_fwd(x, w, bias=None, relu_in=False, dtype=np.float64, m=2):
    uf, _ = pack_weight(w, dtype, m)
    B, H, W, _ = np.asarray(x).shape
    return output_transform(products_nt(input_transform(x, relu_in, dtype, m), uf), B, H, W, bias, dtype, m)


def conv_dgrad(dy, w, dtype=np.float64, m=2):
    _, ud = pack_weight(w, dtype, m)
    B, H, W, _ = np.asarray(dy).shape
    return output_transform(products_nt(input_transform(dy, False, dtype, m), ud), B, H, W, None, dtype, m)


def conv_wgrad(x, dy, relu_in=False, dtype=np.float64, m=2):
    """-> (dw (Co,Ci,3,3), db (Co)); db is the column sum of dM at position (1,1)."""
    V = input_transform(x, relu_in, dtype, m)
    dM = outgrad_transform(dy, dtype, m)
    dU = np.einsum('pto,pti->poi', dM, V)
    return filter_grad(dU, dtype, m), dM[bias_position(m)].sum(axis=0)
