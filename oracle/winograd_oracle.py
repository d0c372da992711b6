"""CPU restatement (numpy, fp64 by default) of the Winograd F(2x2, 3x3) passes the HIP path uses for the wide conv layers.

TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the product path).  It pins, stage by stage, what the entries
wesup_winograd_input_transform / _outgrad_transform / _pack_weight / _output_transform / _filter_grad and the batched
products between them compute; the end result is Conv2d(k=3, pad=1) and its autograd (the reference's
torchvision VGG16 convs, models/wesup.py:199,279, models/base.py:207), which tests/test_winograd_oracle_cpu.py checks
against torch on the CPU.

Layouts follow the HIP side: activations NHWC (B,H,W,C); a transformed tensor is (16, tiles, C) with position
p = 4*xi + nu and tiles in (image, tile row, tile column) order, tiles = B * ceil(H/2) * ceil(W/2); filters (Co,Ci,3,3).
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def tiles(B, H, W):
    return B * ((H + 1) // 2) * ((W + 1) // 2)


def input_transform(x, relu=False, dtype=np.float64):
    """x (B,H,W,C) -> V (16, tiles, C):  B^T d B of every 4x4 patch (rows 2i-1..2i+2, zero outside the image)."""
    x = np.asarray(x, dtype=dtype)
    if relu:
        x = np.maximum(x, 0)
    B, H, W, C = x.shape
    Th, Tw = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((B, 2 * Th + 2, 2 * Tw + 2, C), dtype=dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    d = np.empty((B, Th, Tw, 4, 4, C), dtype=dtype)
    for r in range(4):
        for c in range(4):
            d[:, :, :, r, c] = xp[:, r:r + 2 * Th:2, c:c + 2 * Tw:2]
    bt = BT.astype(dtype)
    v = np.einsum('ar,bijrcC,dc->bijadC', bt, d, bt)
    return v.reshape(B * Th * Tw, 16, C).transpose(1, 0, 2).copy()


def outgrad_transform(dy, dtype=np.float64):
    """dy (B,H,W,C) -> dM (16, tiles, C):  A dY A^T of every 2x2 output tile (zero outside the image)."""
    dy = np.asarray(dy, dtype=dtype)
    B, H, W, C = dy.shape
    Th, Tw = (H + 1) // 2, (W + 1) // 2
    yp = np.zeros((B, 2 * Th, 2 * Tw, C), dtype=dtype)
    yp[:, :H, :W] = dy
    q = np.empty((B, Th, Tw, 2, 2, C), dtype=dtype)
    for r in range(2):
        for c in range(2):
            q[:, :, :, r, c] = yp[:, r::2, c::2]
    a = AT.T.astype(dtype)
    m = np.einsum('ar,bijrcC,dc->bijadC', a, q, a)
    return m.reshape(B * Th * Tw, 16, C).transpose(1, 0, 2).copy()


def pack_weight(w, dtype=np.float64):
    """w (Co,Ci,3,3) -> (u_fwd (16,Co,Ci) = G g G^T, u_dgrad (16,Ci,Co) = the same of the 180-degree rotated filter)."""
    w = np.asarray(w, dtype=dtype)
    g = G.astype(dtype)
    Co, Ci = w.shape[:2]
    uf = np.einsum('ar,oirc,dc->adoi', g, w, g).reshape(16, Co, Ci)
    ud = np.einsum('ar,oirc,dc->adio', g, w[:, :, ::-1, ::-1], g).reshape(16, Ci, Co)
    return uf, ud


def products_nt(V, U):
    """M_p = V_p . U_p^T for the 16 positions: (16,tiles,K) x (16,N,K) -> (16,tiles,N)."""
    return np.einsum('ptk,pnk->ptn', V, U)


def output_transform(M, B, H, W, bias=None, dtype=np.float64):
    """M (16, tiles, C) -> y (B,H,W,C) = A^T M A per tile (+ bias), outputs outside the image dropped."""
    M = np.asarray(M, dtype=dtype)
    C = M.shape[2]
    Th, Tw = (H + 1) // 2, (W + 1) // 2
    m = M.transpose(1, 0, 2).reshape(B, Th, Tw, 4, 4, C)
    at = AT.astype(dtype)
    y = np.einsum('ar,bijrcC,dc->bijadC', at, m, at)                   # (B,Th,Tw,2,2,C)
    y = y.transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * Th, 2 * Tw, C)[:, :H, :W]
    return y + (0 if bias is None else np.asarray(bias, dtype=dtype))


def filter_grad(dU, dtype=np.float64):
    """dU (16,Co,Ci) -> dw (Co,Ci,3,3) = G^T dU G."""
    dU = np.asarray(dU, dtype=dtype)
    Co, Ci = dU.shape[1:]
    g = G.astype(dtype)
    return np.einsum('ra,rcoi,cb->oiab', g, dU.reshape(4, 4, Co, Ci), g)


def conv_fwd(x, w, bias=None, relu_in=False, dtype=np.float64):
    uf, _ = pack_weight(w, dtype)
    B, H, W, _ = np.asarray(x).shape
    return output_transform(products_nt(input_transform(x, relu_in, dtype), uf), B, H, W, bias, dtype)


def conv_dgrad(dy, w, dtype=np.float64):
    _, ud = pack_weight(w, dtype)
    B, H, W, _ = np.asarray(dy).shape
    return output_transform(products_nt(input_transform(dy, False, dtype), ud), B, H, W, None, dtype)


def conv_wgrad(x, dy, relu_in=False, dtype=np.float64):
    """-> (dw (Co,Ci,3,3), db (Co)); db is the column sum of dM at position 5."""
    V = input_transform(x, relu_in, dtype)
    dM = outgrad_transform(dy, dtype)
    dU = np.einsum('pto,pti->poi', dM, V)
    return filter_grad(dU, dtype), dM[5].sum(axis=0)
