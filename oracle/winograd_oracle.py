"""CPU restatement (numpy, fp64 by default) of the Winograd F(m x m, 3x3) passes, m = 2 or 4, that the HIP path uses for
the wide conv layers.

TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the product path).  It pins, stage by stage, what the entries
wesup_winograd_input_transform / _outgrad_transform / _pack_weight / _output_transform / _filter_grad and the batched
products between them compute; the end result is Conv2d(k=3, pad=1) and its autograd (the reference's
torchvision VGG16 convs, models/wesup.py:199,279, models/base.py:207), which tests/test_winograd_oracle_cpu.py checks
against torch on the CPU.

Per m x m output tile:  Y = A^T [ (G g G^T) o (B^T d B) ] A  with d the (m+2) x (m+2) input patch whose first row / column
is one pixel above / left of the tile (pad 1).  F(2x2,3x3) uses the published minimal-filtering matrices (Lavin & Gray,
"Fast Algorithms for Convolutional Neural Networks", 2016; interpolation points 0, +-1, inf).  F(4x4,3x3) is the same
Toom-Cook construction over the points 0, +-3/4, +-3/2, inf instead of the textbook 0, +-1, +-2, inf: in fp32 its error is
~4x smaller (8e-7 instead of 3.5e-6 of the tensor's maximum at 256..512 channels; /tmp-style search over dyadic symmetric
point pairs), every entry of B^T and A^T is a dyadic rational (exact in fp32) and the +- pairs keep the
transforms cheap.  ``toom_cook(points, m)`` below derives (A^T, G, B^T) for any point set in exact rational arithmetic and
``tests/test_winograd_oracle_cpu.py`` checks the literal matrices against it.

Layouts follow the HIP side: activations NHWC (B,H,W,C); a transformed tensor is (n^2, tiles, C) with n = m + 2,
position p = n*xi + nu and tiles in (image, tile row, tile column) order, tiles = B * ceil(H/m) * ceil(W/m); filters
(Co,Ci,3,3).
"""
import numpy as np

_BT = {
    2: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64),
    4: np.array([[81 / 64, 0, -45 / 16, 0, 1, 0], [0, -27 / 16, -9 / 4, 3 / 4, 1, 0], [0, 27 / 16, -9 / 4, -3 / 4, 1, 0],
                 [0, -27 / 32, -9 / 16, 3 / 2, 1, 0], [0, 27 / 32, -9 / 16, -3 / 2, 1, 0], [0, 81 / 64, 0, -45 / 16, 0, 1]],
                dtype=np.float64),
}
_G = {
    2: np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64),
    4: np.array([[64 / 81, 0, 0], [-128 / 243, -32 / 81, -8 / 27], [-128 / 243, 32 / 81, -8 / 27], [32 / 243, 16 / 81, 8 / 27],
                 [32 / 243, -16 / 81, 8 / 27], [0, 0, 1]], dtype=np.float64),
}
_AT = {
    2: np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64),
    4: np.array([[1, 1, 1, 1, 1, 0], [0, 3 / 4, -3 / 4, 3 / 2, -3 / 2, 0], [0, 9 / 16, 9 / 16, 9 / 4, 9 / 4, 0],
                 [0, 27 / 64, -27 / 64, 27 / 8, -27 / 8, 1]], dtype=np.float64),
}
F4_POINTS = (0, (3, 4), (-3, 4), (3, 2), (-3, 2))      # + infinity
BT, G, AT = _BT[2], _G[2], _AT[2]           # the F(2x2,3x3) matrices under their round-2 names


def toom_cook(points, m, r=3):
    """(A^T (m x n), G (n x r), B^T (n x n)), n = m + r - 1, of the Toom-Cook / Winograd algorithm F(m, r) over the n - 1
    finite interpolation ``points`` (ints or (numerator, denominator) pairs) plus infinity, in exact rational arithmetic:
    A^T[i][j] = a_j^i, G[j][k] = a_j^k / prod_{l != j}(a_j - a_l), last column of A^T / last row of G the point at
    infinity, and B^T the unique solution of  sum_j A^T[i][j] G[j][k] B^T[j][l] = [l == i + k]  (the algorithm computes
    the correlation y_i = sum_k g_k d_{i+k})."""
    from fractions import Fraction as Fr
    n = m + r - 1
    a = [Fr(*p) if isinstance(p, tuple) else Fr(p) for p in points]
    assert len(a) == n - 1
    AT = [[a[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for l in range(n - 1):
            if l != j:
                N *= a[j] - a[l]
        G.append([a[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    rows = [(i, k) for i in range(m) for k in range(r)]
    BT = [[None] * n for _ in range(n)]
    for l in range(n):                                   # Gauss-Jordan on the consistent (m r) x n system of column l
        A = [[AT[i][j] * G[j][k] for j in range(n)] + [Fr(1 if l == i + k else 0)] for (i, k) in rows]
        rix = 0
        for c in range(n):
            p = next(q for q in range(rix, len(A)) if A[q][c] != 0)
            A[rix], A[p] = A[p], A[rix]
            A[rix] = [v / A[rix][c] for v in A[rix]]
            for q in range(len(A)):
                if q != rix and A[q][c] != 0:
                    f = A[q][c]
                    A[q] = [x - f * y for x, y in zip(A[q], A[rix])]
            rix += 1
        assert all(v == 0 for row in A[n:] for v in row), 'the point set does not give an F(m, r) algorithm'
        for j in range(n):
            BT[j][l] = A[j][n]
    f = lambda X: np.array([[float(v) for v in row] for row in X], dtype=np.float64)      # noqa: E731
    return f(AT), f(G), f(BT)


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
    """F(2x2,3x3) only: the position of dM whose column sum over the tiles is the bias gradient -- A dY A^T at (1,1) is the
    plain sum of the tile's gradients (the point 1: column 1 of A^T is all ones).  The F(4x4,3x3) point set has no point
    1; there the bias gradient is the column sum of dy itself (summed per block inside the outgrad transform)."""
    assert m == 2
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
    """-> (dw (Co,Ci,3,3), db (Co)); db is the column sum of dM at position (1,1) for m = 2, of dy for m = 4."""
    V = input_transform(x, relu_in, dtype, m)
    dM = outgrad_transform(dy, dtype, m)
    dU = np.einsum('pto,pti->poi', dM, V)
    db = dM[bias_position(m)].sum(axis=0) if m == 2 else np.asarray(dy, dtype=dtype).sum(axis=(0, 1, 2))
    return filter_grad(dU, dtype, m), db
