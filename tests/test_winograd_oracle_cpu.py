"""The numpy restatement of the Winograd F(2x2,3x3) and F(4x4,3x3) passes (oracle/winograd_oracle.py) against torch's Conv2d(k=3,pad=1)
and its autograd on the CPU, fp64: the transform matrices, the tile / position layout and the border handling the HIP
entries are held to in tests/test_kernels_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import winograd_oracle as wo          # noqa: E402


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


@pytest.mark.parametrize('B,H,W,Ci,Co', [(1, 4, 4, 3, 5), (2, 7, 9, 4, 6), (1, 1, 1, 2, 3), (2, 6, 5, 8, 4), (1, 8, 12, 2, 2)])
@pytest.mark.parametrize('relu_in', [False, True])
@pytest.mark.parametrize('m', [2, 4])
def test_three_passes_equal_conv2d_and_its_autograd(B, H, W, Ci, Co, relu_in, m):
    g = torch.Generator().manual_seed(B * 100 + H * 10 + W)
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(Co, generator=g, dtype=torch.float64).requires_grad_(True)
    dy = torch.randn(B, Co, H, W, generator=g, dtype=torch.float64)
    xin = torch.relu(x) if relu_in else x
    y = F.conv2d(xin, w, b, padding=1)
    dxin, dw, db = torch.autograd.grad(y, (xin, w, b), dy)
    xn, dyn = nhwc(x.detach()), nhwc(dy)
    assert np.allclose(wo.conv_fwd(xn, w.detach().numpy(), b.detach().numpy(), relu_in, m=m), nhwc(y.detach()), atol=1e-11)
    assert np.allclose(wo.conv_dgrad(dyn, w.detach().numpy(), m=m), nhwc(dxin), atol=1e-11)
    dw_w, db_w = wo.conv_wgrad(xn, dyn, relu_in, m=m)
    assert np.allclose(dw_w, dw.numpy(), atol=1e-10) and np.allclose(db_w, db.numpy(), atol=1e-10)


def test_layout_of_a_transformed_tensor():
    """(16, tiles, C): position p = 4*xi + nu, tiles in (image, tile row, tile column) order; odd borders are zero-padded."""
    x = np.arange(2 * 3 * 5 * 1, dtype=np.float64).reshape(2, 3, 5, 1) + 1
    V = wo.input_transform(x)
    assert V.shape == (16, wo.tiles(2, 3, 5), 1) == (16, 2 * 2 * 3, 1)
    # position (1,1) of tile (image 1, row 0, column 2) = d[1][1] + d[1][2] + d[2][1] + d[2][2] of its patch: pixels
    # (0,4), (0,5: outside), (1,4), (1,5: outside)
    t = 1 * (2 * 3) + 0 * 3 + 2
    assert V[5, t, 0] == x[1, 0, 4, 0] + x[1, 1, 4, 0]
    dM = wo.outgrad_transform(x)
    assert dM[0, t, 0] == x[1, 0, 4, 0]                       # A dY A^T at (0,0) is the tile's first pixel
    assert dM[5, t, 0] == x[1, 0, 4, 0] + x[1, 1, 4, 0]       # ... at (1,1) the sum of its (in-image) pixels


def test_layout_of_an_f4_transformed_tensor():
    """m = 4: (36, tiles, C), position p = 6*xi + nu, tiles = B * ceil(H/4) * ceil(W/4); the patch of tile (i, j) starts at
    pixel (4i-1, 4j-1)."""
    x = np.arange(1 * 6 * 9 * 1, dtype=np.float64).reshape(1, 6, 9, 1) + 1
    V = wo.input_transform(x, m=4)
    assert V.shape == (36, wo.tiles(1, 6, 9, 4), 1) == (36, 2 * 3, 1)
    # position (5,5) = row 5 of B^T on both sides = 81/64 d1 - 45/16 d3 + d5 along each axis; tile (1, 2): patch rows 3..8,
    # columns 7..12 (pixels with row > 5 or column > 8 are outside)
    bt5 = np.array([0, 81 / 64, 0, -45 / 16, 0, 1.0])
    patch = np.zeros((6, 6))
    patch[:3, :2] = x[0, 3:6, 7:9, 0]
    assert np.isclose(V[35, 1 * 3 + 2, 0], bt5 @ patch @ bt5)
    dM = wo.outgrad_transform(x, m=4)
    t = 0 * 3 + 1                                               # tile (0, 1): pixels rows 0..3, columns 4..7
    assert dM[0, t, 0] == x[0, 0, 4, 0]                         # A dY A^T at (0,0) is the tile's first pixel (point 0)
    assert dM[35, t, 0] == x[0, 3, 7, 0]                        # ... at (5,5) its last one (the point at infinity)
    assert wo.bias_position(2) == 5


def test_literal_matrices_are_the_toom_cook_construction():
    """The matrices typed into the oracle (and, as formulas, into csrc/winograd.hip) equal the exact Toom-Cook construction
    over their interpolation points; every entry of the F(4x4,3x3) B^T and A^T is a dyadic rational, i.e. exact in fp32."""
    at, g, bt = wo.toom_cook(wo.F4_POINTS, 4)
    assert np.array_equal(at, wo._AT[4]) and np.allclose(g, wo._G[4], rtol=1e-15, atol=0) and np.array_equal(bt, wo._BT[4])
    # (the published F(2x2,3x3) matrices are the construction over 0, +-1, inf with the signs of two points flipped in pairs)
    for m, mats in ((2, wo.toom_cook((0, 1, -1), 2)), (2, (wo._AT[2], wo._G[2], wo._BT[2])), (4, (wo._AT[4], wo._G[4], wo._BT[4]))):
        at, g, bt = mats
        ident = np.einsum('ij,jk,jl->ikl', at, g, bt)                 # the defining identity: y_i = sum_k g_k d_{i+k}
        want = np.zeros_like(ident)
        for i in range(m):
            for k in range(3):
                want[i, k, i + k] = 1
        assert np.allclose(ident, want, atol=1e-14)
    for M in (wo._BT[4], wo._AT[4]):
        assert np.array_equal(M.astype(np.float32).astype(np.float64), M)
    # the textbook F(4x4,3x3) point set gives a valid algorithm too (it is what this one was compared against)
    at, g, bt = wo.toom_cook((0, 1, -1, 2, -2), 4)
    assert at[3].tolist() == [0, 1, -1, 8, -8, 1] and bt[0].tolist() == [4, 0, -5, 0, 1, 0]
