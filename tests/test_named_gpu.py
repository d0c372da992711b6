"""The entries that carry the names of SURVEY.md 8(b) (csrc/named.hip): one call per ATen op of the reference, each
against that op in torch on the CPU (fp32), through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope='module')
def L():
    from wesup_amd import _lib
    _lib.load()
    return _lib


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32))


def test_sp_stats(L):
    from wesup_amd import synth
    d = torch.device('cuda:0')
    B, H, W, g = 2, 48, 40, 5
    labs = np.stack([synth.voronoi_labels(b, H, W, g) for b in range(B)])
    masks = np.stack([synth.point_mask(b, labs[b], 0.5, 2, tie_every=2) for b in range(B)])
    K = g * g + 3
    area = torch.empty(B, K, dtype=torch.int32, device=d)
    counts = torch.empty(B, K, 2, dtype=torch.int32, device=d)
    status = torch.empty(B, dtype=torch.int32, device=d)
    labs_d, masks_d = torch.from_numpy(labs).to(d), torch.from_numpy(masks).to(d)      # keep them alive across the call
    L.call('wesup_sp_stats', P(labs_d), P(masks_d), B, H * W, 2, K, P(area), P(counts), P(status), S())
    for b in range(B):
        assert np.array_equal(area[b].cpu().numpy(), np.bincount(labs[b].ravel(), minlength=K))
        for c in range(2):
            want = np.bincount(labs[b].ravel(), weights=masks[b, c].ravel(), minlength=K).astype(np.int32)
            assert np.array_equal(counts[b, :, c].cpu().numpy(), want)
    assert status.tolist() == [0, 0]
    bad = labs_d.clone()
    bad[1, 0, 0] = K
    L.call('wesup_sp_stats', P(bad), None, B, H * W, 2, K, P(area), None, P(status), S())
    assert status.tolist() == [0, 1]


@pytest.mark.parametrize('Pn,Cin,Cout', [(4800, 64, 32), (3600, 512, 256), (14400, 128, 64)])
def test_conv1x1(L, Pn, Cin, Cout):
    d = torch.device('cuda:0')
    x, w, b = rnd(Pn, Cin, seed=1), rnd(Cout, Cin, seed=2, scale=Cin ** -0.5).requires_grad_(True), rnd(Cout, seed=3)
    xr = x.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr.t().reshape(1, Cin, Pn, 1), w.reshape(Cout, Cin, 1, 1), br).reshape(Cout, Pn).t()
    dy = rnd(Pn, Cout, seed=4)
    y_ref.backward(dy)
    nb = L.load().wesup_conv1x1_workspace_bytes(Pn, Cin, Cout)
    ws = torch.zeros(max(nb, 256), dtype=torch.uint8, device=d)
    xg, wg, dyg = x.to(d), w.detach().to(d), dy.to(d)
    y = torch.empty(Pn, Cout, device=d)
    bg = b.to(d)
    L.call('wesup_conv1x1_fwd', P(xg), P(wg), P(bg), P(y), Pn, Cin, Cout, P(ws), nb, S())
    assert rel_err(y, y_ref) < TOL
    dx = torch.ones(Pn, Cin, device=d)
    wt = wg.t().contiguous()
    L.call('wesup_conv1x1_dgrad', P(dyg), P(wt), P(dx), Pn, Cin, Cout, 1, P(ws), nb, S())
    assert rel_err(dx - 1.0, xr.grad) < TOL
    dw, db = torch.empty(Cout, Cin, device=d), torch.empty(Cout, device=d)
    L.call('wesup_conv1x1_wgrad', P(dyg), P(xg), P(dw), P(db), Pn, Cin, Cout, P(ws), nb, S())
    assert rel_err(dw, w.grad) < TOL and rel_err(db, br.grad) < TOL


def test_linear(L):
    d = torch.device('cuda:0')
    R, In, Out = 1152, 2112, 1024
    x0 = rnd(R, In, seed=1)
    xr = x0.clone().requires_grad_(True)
    w = rnd(Out, In, seed=2, scale=In ** -0.5).requires_grad_(True)
    b = rnd(Out, seed=3, scale=0.1).requires_grad_(True)
    h = F.relu(xr)                                      # the layer's input came out of a ReLU
    y_ref = F.relu(F.linear(h, w, b))
    dy = rnd(R, Out, seed=4) * (y_ref.detach() > 0)      # gradient behind this layer's own ReLU
    y_ref.backward(rnd(R, Out, seed=4))
    nb = L.load().wesup_linear_workspace_bytes(R, In, Out)
    ws = torch.zeros(max(nb, 256), dtype=torch.uint8, device=d)
    hg, wg = h.detach().to(d), w.detach().to(d)
    y = torch.empty(R, Out, device=d)
    bg = b.detach().to(d)
    L.call('wesup_linear_fwd', P(hg), P(wg), P(bg), P(y), R, In, Out, 1, P(ws), nb, S())
    assert rel_err(y, y_ref) < TOL
    dx, dw, db = torch.empty(R, In, device=d), torch.empty(Out, In, device=d), torch.empty(Out, device=d)
    dyg, wt, x0g = dy.to(d), wg.t().contiguous(), x0.to(d)
    L.call('wesup_linear_bwd', P(dyg), P(hg), P(wt), P(x0g), P(dx), P(dw), P(db), R, In, Out, P(ws), nb, S())
    assert rel_err(dw, w.grad) < TOL and rel_err(db, b.grad) < TOL and rel_err(dx, xr.grad) < TOL


def test_upsample_bilinear_ac(L):
    d = torch.device('cuda:0')
    B, h, w, H, W, C, ld, off = 2, 15, 12, 60, 47, 32, 48, 8
    s = rnd(B, C, h, w, seed=1).requires_grad_(True)
    ref = F.interpolate(s, (H, W), mode='bilinear', align_corners=True)
    dout = rnd(B, C, H, W, seed=2)
    ref.backward(dout)
    out = torch.zeros(B, H, W, ld, device=d)
    sg = s.detach().permute(0, 2, 3, 1).contiguous().to(d)
    L.call('wesup_upsample_bilinear_ac_fwd', P(sg), P(out), B, h, w, H, W, C, ld, off, S())
    assert rel_err(out[..., off:off + C].permute(0, 3, 1, 2), ref) < TOL and float(out[..., :off].abs().max()) == 0.0
    dfull = torch.zeros(B, H, W, ld, device=d)
    dfull[..., off:off + C] = dout.permute(0, 2, 3, 1).to(d)
    ds = torch.empty(B, h, w, C, device=d)
    L.call('wesup_upsample_bilinear_ac_bwd', P(dfull), P(ds), B, h, w, H, W, C, ld, off, S())
    assert rel_err(ds.permute(0, 3, 1, 2), s.grad) < TOL


@pytest.mark.parametrize('weights', [None, (3.0, 1.0)])
def test_softmax_ce(L, weights):
    d = torch.device('cuda:0')
    n, C = 300, 2
    z = (rnd(n, C, seed=1) * 4).requires_grad_(True)
    y = torch.from_numpy((np.random.RandomState(2).rand(n, C) < 0.35).astype(np.float32))
    cw = None if weights is None else torch.tensor(weights)
    p = torch.softmax(z, dim=1)
    pc = torch.clamp(p, 1e-7, 1 - 1e-7)
    ce = -y * torch.log(pc)
    if cw is not None:
        ce = ce * cw.unsqueeze(0)
    ref = ce.sum() / (y.sum(dim=1) > 0).sum().float()
    ref.backward()
    probs, out2, dz = torch.empty(n, C, device=d), torch.empty(4, device=d), torch.empty(n, C, device=d)
    cwg = None if cw is None else cw.to(d)
    zg, yg = z.detach().to(d), y.to(d)
    L.call('wesup_softmax_ce_fwd', P(zg), P(yg), P(cwg), ctypes.c_float(1e-7), P(probs), P(out2), n, C, S())
    assert rel_err(probs, p) < 1e-6 and abs(float(out2[2]) - float(ref)) <= 1e-5 * abs(float(ref))
    assert float(out2[1]) == float((y.sum(dim=1) > 0).sum())
    one = torch.ones(1, device=d)
    L.call('wesup_softmax_ce_bwd', P(probs), P(yg), P(cwg), P(out2), P(one), ctypes.c_float(1e-7), P(dz), n, C, S())
    assert rel_err(dz, z.grad) < TOL
