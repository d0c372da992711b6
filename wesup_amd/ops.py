"""Tensor-level wrappers over the C ABI (include/wesup_hip.h).

Each wrapper checks on the HOST that operand shapes/dtypes/devices match what
the kernel and its grid assume (a faulting kernel can reset the GPU), then calls
the entry on torch's current stream.  torch is only used for device memory and
streams here.  No CPU fallback: a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib

RELU_IN, RELU_OUT, ACCUM, MASK = 1, 2, 4, 8
# Stream-K tail of the NT GEMM family (the tiles of a short last round cut along K over all block slots + a fix-up launch): a
# kernel ALONE on the GPU gains 6-7 % from it, inside the 3-stream training step the other streams fill a short round anyway and
# the fix-up launches sit on the chain (17.52 vs 17.34 ms, round 2).  The step never uses it; a single-stream caller (a tool, an
# inference experiment, the tests of the path) switches it on for its own calls with set_streamk(True).
STREAMK = False


def set_streamk(on):
    global STREAMK
    STREAMK = bool(on)


if hasattr(torch._C, '_cuda_getCurrentRawStream') and hasattr(torch._C, '_cuda_getDevice'):
    def _stream():
        # the raw handle of torch's current stream on the current device (torch.cuda.current_stream().cuda_stream builds a Stream
        # object per call: 2.5 us, ~500 times per training step -- a quarter of the host time of a step at batch 1).  The engine
        # checks once per forward that the current device is the model's.
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
else:                                                  # a torch build without the private accessors
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# TIMING-ONLY diagnostics (bench.py --diag-skip, results wrong): 'tout' leaves out the separate output transforms of the K = 512
# Winograd layers, 'tin' the input transforms of the forward -- what those launches cost the step
DIAG = set()


# Optional HIP-event timing of the kernel classes that are launched outside the engine's schedule (superpixel
# preprocessing, label propagation + loss, SGD): bench.py hands the engine's KernelTimer over; None = no events.
_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


def _tbegin(tag):
    return _timer.begin(tag) if _timer is not None else None


def _tend(tok, work):
    if tok is not None:
        _timer.end(tok, work)


def _chk(t, dtype=torch.float32, name='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.WesupHipError(f'{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)')
    if t.dtype != dtype:
        raise _lib.WesupHipError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.WesupHipError(f'{name}: expected a contiguous tensor')
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_ws_cache = {}
_ws_first = {}         # key -> size of the first request
_ws_retired = []       # buffers replaced by a growth: kept for a while (another stream's queued kernels may still use them)
ws_generation = 0      # bumped whenever a workspace is (re)allocated: a recorded step plan holds the old addresses


ws_headroom = 1        # 2 once the step runner has met a second shape (set_workspace_headroom): see workspace()


def set_workspace_headroom(factor):
    global ws_headroom
    ws_headroom = max(1, int(factor))


def workspace(nbytes, device, tag='default'):
    """Grow-only byte workspace per (device, tag), served exactly while a run has one shape (``ws_headroom`` 1: the benchmark
    configurations pay nothing extra).  Once the step runner has met a second shape it sets the headroom to 2: from then on a
    workspace that has to grow becomes twice what is asked for and, in the same go, every other workspace of the device at least
    twice its first size.  Every growth bumps ``ws_generation`` and with it drops every recorded step plan and every plan waiting
    for its twin; under multi-scale training (utils/data.py:98-101: shapes within a factor 1.8 of each other in area) exact-fit
    growth did that once per new largest shape and tag, spread over the first dozens of shapes -- a shape's first recording hardly
    ever met its twin (3.4 -> 2.2 recordings per shape over the 94-shape rotation; the cold line itself moves with the box, 262 ... 297 img/s)."""
    global ws_generation
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        regrow = buf is not None and ws_headroom > 1
        buf = None                                       # (the tag's own stream orders the release behind its queued kernels)
        _ws_cache.pop(key, None)
        _ws_first.setdefault(key, int(nbytes))
        buf = torch.empty(max((ws_headroom if regrow else 1) * int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        if regrow:
            for k2 in [k for k in _ws_cache if k[0] == key[0] and k != key]:
                if _ws_cache[k2].numel() < ws_headroom * _ws_first.get(k2, 0):
                    _ws_retired.append(_ws_cache.pop(k2))      # (ITS stream may still run kernels of this very walk on it)
                    _ws_cache[k2] = torch.empty(ws_headroom * _ws_first[k2], dtype=torch.uint8, device=device)
        del _ws_retired[:-64]
        ws_generation += 1
    return buf


def _nt_workspace(nbytes, device):
    """Stream-K workspace of the NT GEMM family: one per stream, because launches on different streams overlap."""
    if not nbytes:
        return None
    return workspace(nbytes, device, 'nt%x' % torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------- input pipeline
def augment(img_u8, mask_u8, params, n_classes=2, elastic=None):
    """img (B,H,W,3) uint8, mask (B,H,W) uint8 class index or None, params (B,12) fp32 -> img f32 (B,3,H,W) in [0,1],
    one-hot mask uint8 (B,C,H,W) or None.  elastic = (field (B,2,hc,wc) fp32, params (B,12) fp32, cell): the displacement
    field of ElasticTransform on a coarse grid (utils.data.elastic_field) for the images whose params row has on = 1."""
    _chk(img_u8, torch.uint8, 'img'); _chk(params, name='params')
    B, H, W, three = img_u8.shape
    assert three == 3 and params.shape == (B, 12)
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=img_u8.device)
    om = None
    if mask_u8 is not None:
        _chk(mask_u8, torch.uint8, 'mask'); assert mask_u8.shape == (B, H, W)
        om = torch.empty(B, n_classes, H, W, dtype=torch.uint8, device=img_u8.device)
    ef = ep = None
    hc = wc = cell = 0
    if elastic is not None:
        ef, ep, cell = elastic
        _chk(ef, name='elastic field'); _chk(ep, name='elastic params')
        hc, wc = ef.shape[2:]
        assert ef.shape == (B, 2, hc, wc) and ep.shape == (B, 12) and cell > 0 and hc * cell >= H and wc * cell >= W
    _lib.call('wesup_augment', _p(img_u8), _p(mask_u8), _p(params), _p(ef), _p(ep), hc, wc, int(cell), _p(out), _p(om), B, H, W,
              n_classes, _stream())
    return out, om


def appearance(img_u8, params):
    """img (B,H,W,3) uint8, params (B,8) fp32 {alpha, beta, hue, sat, val, clahe_clip, blur, 0} -> (B,H,W,3) uint8 after
    HueSaturationValue, RandomBrightnessContrast, CLAHE and Blur(3) in the reference's order."""
    _chk(img_u8, torch.uint8, 'img'); _chk(params, name='params')
    B, H, W, three = img_u8.shape
    assert three == 3 and params.shape == (B, 8)
    out = torch.empty_like(img_u8)
    nb = _lib.load().wesup_appearance_workspace_bytes(B, H, W)
    ws = workspace(nb, img_u8.device, 'appearance')
    _lib.call('wesup_appearance', _p(img_u8), _p(params), _p(out), B, H, W, _p(ws), nb, _stream())
    return out


# ---------------------------------------------------------------- packing
def pack_input(img, out=None):
    _chk(img, name='img')
    B, C, H, W = img.shape
    assert C == 3
    if out is None:
        out = torch.empty(B, H, W, 4, dtype=torch.float32, device=img.device)
    _lib.call('wesup_pack_input', _p(img), _p(out), B, H, W, _stream())
    return out


def conv3x3_kpad(Ci):
    return _lib.load().wesup_conv3x3_kpad(int(Ci))


def pack_conv3x3_weight(w, w_fwd=None, w_dgrad=None, need_dgrad=True, need_fwd=True):
    _chk(w, name='w')
    Co, Ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and (need_fwd or need_dgrad)
    if need_fwd and w_fwd is None:
        w_fwd = torch.empty(Co, conv3x3_kpad(Ci), dtype=torch.float32, device=w.device)
    if need_dgrad and w_dgrad is None:
        w_dgrad = torch.empty(Ci, 9 * Co, dtype=torch.float32, device=w.device)
    _lib.call('wesup_pack_conv3x3_weight', _p(w), _p(w_fwd) if need_fwd else None, _p(w_dgrad) if need_dgrad else None,
              Co, Ci, _stream())
    return w_fwd, w_dgrad


def transpose(a, out=None):
    _chk(a, name='a')
    r, c = a.shape
    if out is None:
        out = torch.empty(c, r, dtype=torch.float32, device=a.device)
    _lib.call('wesup_transpose', _p(a), _p(out), r, c, _stream())
    return out


TRANSPOSE_MAX = 40      # items per wesup_transpose_batched launch (csrc/spatial.hip)


def transpose_batched(pairs):
    """[(a (r, c), out (c, r)), ...]: every out = a^T, TRANSPOSE_MAX pairs per launch (a batch of 32 images at 480 x 480 hands over
    64 interpolation-pooling matrices: two launches)."""
    for a, out in pairs:
        _chk(a, name='a'); _chk(out, name='out')
        assert out.shape == (a.shape[1], a.shape[0])
    for i0 in range(0, len(pairs), TRANSPOSE_MAX):
        part = pairs[i0:i0 + TRANSPOSE_MAX]
        n = len(part)
        arr = (_lib.TransposeItem * n)()
        for i, (a, out) in enumerate(part):
            arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols = a.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1]
        _lib.call('wesup_transpose_batched', ctypes.cast(arr, ctypes.c_void_p), n, _stream())


def scale_rows_by_area(x, area):
    """x (B,Kmax,C) *= 1 / area (B,Kmax) per row, in place (0 for rows of area 0): the pre-scaled form of the side-branch gradient
    rows that conv3x3_dgrad_winograd_gather(area_new=None) gathers."""
    _chk(x, name='x'); _chk(area, torch.int32, 'area')
    assert x.dim() == 3 and area.shape == x.shape[:2]
    _lib.call('wesup_scale_rows_by_area', _p(x), _p(area), x.shape[0] * x.shape[1], x.shape[2], _stream())
    return x


def winograd_fused_min_blocks():
    """The process-wide rule behind wesup_winograd_fused_route: grids below this many blocks take the two-kernel route."""
    return int(_lib.load().wesup_winograd_set_fused_min_blocks(-1))


# ---------------------------------------------------------------- sync edges / step plans (csrc/plan.hip)
def sync_record(slot, stream=None):
    """Mark the work queued so far on ``stream`` (raw handle; None: torch's current stream) in event slot ``slot``."""
    _lib.call('wesup_sync_record', int(slot), _stream() if stream is None else ctypes.c_void_p(stream))


def sync_wait(slot, stream=None):
    """``stream`` waits for the latest mark of ``slot``."""
    _lib.call('wesup_sync_wait', int(slot), _stream() if stream is None else ctypes.c_void_p(stream))


# ---------------------------------------------------------------- conv 3x3
def conv3x3_fwd(x, w_fwd, bias, Cout, relu_in, out=None, out_relu=None, side=None):
    """side = (side_w (Cout/2, Cout), side_bias or None, side_out 2-D view with row stride >= Cout/2): the layer's 1x1
    side conv computed in the conv's epilogue (Cout 64 / 128 only)."""
    _chk(x, name='x'); _chk(w_fwd, name='w_fwd')
    B, H, W, Cin = x.shape
    assert w_fwd.shape == (Cout, conv3x3_kpad(Cin)), (w_fwd.shape, Cout, Cin)
    if bias is not None:
        _chk(bias, name='bias'); assert bias.numel() == Cout
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    assert out.shape == (B, H, W, Cout) and out.is_contiguous()
    if out_relu is not None:
        _chk(out_relu, name='out_relu'); assert out_relu.shape == out.shape
    if side is not None:
        sw, sbias, sout = side
        _chk(sw, name='side_w'); assert sw.shape == (Cout // 2, Cout)
        if sbias is not None:
            _chk(sbias, name='side_bias'); assert sbias.numel() == Cout // 2
        assert sout.dtype == torch.float32 and sout.is_cuda and sout.dim() == 2 and sout.stride(1) == 1
        assert sout.shape == (B * H * W, Cout // 2) and sout.stride(0) >= Cout // 2
        _lib.call('wesup_conv3x3_fwd_side', _p(x), _p(w_fwd), _p(bias), _p(out), _p(out_relu), _p(sw), _p(sbias), _p(sout),
                  sout.stride(0), B, H, W, Cin, Cout, int(relu_in), _stream())
        return out
    nb = _lib.load().wesup_conv3x3_workspace_bytes(B, H, W, Cin, Cout) if STREAMK else 0
    _lib.call('wesup_conv3x3_fwd', _p(x), _p(w_fwd), _p(bias), _p(out), _p(out_relu), B, H, W, Cin, Cout, int(relu_in),
              _p(_nt_workspace(nb, x.device)), nb, _stream())
    return out


def conv3x3_dgrad(dy, w_dgrad, Cin, mask_src=None, out=None, accumulate=False):
    _chk(dy, name='dy'); _chk(w_dgrad, name='w_dgrad')
    B, H, W, Cout = dy.shape
    assert w_dgrad.shape == (Cin, 9 * Cout)
    if mask_src is not None:
        _chk(mask_src, name='mask_src'); assert mask_src.shape == (B, H, W, Cin)
    if out is None:
        assert not accumulate
        out = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device)
    assert out.shape == (B, H, W, Cin) and out.is_contiguous()
    nb = _lib.load().wesup_conv3x3_workspace_bytes(B, H, W, Cout, Cin) if STREAMK else 0
    _lib.call('wesup_conv3x3_dgrad', _p(dy), _p(w_dgrad), _p(mask_src), _p(out), B, H, W, Cin, Cout, int(accumulate),
              _p(_nt_workspace(nb, dy.device)), nb, _stream())
    return out


def conv3x3_wgrad(x, dy, Ci, relu_in, dw=None, db=None, ws_tag='default'):
    _chk(x, name='x'); _chk(dy, name='dy')
    B, H, W, Cx = x.shape
    Cout = dy.shape[3]
    assert dy.shape[:3] == (B, H, W) and Cx == (4 if Ci == 3 else Ci)
    if dw is None:
        dw = torch.empty(Cout, Ci, 3, 3, dtype=torch.float32, device=x.device)
    if db is None:
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == Cout * Ci * 9 and db.numel() == Cout
    nb = _lib.load().wesup_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Cout)
    ws = workspace(nb, x.device, ws_tag)
    _lib.call('wesup_conv3x3_wgrad', _p(x), _p(dy), _p(dw), _p(db), B, H, W, Ci, Cout, int(relu_in), _p(ws), nb, _stream())
    return dw, db


def winograd_positions(m=2):
    """Positions of the F(m x m, 3x3) domain: (m+2)^2 = 16 for m = 2, 36 for m = 4."""
    assert m in (2, 4), m
    return (m + 2) * (m + 2)


def winograd_pack_weight(w, need_fwd=True, need_dgrad=True, u_fwd=None, u_dgrad=None, m=2):
    """w (Cout,Cin,3,3) -> (u_fwd (P,Cout,Cin), u_dgrad (P,Cin,Cout)): G g G^T per channel pair, the second from the
    rotated filter; P = (m+2)^2 positions of F(m x m, 3x3)."""
    _chk(w, name='w')
    Cout, Cin = w.shape[:2]
    P = winograd_positions(m)
    if need_fwd and u_fwd is None:
        u_fwd = torch.empty(P, Cout, Cin, dtype=torch.float32, device=w.device)
    if need_dgrad and u_dgrad is None:
        u_dgrad = torch.empty(P, Cin, Cout, dtype=torch.float32, device=w.device)
    for u, shape in ((u_fwd if need_fwd else None, (P, Cout, Cin)), (u_dgrad if need_dgrad else None, (P, Cin, Cout))):
        if u is not None:
            _chk(u, name='u'); assert u.shape == shape, (u.shape, shape)
    _lib.call('wesup_winograd_pack_weight', _p(w), _p(u_fwd if need_fwd else None),
              _p(u_dgrad if need_dgrad else None), Cout, Cin, m, _stream())
    return u_fwd if need_fwd else None, u_dgrad if need_dgrad else None


def winograd_pack_weights(items):
    """[(w (Cout,Cin,3,3), u_fwd (36,Cout,Cin) or None, u_dgrad (36,Cin,Cout) or None), ...]: the F(4x4,3x3) filters of
    several layers in one launch (at most 32 panels)."""
    n = len(items)
    arr = (_lib.WinoFilter * n)()
    for i, (w, uf, ud) in enumerate(items):
        _chk(w, name='w')
        Cout, Cin = w.shape[:2]
        for u, shape in ((uf, (36, Cout, Cin)), (ud, (36, Cin, Cout))):
            if u is not None:
                _chk(u, name='u'); assert u.shape == shape, (u.shape, shape)
        arr[i].w, arr[i].u_fwd, arr[i].u_dgrad = w.data_ptr(), (uf.data_ptr() if uf is not None else None), (ud.data_ptr() if ud is not None else None)
        arr[i].Cout, arr[i].Cin = Cout, Cin
    _lib.call('wesup_winograd_pack_weights', ctypes.cast(arr, ctypes.c_void_p), n, _stream())


def winograd_tiles(B, H, W, m=2):
    return B * ((H + m - 1) // m) * ((W + m - 1) // m)


def winograd_input_transform(x, relu=False, out=None, m=2):
    """x (B,H,W,C) -> V (P, tiles, C) = B^T d B of every (m+2) x (m+2) input patch."""
    _chk(x, name='x')
    B, H, W, C = x.shape
    T, P = winograd_tiles(B, H, W, m), winograd_positions(m)
    if out is None:
        out = torch.empty(P, T, C, dtype=torch.float32, device=x.device)
    assert out.shape == (P, T, C) and out.is_contiguous()
    _lib.call('wesup_winograd_input_transform', _p(x), _p(out), 0, B, H, W, C, int(relu), m, _stream())
    return out


def winograd_outgrad_transform(dy, out=None, m=2, db=None):
    """dy (B,H,W,C) -> dM (P, tiles, C) = A dY A^T of every m x m output tile.  db (C,), m = 4 only: also the bias gradient
    sum_pixels dy (m = 2 gets it from winograd_filter_grad: position (1,1) of dM)."""
    _chk(dy, name='dy')
    B, H, W, C = dy.shape
    T, P = winograd_tiles(B, H, W, m), winograd_positions(m)
    if out is None:
        out = torch.empty(P, T, C, dtype=torch.float32, device=dy.device)
    assert out.shape == (P, T, C) and out.is_contiguous()
    ws, nb = None, 0
    if db is not None:
        _chk(db, name='db'); assert db.numel() == C and m == 4
        nb = _lib.load().wesup_winograd_outgrad_workspace_bytes(B, H, W, C, m)
        ws = workspace(nb, dy.device, 'wino_outgrad')
    _lib.call('wesup_winograd_outgrad_transform', _p(dy), _p(out), _p(db), B, H, W, C, m, _p(ws), nb, _stream())
    return out


def winograd_output_transform(Mt, B, H, W, bias=None, mask_src=None, out=None, out_relu=None, out_pool=None, pool_relu=False,
                              accumulate=False, m=2):
    """Mt (P, tiles, C) -> y (B,H,W,C) = A^T M A + bias [masked by mask_src > 0] [+ old y]."""
    _chk(Mt, name='Mt')
    C = Mt.shape[2]
    assert Mt.shape == (winograd_positions(m), winograd_tiles(B, H, W, m), C)
    if out is None:
        assert not accumulate
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=Mt.device)
    assert out.shape == (B, H, W, C) and out.is_contiguous()
    for t, shape in ((out_relu, out.shape), (mask_src, out.shape), (out_pool, (B, H // 2, W // 2, C))):
        if t is not None:
            _chk(t, name='operand'); assert t.shape == shape, (t.shape, shape)
    _lib.call('wesup_winograd_output_transform', _p(Mt), 0, _p(bias), _p(mask_src), _p(out), _p(out_relu), _p(out_pool),
              int(pool_relu), B, H, W, C, int(accumulate), m, _stream())
    return out


def winograd_gemm_output_transform(V, U, B, H, W, bias=None, mask_src=None, out=None, out_pool=None, pool_relu=False,
                                   accumulate=False, unpool=None):
    """V (36, tiles, K) x U (36, N, K) -> y (B,H,W,N) = A^T (V_p . U_p^T) A + epilogue, in one kernel (F(4x4,3x3); K = 64 or
    128, N % 64 == 0).  unpool = (src, dst) (B,Hu,Wu,N): the max-pool backward as the epilogue instead of the store."""
    _chk(V, name='V'); _chk(U, name='U')
    P, T, K = V.shape
    N = U.shape[1]
    assert P == 36 and U.shape == (36, N, K) and T == winograd_tiles(B, H, W, 4)
    Hu = Wu = 0
    us = ud = None
    if unpool is not None:
        us, ud = unpool
        _chk(us, name='unpool_src'); _chk(ud, name='unpool_dst')
        _, Hu, Wu, _ = us.shape
        assert us.shape == (B, Hu, Wu, N) == ud.shape and (Hu // 2, Wu // 2) == (H, W) and out is None
    else:
        if out is None:
            assert not accumulate
            out = torch.empty(B, H, W, N, dtype=torch.float32, device=V.device)
        assert out.shape == (B, H, W, N) and out.is_contiguous()
    for t, shape in ((mask_src, (B, H, W, N)), (out_pool, (B, H // 2, W // 2, N))):
        if t is not None:
            _chk(t, name='operand'); assert t.shape == shape, (t.shape, shape)
    _lib.call('wesup_winograd_gemm_output_transform', _p(V), 0, _p(U), _p(bias), _p(mask_src), _p(out), _p(out_pool), int(pool_relu),
              _p(us), _p(ud), Hu, Wu, B, H, W, K, N, int(accumulate), _stream())
    return ud if unpool is not None else out


def gemm_nt_batched(A, Bw, out=None):
    """out[b] = A[b] @ Bw[b]^T for contiguous (nbatch, M, K) x (nbatch, N, K) -> (nbatch, M, N), one launch."""
    _chk(A, name='A'); _chk(Bw, name='B')
    nb, M, K = A.shape
    N = Bw.shape[1]
    assert Bw.shape == (nb, N, K)
    if out is None:
        out = torch.empty(nb, M, N, dtype=torch.float32, device=A.device)
    assert out.shape == (nb, M, N) and out.is_contiguous()
    _lib.call('wesup_gemm_nt_batched', _p(A), K, M * K, _p(Bw), K, N * K, _p(out), N, M * N, nb, M, N, K, _stream())
    return out


def gemm_nt_group(As, Bs, biases, outs):
    """outs[i] = As[i] @ Bs[i]^T + biases[i] for equally shaped 2-D operands that sit at a constant element stride from
    each other (row-strided views allowed: column slices of one buffer, parameters of one flat buffer): ONE launch.
    Returns False -- and launches nothing -- when the operands are not equally spaced; the caller then loops."""
    n = len(As)
    M, K = As[0].shape
    N = Bs[0].shape[0]

    def stride(ts):
        if any(t.shape != ts[0].shape or t.stride() != ts[0].stride() or t.dtype != torch.float32 or not t.is_cuda for t in ts):
            return None
        d = [(ts[i + 1].data_ptr() - ts[i].data_ptr()) for i in range(n - 1)]
        if any(v != d[0] for v in d) or d[0] % 16:
            return None
        return d[0] // 4
    sA, sB, sC = stride(As), stride(Bs), stride(outs)
    sb = 0 if biases is None else stride(biases)
    if n < 2 or None in (sA, sB, sC, sb) or K % 32 or min(sA, sB, sC) < 0:
        return False
    for t in (As[0], Bs[0], outs[0]):
        assert t.dim() == 2 and t.stride(1) == 1
    assert Bs[0].shape == (N, K) and outs[0].shape == (M, N)
    _lib.call('wesup_gemm_nt_batched_bias', _p(As[0]), As[0].stride(0), sA, _p(Bs[0]), Bs[0].stride(0), sB,
              _p(None if biases is None else biases[0]), sb, _p(outs[0]), outs[0].stride(0), sC, n, M, N, K, _stream())
    return True


def winograd_filter_grad(slabs, dw=None, db=None, m=2):
    """slabs (P, S, Cout*Cin + Cout): split-K partial products of the P transformed filter gradients, each followed by
    the column sums of its dM operand -> (dw (Cout,Cin,3,3) = G^T (sum over S) G, db (Cout) from position (1,1): m = 2
    only, pass None for m = 4 -- winograd_outgrad_transform(db=...) has it there).  Cout, Cin are taken from dw."""
    _chk(slabs, name='slabs'); _chk(dw, name='dw')
    Cout, Cin = dw.shape[:2]
    S = slabs.shape[1]
    assert slabs.shape == (winograd_positions(m), S, Cout * Cin + Cout) and slabs.is_contiguous()
    assert dw.shape == (Cout, Cin, 3, 3)
    if db is not None:
        _chk(db, name='db'); assert db.numel() == Cout and m == 2
    _lib.call('wesup_winograd_filter_grad', _p(slabs), slabs.stride(1), slabs.stride(0), S, _p(dw), _p(db), Cout, Cin, m,
              _stream())
    return dw, db


def _winograd_conv(inp, u, bias, mask_src, out, out_relu, v_keep, relu_in, accumulate, ws_tag, timer, out_pool=None,
                   pool_relu=False, m=2, relu_bits_out=None, pool_code_out=None, mask_bits=None, v_ready=False,
                   after_transform=None):
    """The three passes of a Winograd-domain conv (input transform, P batched NT GEMMs, output transform + epilogue) -- or, for
    the short products (K <= 256 channels), the input transform and ONE kernel for products + output transform.
    timer (optional, engine.KernelTimer-like): the GEMM and the transforms are bracketed as classes of their own.
    after_transform: called once, when the input transform has been queued (the engine queues the previous layer's side
    branch there)."""
    B, H, W, Cin = inp.shape
    Cout = u.shape[1]
    lib = _lib.load()
    nb = lib.wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cin, Cout, m)
    if not nb:
        raise _lib.WesupHipError(f'winograd conv: unsupported shape {(B, H, W, Cin, Cout, m)}')
    T, P = winograd_tiles(B, H, W, m), winograd_positions(m)
    ws = workspace(nb, inp.device, ws_tag)
    v_bytes = (P * T * Cin * 4 + 255) // 256 * 256
    V = v_keep if v_keep is not None else ws[:v_bytes]
    Mt = ws[v_bytes:]
    n_io = 1 + (out_relu is not None) + (mask_src is not None) + bool(accumulate)

    hook = [after_transform]      # called once, when the (first) input transform has been queued

    def t_in(b0, nb_, st):
        if not v_ready and 'tin' not in DIAG:
            _t_in(b0, nb_, st)
        if hook[0] is not None:
            hook[0]()
            hook[0] = None

    def _t_in(b0, nb_, st):
        tok = timer.begin('winograd_transform') if timer else None
        t0 = winograd_tiles(b0, H, W, m)
        if relu_bits_out is not None:      # the sign bits of the input ride along (the ReLU mask of the layer below, for its backward)
            _lib.call('wesup_winograd_input_transform_bits', _p(inp[b0:b0 + nb_]), ctypes.c_void_p(V.data_ptr() + 4 * t0 * Cin),
                      T * Cin, _p(relu_bits_out[b0:b0 + nb_]), nb_, H, W, Cin, int(relu_in), st)
        else:
            _lib.call('wesup_winograd_input_transform', _p(inp[b0:b0 + nb_]), ctypes.c_void_p(V.data_ptr() + 4 * t0 * Cin), T * Cin,
                      nb_, H, W, Cin, int(relu_in), m, st)
        if timer:       # bytes: read x, write the P / m^2-fold expansion (4x for m = 2, 2.25x for m = 4)
            timer.end(tok, 4.0 * (nb_ * H * W + P * winograd_tiles(nb_, H, W, m)) * Cin)

    def gemm(b0, nb_, st):
        tok = timer.begin('winograd_gemm') if timer else None
        t0, tn = winograd_tiles(b0, H, W, m), winograd_tiles(nb_, H, W, m)
        _lib.call('wesup_gemm_nt_batched', ctypes.c_void_p(V.data_ptr() + 4 * t0 * Cin), Cin, T * Cin, _p(u), Cin, Cout * Cin,
                  ctypes.c_void_p(Mt.data_ptr() + 4 * t0 * Cout), Cout, T * Cout, P, tn, Cout, Cin, st)
        if timer:
            timer.end(tok, 2.0 * P * tn * Cin * Cout)

    def t_out(b0, nb_, st):
        if 'tout' in DIAG:
            return
        tok = timer.begin('winograd_transform') if timer else None
        t0 = winograd_tiles(b0, H, W, m)
        sl = slice(b0, b0 + nb_)
        _lib.call('wesup_winograd_output_transform', ctypes.c_void_p(Mt.data_ptr() + 4 * t0 * Cout), T * Cout, _p(bias),
                  _p(None if mask_src is None else mask_src[sl]), _p(out[sl]), _p(None if out_relu is None else out_relu[sl]),
                  _p(None if out_pool is None else out_pool[sl]), int(pool_relu), nb_, H, W, Cout, int(accumulate), m, st)
        if timer:
            timer.end(tok, 4.0 * (P * winograd_tiles(nb_, H, W, m) + (n_io + (0.25 if out_pool is not None else 0)) * nb_ * H * W) * Cout)

    # (the compact forms -- sign bits in, pooling codes out -- exist on the one-kernel route only: a caller that hands them over
    #  has chosen it; otherwise the size of the grid decides, wesup_winograd_fused_route)
    compact = mask_bits is not None or pool_code_out is not None
    fused = 0 if out_relu is not None else (lib.wesup_winograd_fused_supported(Cin, Cout, m) if compact
                                            else lib.wesup_winograd_fused_route(Cin, Cout, m, T))
    masked = mask_src is not None or mask_bits is not None
    if relu_bits_out is not None:
        assert m == 4 and relu_bits_out.dtype == torch.uint8 and relu_bits_out.shape == (B, H, W, Cin // 4) and relu_bits_out.is_contiguous()
    if mask_bits is not None or pool_code_out is not None:       # compact mask in / pooling decisions out: one-kernel route only
        if not (fused == 2 or (fused == 1 and not masked and not accumulate)):
            raise _lib.WesupHipError(f'winograd conv {Cin} -> {Cout}: mask_bits / pool_code_out need the one-kernel product route')
        assert mask_bits is None or (mask_bits.dtype == torch.uint8 and mask_bits.shape == (B, H, W, Cout // 4) and mask_bits.is_contiguous())
        assert pool_code_out is None or (out_pool is not None and pool_code_out.dtype == torch.int16
                                         and pool_code_out.shape == (B, H // 2, W // 2, Cout // 4) and pool_code_out.is_contiguous())
    if fused == 2 or (fused == 1 and not masked and not accumulate):
        # short products (64 ... 256 channels): the batched products and the output transform in one kernel
        st = _stream()
        for b0, nb_ in ((0, B),):
            t_in(b0, nb_, st)
            tok = timer.begin('winograd_gemm') if timer else None
            t0, tn = winograd_tiles(b0, H, W, m), winograd_tiles(nb_, H, W, m)
            sl = slice(b0, b0 + nb_)
            if mask_bits is not None or pool_code_out is not None:
                _lib.call('wesup_winograd_gemm_output_transform_ex', ctypes.c_void_p(V.data_ptr() + 4 * t0 * Cin), T * Cin, _p(u),
                          _p(bias), _p(None if mask_src is None else mask_src[sl]), _p(None if mask_bits is None else mask_bits[sl]),
                          _p(out[sl]), _p(None if out_pool is None else out_pool[sl]), int(pool_relu),
                          _p(None if pool_code_out is None else pool_code_out[sl]), None, None, None, 0, 0, None, None, None, 0,
                          nb_, H, W, Cin, Cout, int(accumulate), st)
            else:
                _lib.call('wesup_winograd_gemm_output_transform', ctypes.c_void_p(V.data_ptr() + 4 * t0 * Cin), T * Cin, _p(u), _p(bias),
                          _p(None if mask_src is None else mask_src[sl]), _p(out[sl]), _p(None if out_pool is None else out_pool[sl]),
                          int(pool_relu), None, None, 0, 0, nb_, H, W, Cin, Cout, int(accumulate), st)
            if timer:
                timer.end(tok, 2.0 * P * tn * Cin * Cout)
        return out
    st = _stream()
    t_in(0, B, st); gemm(0, B, st); t_out(0, B, st)
    return out


def conv3x3_fwd_winograd(x, u_fwd, bias, relu_in, out=None, out_relu=None, v_keep=None, ws_tag='default', timer=None,
                         out_pool=None, pool_relu=False, m=2, relu_bits_out=None, pool_code_out=None, after_transform=None):
    """conv3x3_fwd through the Winograd F(m x m, 3x3) domain (deep layers); v_keep (P, tiles, Cin) receives the transformed
    input; out_pool (B, H//2, W//2, Cout) the 2x2 max-pool of the output (ReLU'd if pool_relu), written by the output
    transform."""
    _chk(x, name='x'); _chk(u_fwd, name='u_fwd')
    B, H, W, Cin = x.shape
    Cout = u_fwd.shape[1]
    assert u_fwd.shape == (winograd_positions(m), Cout, Cin)
    if bias is not None:
        _chk(bias, name='bias'); assert bias.numel() == Cout
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    assert out.shape == (B, H, W, Cout) and out.is_contiguous()
    if out_relu is not None:
        _chk(out_relu, name='out_relu'); assert out_relu.shape == out.shape
    if v_keep is not None:
        _chk(v_keep, name='v_keep'); assert v_keep.numel() == winograd_positions(m) * winograd_tiles(B, H, W, m) * Cin
    if out_pool is not None:
        _chk(out_pool, name='out_pool'); assert out_pool.shape == (B, H // 2, W // 2, Cout) and out_pool.is_contiguous()
    # relu_bits_out (B,H,W,Cin/4) uint8: the sign bits of x; pool_code_out (B,H/2,W/2,Cout/4) int16: the pooling's decisions
    # (include/wesup_hip.h: wesup_winograd_input_transform_bits, wesup_winograd_gemm_output_transform_ex)
    return _winograd_conv(x, u_fwd, bias, None, out, out_relu, v_keep, relu_in, False, ws_tag, timer, out_pool, pool_relu, m,
                          relu_bits_out=relu_bits_out, pool_code_out=pool_code_out, after_transform=after_transform)


def conv3x3_dgrad_winograd(dy, u_dgrad, mask_src=None, out=None, accumulate=False, ws_tag='default', timer=None, m=2,
                           mask_bits=None, v_pre=None):
    _chk(dy, name='dy'); _chk(u_dgrad, name='u_dgrad')
    B, H, W, Cout = dy.shape
    Cin = u_dgrad.shape[1]
    assert u_dgrad.shape == (winograd_positions(m), Cin, Cout)
    if mask_src is not None:
        _chk(mask_src, name='mask_src'); assert mask_src.shape == (B, H, W, Cin)
    if out is None:
        assert not accumulate
        out = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device)
    assert out.shape == (B, H, W, Cin) and out.is_contiguous()
    # v_pre (P,tiles,Cout): dy's input transform, done already (winograd_dual_transform): only the products and the way back
    if v_pre is not None:
        _chk(v_pre, name='v_pre'); assert v_pre.shape == (winograd_positions(m), winograd_tiles(B, H, W, m), Cout)
    return _winograd_conv(dy, u_dgrad, None, mask_src, out, None, v_pre, False, accumulate, ws_tag, timer, m=m, mask_bits=mask_bits,
                          v_ready=v_pre is not None)


def conv3x3_dgrad_winograd_unpool(dy, u_dgrad, unpool_src, unpool_dst, ws_tag='default', timer=None, m=4, unpool_code=None,
                                  v_pre=None):
    """Input gradient of a layer that follows a 2x2 max-pool, added straight into the gradient of the PRE-pool activations:
    dy (B,H,W,Cout) at pooled resolution, unpool_src / unpool_dst (B,Hu,Wu,Cin) with H == Hu // 2, W == Wu // 2.  Equals
    conv3x3_dgrad_winograd(out=dxp) followed by maxpool2_bwd(unpool_src, dxp, unpool_dst, accumulate=True)."""
    _chk(dy, name='dy'); _chk(u_dgrad, name='u_dgrad'); _chk(unpool_dst, name='unpool_dst')
    B, H, W, Cout = dy.shape
    Cin = u_dgrad.shape[1]
    assert m == 4 and u_dgrad.shape == (winograd_positions(m), Cin, Cout)
    _, Hu, Wu, _ = unpool_dst.shape
    assert unpool_dst.shape == (B, Hu, Wu, Cin) and (Hu // 2, Wu // 2) == (H, W)
    lib = _lib.load()
    route = lib.wesup_winograd_fused_route(Cout, Cin, m, winograd_tiles(B, H, W, m))
    if unpool_code is not None or (v_pre is not None and route == 2):
        # the pooling's decisions as codes (conv3x3_fwd_winograd(pool_code_out=...)): no read of unpool_src; and / or the input
        # transform of dy done already (winograd_dual_transform).  One-kernel product route only.
        if unpool_code is not None:
            assert unpool_code.dtype == torch.int16 and unpool_code.shape == (B, H, W, Cin // 4) and unpool_code.is_contiguous()
        else:
            _chk(unpool_src, name='unpool_src'); assert unpool_src.shape == unpool_dst.shape
        if lib.wesup_winograd_fused_supported(Cout, Cin, m) != 2:
            raise _lib.WesupHipError(f'winograd dgrad {Cout} -> {Cin}: unpool_code needs the one-kernel product route')
        T, P = winograd_tiles(B, H, W, m), winograd_positions(m)
        st = _stream()
        if v_pre is None:
            nb = lib.wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cout, Cin, m)
            ws = workspace(nb, dy.device, ws_tag)
            tok = timer.begin('winograd_transform') if timer else None
            _lib.call('wesup_winograd_input_transform', _p(dy), _p(ws), 0, B, H, W, Cout, 0, m, st)
            if timer:
                timer.end(tok, 4.0 * (B * H * W + P * T) * Cout)
        else:
            _chk(v_pre, name='v_pre'); assert v_pre.shape == (P, T, Cout)
            ws = v_pre
        tok = timer.begin('winograd_gemm') if timer else None
        _lib.call('wesup_winograd_gemm_output_transform_ex', _p(ws), 0, _p(u_dgrad), None, None, None, None, None, 0, None,
                  _p(None if unpool_code is not None else unpool_src),
                  _p(unpool_code), _p(unpool_dst), Hu, Wu, None, None, None, 0, B, H, W, Cout, Cin, 0, st)
        if timer:
            timer.end(tok, 2.0 * P * T * Cin * Cout)
        return unpool_dst
    _chk(unpool_src, name='unpool_src')
    assert unpool_src.shape == unpool_dst.shape
    nb = lib.wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cout, Cin, m)
    if not nb:
        raise _lib.WesupHipError(f'winograd dgrad: unsupported shape {(B, H, W, Cout, Cin, m)}')
    ws = workspace(nb, dy.device, ws_tag)
    if timer is None and v_pre is None:
        _lib.call('wesup_conv3x3_dgrad_winograd_unpool', _p(dy), _p(u_dgrad), _p(unpool_src), _p(unpool_dst), B, H, W, Hu, Wu,
                  Cin, Cout, m, _p(ws), nb, _stream())
        return unpool_dst
    # the three passes bracketed as classes of their own, as in _winograd_conv
    class _NoTimer:
        def begin(self, tag): return None
        def end(self, tok, work): pass
    timer = timer or _NoTimer()
    T, P = winograd_tiles(B, H, W, m), winograd_positions(m)
    v_bytes = (P * T * Cout * 4 + 255) // 256 * 256
    V, Mt = ws[:v_bytes], ws[v_bytes:]
    st = _stream()
    if v_pre is not None:      # dy's input transform, done already (winograd_dual_transform)
        _chk(v_pre, name='v_pre'); assert v_pre.shape == (P, T, Cout)
        V = v_pre
    else:
        tok = timer.begin('winograd_transform')
        _lib.call('wesup_winograd_input_transform', _p(dy), _p(V), 0, B, H, W, Cout, 0, m, st)
        timer.end(tok, 4.0 * (B * H * W + P * T) * Cout)
    if route == 2:
        tok = timer.begin('winograd_gemm')
        _lib.call('wesup_winograd_gemm_output_transform', _p(V), 0, _p(u_dgrad), None, None, None, None, 0, _p(unpool_src),
                  _p(unpool_dst), Hu, Wu, B, H, W, Cout, Cin, 0, st)
        timer.end(tok, 2.0 * P * T * Cin * Cout)
        return unpool_dst
    tok = timer.begin('winograd_gemm')
    _lib.call('wesup_gemm_nt_batched', _p(V), Cout, T * Cout, _p(u_dgrad), Cout, Cin * Cout, _p(Mt), Cin, T * Cin, P, T, Cin, Cout, st)
    timer.end(tok, 2.0 * P * T * Cin * Cout)
    tok = timer.begin('winograd_transform')
    _lib.call('wesup_winograd_output_transform_unpool', _p(Mt), 0, None, None, _p(unpool_src), _p(unpool_dst), B, H, W, Hu, Wu,
              Cin, m, st)
    # bytes: the transformed gradient in, the windows of the pre-pool activations, one position of four read and re-written
    timer.end(tok, 4.0 * (P * T + (4 + 2) * B * H * W) * Cin)
    return unpool_dst


def winograd_fused_supported(K, N, m=4, tiles=0):
    """0: the products of this shape go through the batched GEMM + output transform; 1 / 2: through the one-kernel route in the
    forward / in every pass (wesup_winograd_fused_supported).  tiles > 0: ... of a problem of that many tiles (0 when the
    one-kernel route's grid would be too small, wesup_winograd_fused_route)."""
    return int(_lib.load().wesup_winograd_fused_route(K, N, m, int(tiles)))


def conv3x3_dgrad_winograd_gather(dy, u_dgrad, side, new_row, area_new, out, mask_src=None, unpool_src=None,
                                  ws_tag='default', timer=None, mask_bits=None, unpool_code=None, v_pre=None):
    """conv3x3_dgrad_winograd(accumulate=True) / conv3x3_dgrad_winograd_unpool (unpool_src given) with out's old content
    replaced by the gather side[b][new_row[b][pixel]] / area_new[...] (side (B,Kmax,Cin): the commuted side-branch gradient of
    a native-resolution layer): out is written, never read.  m = 4, shapes of the one-kernel product route only."""
    for t, n in ((dy, 'dy'), (u_dgrad, 'u_dgrad'), (side, 'side'), (out, 'out')):
        _chk(t, name=n)
    _chk(new_row, torch.int32, 'new_row')
    if area_new is not None:       # None: the rows of side are divided by their areas already (scale_rows_by_area)
        _chk(area_new, torch.int32, 'area_new')
    B, H, W, Cout = dy.shape
    Cin = u_dgrad.shape[1]
    Kmax = side.shape[1]
    assert u_dgrad.shape == (36, Cin, Cout) and side.shape == (B, Kmax, Cin) and (area_new is None or area_new.shape == (B, Kmax))
    Hu = Wu = 0
    if unpool_src is not None or unpool_code is not None:
        _, Hu, Wu, _ = out.shape
        assert out.shape == (B, Hu, Wu, Cin) and (Hu // 2, Wu // 2) == (H, W) and Hu % 2 == 0 and Wu % 2 == 0
        assert mask_src is None and mask_bits is None and new_row.shape == (B, Hu * Wu)
        if unpool_code is not None:
            assert unpool_code.dtype == torch.int16 and unpool_code.shape == (B, H, W, Cin // 4) and unpool_code.is_contiguous()
            unpool_src = None
        else:
            _chk(unpool_src, name='unpool_src'); assert unpool_src.shape == out.shape
    else:
        assert out.shape == (B, H, W, Cin) and new_row.shape == (B, H * W)
        if mask_bits is not None:
            assert mask_bits.dtype == torch.uint8 and mask_bits.shape == (B, H, W, Cin // 4) and mask_bits.is_contiguous()
            mask_src = None
        elif mask_src is not None:
            _chk(mask_src, name='mask_src'); assert mask_src.shape == out.shape
    lib = _lib.load()
    if lib.wesup_winograd_fused_supported(Cout, Cin, 4) != 2:
        raise _lib.WesupHipError(f'conv3x3_dgrad_winograd_gather: product {Cout} -> {Cin} is not on the one-kernel route')
    nb = lib.wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cout, Cin, 4)
    ws = workspace(nb, dy.device, ws_tag)
    compact = mask_bits is not None or unpool_code is not None
    if v_pre is not None:      # dy's input transform, done already (winograd_dual_transform)
        _chk(v_pre, name='v_pre'); assert v_pre.shape == (36, winograd_tiles(B, H, W, 4), Cout)
    if timer is None and not compact and v_pre is None:
        _lib.call('wesup_conv3x3_dgrad_winograd_gather', _p(dy), _p(u_dgrad), _p(mask_src), _p(unpool_src), _p(out), _p(side),
                  _p(new_row), _p(area_new), Kmax, B, H, W, Hu, Wu, Cin, Cout, _p(ws), nb, _stream())
        return out
    T, P = winograd_tiles(B, H, W, 4), 36
    st = _stream()
    if v_pre is None:
        tok = timer.begin('winograd_transform') if timer else None
        _lib.call('wesup_winograd_input_transform', _p(dy), _p(ws), 0, B, H, W, Cout, 0, 4, st)
        if timer:
            timer.end(tok, 4.0 * (B * H * W + P * T) * Cout)
    else:
        ws = v_pre
    tok = timer.begin('winograd_gemm') if timer else None
    if compact:
        pooled = Hu > 0
        _lib.call('wesup_winograd_gemm_output_transform_ex', _p(ws), 0, _p(u_dgrad), None, _p(mask_src), _p(mask_bits),
                  _p(None if pooled else out), None, 0, None, _p(unpool_src), _p(unpool_code), _p(out if pooled else None), Hu, Wu,
                  _p(side), _p(new_row), _p(area_new), Kmax, B, H, W, Cout, Cin, 0, st)
    else:
        _lib.call('wesup_winograd_gemm_output_transform_gather', _p(ws), 0, _p(u_dgrad), _p(mask_src), _p(out), _p(unpool_src), Hu, Wu,
                  _p(side), _p(new_row), _p(area_new), Kmax, B, H, W, Cout, Cin, st)
    if timer:
        timer.end(tok, 2.0 * P * T * Cin * Cout)
    return out


def winograd_bias_rows(B, H, W, C):
    """Rows of per-block column sums the F(4x4) gradient transforms leave for the bias gradient (0: C is not covered)."""
    return int(_lib.load().wesup_winograd_bias_rows(B, H, W, C))


def winograd_dual_transform(dy, V, dM, bias_part=None):
    """One pass over dy (B,H,W,C) for both of its F(4x4) transforms: V (36,tiles,C) = B^T dY B (input of the layer's input
    gradient) and dM (36,tiles,C) = A dY A^T (operand of its weight gradient); bias_part (winograd_bias_rows, C): per-block
    column sums of dy for conv3x3_wgrad_winograd_pre."""
    _chk(dy, name='dy'); _chk(V, name='V'); _chk(dM, name='dM')
    B, H, W, C = dy.shape
    T = winograd_tiles(B, H, W, 4)
    assert V.shape == (36, T, C) == dM.shape
    if bias_part is not None:
        _chk(bias_part, name='bias_part'); assert bias_part.shape == (winograd_bias_rows(B, H, W, C), C) and bias_part.shape[0] > 0
    _lib.call('wesup_winograd_dual_transform', _p(dy), _p(V), _p(dM), _p(bias_part), B, H, W, C, _stream())
    return V, dM


def conv3x3_wgrad_winograd_pre(v_pre, dm_pre, bias_part, B, H, W, dw, db=None, ws_tag='default'):
    """The F(4x4) weight gradient from operands that exist: v_pre (36,tiles,Ci) kept by the forward, dm_pre (36,tiles,Cout) and
    bias_part from winograd_dual_transform.  dw (Cout,Ci,3,3); db (Cout) needs bias_part."""
    _chk(v_pre, name='v_pre'); _chk(dm_pre, name='dm_pre'); _chk(dw, name='dw')
    T = winograd_tiles(B, H, W, 4)
    Ci, Cout = v_pre.shape[2], dm_pre.shape[2]
    assert v_pre.shape == (36, T, Ci) and dm_pre.shape == (36, T, Cout) and dw.shape == (Cout, Ci, 3, 3)
    rows = 0
    if db is not None:
        _chk(db, name='db'); _chk(bias_part, name='bias_part')
        rows = bias_part.shape[0]
        assert db.numel() == Cout and bias_part.shape == (winograd_bias_rows(B, H, W, Cout), Cout)
    nb = _lib.load().wesup_conv3x3_wgrad_winograd_workspace_bytes(B, H, W, Ci, Cout, 4)
    ws = workspace(nb, v_pre.device, ws_tag)
    _lib.call('wesup_conv3x3_wgrad_winograd_pre', _p(v_pre), _p(dm_pre), _p(bias_part if db is not None else None), rows, _p(dw),
              _p(db), B, H, W, Ci, Cout, _p(ws), nb, _stream())
    return dw, db


def conv3x3_wgrad_winograd(x, dy, relu_in, dw=None, db=None, ws_tag='default', v_pre=None, m=2):
    """The same (dw, db) as conv3x3_wgrad through the Winograd F(m x m, 3x3) domain: 2.25x (m = 2) / 4x (m = 4) fewer
    multiply-adds, 4x / 2.25x the operand bytes; for the wide layers (Ci, Cout >= 128)."""
    _chk(x, name='x'); _chk(dy, name='dy')
    B, H, W, Ci = x.shape
    Cout = dy.shape[3]
    assert dy.shape[:3] == (B, H, W)
    if dw is None:
        dw = torch.empty(Cout, Ci, 3, 3, dtype=torch.float32, device=x.device)
    if db is None:
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == Cout * Ci * 9 and db.numel() == Cout
    nb = _lib.load().wesup_conv3x3_wgrad_winograd_workspace_bytes(B, H, W, Ci, Cout, m)
    if not nb:
        raise _lib.WesupHipError(f'conv3x3_wgrad_winograd: unsupported shape {(B, H, W, Ci, Cout, m)}')
    ws = workspace(nb, x.device, ws_tag)
    if v_pre is not None:
        _chk(v_pre, name='v_pre'); assert v_pre.numel() == winograd_positions(m) * winograd_tiles(B, H, W, m) * Ci
    _lib.call('wesup_conv3x3_wgrad_winograd', _p(x), _p(v_pre), _p(dy), _p(dw), _p(db), B, H, W, Ci, Cout, int(relu_in), m,
              _p(ws), nb, _stream())
    return dw, db


# ---------------------------------------------------------------- GEMMs
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def gemm_nt(A, Bw, bias=None, out=None, mask=None, flags=0, streamk=None):
    """out[M][N] = epi(A[M][K] @ Bw[N][K]^T + bias).  A/out/mask may be row-strided 2-D views.
    streamk: True / False overrides ops.STREAMK for this call."""
    for t, n in ((A, 'A'), (Bw, 'B')):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.WesupHipError(f'{n}: expected float32 CUDA/HIP tensor')
    M, K = A.shape
    N, K2 = Bw.shape
    assert K == K2 and K % 32 == 0, (A.shape, Bw.shape)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    assert out.shape == (M, N) and out.is_cuda
    ldmask = 0
    if mask is not None:
        assert mask.shape == (M, N) and mask.is_cuda
        ldmask = _ld(mask)
        flags |= MASK
    if bias is not None:
        _chk(bias, name='bias'); assert bias.numel() == N
    nb = _lib.load().wesup_gemm_nt_workspace_bytes(M, N, K) if (STREAMK if streamk is None else streamk) else 0
    _lib.call('wesup_gemm_nt', _p(A), _ld(A), _p(Bw), _ld(Bw), _p(bias), _p(out), _ld(out), _p(mask), ldmask, M, N, K,
              flags, _p(_nt_workspace(nb, A.device)), nb, _stream())
    return out


def gemm_tn(A, Bm, out=None, relu_b=False, ws_tag='default', colsum=None):
    """out[M][N] = A[K][M]^T @ Bm[K][N]  (deterministic split-K); colsum (M,), optional: also sum_k A[k][m]."""
    K, M = A.shape
    K2, N = Bm.shape
    assert K == K2 and A.is_cuda and Bm.is_cuda
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    assert out.shape == (M, N)
    nb = _lib.load().wesup_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(nb, A.device, ws_tag)      # one workspace per stream that may run concurrently
    if colsum is not None:
        _chk(colsum, name='colsum'); assert colsum.numel() == M
    _lib.call('wesup_gemm_tn', _p(A), _ld(A), _p(Bm), _ld(Bm), _p(out), _ld(out), _p(colsum), M, N, K, int(relu_b),
              _p(ws), nb, _stream())
    return out


def gemm_tn_batched(A, Bm, out, ws_tag='default'):
    """out[b] = A[b]^T @ Bm[b] for 3-D operands (nb,K,M), (nb,K,N) -> (nb,M,N); the last two dims may be row-strided
    views (unit stride along the last), the batch stride is free.  One launch + one reduce for all nb products."""
    nb, K, M = A.shape
    nb2, K2, N = Bm.shape
    assert nb == nb2 and K == K2 and out.shape == (nb, M, N)
    for t in (A, Bm, out):
        assert t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1
    nbytes = _lib.load().wesup_gemm_tn_batched_workspace_bytes(nb, M, N, K)
    ws = workspace(nbytes, A.device, ws_tag)
    _lib.call('wesup_gemm_tn_batched', _p(A), A.stride(1), A.stride(0), _p(Bm), Bm.stride(1), Bm.stride(0), _p(out),
              out.stride(1), out.stride(0), nb, M, N, K, 0, _p(ws), nbytes, _stream())
    return out


def colsum(A, out=None, ws_tag='colsum'):
    M, N = A.shape
    assert A.is_cuda
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=A.device)
    nb = _lib.load().wesup_colsum_workspace_bytes(M, N)
    ws = workspace(nb, A.device, ws_tag)
    _lib.call('wesup_colsum', _p(A), _ld(A), _p(out), M, N, _p(ws), nb, _stream())
    return out


# ---------------------------------------------------------------- pooling / upsampling
def maxpool2_fwd(y, out=None, relu=False):
    _chk(y, name='y')
    B, H, W, C = y.shape
    if out is None:
        out = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=y.device)
    assert out.shape == (B, H // 2, W // 2, C)
    _lib.call('wesup_maxpool2_fwd', _p(y), _p(out), B, H, W, C, int(relu), _stream())
    return out


def maxpool2_bwd(y, dyp, dy=None, accumulate=False):
    _chk(y, name='y'); _chk(dyp, name='dyp')
    B, H, W, C = y.shape
    assert dyp.shape == (B, H // 2, W // 2, C)
    if dy is None:
        assert not accumulate
        dy = torch.empty_like(y)
    assert dy.shape == y.shape and dy.is_contiguous()
    _lib.call('wesup_maxpool2_bwd', _p(y), _p(dyp), _p(dy), B, H, W, C, int(accumulate), _stream())
    return dy


def upsample_fwd(s, fm, coff):
    _chk(s, name='s'); _chk(fm, name='fm')
    B, h, w, C = s.shape
    B2, H, W, ldf = fm.shape
    assert B == B2 and coff + C <= ldf
    _lib.call('wesup_upsample_fwd', _p(s), _p(fm), B, h, w, H, W, C, ldf, coff, _stream())
    return fm


def upsample_bwd(dfm, coff, h, w, C, out=None):
    _chk(dfm, name='dfm')
    B, H, W, ldf = dfm.shape
    assert coff + C <= ldf
    if out is None:
        out = torch.empty(B, h, w, C, dtype=torch.float32, device=dfm.device)
    assert out.shape == (B, h, w, C)
    _lib.call('wesup_upsample_bwd', _p(dfm), None, None, _p(out), B, h, w, H, W, C, ldf, coff, 0, _stream())
    return out


def upsample_bwd_fused(g, new_row, area_new, H, W, coff, h, w, C, out=None):
    """pool-backward fused in: reads g[B][Kmax][ldf] through new_row instead of a materialised dfm."""
    _chk(g, name='g'); _chk(new_row, torch.int32, 'new_row'); _chk(area_new, torch.int32, 'area_new')
    B, Kmax, ldf = g.shape
    assert new_row.shape == (B, H * W) and area_new.shape == (B, Kmax) and coff + C <= ldf
    if out is None:
        out = torch.empty(B, h, w, C, dtype=torch.float32, device=g.device)
    assert out.shape == (B, h, w, C)
    _lib.call('wesup_upsample_bwd', _p(g), _p(new_row), _p(area_new), _p(out), B, h, w, H, W, C, ldf, coff, Kmax, _stream())
    return out


def upsample_bwd_fused_group(gs, new_row, area_new, H, W, h, w, outs):
    """upsample_bwd_fused for up to three layers of one coarse resolution in one launch: gs[i] (B,Kmax,C_i) dense ->
    outs[i] (B,h,w,C_i).  The window scan per coarse cell is shared by the layers."""
    n = len(gs)
    assert 1 <= n <= 3 and len(outs) == n and (h, w) != (H, W)
    B, Kmax = gs[0].shape[:2]
    _chk(new_row, torch.int32, 'new_row'); _chk(area_new, torch.int32, 'area_new')
    assert new_row.shape == (B, H * W) and area_new.shape == (B, Kmax)
    for g, o in zip(gs, outs):
        _chk(g, name='g'); _chk(o, name='out')
        assert g.is_contiguous() and o.is_contiguous() and g.shape[:2] == (B, Kmax) and o.shape == (B, h, w, g.shape[2])
    assert sum(g.shape[2] for g in gs) <= 768
    pg = [_p(g) for g in gs] + [_p(None)] * (3 - n)
    po = [_p(o) for o in outs] + [_p(None)] * (3 - n)
    cs = [g.shape[2] for g in gs] + [0] * (3 - n)
    _lib.call('wesup_upsample_bwd_group', *pg, *po, *cs, n, _p(new_row), _p(area_new), B, h, w, H, W, Kmax, _stream())
    return outs


# ---------------------------------------------------------------- superpixels
class SuperpixelMeta:
    """Device-side result of wesup_sp_preprocess for a batch of label maps (padded to Kmax rows per image)."""
    __slots__ = ('B', 'H', 'W', 'C', 'Kmax', 'labels', 'mask', 'n_sp', 'n_l', 'perm', 'inv_perm', 'area_new',
                 'sp_labels', 'new_row', 'row_start', 'pix_sorted', 'status', 'n_sp_host', 'seg_start', 'unit_row',
                 'Umax', 'counts', 'tiles')

    def check(self):
        """Host sync: raise on label-map errors the reference would turn into NaNs (models/wesup.py:57-61)."""
        st = self.status.cpu()
        if int(st.max()) != 0:
            bad = [(b, int(v)) for b, v in enumerate(st.tolist()) if v]
            raise ValueError(f'invalid label map (image, code): {bad}; 1 = id >= Kmax, 2 = empty id below the '
                             'maximum id (ids must be contiguous 0..K-1)')


def sp_preprocess(labels, mask, Kmax, n_classes=2, n_sp_host=None, into=None, counts=None):
    """labels (B,H,W) int32; mask (B,C,H,W) uint8 or None.  ``into``: a SuperpixelMeta of the same (B,H,W,C,Kmax) whose
    buffers are written again (a recorded step plan needs every buffer at a fixed address); ``counts``: an int32 (3,B) tensor
    that receives n_sp | n_l | status (the runner's read-back block)."""
    _chk(labels, torch.int32, 'labels')
    B, H, W = labels.shape
    HW = H * W
    dev = labels.device
    C = n_classes
    Kmax = int(Kmax)
    if mask is not None:
        _chk(mask, torch.uint8, 'mask')
        assert mask.shape == (B, C, H, W)
    m = into
    if m is None or (m.B, m.H, m.W, m.C, m.Kmax) != (B, H, W, C, Kmax) or m.n_sp.device != dev:
        m = SuperpixelMeta()
        m.B, m.H, m.W, m.C, m.Kmax = B, H, W, C, Kmax
        i32 = dict(dtype=torch.int32, device=dev)
        # n_sp | n_l | status side by side: the trainer reads the three back with one copy
        if counts is not None:
            assert counts.shape == (3, B) and counts.dtype == torch.int32 and counts.is_contiguous()
        m.counts = counts if counts is not None else torch.empty(3, B, **i32)
        m.n_sp, m.n_l, m.status = m.counts[0], m.counts[1], m.counts[2]
        m.perm = torch.empty(B, Kmax, **i32); m.inv_perm = torch.empty(B, Kmax, **i32)
        m.area_new = torch.empty(B, Kmax, **i32)
        m.sp_labels = torch.empty(B, Kmax, C, dtype=torch.float32, device=dev)
        m.new_row = torch.empty(B, HW, **i32); m.row_start = torch.empty(B, Kmax + 1, **i32)
        m.pix_sorted = torch.empty(B, HW, **i32)
        # segment table (rows cut into <= 512-pixel segments) for the load-balanced pooling kernels
        m.Umax = _lib.load().wesup_sp_max_units(HW, Kmax)
        m.seg_start = torch.empty(B, Kmax + 1, **i32)
        m.unit_row = torch.empty(B, m.Umax, **i32)
        m.tiles = None
    m.labels, m.mask, m.n_sp_host = labels, mask, n_sp_host
    nb = _lib.load().wesup_sp_preprocess_workspace_bytes(B, HW, C, Kmax)
    ws = workspace(nb, dev, 'sp')
    tok = _tbegin('sp_preprocess')
    _lib.call('wesup_sp_preprocess', _p(labels), _p(mask), B, HW, C, Kmax, _p(m.n_sp), _p(m.n_l), _p(m.perm),
              _p(m.inv_perm), _p(m.area_new), _p(m.sp_labels), _p(m.new_row), _p(m.row_start), _p(m.pix_sorted),
              _p(m.status), _p(m.seg_start), _p(m.unit_row), m.Umax, _p(ws), nb, _stream())
    # algorithmic bytes: the label map and the mask in, the row of every pixel and the row-sorted pixel list out
    _tend(tok, float(B) * HW * (4 + (C if mask is not None else 0) + 4 + 4))
    return m


def spmaps_to_labels(sp_maps):
    _chk(sp_maps, name='sp_maps')
    N, H, W = sp_maps.shape
    labels = torch.empty(1, H, W, dtype=torch.int32, device=sp_maps.device)
    _lib.call('wesup_spmaps_to_labels', _p(sp_maps), _p(labels), N, H * W, _stream())
    return labels


def sp_pool_fwd(fm, meta, C=None, out=None):
    _chk(fm, name='fm')
    B, H, W, ldf = fm.shape
    C = ldf if C is None else C
    assert (B, H, W) == (meta.B, meta.H, meta.W)
    if out is None:
        out = torch.empty(B, meta.Kmax, C, dtype=torch.float32, device=fm.device)
    assert out.shape == (B, meta.Kmax, C) and out.is_contiguous()
    nb = _lib.load().wesup_sp_pool_workspace_bytes(B, meta.Umax, C)
    ws = workspace(nb, fm.device, 'pool')
    _lib.call('wesup_sp_pool_fwd', _p(fm), _p(meta.pix_sorted), _p(meta.row_start), _p(meta.seg_start), _p(meta.unit_row),
              _p(out), B, H * W, ldf, C, meta.Kmax, meta.Umax, _p(ws), nb, _stream())
    return out


def sp_pool_upsample_fwd(s, meta, out, coff):
    """Fused upsample + scatter-mean of one side output s (B,h,w,C) into out[..., coff:coff+C] (out: (B,Kmax,ldo))."""
    _chk(s, name='s'); _chk(out, name='out')
    B, h, w, C = s.shape
    assert out.shape[:2] == (B, meta.Kmax) and B == meta.B and coff + C <= out.shape[2]
    nb = _lib.load().wesup_sp_pool_workspace_bytes(B, meta.Umax, C)
    ws = workspace(nb, s.device, 'pool_up')
    _lib.call('wesup_sp_pool_upsample_fwd', _p(s), _p(meta.pix_sorted), _p(meta.row_start), _p(meta.seg_start),
              _p(meta.unit_row), _p(out), B, h, w, meta.H, meta.W, C, out.shape[2], coff, meta.Kmax, meta.Umax, _p(ws), nb,
              _stream())
    return out


def sp_interp_matrix(meta, h, w, out=None):
    """Wm (B,Kmax,h*w): upsample-to-(H,W)-then-average-over-superpixel as a matrix over the coarse cells."""
    B, Kmax = meta.B, meta.Kmax
    assert 0 < h <= meta.H and 0 < w <= meta.W and h * w <= 8192
    if out is None:
        out = torch.empty(B, Kmax, h * w, dtype=torch.float32, device=meta.pix_sorted.device)
    _chk(out, name='out')
    assert out.shape == (B, Kmax, h * w)
    _lib.call('wesup_sp_interp_matrix', _p(meta.pix_sorted), _p(meta.row_start), _p(out), B, meta.H, meta.W, h, w, Kmax,
              _stream())
    return out


def sp_pool_bwd(g, meta, out=None):
    _chk(g, name='g')
    B, Kmax, C = g.shape
    assert B == meta.B and Kmax == meta.Kmax
    if out is None:
        out = torch.empty(B, meta.H, meta.W, C, dtype=torch.float32, device=g.device)
    assert out.shape == (B, meta.H, meta.W, C) and out.is_contiguous()
    _lib.call('wesup_sp_pool_bwd', _p(g), _p(meta.new_row), _p(meta.area_new), _p(out), B, meta.H * meta.W, C, C, Kmax,
              _stream())
    return out


def paint_fwd(sp_pred, meta, cls=1, out=None):
    _chk(sp_pred, name='sp_pred')
    B, Kmax, C = sp_pred.shape
    assert B == meta.B and Kmax == meta.Kmax
    if out is None:
        out = torch.empty(B, meta.H, meta.W, dtype=torch.float32, device=sp_pred.device)
    tok = _tbegin('paint')
    _lib.call('wesup_paint_fwd', _p(sp_pred), _p(meta.new_row), _p(out), B, meta.H * meta.W, Kmax, C, cls, _stream())
    _tend(tok, 8.0 * B * meta.H * meta.W + 4.0 * B * Kmax * C)
    return out


# ---------------------------------------------------------------- SLIC
def slic(img, n_segments, compactness=40.0, max_iter=10, enforce_connectivity=True, min_size_factor=0.5):
    """GPU SLIC: img (B,3,H,W) RGB in [0,1] -> (labels (B,H,W) int32 with contiguous ids, n_labels (B,) int32)."""
    _chk(img, name='img')
    B, C, H, W = img.shape
    assert C == 3
    labels = torch.empty(B, H, W, dtype=torch.int32, device=img.device)
    n_labels = torch.empty(B, dtype=torch.int32, device=img.device)
    nb = _lib.load().wesup_slic_workspace_bytes(B, H, W, int(n_segments))
    ws = workspace(nb, img.device, 'slic')
    _lib.call('wesup_slic', _p(img), _p(labels), _p(n_labels), B, H, W, int(n_segments), float(compactness), int(max_iter),
              int(enforce_connectivity), float(min_size_factor), _p(ws), nb, _stream())
    return labels, n_labels


# ---------------------------------------------------------------- head / loss / optimiser
def classifier_fwd(feat, Wc, bc, out=None):
    _chk(feat, name='feat'); _chk(Wc, name='Wc'); _chk(bc, name='bc')
    R, D = feat.shape
    assert Wc.shape == (2, D) and bc.numel() == 2
    if out is None:
        out = torch.empty(R, 2, dtype=torch.float32, device=feat.device)
    _lib.call('wesup_classifier_fwd', _p(feat), _p(Wc), _p(bc), _p(out), R, D, _stream())
    return out


def classifier_bwd(feat, Wc, pred, dpred, dfeat_extra=None, dfeat=None, dWc=None, dbc=None):
    _chk(feat, name='feat'); _chk(pred, name='pred'); _chk(dpred, name='dpred')
    R, D = feat.shape
    assert pred.shape == (R, 2) and dpred.shape == (R, 2)
    if dfeat_extra is not None:
        _chk(dfeat_extra, name='dfeat_extra'); assert dfeat_extra.shape == (R, D)
    dev = feat.device
    dfeat = torch.empty(R, D, dtype=torch.float32, device=dev) if dfeat is None else dfeat
    dWc = torch.empty(2, D, dtype=torch.float32, device=dev) if dWc is None else dWc
    dbc = torch.empty(2, dtype=torch.float32, device=dev) if dbc is None else dbc
    nb = _lib.load().wesup_classifier_bwd_workspace_bytes(R, D)
    ws = workspace(nb, dev, 'cls')
    _lib.call('wesup_classifier_bwd', _p(feat), _p(Wc), _p(pred), _p(dpred), _p(dfeat_extra), _p(dfeat), _p(dWc), _p(dbc),
              R, D, _p(ws), nb, _stream())
    return dfeat, dWc, dbc


def propagate(feat, meta, threshold, enable=True, out=None):
    """feat (B,Kmax,D).  Returns y_all (B,Kmax,C), src_idx (B,Kmax) int32, max_sim (B,Kmax) (``out``: the three, reused)."""
    _chk(feat, name='feat')
    B, Kmax, D = feat.shape
    assert B == meta.B and Kmax == meta.Kmax
    dev = feat.device
    if out is not None:
        y_all, src, sim = out
        assert y_all.shape == (B, Kmax, meta.C) and src.shape == (B, Kmax) == sim.shape and src.dtype == torch.int32
    else:
        y_all = torch.empty(B, Kmax, meta.C, dtype=torch.float32, device=dev)
        src = torch.empty(B, Kmax, dtype=torch.int32, device=dev)
        sim = torch.empty(B, Kmax, dtype=torch.float32, device=dev)
    tok = _tbegin('propagate')
    _lib.call('wesup_propagate', _p(feat), _p(meta.sp_labels), _p(meta.n_sp), _p(meta.n_l), float(threshold), int(enable),
              _p(y_all), _p(src), _p(sim), B, Kmax, D, meta.C, _stream())
    _tend(tok, 4.0 * B * Kmax * (D + 2 * meta.C + 2))
    return y_all, src, sim


def loss_fwd(pred, y_all, meta, eps, prop_weight, out=None):
    _chk(pred, name='pred'); _chk(y_all, name='y_all')
    B, Kmax, C = pred.shape
    assert y_all.shape == (B, Kmax, C) and B == meta.B and Kmax == meta.Kmax
    if out is not None:
        loss, terms = out                  # (loss None: only the per-image terms; the caller forms their mean)
        assert (loss is None or loss.numel() == 1) and terms.shape == (B, 8) and terms.is_contiguous()
    else:
        terms = torch.empty(B, 8, dtype=torch.float32, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    _lib.call('wesup_loss_fwd', _p(pred), _p(y_all), _p(meta.n_sp), _p(meta.n_l), float(eps), float(prop_weight),
              _p(terms), _p(loss), B, Kmax, C, _stream())
    return loss, terms


def loss_bwd(pred, y_all, meta, terms, dloss, eps, prop_weight, out=None):
    _chk(dloss, name='dloss')
    B, Kmax, C = pred.shape
    if out is None:
        out = torch.empty_like(pred)
    _lib.call('wesup_loss_bwd', _p(pred), _p(y_all), _p(meta.n_sp), _p(meta.n_l), _p(terms), _p(dloss), float(eps),
              float(prop_weight), _p(out), B, Kmax, C, _stream())
    return out


def head_fwd(feat, Wc, bc, pred, meta, threshold, enable=True, out=None):
    """classifier_fwd + propagate in one launch (wesup_head_fwd): feat (B,Kmax,D) -> pred (B*Kmax,2) and (y_all, src_idx, max_sim)."""
    _chk(feat, name='feat'); _chk(Wc, name='Wc'); _chk(bc, name='bc'); _chk(pred, name='pred')
    B, Kmax, D = feat.shape
    assert B == meta.B and Kmax == meta.Kmax and Wc.shape == (2, D) and bc.numel() == 2 and pred.numel() == B * Kmax * 2
    y_all, src, sim = out
    assert y_all.shape == (B, Kmax, meta.C) and src.shape == (B, Kmax) == sim.shape and src.dtype == torch.int32
    tok = _tbegin('propagate')
    _lib.call('wesup_head_fwd', _p(feat), _p(Wc), _p(bc), _p(pred), _p(meta.sp_labels), _p(meta.n_sp), _p(meta.n_l), float(threshold),
              int(enable), _p(y_all), _p(src), _p(sim), B, Kmax, D, meta.C, _stream())
    _tend(tok, 4.0 * B * Kmax * (D + 2 * meta.C + 2))
    return y_all, src, sim


def head_bwd_supported(Kmax, C):
    return Kmax % 64 == 0 and C == 2


def head_bwd_partials(R, D, device):
    """The buffer wesup_head_bwd leaves the partial sums of dWc / dbc in and wesup_classifier_bwd_finish reads (on another stream,
    launches later): owned by the caller -- the engine keeps one per buffer set -- never a shared grow-only workspace, which another
    tag's growth may replace between the two calls."""
    return torch.empty(max(int(_lib.load().wesup_classifier_bwd_workspace_bytes(R, D)), 256), dtype=torch.uint8, device=device)


def head_bwd(feat, Wc, pred, y_all, meta, dloss, eps, prop_weight, terms, dpred, dfeat, partials):
    """loss_fwd (terms) + loss_bwd (dpred) + the first kernel of classifier_bwd (dfeat, partial sums of dWc / dbc into ``partials``,
    see head_bwd_partials) in one launch (wesup_head_bwd); classifier_bwd_finish adds the partial sums up."""
    _chk(feat, name='feat'); _chk(pred, name='pred'); _chk(y_all, name='y_all'); _chk(dloss, name='dloss')
    _chk(partials, torch.uint8, 'partials')
    B, Kmax, C = y_all.shape
    R, D = feat.shape
    assert R == B * Kmax and pred.numel() == R * C and dpred.numel() == R * C and dfeat.shape == (R, D) and terms.shape == (B, 8)
    assert head_bwd_supported(Kmax, C) and B == meta.B and Kmax == meta.Kmax
    nb = _lib.load().wesup_classifier_bwd_workspace_bytes(R, D)
    assert partials.numel() >= nb
    _lib.call('wesup_head_bwd', _p(feat), _p(Wc), _p(pred), _p(y_all), _p(meta.n_sp), _p(meta.n_l), _p(dloss), float(eps),
              float(prop_weight), _p(terms), _p(dpred), _p(dfeat), B, Kmax, D, C, _p(partials), nb, _stream())


def classifier_bwd_finish(partials, R, D, dWc, dbc):
    _chk(partials, torch.uint8, 'partials')
    nb = _lib.load().wesup_classifier_bwd_workspace_bytes(R, D)
    assert partials.numel() >= nb
    _lib.call('wesup_classifier_bwd_finish', _p(partials), nb, _p(dWc), _p(dbc), R, D, _stream())


def cross_entropy_fwd(y_hat, y_true, eps, class_weights=None):
    _chk(y_hat, name='y_hat'); _chk(y_true, name='y_true')
    n, C = y_hat.shape
    assert y_true.shape == (n, C)
    if class_weights is not None:
        _chk(class_weights, name='class_weights'); assert class_weights.numel() == C
    out2 = torch.empty(4, dtype=torch.float32, device=y_hat.device)
    _lib.call('wesup_cross_entropy_fwd', _p(y_hat), _p(y_true), _p(class_weights), float(eps), _p(out2), n, C, _stream())
    return out2


def cross_entropy_bwd(y_hat, y_true, out2, dloss, eps, class_weights=None):
    n, C = y_hat.shape
    dy = torch.empty_like(y_hat)
    if n > 0:
        _lib.call('wesup_cross_entropy_bwd', _p(y_hat), _p(y_true), _p(class_weights), _p(out2), _p(dloss), float(eps),
                  _p(dy), n, C, _stream())
    return dy


def sgd_step(p, g, v, lr, momentum, weight_decay, grad_scale, first_step):
    for t, n in ((p, 'p'), (g, 'g'), (v, 'v')):
        _chk(t, name=n)
    assert p.numel() == g.numel() == v.numel()
    tok = _tbegin('sgd')
    _lib.call('wesup_sgd_step', _p(p), _p(g), _p(v), p.numel(), float(lr), float(momentum), float(weight_decay),
              float(grad_scale), int(first_step), _stream())
    _tend(tok, 20.0 * p.numel())                     # read p, g, v; write p, v


def seg_metrics(pred, mask, out=None):
    """pred (B,H,W) f32, mask (B,C,H,W) uint8 -> (B,4) sums {#(P==G), sum(P*G), sum(P), sum(G)}."""
    _chk(pred, name='pred'); _chk(mask, torch.uint8, 'mask')
    B, H, W = pred.shape
    C = mask.shape[1]
    if out is None:
        out = torch.empty(B, 4, dtype=torch.float32, device=pred.device)
    assert out.shape == (B, 4) and out.is_contiguous()
    nb = _lib.load().wesup_seg_metrics_workspace_bytes(B)
    ws = workspace(nb, pred.device, 'seg')
    _lib.call('wesup_seg_metrics', _p(pred), _p(mask), _p(out), B, H * W, C, _p(ws), nb, _stream())
    return out
