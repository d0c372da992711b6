"""Round-4 entries that the step uses in batched / recorded form, each against the single form it replaces:
wesup_winograd_pack_weights, wesup_transpose_batched, wesup_winograd_fused_route, and a step plan recorded and replayed by
hand through the C ABI (wesup_plan_* / wesup_sync_* / wesup_copy / wesup_fill_words) across two streams."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def test_filter_transforms_of_several_layers_in_one_launch():
    from wesup_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64), (128, 64), (64, 128), (256, 256), (512, 256), (96, 32)]
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5).to(d) for co, ci in shapes]
    items, want = [], []
    for i, w in enumerate(ws):
        co, ci = w.shape[:2]
        uf = torch.full((36, co, ci), float('nan'), device=d) if i != 2 else None       # one panel without a forward filter ...
        ud = torch.full((36, ci, co), float('nan'), device=d) if i != 4 else None       # ... one without the rotated one
        items.append((w, uf, ud))
        want.append(ops.winograd_pack_weight(w, m=4))
    ops.winograd_pack_weights(items)
    for (w, uf, ud), (rf, rd) in zip(items, want):
        if uf is not None:
            assert torch.equal(uf, rf)
        if ud is not None:
            assert torch.equal(ud, rd)
    with pytest.raises(Exception):
        ops.winograd_pack_weights(items * 6)           # more than 32 panels per launch


def test_transposes_of_several_matrices_in_one_launch():
    from wesup_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(32, 64), (1024, 2112), (1024, 1024), (32, 1024), (7, 5), (1, 300), (257, 129), (64, 64)]
    pairs = [(torch.randn(r, c, generator=g).to(d), torch.full((c, r), float('nan'), device=d)) for r, c in shapes]
    ops.transpose_batched(pairs)
    for a, out in pairs:
        assert torch.equal(out, a.t().contiguous())


def test_more_transposes_than_one_launch_takes():
    """ops.transpose_batched cuts its list into launches of TRANSPOSE_MAX items (the C entry takes at most 40)."""
    from wesup_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(6)
    n = 2 * ops.TRANSPOSE_MAX + 7
    pairs = [(torch.randn(5 + i % 11, 3 + i % 7, generator=g).to(d), None) for i in range(n)]
    pairs = [(a, torch.full((a.shape[1], a.shape[0]), float('nan'), device=d)) for a, _ in pairs]
    ops.transpose_batched(pairs)
    for a, out in pairs:
        assert torch.equal(out, a.t().contiguous())


@pytest.mark.parametrize('B,H,W,g', [(16, 256, 256, 12), (32, 480, 480, 24)])
def test_large_batches_step_through_the_matrix_pooling(B, H, W, g):
    """The forward hands B x (matrix-pooling groups) interpolation-matrix transposes to ops.transpose_batched: 48 at B = 16,
    256 x 256 (three groups), 64 at B = 32, 480 x 480 (two) -- more than one launch holds.  One training iteration; the superpixel
    features of the first four images against the same images as a batch of four (1e-4 of the tensor's max: the product kernels'
    tilings depend on the batch)."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    from wesup_amd.models import initialize_trainer
    d = _dev()
    weights = orc.make_weights(2, feat_scale=0.05)
    imgs, labs, pts, pix = synth.make_batch(21, B, H, W, g)

    def run(n):
        t = initialize_trainer('wesup', device='cuda:0', max_superpixels=g * g)
        t.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
        t.optimizer, t.scheduler = t.get_default_optimizer()
        t.model.train(); t.tracker.train()
        t.train_one_iteration('train', *(torch.from_numpy(a[:n]).to(d) for a in (imgs, pix, pts, labs)))
        torch.cuda.synchronize()
        feats = t.model._padded[0].detach().clone()
        loss = t.tracker.history['loss'][-1]
        t.model.engine.release_buffers()
        return feats, loss
    full, loss = run(B)
    part, _ = run(4)
    assert np.isfinite(loss) and full.shape[0] == B
    assert float((full[:4] - part).abs().max()) <= 1e-4 * float(part.abs().max())
    torch.cuda.empty_cache()


def test_one_kernel_route_needs_a_grid_that_fills_the_chip():
    from wesup_amd import _lib
    h = _lib.load()
    assert h.wesup_winograd_fused_supported(64, 64, 4) == 2 and h.wesup_winograd_fused_supported(512, 512, 4) == 0
    big = 4 * 120 * 120                      # conv1_2 at the bench shape: 57 600 tiles
    assert h.wesup_winograd_fused_route(64, 64, 4, big) == 2
    assert h.wesup_winograd_fused_route(256, 256, 4, 900) == 0            # one 120 x 120 image: 29 x 4 blocks
    assert h.wesup_winograd_fused_route(256, 256, 4, 3600) == h.wesup_winograd_fused_supported(256, 256, 4)
    assert h.wesup_winograd_fused_route(512, 512, 4, big) == 0 and h.wesup_winograd_fused_route(64, 64, 2, big) == 0


def test_plan_recorded_and_replayed_through_the_c_abi():
    """A small 'step' on two streams: fill, GEMM on stream A; record slot; stream B waits, transposes the product, copies it to the
    host and device-to-device.  Recorded twice (identical), replayed with new inputs in the SAME buffers: the replay computes from
    what the buffers hold at replay time, and its two halves can be issued with host work in between (a cut)."""
    from wesup_amd import _lib, ops
    d = _dev()
    h = _lib.load()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    A = torch.randn(256, 128, device=d); B = torch.randn(64, 128, device=d)
    C = torch.empty(256, 64, device=d); Ct = torch.empty(64, 256, device=d); Ct2 = torch.empty_like(Ct)
    pad = torch.empty(64, dtype=torch.int32, device=d)
    host = torch.empty(64, 256).pin_memory()
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def step():
        with torch.cuda.stream(sa):
            _lib.call('wesup_fill_words', P(pad), 7, pad.numel(), ctypes.c_void_p(sa.cuda_stream))
            ops.gemm_nt(A, B, None, out=C)
            ops.sync_record(33)
        with torch.cuda.stream(sb):
            ops.sync_wait(33)
            ops.transpose_batched([(C, Ct)])
            _lib.call('wesup_copy_to_host', ctypes.c_void_p(host.data_ptr()), P(Ct), Ct.numel() * 4, ctypes.c_void_p(sb.cuda_stream))
            _lib.call('wesup_copy', P(Ct2), P(Ct), Ct.numel() * 4, ctypes.c_void_p(sb.cuda_stream))
            ops.sync_record(34)

    step(); torch.cuda.synchronize()                      # warm-up: workspaces exist
    plans = []
    for _ in range(2):
        p = ctypes.c_void_p()
        _lib.call('wesup_plan_create', ctypes.byref(p))
        _lib.call('wesup_plan_begin', p)
        step()
        _lib.check(h.wesup_plan_end(p), 'wesup_plan_end')
        plans.append(p)
    torch.cuda.synchronize()
    assert h.wesup_plan_diff(plans[0], plans[1]) == 0
    n = h.wesup_plan_size(plans[0])
    assert n >= 8 and h.wesup_plan_kernels(plans[0]) >= 3
    A.normal_(); B.normal_(); pad.zero_(); Ct2.zero_()
    torch.cuda.synchronize()
    cut = n // 2
    _lib.check(h.wesup_plan_replay(plans[0], 0, cut), 'wesup_plan_replay')
    _lib.check(h.wesup_plan_replay(plans[0], cut, n), 'wesup_plan_replay')
    _lib.check(h.wesup_sync_synchronize(34), 'wesup_sync_synchronize')
    want = (A.double() @ B.double().t()).t().float()
    assert float((Ct - want).abs().max()) < 1e-3 and torch.equal(Ct2, Ct) and torch.equal(host.to(d), Ct)
    assert bool((pad == 7).all())
    # a recording with another operand is not the same plan
    p3 = ctypes.c_void_p()
    _lib.call('wesup_plan_create', ctypes.byref(p3))
    _lib.call('wesup_plan_begin', p3)
    A2 = torch.randn_like(A)
    keep, A = A, A2
    step()
    _lib.check(h.wesup_plan_end(p3), 'wesup_plan_end')
    torch.cuda.synchronize()
    assert h.wesup_plan_diff(plans[0], p3) != 0
    for p in plans + [p3]:
        h.wesup_plan_destroy(p)
