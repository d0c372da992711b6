"""The drop-in boundary on the GPU: what `preprocess` hands back, `_cross_entropy(class_weights=...)`,
`freeze_backbone`, a bounded engine-buffer cache over changing shapes, `feature_maps` of the LAST forward, and
`train.fit(...)` end to end on a synthetic dataset."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_trainer(weights, **kw):
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    trainer = initialize_trainer('wesup', device='cuda:0', **kw)
    trainer.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    trainer.optimizer, trainer.scheduler = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.model.train()
    trainer.tracker.train()
    return trainer


def test_preprocess_returns_sp_labels_like_the_reference():
    """models/wesup.py:487-490: ((img, sp_maps), (pixel_mask, sp_labels)) with sp_labels the (N_l, C) float tensor of
    the labelled superpixels -- here a lazy wrapper that behaves like it -- and the reference-style compute_loss on
    the plain tensor gives the same loss as the batched device path."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    from wesup_amd.utils import is_empty_tensor
    weights = orc.make_weights(3, feat_scale=0.03)
    trainer = make_trainer(weights)
    H, W, g = 64, 48, 5
    img = synth.synth_image(1, H, W)[None]
    seg = synth.voronoi_labels(2, H, W, g)[None]
    pts = synth.point_mask(3, seg[0], 0.3, 2, tie_every=3)[None]
    pix = synth.pixel_mask(4, H, W)[None]
    (x, sp_maps), (pixel_mask, sp_labels) = trainer.preprocess(torch.from_numpy(img), torch.from_numpy(pix).long(),
                                                               torch.from_numpy(pts).long(), torch.from_numpy(seg))
    pp = orc.preprocess_superpixels(torch.from_numpy(seg[0].astype(np.int64)), torch.from_numpy(pts[0].astype(np.int64)))
    assert pixel_mask.shape == (1, 2, H, W)
    assert sp_labels.size(0) == pp['n_l'] and tuple(sp_labels.size()) == (pp['n_l'], 2) and len(sp_labels) == pp['n_l']
    assert sp_labels.dim() == 2 and sp_labels.dtype == torch.float32
    assert torch.equal(sp_labels[:].cpu(), pp['sp_labels'])                  # indexing
    assert float(torch.sum(sp_labels)) == float(pp['sp_labels'].sum())      # torch functions
    assert float(sp_labels.sum(dim=1).min()) >= 1.0                           # tensor methods
    assert sp_maps.size() == (g * g, H, W)
    pred = trainer.model((x, sp_maps))
    l_batched = trainer.compute_loss(pred, (pixel_mask, sp_labels), metrics={})
    pred = trainer.model((x, sp_maps))
    m2 = {}
    l_ref_style = trainer.compute_loss(pred, (pixel_mask, sp_labels.tensor().clone()), metrics=m2)
    assert abs(float(l_batched) - float(l_ref_style)) <= 1e-6 * abs(float(l_ref_style))
    assert 'labeled_sp_ratio' in m2
    # without any mask the reference's sp_labels is the 0-dim empty tensor (models/wesup.py:54)
    _, (pm, lab) = trainer.preprocess(torch.from_numpy(img), torch.tensor(0), torch.tensor(0), torch.from_numpy(seg))
    assert is_empty_tensor(lab.tensor()) and lab.dim() == 0


def test_cross_entropy_class_weights():
    """models/wesup.py:93-94: ce * class_weights, forward and backward, against the oracle's formula in torch."""
    from wesup_amd.models.wesup import _cross_entropy
    d = torch.device('cuda:0')
    rs = np.random.RandomState(0)
    y_hat = torch.softmax(torch.from_numpy(rs.randn(37, 2).astype(np.float32) * 3), dim=1)
    y_hat[3] = torch.tensor([1.0, 0.0])                                       # clamped entries: zero gradient
    y_true = torch.from_numpy((rs.rand(37, 2) < 0.4).astype(np.float32))
    cw = torch.tensor([3.0, 1.0])
    a = y_hat.clone().requires_grad_(True)
    yc = torch.clamp(a, min=1e-7, max=1 - 1e-7)
    ref = torch.sum(-y_true * torch.log(yc) * cw.unsqueeze(0)) / torch.sum(y_true.sum(dim=1) > 0).float()
    ref.backward()
    b = y_hat.clone().to(d).requires_grad_(True)
    got = _cross_entropy(b, y_true.to(d), class_weights=cw.to(d))
    (2.0 * got).backward()
    assert abs(float(got) - float(ref)) <= 1e-6 * abs(float(ref))
    assert rel_err(b.grad, 2.0 * a.grad) < 1e-6
    # plain tuple of weights (the config's class_weights = (3, 1)) and no weights
    assert abs(float(_cross_entropy(y_hat.to(d), y_true.to(d), class_weights=(3, 1))) - float(ref)) <= 1e-6 * abs(float(ref))
    ref0 = torch.sum(-y_true * torch.log(torch.clamp(y_hat, 1e-7, 1 - 1e-7))) / torch.sum(y_true.sum(dim=1) > 0).float()
    assert abs(float(_cross_entropy(y_hat.to(d), y_true.to(d))) - float(ref0)) <= 1e-6 * abs(float(ref0))


def test_freeze_backbone():
    """models/wesup.py:427-429,447: frozen backbone parameters get no gradient and no update; everything else trains
    exactly as in the unfrozen run (the side / head gradients do not depend on the backbone's weight gradients)."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    weights = orc.make_weights(9, feat_scale=0.03)
    imgs, labs, pts, pix = synth.make_batch(4, 2, 64, 64, 5)
    data = (torch.from_numpy(imgs), torch.from_numpy(pix).long(), torch.from_numpy(pts).long(), torch.from_numpy(labs))
    free = make_trainer(weights)
    free.train_one_iteration('train', *data)
    frozen = make_trainer(weights, freeze_backbone=True)
    assert all(not p.requires_grad for p in frozen.model.backbone.parameters())
    assert all(id(p) not in {id(q) for g in frozen.optimizer.param_groups for q in g['params']}
               for p in frozen.model.backbone.parameters())
    launches = []
    from wesup_amd import ops
    real_wgrad, real_dgrad = ops.conv3x3_wgrad, ops.conv3x3_dgrad
    ops.conv3x3_wgrad = lambda *a, **k: (launches.append('wgrad'), real_wgrad(*a, **k))[1]
    ops.conv3x3_dgrad = lambda *a, **k: (launches.append('dgrad'), real_dgrad(*a, **k))[1]
    try:
        frozen.train_one_iteration('train', *data)
    finally:
        ops.conv3x3_wgrad, ops.conv3x3_dgrad = real_wgrad, real_dgrad
    assert launches == []                                   # no backbone wgrad, no dgrad chain at all
    assert frozen.tracker.history['loss'][0] == free.tracker.history['loss'][0]
    sd0 = {k: torch.from_numpy(v) for k, v in weights.items()}
    for (k, a), (_, b) in zip(frozen.model.state_dict().items(), free.model.state_dict().items()):
        if k.startswith('backbone.'):
            assert torch.equal(a.cpu(), sd0[k]), k           # untouched (not even weight decay)
            assert float(frozen.model._grad_views[k].abs().max()) == 0.0, k
        else:
            assert torch.equal(frozen.model._grad_views[k], free.model._grad_views[k]), k
            assert torch.equal(a, b), k
    # a partially frozen backbone: conv1_1..conv3_3 frozen, conv4_1.. train; their gradients equal the free run's
    part = make_trainer(weights)
    for name, p in part.model.named_parameters():
        if name.startswith('backbone.') and int(name.split('.')[1]) < 17:
            p.requires_grad = False
    part.optimizer, _ = part.get_default_optimizer()
    part.train_one_iteration('train', *data)
    for k in free.model._grad_views:
        if k.startswith('backbone.') and int(k.split('.')[1]) < 17:
            assert float(part.model._grad_views[k].abs().max()) == 0.0, k
        else:
            assert torch.equal(part.model._grad_views[k], free.model._grad_views[k]), k


def test_buffer_cache_is_bounded_over_changing_shapes():
    """Multi-scale training meets a new (H, W, superpixel count) almost every iteration: the engine keeps the buffers
    of the most recent shapes only, so device memory stays bounded instead of growing by a buffer set per shape."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    trainer = make_trainer(orc.make_weights(2, feat_scale=0.05))
    trainer.model._ensure_engine()
    eng = trainer.model.engine
    assert eng.max_cached_shapes >= 2 and eng.max_cached_pixels >= 2 * 4 * 480 * 480      # training + validation shape of configs[1]
    eng.max_cached_shapes = 2                # (default 16: small crops that come back find their buffers and their step plan)
    rs = np.random.RandomState(0)
    peak = []
    shapes = set()
    for it in range(20):
        H, W, g = int(rs.randint(20, 41)) * 4, int(rs.randint(20, 41)) * 4, int(rs.randint(4, 13))
        img = synth.synth_image(it, H, W)[None]
        seg = synth.voronoi_labels(it, H, W, g)[None]
        pts = synth.point_mask(it, seg[0], 0.3, 2)[None]
        pix = synth.pixel_mask(it, H, W)[None]
        trainer.train_one_iteration('train', torch.from_numpy(img), torch.from_numpy(pix).long(),
                                    torch.from_numpy(pts).long(), torch.from_numpy(seg))
        torch.cuda.synchronize()
        shapes.add((H, W, trainer.model._last_meta.Kmax))
        assert len(trainer.model.engine._bufs) <= trainer.model.engine.max_cached_shapes
        peak.append(torch.cuda.memory_allocated())
    assert len(shapes) >= 15                                        # really different shapes
    assert all(k % 64 == 0 for _, _, k in shapes)                   # superpixel rows padded to a coarse multiple
    # a 160x160 image needs ~80 MB of training buffers; twenty cached shapes would add > 1 GB, two add < 0.4 GB
    assert max(peak) - peak[1] < (400 << 20), [p >> 20 for p in peak]
    assert len(trainer.tracker.history['loss']) == 20 and np.isfinite(trainer.tracker.history['loss']).all()


def test_feature_maps_belong_to_the_last_forward():
    """Shapes A, B, then A again (a cache hit): feature_maps must be A's, not the most recently CREATED entry's."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    from wesup_amd.models.wesup import preprocess_label_maps, SuperpixelMaps
    d = torch.device('cuda:0')
    trainer = make_trainer(orc.make_weights(2, feat_scale=0.05))
    model = trainer.model
    model.eval()

    def run(seed, H, W):
        img = torch.from_numpy(synth.synth_image(seed, H, W))[None].to(d)
        seg = torch.from_numpy(synth.voronoi_labels(seed, H, W, 4))[None].to(d)
        meta = preprocess_label_maps(seg, None, Kmax=16, n_sp_host=[16])
        with torch.no_grad():
            model((img, SuperpixelMaps(meta)))
        return model.feature_maps.clone()
    a1 = run(1, 48, 80)
    b1 = run(2, 80, 48)                    # same number of elements, other shape
    a2 = run(1, 48, 80)                    # cache hit on A's buffers
    assert a2.shape == a1.shape == (2112, 48, 80) and b1.shape == (2112, 80, 48)
    assert torch.equal(a1, a2)


def test_train_fit_end_to_end(tmp_path, monkeypatch):
    """train.fit('synthetic:...', smoke=True, epochs=2) through the reference's trainer surface (train.py:14-32,
    models/base.py:252-333): datasets, epochs, history.csv, checkpoint, and the record directory removed by --smoke."""
    monkeypatch.setenv('RECORD_ROOT', str(tmp_path))
    from wesup_amd.train import fit
    trainer = fit('synthetic:64:64:4:6', model='wesup', epochs=2, batch_size=2, smoke=True, num_workers=0)
    assert trainer.initial_epoch == 1
    assert not trainer.record_dir.exists()                                   # --smoke: removed (train.py:26-27)
    trainer = fit('synthetic:64:64:4:6', model='wesup', epochs=2, batch_size=2, num_workers=0)
    rd = trainer.record_dir
    rows = (rd / 'history.csv').read_text().strip().splitlines()
    assert len(rows) == 3                                                    # header + one row per epoch
    cols = rows[0].split(',')
    assert cols[-1] == 'lr' and cols[:-1] == sorted(cols[:-1])
    assert {'loss', 'accuracy', 'dice', 'labeled_sp_ratio', 'propagated_labels', 'propagate_loss', 'val_accuracy',
            'val_dice'} <= set(cols)
    ckpts = sorted((rd / 'checkpoints').glob('*.pth'))
    assert [c.name for c in ckpts] == ['ckpt.0002.pth']                      # older checkpoints are pruned
    # resume: one more epoch from the checkpoint continues the numbering
    t2 = fit('synthetic:64:64:4:6', model='wesup', epochs=1, batch_size=2, checkpoint=str(ckpts[0]), num_workers=0)
    assert t2.initial_epoch == 3 and t2.record_dir == rd
    assert sorted(p.name for p in (rd / 'checkpoints').glob('*.pth')) == ['ckpt.0003.pth']


def test_nan_loss_raises_before_the_weights_are_touched():
    """models/base.py:202-203: `if torch.isnan(loss): raise ValueError('Loss is nan!')` sits in front of backward and
    the optimiser.  A NaN anywhere in the network must reach the loss as it does through torch's NaN-propagating relu
    and clamp (here: through the pre-ReLU side tap and NaN-preserving epilogue ReLU / clamp), and the parameters must be
    bit-identical afterwards."""
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    trainer = make_trainer(orc.make_weights(9, feat_scale=0.03))
    imgs, labs, pts, pix = synth.make_batch(4, 2, 64, 64, 5)
    imgs[1, 2, 40, 17] = np.nan
    before = trainer.model._flat.clone()
    with pytest.raises(ValueError, match='nan'):
        trainer.train_one_iteration('train', torch.from_numpy(imgs), torch.from_numpy(pix).long(),
                                    torch.from_numpy(pts).long(), torch.from_numpy(labs))
    torch.cuda.synchronize()
    assert torch.equal(before, trainer.model._flat)
    imgs[1, 2, 40, 17] = 0.5                                      # and the trainer is usable afterwards
    trainer.train_one_iteration('train', torch.from_numpy(imgs), torch.from_numpy(pix).long(),
                                torch.from_numpy(pts).long(), torch.from_numpy(labs))
    assert np.isfinite(trainer.tracker.history['loss'][-1]) and not torch.equal(before, trainer.model._flat)
