"""Edge cases of the training step on the GPU, each against the CPU oracle: a single superpixel, every superpixel
labelled (fully supervised branch), no annotation at all, odd image sizes (floor-mode pooling, ragged upsampling),
more labelled rows than one propagation tile, label ids given with gaps (rejected), and an all-zero mask."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-4


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
    trainer.tracker.train()
    return trainer


def run_both(imgs, segs, masks, pix, seed=7, feat_scale=0.03, check_grads=('backbone.0.weight', 'backbone.28.weight',
                                                                            'side_conv576.weight', 'fc_layers.0.weight',
                                                                            'classifier.0.weight')):
    from oracle import wesup_oracle as orc
    weights = orc.make_weights(seed, feat_scale=feat_scale)
    ref_loss, ref_grads, ref_new, _, outs, mets = orc.train_step(weights, imgs, segs.astype(np.int64),
                                                                 None if masks is None else masks.astype(np.int64))
    trainer = make_trainer(weights)
    d = torch.device('cuda:0')
    data = [torch.from_numpy(imgs).to(d), torch.from_numpy(pix).long().to(d),
            torch.from_numpy(masks).long().to(d) if masks is not None else torch.tensor(0), torch.from_numpy(segs)]
    trainer.train_one_iteration('train', *data)
    loss = trainer.tracker.history['loss'][0]
    assert abs(loss - ref_loss) <= TOL * max(abs(ref_loss), 1e-3), (loss, ref_loss)
    # Gradients against an fp64 evaluation that takes the GPU's ReLU signs and pooling arg-maxes (tests/_gradcheck.py):
    # decisions that differ from fp64's own must be genuine near-ties (they are named in the failure message), and
    # under equal decisions every checked gradient is within 1e-4 of its tensor's max.
    import _gradcheck
    ys = _gradcheck.gpu_preactivations(trainer.model.engine)
    import torch.nn.functional as F
    w32 = orc.to_torch(weights)
    for b in range(imgs.shape[0]):                                   # the activations themselves are tight
        h = torch.from_numpy(imgs[b])[None]
        for li, idx in enumerate(orc.CONV_IDX):
            y = F.conv2d(h, w32[f'backbone.{idx}.weight'], w32[f'backbone.{idx}.bias'], padding=1)
            assert rel_err(ys[li][b:b + 1], y) < 1e-5, li
            h = F.relu(y)
            if orc.POOL_AFTER[li] and li != 12:
                h = F.max_pool2d(h, 2, 2)
    _gradcheck.check_gradients(trainer.model, weights, imgs, segs, masks, names=list(check_grads))
    return trainer, outs, mets


def test_single_superpixel():
    from wesup_amd import synth
    H = W = 32
    imgs = synth.synth_image(1, H, W)[None]
    segs = np.zeros((1, H, W), dtype=np.int32)
    masks = np.zeros((1, 2, H, W), dtype=np.uint8)
    masks[0, 1, 5, 7] = 1
    pix = synth.pixel_mask(1, H, W)[None]
    trainer, outs, _ = run_both(imgs, segs, masks, pix)          # N = 1, labelled: fully supervised branch
    assert int(trainer.model._last_meta.n_sp[0]) == 1 and trainer.model._last_meta.Kmax == 64      # padded rows are inert
    assert 'propagated_labels' not in trainer.tracker.history


def test_all_superpixels_labelled_full_mask():
    from wesup_amd import synth
    H, W, g = 48, 64, 5
    imgs = np.stack([synth.synth_image(10 + b, H, W) for b in range(2)])
    segs = np.stack([synth.voronoi_labels(20 + b, H, W, g) for b in range(2)])
    pix = np.stack([synth.pixel_mask(30 + b, H, W) for b in range(2)])
    trainer, _, _ = run_both(imgs, segs, pix.copy(), pix)          # dense mask: n_l == N (models/wesup.py:525-526)
    assert int(trainer.model._last_meta.n_l.min()) == g * g


def test_odd_sizes_and_ragged_counts():
    from wesup_amd import synth
    H, W = 70, 50                                                  # 70->35->17->8->4, 50->25->12->6->3
    imgs = np.stack([synth.synth_image(40 + b, H, W) for b in range(2)])
    segs = np.stack([synth.voronoi_labels(50, H, W, 5), synth.voronoi_labels(51, H, W, 3)])
    masks = np.stack([synth.point_mask(60 + b, segs[b], 0.4, 2, tie_every=2) for b in range(2)])
    pix = np.stack([synth.pixel_mask(70 + b, H, W) for b in range(2)])
    run_both(imgs, segs, masks, pix)


def test_many_labelled_rows_span_propagation_tiles():
    """n_l > 256 labelled rows: the propagation kernel walks more than one LDS tile; argmax indices bit-exact."""
    from oracle import wesup_oracle as orc
    from wesup_amd import ops
    d = torch.device('cuda:0')
    N, n_l, D = 900, 700, 32
    g = torch.Generator().manual_seed(3)
    feat = torch.relu(torch.randn(N, D, generator=g) * 0.08)
    feat[800] = feat[650]                                           # exact duplicate of a labelled row in tile 2
    feat[801] = feat[10]
    y_l = torch.zeros(n_l, 2)
    y_l[torch.arange(n_l), torch.randint(0, 2, (n_l,), generator=g)] = 1
    from wesup_amd.models.wesup import _label_propagate
    got = _label_propagate(feat.to(d), y_l.to(d), 0.8)
    ref, W_ul, max_sim, src = orc.label_propagate(feat, y_l, 0.8, return_aux=True)
    assert torch.equal(got.cpu(), ref)
    meta = ops.SuperpixelMeta()
    meta.B, meta.Kmax, meta.C = 1, N, 2
    meta.sp_labels = torch.zeros(1, N, 2, device=d); meta.sp_labels[0, :n_l] = y_l.to(d)
    meta.n_sp = torch.tensor([N], dtype=torch.int32, device=d); meta.n_l = torch.tensor([n_l], dtype=torch.int32, device=d)
    _, s, sim = ops.propagate(feat.view(1, N, D).to(d), meta, 0.8)
    assert torch.equal(s[0, n_l:].cpu().long(), src)
    assert int(s[0, 800]) == 650 and int(s[0, 801]) == 10


def test_no_annotation_forward_only_and_zero_mask():
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    d = torch.device('cuda:0')
    H, W, g = 32, 32, 4
    weights = orc.make_weights(9, feat_scale=0.05)
    trainer = make_trainer(weights)
    img = torch.from_numpy(synth.synth_image(3, H, W))[None].to(d)
    seg = torch.from_numpy(synth.voronoi_labels(3, H, W, g))[None]
    # (img,) only: preprocess arity 1 (models/wesup.py:464-467) -> prediction without labels
    (x, sp_maps), (pixel_mask, sp_lab) = trainer.preprocess(img, torch.tensor(0), torch.tensor(0), seg)
    with torch.no_grad():
        pred = trainer.model((x, sp_maps))
    o = orc.forward_image(orc.to_torch(weights), img[0].cpu(), seg[0].long(), None)
    assert rel_err(pred[0], o['pred']) < TOL
    assert int(sp_maps.meta.n_l[0]) == 0
    # a mask that labels nothing: loss is exactly 0 and so are all gradients (the reference returns tensor(0.),
    # models/wesup.py:88-89)
    zero = torch.zeros(1, 2, H, W, dtype=torch.long)
    trainer.train_one_iteration('train', img, zero, zero, seg)
    assert trainer.tracker.history['loss'][0] == 0.0
    assert float(trainer.model._flat_grad.abs().max()) == 0.0


def test_gapped_label_ids_are_rejected():
    from wesup_amd.models.wesup import _preprocess_superpixels
    d = torch.device('cuda:0')
    seg = torch.zeros(16, 16, dtype=torch.long)
    seg[8:] = 2                                                     # id 1 missing: NaN row in the reference (:57-61)
    with pytest.raises(ValueError):
        _preprocess_superpixels(seg.to(d), None)


def test_long_thin_diagonal_superpixel():
    """A superpixel that is a 3-pixel diagonal band: its box of the coarse grids (80x80: 6400 cells, 40x40: 1600) does
    not fit the per-segment LDS cell table of the fused upsample+scatter-mean, so that segment takes the per-pixel loop
    while every other superpixel of the image is pooled cell by cell; both against the oracle."""
    from wesup_amd import synth
    H = W = 160
    imgs = synth.synth_image(81, H, W)[None]
    seg = synth.voronoi_labels(82, H, W, 8).astype(np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    band = np.abs(yy - xx) <= 1
    seg[band] = seg.max() + 1
    ids, seg = np.unique(seg, return_inverse=True)                 # contiguous ids again should a cell have vanished
    segs = seg.reshape(1, H, W).astype(np.int32)
    assert int((segs == segs.max()).sum()) == int(band.sum()) <= 512           # one segment
    masks = synth.point_mask(83, segs[0], 0.3, 2)[None]
    pix = synth.pixel_mask(84, H, W)[None]
    run_both(imgs, segs, masks, pix)


def test_one_block_per_cu_shape_of_the_nt_gemm():
    """WESUP_NT_SHAPE=big (256x128 tiles, 8 waves, one block per CU; measurement knob) computes the same
    convolution: run in a child process because the choice is read once per process."""
    import subprocess, sys, os
    code = (
        "import torch, torch.nn.functional as F\n"
        "from wesup_amd import ops\n"
        "torch.manual_seed(0)\n"
        "B,H,W,Ci,Co = 2,60,60,128,256\n"
        "x = torch.randn(B,Ci,H,W); w = torch.randn(Co,Ci,3,3)*(9*Ci)**-0.5; b = torch.randn(Co)\n"
        "ref = F.conv2d(F.relu(x).double(), w.double(), b.double(), padding=1).float().permute(0,2,3,1)\n"
        "wf, wd = ops.pack_conv3x3_weight(w.cuda())\n"
        "y = ops.conv3x3_fwd(x.permute(0,2,3,1).contiguous().cuda(), wf, b.cuda(), Co, True)\n"
        "e = float((y.cpu().double()-ref.double()).abs().max()/ref.abs().max())\n"
        "A = torch.randn(5000, 512); Bw = torch.randn(384, 512)*512**-0.5\n"
        "o = ops.gemm_nt(A.cuda(), Bw.cuda(), None)\n"
        "e2 = float((o.cpu().double()-(A.double()@Bw.double().t())).abs().max()/ (A.double()@Bw.double().t()).abs().max())\n"
        "assert e < 1e-4 and e2 < 1e-4, (e, e2)\n"
        "print('ok', e, e2)\n")
    env = dict(os.environ, WESUP_NT_SHAPE='big')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr
