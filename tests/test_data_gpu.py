"""GPU side of the input pipeline: wesup_augment against its numpy restatement, and the DevicePrefetcher feeding
the trainer from an on-disk dataset in the reference's layout."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_augment_kernel_matches_the_numpy_restatement():
    from oracle import augment_oracle as ao
    from wesup_amd import ops
    from wesup_amd.utils import data as D
    d = torch.device('cuda:0')
    rs = np.random.RandomState(3)
    B, H, W = 5, 61, 83
    img = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    # smooth images as well: random noise hides interpolation mistakes behind large local contrast
    yy, xx = np.mgrid[0:H, 0:W]
    img[1] = np.stack([(xx * 3) % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    mask = rs.randint(0, 2, (B, H, W)).astype(np.uint8)
    mask[2] = 255                                                     # an image without a mask: no class anywhere
    rows = []
    for b in range(B):
        row, _ = D.sample_params(rs, H, W, train=(b > 0))
        if b == 3:
            row[8:11] = 0                                             # geometry + brightness/contrast only
        rows.append(row)
    params = np.stack(rows)
    out, om = ops.augment(torch.from_numpy(img).to(d), torch.from_numpy(mask).to(d), torch.from_numpy(params).to(d))
    out, om = out.cpu().numpy(), om.cpu().numpy()
    for b in range(B):
        want, wm = ao.augment(img[b], mask[b], params[b])
        assert np.array_equal(om[b], wm), b                           # nearest-neighbour labels: exact
        diff = np.abs(out[b] - want)
        if params[b, 8] == 0 and params[b, 9] == 0 and params[b, 10] == 0:
            assert diff.max() < 2e-5, (b, diff.max())                 # no HSV: affine + gain only (fma contraction of the coordinates)
        else:
            # the HSV round trip is piecewise: a pixel exactly on a hue-sector boundary may take the other branch
            # under different fp32 contraction; everything else agrees to rounding
            assert np.mean(diff > 1e-4) < 1e-3 and np.median(diff) < 1e-6, (b, np.mean(diff > 1e-4))
    assert om[2].sum() == 0
    # identity parameters are exactly ToTensor
    assert np.array_equal(out[0], (img[0].transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)))


def test_augment_kernel_with_the_elastic_displacement_field():
    """ElasticTransform (utils/data.py:124) = its random affine (folded into the one inverse map) + a displacement field, the
    kernel's bilinear look-up into the coarse grid against the numpy restatement: at the reference's alpha = 1 (a field of
    ~0.01 px) and at alpha = 400 (several pixels, so that a wrong sign / axis / frame of the look-up cannot hide); images
    of the batch that did not draw the transform are untouched by the field argument."""
    from oracle import augment_oracle as ao
    from wesup_amd import ops
    from wesup_amd.utils import data as D
    d = torch.device('cuda:0')
    rs = np.random.RandomState(11)
    B, H, W = 4, 72, 100
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([np.stack([(xx * 3 + b * 17) % 256, (yy * 4) % 256, (xx + 2 * yy) % 256], -1) for b in range(B)]).astype(np.uint8)
    mask = ((xx // 9 + yy // 7) % 2).astype(np.uint8)[None].repeat(B, 0)
    cell = D.ELASTIC_CELL
    hc, wc = -(-H // cell), -(-W // cell)
    for alpha in (1.0, 400.0):
        rows, fields, pars = [], np.zeros((B, 2, hc, wc), np.float32), np.zeros((B, 12), np.float32)
        for b in range(B):
            got = []
            while True:                                   # images 0..2 with the transform, image 3 without
                got.clear()
                row, _ = D.sample_params(rs, H, W, True, point_pipeline=False, elastic_out=got)
                if (got[0] is not None) == (b < 3):
                    break
            row[8:11] = 0                                 # geometry + gain only: colours comparable to 2e-5
            rows.append(row)
            if got[0] is not None:
                fields[b], pars[b] = got[0][0] * alpha, got[0][1]
        params = np.stack(rows)
        out, om = ops.augment(torch.from_numpy(img).to(d), torch.from_numpy(mask).to(d), torch.from_numpy(params).to(d),
                              elastic=(torch.from_numpy(fields).to(d), torch.from_numpy(pars).to(d), cell))
        plain, pm = ops.augment(torch.from_numpy(img).to(d), torch.from_numpy(mask).to(d), torch.from_numpy(params).to(d))
        out, om, plain, pm = out.cpu().numpy(), om.cpu().numpy(), plain.cpu().numpy(), pm.cpu().numpy()
        for b in range(B):
            want, wm = ao.augment(img[b], mask[b], params[b], elastic=(fields[b], pars[b], cell))
            assert np.mean(om[b] != wm) < (1e-3 if alpha > 1 else 1e-4), (alpha, b)      # a label may sit within rounding of a cell edge
            assert np.abs(out[b] - want).max() < (2e-3 if alpha > 1 else 2e-5), (alpha, b, np.abs(out[b] - want).max())
        assert np.array_equal(out[3], plain[3]) and np.array_equal(om[3], pm[3])
        moved = np.abs(out[0] - plain[0]).max()
        assert (moved > 0.02) if alpha > 1 else (moved < 0.02)          # the field does something at 400x, next to nothing at 1x


def test_prefetcher_feeds_the_trainer_from_disk(tmp_path):
    from tests.test_data_cpu import _make_dataset
    from wesup_amd.utils import data as D
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    _make_dataset(str(tmp_path), n=5, H=64, W=64)
    ds = D.get_dataset(tmp_path, train=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
    pf = D.DevicePrefetcher(loader, 'cuda:0', train=True, with_points=True, has_masks=True, seed=0)
    batches = list(pf)
    assert [b[0].shape[0] for b in batches] == [2, 2, 1]
    img, pixel_mask, point_mask = batches[0]
    assert img.shape == (2, 3, 64, 64) and img.dtype == torch.float32 and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    assert pixel_mask.shape == (2, 2, 64, 64) and int(pixel_mask.sum(1).max()) == 1
    assert point_mask.shape == (2, 2, 64, 64) and 0 < int(point_mask.sum()) <= 14
    # validation mode: no augmentation -> the reference's CPU item, and points exactly where the CSV puts them
    pv = D.DevicePrefetcher(torch.utils.data.DataLoader(ds, batch_size=1), 'cuda:0', train=False, seed=0)
    vimg, vmask, vpts = next(iter(pv))
    raw = ds[0]
    ref_img, ref_mask = ds.to_reference_item(raw)
    assert torch.allclose(vimg[0].cpu(), ref_img, atol=1e-7, rtol=0) and torch.equal(vmask[0].cpu().long(), ref_mask)
    want = torch.zeros(2, 64, 64, dtype=torch.uint8)
    for x, y, c in raw[2][raw[2][:, 2] >= 0].tolist():
        want[c, y, x] = 1
    assert torch.equal(vpts[0].cpu(), want)
    # and a training iteration straight from the prefetcher (GPU SLIC inside preprocess)
    trainer = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    trainer.optimizer, _ = trainer.get_default_optimizer()
    trainer.metric_funcs = [accuracy, dice]
    trainer.tracker.train()
    for data in pf:
        trainer.train_one_iteration('train', *data)
    hist = trainer.tracker.history
    assert len(hist['loss']) == 3 and all(np.isfinite(hist['loss']))
    # SLIC one batch ahead on the copy stream: label maps + exact counts arrive with the batch and the step is the same
    # step (same augmentation stream, same deterministic SLIC) as with the segmentation inside preprocess
    t2 = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    t2.model.load_state_dict(initialize_trainer('wesup', device='cuda:0', sp_area=64).model.state_dict())
    ta = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    ta.model.load_state_dict(t2.model.state_dict())
    for t in (t2, ta):
        t.optimizer, _ = t.get_default_optimizer()
        t.metric_funcs = [accuracy, dice]
        t.tracker.train()
    pf_in = D.DevicePrefetcher(loader, 'cuda:0', train=True, with_points=True, has_masks=True, seed=3)
    pf_ahead = D.DevicePrefetcher(loader, 'cuda:0', train=True, with_points=True, has_masks=True, seed=3,
                                  segment_fn=ta.prefetch_segment_fn())
    for data in pf_in:
        t2.train_one_iteration('train', *data)
    n_batches = 0
    for data in pf_ahead:
        assert len(data) == 4 and isinstance(data[3], D.LabelMaps) and data[3].labels.shape == data[0].shape[:1] + (64, 64)
        assert all(isinstance(c, int) and c == int(data[3].labels[b].max()) + 1 for b, c in enumerate(data[3].counts))
        ta.train_one_iteration('train', *data)
        n_batches += 1
    assert n_batches == 3
    assert np.allclose(t2.tracker.history['loss'], ta.tracker.history['loss'], rtol=1e-5)


def test_superpixel_inference_on_a_directory(tmp_path):
    """infer.py mirror: fixed input size, multi-scale average + opening, saving, challenge metrics."""
    from PIL import Image
    from tests.test_data_cpu import _make_dataset
    from wesup_amd import infer as I
    from wesup_amd.models import initialize_trainer
    _make_dataset(str(tmp_path / 'val'), n=2, H=72, W=88, with_points=False)
    trainer = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    preds = I.infer(trainer, tmp_path / 'val', output_dir=tmp_path / 'out', input_size=(64, 64), device='cuda:0')
    assert len(preds) == 2 and preds[0].shape == (72, 88) and set(np.unique(preds[0])) <= {0.0, 1.0}
    saved = np.asarray(Image.open(tmp_path / 'out' / 'im00.png'))
    assert saved.shape == (72, 88) and np.array_equal(saved > 0, preds[0] > 0)
    ds = I.SegmentationDataset(tmp_path / 'val', train=False)
    multi = I.predict(trainer, ds, scales=(0.5, 1.0), device='cuda:0')
    assert multi[1].shape == (72, 88) and set(np.unique(multi[1])) <= {0.0, 1.0}
    mean, rows = I.evaluate_predictions(multi, ds)
    assert set(mean) == {'accuracy', 'dice', 'detection_f1', 'object_dice', 'object_hausdorff'} and len(rows) == 2
    assert 0.0 <= mean['accuracy'] <= 1.0 and 0.0 <= mean['object_dice'] <= 1.0
    # same images, same weights, same scale -> same prediction (deterministic GPU SLIC + forward)
    again = I.predict(trainer, ds, scales=(0.5, 1.0), device='cuda:0')
    assert all(np.array_equal(a, b) for a, b in zip(multi, again))


def test_window_inference_matches_per_window_forward(tmp_path):
    """infer_tile.py mirror: every window goes through preprocess -> forward -> postprocess on its own and the merge is
    the reference's running average; the pixel-wise variant merges class-1 probabilities."""
    from PIL import Image
    from wesup_amd import infer_tile as T
    from wesup_amd.models import initialize_trainer
    from wesup_amd.models.wesup import WESUPPixelInference
    from wesup_amd import synth
    H, W, ps = 80, 112, 64
    img = (synth.synth_image(5, H, W).transpose(1, 2, 0) * 255).astype(np.uint8)
    (tmp_path / 'd' / 'images').mkdir(parents=True)
    Image.fromarray(img).save(tmp_path / 'd' / 'images' / 'a.png')
    trainer = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    preds = T.infer(trainer, tmp_path / 'd', ps, output_dir=tmp_path / 'out', device='cuda:0')
    assert len(preds) == 1 and preds[0].shape == (H, W) and preds[0].min() >= 0.0 and preds[0].max() <= 1.0
    # by hand: windows one by one, merged with the pinned function
    singles = []
    with torch.no_grad():
        for patch in T.divide_image_to_patches(img, ps):
            inp, _ = trainer.preprocess(T._to_tensor(patch, 'cuda:0'))
            singles.append(trainer.postprocess(trainer.model(inp)).cpu().numpy()[..., None])
    want = T.combine_patches_to_image(np.concatenate(singles), H, W)
    assert np.array_equal(preds[0], want)
    assert (tmp_path / 'out' / 'a.png').exists()
    # pixel-wise variant
    pm = WESUPPixelInference().to('cuda:0')
    pm.load_state_dict(trainer.model.state_dict())
    prob = T.pixel_predict_array(pm, img, ps, device='cuda:0')
    assert prob.shape == (H, W) and prob.min() >= 0.0 and prob.max() <= 1.0
    one = pm(T._to_tensor(img[:ps, :ps], 'cuda:0')).cpu().numpy()[..., 1]
    # top-left corner pixels are covered by the first window only
    assert np.allclose(prob[:8, :8], one[:8, :8], atol=1e-6)


def test_appearance_kernel_matches_the_numpy_restatement():
    """wesup_appearance (HSV -> brightness/contrast -> CLAHE -> Blur, uint8 at every stage) against
    oracle/augment_oracle.py; sizes that are not multiples of the 8x8 tile grid exercise the reflected padding."""
    from oracle import augment_oracle as ao
    from wesup_amd import ops
    d = torch.device('cuda:0')
    rs = np.random.RandomState(5)
    B, H, W = 6, 75, 100
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.zeros((B, H, W, 3), dtype=np.uint8)
    for b in range(B):                                   # smooth H&E-like gradients + noise: real histograms, not flat ones
        base = np.stack([150 + 60 * np.sin(xx / (9.0 + b)), 90 + 50 * np.cos(yy / (7.0 + b)), 140 + 40 * np.sin((xx + yy) / 11.0)], -1)
        img[b] = np.clip(base + rs.randint(-25, 26, (H, W, 3)), 0, 255).astype(np.uint8)
    rows = np.zeros((B, 8), dtype=np.float32)
    rows[:, 0] = 1.0
    rows[1, 5] = 2.5                                     # CLAHE only
    rows[2, 6] = 1.0                                     # Blur only
    rows[3] = [1.1, -0.05, 12.0, -20.0, 9.0, 3.7, 1.0, 0.0]     # everything
    rows[4] = [0.8, 0.1, -15.0, 25.0, -12.0, 0.0, 0.0, 0.0]     # colour only
    rows[5, 5] = 1.0                                     # the lowest clip limit
    out = ops.appearance(torch.from_numpy(img).to(d), torch.from_numpy(rows).to(d)).cpu().numpy()
    assert np.array_equal(out[0], img[0])                # identity parameters: untouched
    for b in range(1, B):
        want = ao.appearance(img[b], rows[b]).astype(np.int32)
        diff = np.abs(out[b].astype(np.int32) - want)
        if rows[b, 5] == 0 and (rows[b, 2:5] == 0).all():
            assert diff.max() == 0, (b, diff.max())      # integer paths (blur, gain): exact
        else:
            # pow / cbrt of the sRGB <-> Lab round trip and the piecewise HSV differ in the last bit between libm and the
            # device: single grey levels on a small share of the pixels, never more
            assert diff.max() <= 2 and np.mean(diff > 0) < 0.05, (b, diff.max(), np.mean(diff > 0))
    # CLAHE really equalises: the L histogram of a low-contrast image spreads out
    flat = np.clip(128 + rs.randint(-6, 7, (1, 64, 64, 3)), 0, 255).astype(np.uint8)
    r = np.array([[1, 0, 0, 0, 0, 4.0, 0, 0]], dtype=np.float32)
    eq = ops.appearance(torch.from_numpy(flat).to(d), torch.from_numpy(r).to(d)).cpu().numpy()
    assert eq.std() > 1.5 * flat.std()


def test_prefetcher_runs_clahe_blur_and_negative_images(tmp_path):
    """The training pipeline from disk with CLAHE / Blur drawn (p = 0.5 each) and a Digest-2019 'negative' image whose
    pixel mask is its point annotation (utils/data.py:409-512)."""
    from PIL import Image
    from tests.test_data_cpu import _make_dataset
    from wesup_amd.utils import data as D
    _make_dataset(str(tmp_path), n=4, H=64, W=64)
    # a "negative" image: no csv needed, its (all background) mask is the annotation
    Image.fromarray(np.full((64, 64, 3), 200, dtype=np.uint8)).save(tmp_path / 'images' / 'negative_00.png')
    Image.fromarray(np.zeros((64, 64), dtype=np.uint8)).save(tmp_path / 'masks' / 'negative_00.png')
    (tmp_path / 'points' / 'negative_00.csv').write_text('')
    ds = D.get_dataset(tmp_path, train=True)
    assert isinstance(ds, D.Digest2019PointDataset) and len(ds) == 5
    neg = [i for i in range(5) if ds.img_paths[i].name.startswith('negative')][0]
    loader = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, num_workers=0)
    seen_app = False
    for seed in range(4):
        pf = D.DevicePrefetcher(loader, 'cuda:0', train=True, with_points=True, has_masks=True, seed=seed)
        (img, pixel_mask, point_mask), = list(pf)
        assert img.shape == (5, 3, 64, 64) and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
        assert torch.equal(point_mask[neg], pixel_mask[neg]) and int(point_mask[neg, 0].sum()) == 64 * 64
        others = [i for i in range(5) if i != neg]
        assert int(point_mask[others].sum()) < 5 * 64            # radius-0 dots elsewhere
        seen_app = True
    assert seen_app


def test_glas_test_driver_and_scoring(tmp_path):
    """test_glas.py:13-38 + scripts/evaluate_glas.py: checkpoint -> predictions for testA / testB under the record
    directory -> post-processing and challenge metrics."""
    from tests.test_data_cpu import _make_dataset
    from wesup_amd import evaluate as E
    from wesup_amd.models import initialize_trainer
    for split in ('testA', 'testB'):
        _make_dataset(str(tmp_path / 'glas' / split), n=2, H=72, W=88, with_points=False)
    trainer = initialize_trainer('wesup', device='cuda:0', sp_area=64)
    trainer.optimizer, _ = trainer.get_default_optimizer()
    ck = tmp_path / 'record' / 'checkpoints' / 'ckpt.0001.pth'
    trainer.save_checkpoint(ck, epoch=1)
    out = E.test(ck, input_size=(64, 64), device='cuda:0', data_root=tmp_path / 'glas')
    assert out == tmp_path / 'record' / 'results' and sorted(p.name for p in (out / 'testA').iterdir()) == ['im00.png', 'im01.png']
    out2 = E.test(ck, scales=(0.5, 1.0), device='cuda:0', data_root=tmp_path / 'glas')
    assert out2.name == 'results-2scale' and len(list((out2 / 'testB').iterdir())) == 2
    res = E.evaluate_glas(out, tmp_path / 'glas', min_size=20, log=lambda s: None)
    assert set(res) == {'testA', 'testB'} and 0.0 <= res['testA']['accuracy'] <= 1.0
    assert (out / 'testA.csv').exists() and (tmp_path / 'record' / 'results-new' / 'testB' / 'im01.png').exists()
