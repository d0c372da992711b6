"""Host side of the input pipeline: on-disk format, resize / point rescaling, keypoint
transforms, augmentation oracle invariants, per-rank sharding.  No GPU."""
import csv
import os

import numpy as np
import pytest
import torch


def _make_dataset(root, n=3, H=40, W=56, with_masks=True, with_points=True):
    from PIL import Image
    rs = np.random.RandomState(0)
    for sub in ['images'] + (['masks'] if with_masks else []) + (['points'] if with_points else []):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    truth = []
    for i in range(n):
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, 'images', f'im{i:02d}.png'))
        mask = (rs.random_sample((H, W)) > 0.5).astype(np.uint8)
        if with_masks:
            Image.fromarray(mask).save(os.path.join(root, 'masks', f'im{i:02d}.png'))
        pts = np.stack([rs.randint(0, W, 7), rs.randint(0, H, 7), rs.randint(0, 2, 7)], 1)
        if with_points:
            with open(os.path.join(root, 'points', f'im{i:02d}.csv'), 'w', newline='') as fp:
                csv.writer(fp).writerows(pts.tolist())
        truth.append((img, mask, pts))
    return truth


def test_point_dataset_reads_the_reference_layout(tmp_path):
    from wesup_amd.utils.data import get_dataset, PointSupervisionDataset, NO_CLASS
    truth = _make_dataset(str(tmp_path))
    ds = get_dataset(tmp_path, train=True)
    assert isinstance(ds, PointSupervisionDataset) and len(ds) == 3
    img, mask, pts = ds[1]
    assert img.dtype == torch.uint8 and img.shape == (40, 56, 3) and mask.shape == (40, 56)
    assert np.array_equal(img.numpy(), truth[1][0]) and np.array_equal(mask.numpy(), truth[1][1])
    got = pts.numpy()
    assert np.array_equal(got[:7], truth[1][2]) and (got[7:] == -1).all()
    ref_img, ref_mask = ds.to_reference_item((img, mask, pts))            # utils/data.py:135-152
    assert ref_img.shape == (3, 40, 56) and ref_img.dtype == torch.float32 and float(ref_img.max()) <= 1.0
    assert ref_mask.shape == (2, 40, 56) and ref_mask.dtype == torch.int64
    assert torch.equal(ref_mask.sum(0), torch.ones(40, 56, dtype=torch.int64))
    # no masks: the class-index map is the "no class" value and the reference item carries the empty sentinel
    _make_dataset(str(tmp_path / 'nomask'), with_masks=False)
    ds2 = get_dataset(tmp_path / 'nomask')
    img2, mask2, _ = ds2[0]
    assert int(mask2.min()) == NO_CLASS
    assert len(ds2.to_reference_item((img2, mask2, None))[1].size()) == 0


def test_resize_rescales_points_like_the_reference(tmp_path):
    from wesup_amd.utils.data import PointSupervisionDataset, SegmentationDataset
    truth = _make_dataset(str(tmp_path))
    ds = PointSupervisionDataset(tmp_path, target_size=(80, 84))
    img, mask, pts = ds[0]
    assert img.shape == (80, 84, 3) and mask.shape == (80, 84)
    want = np.floor(truth[0][2] * np.array([[84 / 56, 80 / 40, 1]])).astype(np.int32)      # utils/data.py:340-353
    assert np.array_equal(pts.numpy()[:7], want)
    ds = PointSupervisionDataset(tmp_path, rescale_factor=0.5)
    img, mask, pts = ds[0]
    assert img.shape == (20, 28, 3)
    assert np.array_equal(pts.numpy()[:7], np.floor(truth[0][2] * np.array([[0.5, 0.5, 1]])).astype(np.int32))
    assert set(np.unique(mask.numpy())) <= {0, 1}                                            # order-0 resize keeps labels
    ds = SegmentationDataset(tmp_path, proportion=0.67, seed=3)
    assert len(ds) == 2 and list(ds.picked) == sorted(ds.picked)


def test_augment_oracle_identity_flip_and_keypoints():
    from oracle import augment_oracle as ao
    from wesup_amd.utils import data as D
    rs = np.random.RandomState(1)
    H, W = 24, 30
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    mask = rs.randint(0, 2, (H, W)).astype(np.uint8)
    row, M = D.sample_params(rs, H, W, train=False)
    out, om = ao.augment(img, mask, row)
    assert np.allclose(out, img.transpose(2, 0, 1) / 255.0, atol=1e-6)                       # identity = ToTensor
    assert np.array_equal(om[1], mask) and np.array_equal(om[0], 1 - mask)
    # horizontal flip as an affine map: exact mirror, keypoints land on the mirrored pixel
    Mf = np.array([[-1, 0, W - 1], [0, 1, 0.0]])
    row = np.zeros(12, dtype=np.float32); row[0:3] = [-1, 0, W - 1]; row[3:6] = [0, 1, 0]; row[6] = 1
    out, om = ao.augment(img, mask, row)
    assert np.allclose(out, img[:, ::-1].transpose(2, 0, 1) / 255.0, atol=1e-6) and np.array_equal(om[1], mask[:, ::-1])
    pts = np.array([[3, 5, 1], [29, 0, 0], [0, 23, 1]])
    q = D.transform_points(pts, Mf, H, W)
    assert np.array_equal(q, np.array([[26, 5, 1], [0, 0, 0], [29, 23, 1]]))
    assert np.array_equal(q, ao.transform_points(pts, Mf, H, W))
    # a point pushed outside the image is dropped; padded rows (-1) are ignored
    Ms = np.array([[1, 0, 10.0], [0, 1, 0.0]])
    assert len(D.transform_points(np.array([[25, 2, 0], [-1, -1, -1]]), Ms, H, W)) == 0
    # random training parameters: inverse and forward maps agree, gains inside the albumentations limits
    for _ in range(20):
        row, M = D.sample_params(rs, H, W, train=True)
        Minv = np.array([row[0:3], row[3:6]], dtype=np.float64)
        full = np.vstack([M, [0, 0, 1]]) @ np.vstack([Minv, [0, 0, 1]])
        assert np.allclose(full, np.eye(3), atol=1e-4)
        assert 0.7 <= row[6] <= 1.3 and abs(row[7]) <= 0.3 and abs(row[8]) <= 20 and abs(row[9]) <= 30
    # brightness/contrast only: img*alpha + beta*255 clipped
    row = np.zeros(12, dtype=np.float32); row[0:3] = [1, 0, 0]; row[3:6] = [0, 1, 0]; row[6] = 1.2; row[7] = -0.1
    out, _ = ao.augment(img, None, row)
    assert np.allclose(out, np.clip(img.transpose(2, 0, 1) * 1.2 - 25.5, 0, 255) / 255.0, atol=1e-5)
    # HSV round trip with zero shifts is skipped; a pure value shift moves max(r,g,b) by that amount
    row[6], row[7], row[10] = 1.0, 0.0, 10.0
    out, _ = ao.augment(img, None, row)
    v0 = img.max(-1).astype(np.float32)
    assert np.allclose(out.max(0) * 255.0, np.clip(v0 + 10, 0, 255), atol=1e-2)


def test_rank_sharding_partitions_an_epoch():
    from wesup_amd.ddp import shard_indices
    n, world = 37, 4
    shards = [shard_indices(n, r, world, seed=5, epoch=2) for r in range(world)]
    assert len({len(s) for s in shards}) == 1
    assert set(sum(shards, [])) == set(range(n))
    assert shards != [shard_indices(n, r, world, seed=5, epoch=3) for r in range(world)]


def test_area_v2_compound_and_negative_datasets(tmp_path):
    """The remaining datasets of the reference's utils/data.py: AreaConstraintDataset (:168-277), WESUPV2Dataset
    (:378-406), CompoundDataset (:515-528) and the Digest-2019 'negative' rule (:462-499)."""
    from PIL import Image
    from wesup_amd.utils import data as D
    truth = _make_dataset(str(tmp_path), n=3, H=40, W=56)
    fracs = [float(t[1].mean()) for t in truth]
    with open(tmp_path / 'area.csv', 'w', newline='') as fp:
        w = csv.writer(fp)
        w.writerow(['img', 'area'])
        w.writerows([[f'im{i:02d}.png', fracs[i]] for i in range(3)])
    ds = D.AreaConstraintDataset(tmp_path)
    img, mask, pts, area = ds[1]
    assert img.shape == (40, 56, 3) and area.dtype == torch.float32
    assert torch.allclose(area, torch.tensor([fracs[1], fracs[1]], dtype=torch.float32))       # equality: (a, a)
    ds = D.AreaConstraintDataset(tmp_path, area_type='integer', constraint='individual', margin=0.1)
    n_pos = int(truth[2][1].sum())
    assert ds[2][3].tolist() == [int(n_pos * 0.9), int(n_pos * 1.1)] and ds[2][3].dtype == torch.int64
    ds = D.AreaConstraintDataset(tmp_path, constraint='common')
    assert ds[0][3].tolist() == pytest.approx([min(fracs), max(fracs)])
    ds = D.AreaConstraintDataset(tmp_path, area_type='integer', constraint='common', target_size=(20, 28))
    assert ds[0][3].tolist() == [int(min(fracs) * 560), int(max(fracs) * 560)]
    # WESUPV2: per-pixel label maps from spl-masks/*.npy and the reference's coordinate map
    os.makedirs(tmp_path / 'spl-masks')
    rs = np.random.RandomState(1)
    spl = [(rs.random_sample((40, 56, 2)) > 0.5).astype(np.int64) for _ in range(3)]
    for i, m in enumerate(spl):
        np.save(tmp_path / 'spl-masks' / f'im{i:02d}.npy', m)
    v2 = D.WESUPV2Dataset(tmp_path, train=False)
    img, mask, coords = v2[2]
    assert mask.shape == (2, 40, 56) and mask.dtype == torch.int64 and np.array_equal(mask.numpy(), spl[2].transpose(2, 0, 1))
    x, y = np.linspace(0, 1, 40), np.linspace(0, 1, 56)
    want = np.stack([np.tile(x, 56), np.repeat(y, 40)]).astype(np.float32).reshape(2, 40, 56)     # utils/data.py:385-392
    assert coords.shape == (2, 40, 56) and np.array_equal(coords.numpy(), want)
    # Compound: items in lock step
    both = D.CompoundDataset(D.SegmentationDataset(tmp_path), v2)
    assert len(both) == 3 and len(both[1]) == 2 and torch.equal(both[1][0][0], both[1][1][0])
    # the prefetcher stages raw (img, class-index mask, points) items only: WESUPV2 / Compound items are refused by
    # name instead of being mis-read (their third element is a coordinate map, not points)
    collate = torch.utils.data.default_collate
    with pytest.raises(TypeError, match='mask must be a uint8 class-index map'):
        D.DevicePrefetcher._check_item(collate([v2[0], v2[1]]))
    with pytest.raises(TypeError, match='CompoundDataset'):
        D.DevicePrefetcher._check_item(collate([both[0], both[1]]))
    D.DevicePrefetcher._check_item(collate([D.SegmentationDataset(tmp_path)[0], D.SegmentationDataset(tmp_path)[1]]))
    # Digest 2019: an image named negative* carries the sentinel row instead of csv points
    Image.fromarray(np.zeros((40, 56, 3), dtype=np.uint8)).save(tmp_path / 'images' / 'negative1.png')
    Image.fromarray(np.zeros((40, 56), dtype=np.uint8)).save(tmp_path / 'masks' / 'negative1.png')
    (tmp_path / 'points' / 'negative1.csv').write_text('')
    dg = D.get_dataset(tmp_path, train=True)
    assert isinstance(dg, D.Digest2019PointDataset)
    k = [p.name for p in dg.img_paths].index('negative1.png')
    assert dg[k][2][0].tolist() == [-2, -2, -2] and int(dg[k][2][1:].max()) == -1
    assert dg[0][2][0].tolist() == truth[0][2][0].tolist()
    # validation data is always a plain SegmentationDataset (models/wesup.py:443)
    assert type(D.get_dataset(tmp_path, train=False)) is D.SegmentationDataset


def test_elastic_affine_and_appearance_sampling():
    from wesup_amd.utils import data as D
    rs = np.random.RandomState(0)
    E = D.elastic_affine(rs, 60, 80)
    # an affine map that moves the three control points by at most alpha_affine = 50 pixels each
    c, sq = np.array([30.0, 40.0]), 20
    pts1 = np.array([c + sq, [c[0] + sq, c[1] - sq], c - sq])
    moved = pts1 @ E[:2, :2].T + E[:2, 2]
    assert E.shape == (3, 3) and np.allclose(E[2], [0, 0, 1]) and np.abs(moved - pts1).max() <= 50.0
    assert np.abs(moved - pts1).max() > 1.0
    draws = [D.sample_appearance(rs, True) for _ in range(400)]
    clips = [c for c, _ in draws if c > 0]
    assert 0.4 < len(clips) / 400 < 0.6 and 1.0 <= min(clips) and max(clips) <= 4.0          # CLAHE p = 0.5, clip in [1, 4]
    assert 0.4 < np.mean([b for _, b in draws]) < 0.6                                          # Blur p = 0.5
    assert D.sample_appearance(rs, False) == (0.0, 0.0)
    # mask pipelines draw the elastic affine (p = 0.5), point pipelines never do: their maps stay similarity transforms
    for _ in range(20):
        _, M = D.sample_params(rs, 64, 64, True, point_pipeline=True)
        A = M[:, :2]
        assert abs(abs(np.linalg.det(A)) - np.linalg.norm(A[0]) ** 2) < 1e-9


def test_appearance_oracle_invariants():
    """oracle/augment_oracle.py appearance(): identity, blur of a constant image, CLAHE keeps a constant image constant
    and stretches a low-contrast one, Lab round trip within a grey level or two."""
    from oracle import augment_oracle as ao
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    ident = np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32)
    assert np.array_equal(ao.appearance(img, ident), img)
    const = np.full((40, 48, 3), 137, dtype=np.uint8)
    blur = ident.copy(); blur[6] = 1
    assert np.array_equal(ao.appearance(const, blur), const)
    b = ao.appearance(img, blur).astype(np.int32)
    assert abs(b[10, 10, 0] - int(np.rint(img[9:12, 9:12, 0].mean()))) <= 0
    flat = np.clip(128 + rs.randint(-6, 7, (64, 64, 3)), 0, 255).astype(np.uint8)
    cl = ident.copy(); cl[5] = 4.0
    assert ao.appearance(flat, cl).std() > 1.5 * flat.std()
    L8, a8, b8 = ao.rgb8_to_lab8(img.astype(np.float32))
    back = np.abs(ao.lab8_to_rgb8(L8, a8, b8) - img)          # 8-bit Lab quantises a and b: a few levels on saturated colours
    assert back.max() <= 20 and back.mean() < 1.0 and np.percentile(back, 99) <= 8


def test_elastic_displacement_field():
    """ElasticTransform's second part (utils/data.py:124, albumentations ElasticTransform(alpha=1, sigma=50)): the coarse
    grid the kernel interpolates equals gaussian_filter(per-pixel U(-1,1) noise, 50) on the same noise to < 0.001 px; the
    field is tiny at the reference's parameters (which is why round 2 dropped it -- it is modelled now); the 12 floats
    that place it hold the transform's forward affine and the linear part of its inverse."""
    from oracle import augment_oracle as ao
    from wesup_amd.utils import data as D
    H, W = 120, 200
    rs = np.random.RandomState(3)
    st = rs.get_state()
    f = D.elastic_field(rs, H, W)
    assert f.shape == (2, 15, 25) and f.dtype == np.float32
    rs.set_state(st)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    got = ao.elastic_displacement(f, D.ELASTIC_CELL, xx, yy)
    for a in range(2):
        exact = ao.gaussian_filter_reflect(rs.rand(H, W) * 2.0 - 1.0, 50.0)      # what the reference computes per pixel
        assert np.abs(got[a] - exact).max() < 1e-3
        assert 0.001 < exact.std() < 0.01 and np.abs(exact).max() < 0.05
    # alpha scales it; the sampler hands the field over only for the mask pipeline's ElasticTransform draws
    rs = np.random.RandomState(5)
    assert np.allclose(D.elastic_field(np.random.RandomState(9), 64, 64, alpha=300.0), 300.0 * D.elastic_field(np.random.RandomState(9), 64, 64), rtol=1e-5)
    n_on = 0
    for _ in range(40):
        out = []
        row, M = D.sample_params(rs, 64, 80, True, point_pipeline=False, elastic_out=out)
        assert len(out) == 1
        if out[0] is not None:
            n_on += 1
            field, par = out[0]
            E = np.array([par[0:3], par[3:6], [0, 0, 1]], dtype=np.float64)
            assert par[10] == 1.0 and np.allclose(np.linalg.inv(E)[:2, :2].reshape(-1), par[6:10], atol=1e-6)
            assert field.shape == (2, 8, 10)
        out = []
        D.sample_params(rs, 64, 80, True, point_pipeline=True, elastic_out=out)
        assert out == [None]                                   # the point pipeline has no ElasticTransform
    assert 8 <= n_on <= 32                                     # p = 0.5
    # the oracle's augment with a field: identity geometry + a constant displacement of (+1, 0) px reads the right neighbour
    img = np.random.RandomState(1).randint(0, 256, (16, 24, 3)).astype(np.uint8)
    row = np.zeros(12, dtype=np.float32); row[0] = row[4] = row[6] = 1.0
    par = np.zeros(12, dtype=np.float32); par[0] = par[4] = par[6] = par[9] = par[10] = 1.0
    field = np.zeros((2, 2, 3), dtype=np.float32); field[0] = 1.0
    out, _ = ao.augment(img, None, row, elastic=(field, par, 8))
    want = np.concatenate([img[:, 1:], img[:, -2:-1]], 1).transpose(2, 0, 1) / 255.0         # reflect-101 at the right border
    assert np.allclose(out, want, atol=1e-6)
