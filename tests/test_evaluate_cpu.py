"""GlaS evaluation (scripts/evaluate_glas.py) on the CPU: the small-region post-processing against
outputs of the reference's own function (tests/golden/postprocess.npz, oracle/make_golden.py postprocess), and the
directory driver end to end on synthetic predictions."""
import csv
import os

import numpy as np
import pytest


def test_remove_small_regions_matches_the_reference(golden_dir):
    from wesup_amd.evaluate import remove_small_regions
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    changed = 0
    for i in range(int(g['n'])):
        shape = tuple(int(v) for v in g[f'shape{i}'])
        n = shape[0] * shape[1]
        m = np.unpackbits(g[f'in{i}'])[:n].reshape(shape).astype(np.float64)
        want = np.unpackbits(g[f'out{i}'])[:n].reshape(shape)
        got = remove_small_regions(m.copy())
        assert got.dtype == np.float64 and np.array_equal(got.astype(np.uint8), want), i
        changed += int((want != m).any())
    assert changed >= 3                                  # the cases really exercise both passes


def test_evaluate_glas_driver(tmp_path):
    from PIL import Image
    from wesup_amd.evaluate import evaluate_glas
    from wesup_amd.utils import metrics as M
    rs = np.random.RandomState(0)
    yy, xx = np.mgrid[0:160, 0:200]

    def blobs(k):
        m = np.zeros((160, 200), dtype=np.uint8)
        for j in range(k):
            cy, cx, r = rs.randint(30, 130), rs.randint(30, 170), rs.randint(28, 40)
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = j + 1
        return m
    for split in ('testA', 'testB'):
        os.makedirs(tmp_path / 'pred' / split)
        os.makedirs(tmp_path / 'gt' / split / 'masks')
        for i in range(2):
            gt = blobs(2)
            pred = (gt > 0).astype(np.uint8)
            pred[5:15, 5:15] = 1                          # a 100-pixel speck: must be removed by the post-processing
            Image.fromarray(pred * 255).save(tmp_path / 'pred' / split / f'im{i}.bmp')
            Image.fromarray(gt).save(tmp_path / 'gt' / split / 'masks' / f'im{i}.bmp')
    lines = []
    res = evaluate_glas(tmp_path / 'pred', tmp_path / 'gt', log=lines.append)
    assert set(res) == {'testA', 'testB'} and lines[0] == 'Test A' and any(l.startswith('Object Hausdorff:') for l in lines)
    for split in ('testA', 'testB'):
        assert res[split]['dice'] > 0.999 and res[split]['detection_f1'] == pytest.approx(1.0)     # the speck is gone
        new = np.asarray(Image.open(tmp_path / 'pred-new' / split / 'im0.bmp'))
        assert new[10, 10] == 0 and new.max() == 255
        rows = list(csv.reader(open(tmp_path / 'pred' / f'{split}.csv')))
        assert rows[0] == ['', 'detection_f1', 'object_dice', 'object_hausdorff'] and [r[0] for r in rows[1:]] == ['im0.bmp', 'im1.bmp']
    with pytest.raises(ValueError):
        os.remove(tmp_path / 'gt' / 'testA' / 'masks' / 'im1.bmp')
        evaluate_glas(tmp_path / 'pred', tmp_path / 'gt', log=lambda s: None)
