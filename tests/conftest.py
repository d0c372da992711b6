import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def load_full_size_case(golden_dir, name):
    """Compact full-size golden (oracle/make_golden.py full_size_golden): the reference's outputs plus the seeds of
    its inputs, which are regenerated here from wesup_amd.synth and verified against the stored checksums."""
    import numpy as np
    from wesup_amd import synth
    fx = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    H, W, g, seed = int(fx['H']), int(fx['W']), int(fx['g']), int(fx['seed'])
    assert str(fx['mode']) == 'point20'
    fx['img'] = synth.synth_image(seed, H, W)
    fx['seg'] = synth.voronoi_labels(seed, H, W, g)
    fx['mask'] = synth.point_mask(seed, fx['seg'], 0.2, 2)
    assert float(fx['img'].astype(np.float64).sum()) == float(fx['img_sum']), 'synthetic image generator drifted'
    assert int(fx['seg'].astype(np.int64).sum()) == int(fx['seg_sum']) and int(fx['mask'].sum()) == int(fx['mask_sum'])
    fx['post_pred'] = np.unpackbits(fx['post_pred_bits'])[:H * W].reshape(1, H, W).astype(np.int8)
    return fx


def pytest_sessionfinish(session, exitstatus):
    """The tolerances the parity tests pinned in this session (tests/_tol.py), per tensor class: bar used, worst observed."""
    import json
    try:
        import _tol
    except ImportError:
        return
    if not _tol.RECORDS:
        return
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'tolerances.json'), 'w') as f:
        json.dump({'per_class': _tol.summary(), 'records': _tol.RECORDS}, f, indent=1)
