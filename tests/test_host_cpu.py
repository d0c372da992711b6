"""Host-side mirror of the reference interface, checked without a GPU: config defaults, factory errors,
state_dict key names / parameter count, metric formulas, history bookkeeping, synthetic data contract, and the
"no CPU fallback" rule (the product path raises instead of silently running eager PyTorch)."""
import os

import numpy as np
import pytest
import torch

from oracle import wesup_oracle as orc
from wesup_amd import synth
from wesup_amd.models import initialize_trainer, WESUP, WESUPConfig
from wesup_amd.models.base import BaseConfig
from wesup_amd.utils import empty_tensor, is_empty_tensor, underline
from wesup_amd.utils.history import HistoryTracker
from wesup_amd.utils import metrics as M


def test_config_defaults_match_reference():
    c = WESUPConfig().to_dict()                      # models/wesup.py:142-179, models/base.py:16-36
    assert c['rescale_factor'] == 0.5 and c['multiscale_range'] == (0.3, 0.4) and c['n_classes'] == 2
    assert c['class_weights'] == (3, 1) and c['sp_area'] == 200 and c['sp_compactness'] == 40
    assert c['enable_propagation'] is True and c['propagate_threshold'] == 0.8 and c['propagate_weight'] == 0.5
    assert c['momentum'] == 0.9 and c['weight_decay'] == 0.001 and c['freeze_backbone'] is False
    assert c['batch_size'] == 1 and c['epochs'] == 300 and c['epsilon'] == 1e-7
    assert BaseConfig().to_dict() == {'batch_size': 1, 'epochs': 10, 'epsilon': 1e-7}


def test_factory_and_module_names():
    with pytest.raises(ValueError):
        initialize_trainer('unet', device='cpu')          # models/__init__.py:17
    model = WESUP()
    keys = list(model.state_dict().keys())
    assert keys == orc.param_names()                       # SURVEY.md 8(b) state_dict keys
    assert sum(p.numel() for p in model.parameters()) == 18868194
    assert model.fm_channels_sum == 2112
    assert model.side_conv1856.weight.shape == (256, 512, 1, 1)
    assert model.classifier[0].weight.shape == (2, 32)     # always 2-way (models/wesup.py:230)


def test_no_cpu_fallback():
    model = WESUP()
    img = torch.zeros(1, 3, 16, 16)
    with pytest.raises(RuntimeError, match='HIP'):
        model((img, torch.ones(1, 16, 16)))                # CPU model: the HIP path refuses, no eager fallback
    trainer = initialize_trainer('wesup', device='cpu')
    with pytest.raises(RuntimeError):
        trainer.compute_loss(None, (None, torch.zeros(1, 2)))      # loss before forward (models/wesup.py:498-500)


def test_metric_formulas_and_sentinel():
    P = torch.randint(0, 2, (3, 20, 24))
    G = torch.randint(0, 2, (3, 20, 24))
    sums = np.stack([[float((P[b] == G[b]).sum()), float((P[b] * G[b]).sum()), float(P[b].sum()), float(G[b].sum())]
                     for b in range(3)])
    assert abs(M.accuracy_from_sums(sums, 20 * 24) - np.mean([orc.accuracy(P[b], G[b]) for b in range(3)])) < 1e-7
    assert abs(M.dice_from_sums(sums) - np.mean([orc.dice(P[b], G[b]) for b in range(3)])) < 1e-6
    assert abs(M.accuracy(P[0], G[0]) - orc.accuracy(P[0], G[0])) < 1e-7
    assert abs(M.dice(P, G) - orc.dice(P, G)) < 1e-6
    assert is_empty_tensor(empty_tensor()) and not is_empty_tensor(torch.zeros(1))
    assert underline('ab') == 'ab\n--'


def test_history_tracker(tmp_path):
    t = HistoryTracker(tmp_path / 'history.csv')
    t.start_new_epoch(5e-5)
    t.train()
    t.step({'loss': 1.0, 'accuracy': 0.5})
    t.step({'loss': 3.0, 'accuracy': 0.7})
    t.eval()
    t.step({'accuracy': 0.9})
    assert t.history['val_accuracy'] == [0.9]
    t.train()
    assert 'loss = 2.0000' in t.log().lower()
    t.save()
    t.start_new_epoch(5e-5)
    t.step({'loss': 2.0, 'accuracy': 0.6})
    t.eval(); t.step({'accuracy': 0.8}); t.train()
    t.save()
    assert 'accuracy' in t.report()
    with pytest.raises(RuntimeError):
        HistoryTracker().save()


def test_synthetic_inputs_are_reproducible_and_well_formed():
    lab = synth.voronoi_labels(3, 96, 80, 7)
    assert lab.dtype == np.int32 and lab.min() == 0 and lab.max() == 48 and len(np.unique(lab)) == 49
    assert np.array_equal(lab, synth.voronoi_labels(3, 96, 80, 7))
    pts = synth.point_mask(3, lab, 0.25, 2, tie_every=2)
    assert pts.shape == (2, 96, 80) and pts.max() == 1 and pts.sum() >= 12
    sk = synth.skewed_labels(3, 128, 128, 16)
    areas = np.bincount(sk.ravel())
    assert areas.min() > 0 and areas.max() > 20 * np.median(areas)
    from wesup_amd.utils.data import get_dataset
    ds = get_dataset('synthetic:32:32:4:3/train')
    img, pix, p, seg = ds[0]
    assert img.shape == (3, 32, 32) and pix.shape == (2, 32, 32) and seg.shape == (32, 32) and len(ds) == 3
    with pytest.raises(FileNotFoundError):                      # a real path is read from disk (tests/test_data_cpu.py)
        get_dataset('/data/glas/train')


def test_window_functions_match_reference():
    """infer_tile.py:23-91 (window corners, division, running-average merge) against outputs of the reference's own
    functions (tests/golden/tiles.npz, oracle/make_golden.py tiles)."""
    import os
    from wesup_amd import infer_tile as T
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tiles.npz'))
    for i in range(5):
        H, W, ps = (int(v) for v in g[f'shape{i}'])
        rs = np.random.RandomState(100 + i)
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        coords = np.array(list(T._get_top_left_coordinates(H, W, ps)), dtype=np.int64)
        assert np.array_equal(coords, g[f'coords{i}'])
        patches = T.divide_image_to_patches(img, ps)
        assert patches.dtype == np.uint8 and patches.shape == (len(coords), ps, ps, 3)
        assert np.array_equal(patches.reshape(len(coords), -1).sum(1).astype(np.int64), g[f'patch_sums{i}'])
        preds = rs.rand(patches.shape[0], ps, ps).astype(np.float64)
        combined = T.combine_patches_to_image(preds, H, W)
        assert combined.shape == (H, W) and np.allclose(combined, g[f'combined{i}'], rtol=0, atol=1e-6)
        if f'patches{i}' in g.files:
            assert np.array_equal(patches, g[f'patches{i}'])
            cc = T.combine_patches_to_image(patches.astype(np.float64), H, W)
            assert cc.shape == (H, W, 3) and np.allclose(cc, g[f'combined_c{i}'], rtol=0, atol=1e-4)
    # the merge of the windows of an image is the image again (every pixel is an average of equal values)
    img = np.random.RandomState(0).randint(0, 256, size=(70, 53, 3)).astype(np.uint8)
    back = T.combine_patches_to_image(T.divide_image_to_patches(img, 32).astype(np.float64), 70, 53)
    assert np.allclose(back, img, atol=1e-9)
    with pytest.raises(ValueError):
        T.divide_image_to_patches(img, 64)


def test_record_dir_bookkeeping(tmp_path, monkeypatch):
    """utils/record.py:16-107: run directory under $RECORD_ROOT, one params json per (re)start, a source snapshot,
    learning curves from history.csv."""
    from wesup_amd.utils import record
    from wesup_amd.utils.history import HistoryTracker
    monkeypatch.setenv('RECORD_ROOT', str(tmp_path))
    rd = record.prepare_record_dir()
    assert rd.parent == tmp_path and (rd / 'checkpoints').is_dir()
    record.save_params(rd, {'epochs': 3})
    record.save_params(rd, {'epochs': 5})
    assert sorted(p.name for p in (rd / 'params').iterdir()) == ['0.json', '1.json']
    record.copy_source_files(rd)
    assert (rd / 'source' / 'wesup_amd' / 'engine.py').exists() and (rd / 'source' / 'wesup_amd' / 'csrc' / 'gemm.hip').exists()
    assert (rd / 'source' / 'include' / 'wesup_hip.h').exists() and not list((rd / 'source').rglob('*.so'))
    t = HistoryTracker(rd / 'history.csv')
    for epoch in range(3):
        t.start_new_epoch(5e-5)
        t.train(); t.step({'loss': 1.0 / (epoch + 1), 'dice': 0.5 + 0.1 * epoch})
        t.eval(); t.step({'dice': 0.4 + 0.1 * epoch})
        t.save()
    curves = record.plot_learning_curves(rd / 'history.csv')
    assert sorted(p.name for p in curves) == ['dice.png', 'loss.png'] and all(p.stat().st_size > 1000 for p in curves)
    assert 'dice' in t.report() and 'val_dice' in t.report() and 'loss' not in t.report().split('\n', 3)[-1]


def test_default_route_of_the_conv_layers():
    """wesup_amd.engine.default_route: the algorithm per 3x3 layer -- implicit GEMM for the image layer, Winograd F(4x4,3x3)
    from 64 input channels up (measured per layer and pass at 480 / 800 / 1024: profiles/r03*_wino_table.txt); the class
    attributes are what bench.py's A/B flags change."""
    from wesup_amd.engine import CONV_CH, WesupEngine, default_route
    assert [default_route(ci, co, 480, 480, 4) for ci, co in CONV_CH] == [0] + [4] * 12
    assert WesupEngine.WINOGRAD_CONV_MIN_CI == 64 and WesupEngine.WINOGRAD_TILE == 4
    old = WesupEngine.WINOGRAD_CONV_MIN_CI, WesupEngine.WINOGRAD_TILE
    try:
        WesupEngine.WINOGRAD_CONV_MIN_CI, WesupEngine.WINOGRAD_TILE = 128, 2          # round 2's routing
        assert [default_route(ci, co, 60, 60, 4) for ci, co in CONV_CH] == [0, 0, 0] + [2] * 10
    finally:
        WesupEngine.WINOGRAD_CONV_MIN_CI, WesupEngine.WINOGRAD_TILE = old


def test_kernel_timer_replaces_an_impossible_pair():
    """engine.KernelTimer.collect: a pair whose duration is off by tens of milliseconds (a timestamp glitch seen about once per
    thousand pairs on the GPU) is replaced by its class's median and counted; ordinary spread is left alone."""
    from wesup_amd.engine import KernelTimer

    class _Ev:
        def __init__(self, t): self.t = t
        def elapsed_time(self, other): return other.t - self.t

    T = KernelTimer()
    times = [0.010, 0.012, 0.011, 65.0, 0.013, 0.009]                 # ms; the fourth is the glitch
    T.pending = [('side_bwd', _Ev(0.0), _Ev(t), 1.0) for t in times]
    T.pending += [('winograd_gemm', _Ev(0.0), _Ev(t), 2.0) for t in (0.20, 0.25, 6.0)]       # 6 ms is 24x the median: kept
    tot = T.collect()
    assert T.replaced == 1
    ms, n, work = tot['side_bwd']
    assert n == 6 and work == 6.0 and abs(ms - (0.010 + 0.012 + 0.011 + 0.012 + 0.013 + 0.009)) < 1e-9      # (upper median of the six: 0.012)
    ms, n, work = tot['winograd_gemm']
    assert n == 3 and abs(ms - 6.45) < 1e-9
    assert T.pending == [] and len(T._free) == 18


def test_step_runner_defaults_are_conservative_and_side_effect_free(monkeypatch):
    """Round 6: a step runner made by the library does NOT freeze the host application's garbage collector, does NOT
    seal first recordings without their twin, and audits sealed plans after one replay when a caller opts in; the engine exposes the
    seven documented switches and nothing else that is boolean."""
    import gc
    from wesup_amd import runner
    from wesup_amd.engine import WesupEngine
    calls = []
    monkeypatch.setattr(gc, 'freeze', lambda: calls.append('freeze'))
    monkeypatch.setattr(runner, '_frozen', False)
    t = initialize_trainer('wesup', device='cpu')
    r = runner.StepRunner(t)
    assert calls == [] and r.trust_after is None and r.audit_after == 1 and r.audit_every is None
    t2 = initialize_trainer('wesup', device='cpu', gc_freeze=True, trust_first_recording_after=1, plan_audit_every=50)
    r2 = runner.StepRunner(t2)
    runner.StepRunner(t2)                                   # a second runner of the process does not freeze again
    assert calls == ['freeze'] and r2.trust_after == 1 and r2.audit_every == 50
    p = {'w': torch.zeros(4)}
    eng = WesupEngine(p, {'w': torch.zeros(4)})
    public_bools = sorted(k for k, v in vars(eng).items() if isinstance(v, bool) and not k.startswith('_'))
    assert public_bools == ['conv_winograd', 'fuse_pool_bwd', 'fuse_pool_fwd', 'plain', 'two_streams', 'wgrad_winograd'], public_bools
    assert eng.commute_side and eng.dual_transform and eng.compact_masks and eng.gather_side_grad and eng.fuse_unpool
    eng.plain = True
    assert not (eng.commute_side or eng.dual_transform or eng.compact_masks or eng.gather_side_grad or eng.fuse_unpool)
