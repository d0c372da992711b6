"""The recordable walk of the training iteration (wesup_amd/runner.py) and its replay from a step plan (csrc/plan.hip).

Bar: the runner's walk equals the trainer's general path (preprocess -> forward -> compute_loss -> loss.backward() -> step)
BIT FOR BIT -- the same kernels on the same operands in the same stream order --, and a replayed plan equals the walk bit
for bit: loss, metrics, every parameter after every step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(weights, **kw):
    from wesup_amd.models import initialize_trainer
    from wesup_amd.utils.metrics import accuracy, dice
    t = initialize_trainer('wesup', device='cuda:0', **kw)
    t.model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    t.optimizer, t.scheduler = t.get_default_optimizer()
    t.metric_funcs = [accuracy, dice]
    t.model.train()
    t.tracker.train()
    return t


def _batches(n, B, H, W, g, dev):
    from wesup_amd import synth
    out = []
    for i in range(n):
        imgs, labs, pts, pix = synth.make_batch(100 + i, B, H, W, g)
        out.append(tuple(torch.from_numpy(a).to(dev) for a in (imgs, pix, pts, labs)))
    return out


def _flat(t):
    return t.model._flat.detach().clone()


@pytest.mark.parametrize('B,H,W,g', [(2, 96, 96, 6), (1, 80, 112, 5)])
def test_runner_walk_equals_the_general_path_bit_for_bit(B, H, W, g):
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    weights = orc.make_weights(3, feat_scale=0.05)
    data = _batches(2, B, H, W, g, dev)
    a = _trainer(weights, max_superpixels=g * g, native_step=False)          # the trainer's general path
    b = _trainer(weights, max_superpixels=g * g, step_plan=False)            # the runner, never recording
    for i in range(4):
        a.train_one_iteration('train', *data[i % 2])
        b.train_one_iteration('train', *data[i % 2])
        assert b.step_runner().stats['eager'] == i + 1
        for k in ('loss', 'labeled_sp_ratio', 'propagated_labels', 'propagate_loss', 'accuracy', 'dice'):
            assert a.tracker.history[k][-1] == b.tracker.history[k][-1], (i, k)
        assert torch.equal(_flat(a), _flat(b)), f'parameters differ after step {i}'
        assert torch.equal(a.model._flat_grad, b.model._flat_grad)
    assert a.step_runner() is None


@pytest.mark.parametrize('B,H,W,g', [(2, 96, 96, 6), (4, 64, 64, 4)])
def test_replayed_plan_equals_the_walk_bit_for_bit(B, H, W, g):
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    weights = orc.make_weights(5, feat_scale=0.05)
    data = _batches(3, B, H, W, g, dev)
    a = _trainer(weights, max_superpixels=g * g, step_plan=False)
    b = _trainer(weights, max_superpixels=g * g)
    n = 10
    for i in range(n):
        a.train_one_iteration('train', *data[i % 3])
        b.train_one_iteration('train', *data[i % 3])
        assert a.tracker.history['loss'][-1] == b.tracker.history['loss'][-1], i
        for k in ('labeled_sp_ratio', 'propagated_labels', 'propagate_loss', 'accuracy', 'dice'):
            assert a.tracker.history[k][-1] == b.tracker.history[k][-1], (i, k)
        assert torch.equal(_flat(a), _flat(b)), f'parameters differ after step {i}'
    st = b.step_runner().stats
    # the first iteration of a run is walked only (the optimiser's first step is its own launch), the second and third record the
    # plan and its twin, the fourth replays
    assert st['eager'] == 1 and st['recorded'] == 2 and st['replayed'] == n - 3 and st['dropped'] == 0, st
    # what the module carries after an iteration (models/wesup.py:287-292, :529)
    assert b.model.sp_pred is None and b.model.sp_features.shape[0] == B
    # the plan is a few hundred launches and a handful of cuts
    plan = next(iter(b.step_runner().states.values())).plan
    from wesup_amd import _lib
    assert 150 < _lib.load().wesup_plan_kernels(plan.h) < 600 and len(plan.cuts) == 1


def test_plan_is_dropped_when_the_walk_changes_and_nan_raises_before_the_update():
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    B, H, W, g = 2, 64, 64, 4
    weights = orc.make_weights(7, feat_scale=0.05)
    data = _batches(2, B, H, W, g, dev)
    a = _trainer(weights, max_superpixels=g * g, step_plan=False)
    b = _trainer(weights, max_superpixels=g * g, trust_first_recording_after=None)      # (every recording waits for its twin: the counts below)
    for i in range(6):
        a.train_one_iteration('train', *data[i % 2])
        b.train_one_iteration('train', *data[i % 2])
    assert b.step_runner().stats['replayed'] == 3
    for t in (a, b):                                  # a new learning rate: the recorded SGD launch is stale
        t.optimizer.param_groups[0]['lr'] = 1e-4
    for i in range(6):
        a.train_one_iteration('train', *data[i % 2])
        b.train_one_iteration('train', *data[i % 2])
        assert torch.equal(_flat(a), _flat(b)), i
    st = b.step_runner().stats
    assert st['dropped'] == 1 and st['replayed'] == 7, st          # (a dropped plan is recorded again at once: twin, then 4 replays)
    # an engine switch changes the launch list
    b.model.engine.plain = a.model.engine.plain = True
    for i in range(2):
        a.train_one_iteration('train', *data[i % 2])
        b.train_one_iteration('train', *data[i % 2])
        assert torch.equal(_flat(a), _flat(b)), i
    assert b.step_runner().stats['dropped'] == 2
    # NaN in the input while a plan is being replayed: ValueError before the optimiser, weights untouched
    b.model.engine.plain = a.model.engine.plain = False
    for i in range(5):
        b.train_one_iteration('train', *data[i % 2])
    assert b.step_runner().stats['replayed'] >= 5
    before = _flat(b)
    bad = list(data[0])
    bad[0] = bad[0].clone()
    bad[0][0, 0, 3, 3] = float('nan')
    with pytest.raises(ValueError, match='Loss is nan'):
        b.train_one_iteration('train', *bad)
    torch.cuda.synchronize()
    assert torch.equal(before, _flat(b))
    b.train_one_iteration('train', *data[1])                 # and the next clean iteration goes through
    assert np.isfinite(b.tracker.history['loss'][-1])
    # The NaN check sits in front of the FIRST of the optimiser's two launches, i.e. inside the backward pass (round 5): the rest
    # of that iteration's plan -- the stream joins at its end among it -- is not replayed.  The iterations behind it must not
    # notice: the twin that never replays goes through the same sequence (its NaN iteration raises too) and stays bit-equal.
    for i in range(5):
        a.train_one_iteration('train', *data[i % 2])
    with pytest.raises(ValueError, match='Loss is nan'):
        a.train_one_iteration('train', *bad)
    a.train_one_iteration('train', *data[1])
    assert torch.equal(_flat(a), _flat(b))
    for i in range(4):
        a.train_one_iteration('train', *data[i % 2])
        b.train_one_iteration('train', *data[i % 2])
        assert torch.equal(_flat(a), _flat(b)), i


def test_general_path_still_serves_what_the_runner_does_not_cover():
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    B, H, W, g = 1, 64, 64, 4
    weights = orc.make_weights(9, feat_scale=0.05)
    (img, pix, pts, labs), = _batches(1, B, H, W, g, dev)
    t = _trainer(weights, max_superpixels=g * g)
    t.train_one_iteration('train', img, pix, pts)             # no label maps: SLIC inside preprocess, general path
    assert t.step_runner().stats['eager'] == 0
    t.model.eval(); t.tracker.eval()
    t.train_one_iteration('val', img, pix, pts, labs)         # validation: general path
    assert t.step_runner().stats['eager'] == 0
    t.model.train(); t.tracker.train()
    t.train_one_iteration('train', img, pix, pts, labs)
    st = t.step_runner().stats                                # the runner's first iteration (walked; recorded, the optimiser having stepped)
    assert st['eager'] + st['recorded'] == 1 and st['replayed'] == 0


def test_interleaved_shapes_each_get_a_plan_and_keep_it():
    """Multi-scale training (models/wesup.py:178, utils/data.py:98-101): shapes come interleaved.  A shape's plan depends on ITS
    buffer set and on the shared workspaces only -- other shapes' buffers being created while it waits for the twin of its first
    recording must neither block the second recording nor drop a sealed plan; a workspace that grew (a larger shape arrived)
    drops the plans, and they come back.  Results stay bit-identical to a trainer that never replays."""
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    weights = orc.make_weights(11, feat_scale=0.05)
    shapes = [(1, 48, 64, 4), (1, 64, 48, 4), (1, 40, 56, 3)]
    data = [_batches(1, *s, dev)[0] for s in shapes]
    a = _trainer(weights, max_superpixels=16, step_plan=False)
    b = _trainer(weights, max_superpixels=16)
    order = [0, 1, 0, 1, 0, 2, 1, 0, 2, 1, 0, 2, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2]
    for k, i in enumerate(order):
        a.train_one_iteration('train', *data[i])
        b.train_one_iteration('train', *data[i])
        assert torch.equal(_flat(a), _flat(b)), k
    r = b.step_runner()
    assert len(r.states) == 3 and all(st.plan is not None for st in r.states.values()), r.stats
    # a shape is recorded on its first occurrence, confirmed on its second and replayed from its third: only the run's very first
    # iteration (the optimiser's first step) is walked without a recording
    assert r.stats['eager'] == 1 and r.stats['replayed'] >= len(order) - 1 - 2 * 3 - 2, r.stats
    from wesup_amd import ops
    ops.ws_generation += 1                   # what a workspace that had to grow does (ops.workspace): every recorded address is stale
    for k, i in enumerate([0, 1, 2] * 5):
        a.train_one_iteration('train', *data[i])
        b.train_one_iteration('train', *data[i])
        assert torch.equal(_flat(a), _flat(b)), k
    assert all(st.plan is not None for st in r.states.values()), r.stats
    assert r.stats['dropped'] >= 3


def test_clean_first_recordings_are_sealed_without_a_twin_once_two_shapes_are_confirmed():
    """Multi-scale training's first epochs bring a new shape every other step (utils/data.py:98-101).  With
    trust_first_recording_after=2 (opt-in; the default, None, keeps the twin for every shape) the first two shapes of a run
    are recorded twice and compared node by node; after that a first recording during which no workspace grew and the signature
    did not move is sealed at once: the shape's SECOND occurrence replays.  Bit-identical to a trainer that never replays; a
    trainer with trust_first_recording_after=None keeps the twin for every shape."""
    from oracle import wesup_oracle as orc
    dev = torch.device('cuda:0')
    weights = orc.make_weights(12, feat_scale=0.05)
    shapes = [(1, 64, 64, 4), (1, 64, 48, 4), (1, 48, 48, 3), (1, 40, 56, 3), (1, 40, 40, 3)]      # (no later shape needs a larger workspace)
    data = [_batches(1, *s, dev)[0] for s in shapes]
    a = _trainer(weights, max_superpixels=16, step_plan=False)
    b = _trainer(weights, max_superpixels=16, trust_first_recording_after=2, plan_audit_after=10 ** 9)    # (audits: the next test)
    c = _trainer(weights, max_superpixels=16)
    assert c.step_runner().trust_after is None
    order = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 4, 3, 4, 3, 4, 0, 1, 2]
    seen = {}
    for k, i in enumerate(order):
        before = b.step_runner().stats['replayed'] if k else 0
        for t in (a, b, c):
            t.train_one_iteration('train', *data[i])
        assert torch.equal(_flat(a), _flat(b)) and torch.equal(_flat(a), _flat(c)), k
        seen[i] = seen.get(i, 0) + 1
        if i >= 2 and seen[i] == 2:          # the second occurrence of a shape that came after the two confirmed ones: a replay
            assert b.step_runner().stats['replayed'] == before + 1, (k, b.step_runner().stats)
    rb, rc = b.step_runner().stats, c.step_runner().stats
    assert rb.get('trusted', 0) == 3 and rc.get('trusted', 0) == 0, (rb, rc)
    assert rb['replayed'] == rc['replayed'] + 3, (rb, rc)


def test_sealed_plans_are_audited_and_dropped_when_the_walk_would_differ():
    """A plan sealed without a twin (trust_first_recording_after=1) holds raw device addresses nobody has compared with a second walk.
    (1) After AUDIT_AFTER = 1 replay the shape is walked and recorded once more and the recording compared with the sealed plan node
    by node: a clean audit leaves the plan in place (stats['audited']), bit-identical to a trainer that never replays.
    (2) Negative: a shared workspace that grows while a larger shape is interleaved (realistic sizes: a 224 x 160 crop sealed beside a 256 x 256 one, then two
    320 x 320 images; the lazily allocated V / Ybar / filter-panel buffers of the sealed shape's first walk included) moves addresses the sealed plan
    holds -- the plan must be DROPPED on the shape's next occurrence, not replayed.
    (3) Negative: a sealed plan whose launch list no longer matches what the walk does (a node of the plan tampered with, standing
    for anything the hand-kept validity key misses) fails its audit: dropped, stats['distrusted'], a RuntimeWarning."""
    import warnings
    from oracle import wesup_oracle as orc
    from wesup_amd import ops
    dev = torch.device('cuda:0')
    # the shared workspaces are grow-only per process: start from none, as a fresh training process does, so that the large shape
    # below really has to grow them (earlier tests of this session have left larger ones behind)
    torch.cuda.synchronize()
    ops._ws_cache.clear(); ops._ws_first.clear()
    ops.ws_generation += 1
    ops.set_workspace_headroom(1)
    weights = orc.make_weights(21, feat_scale=0.05)
    shapes = [(1, 256, 256, 12), (1, 224, 160, 10), (2, 320, 320, 14)]        # (the middle one: a GlaS crop at the reference's 0.3 - 0.4 scale)
    data = [_batches(1, *s, dev)[0] for s in shapes]
    a = _trainer(weights, step_plan=False)
    b = _trainer(weights, trust_first_recording_after=1)
    for t in (a, b):
        t.kwargs['max_superpixels'] = 256

    def both(i):
        a.train_one_iteration('train', *data[i])
        b.train_one_iteration('train', *data[i])
        assert torch.equal(_flat(a), _flat(b)), i
    r = b.step_runner()
    for i in (0, 0, 0):               # eager (first optimiser step), recorded, twin-confirmed
        both(i)
    assert r.confirmed == 1
    both(1)                           # shape 1: first walk (allocates its set, V, Ybar lazily) recorded ...
    assert r.stats.get('trusted', 0) == 1, r.stats          # ... and sealed without a twin
    n0 = r.stats['replayed']
    both(1)                           # its second occurrence replays the sealed plan
    assert r.stats['replayed'] == n0 + 1
    both(1)                           # (1) the audit: walked, recorded, compared -- clean
    assert r.stats.get('audited', 0) == 1 and r.stats.get('distrusted', 0) == 0 and r.stats['replayed'] == n0 + 1, r.stats
    both(1)
    assert r.stats['replayed'] == n0 + 2
    # (2) a larger shape grows the shared workspaces: every plan recorded before holds stale addresses
    g0, d0 = ops.ws_generation, r.stats['dropped']
    both(2)
    assert ops.ws_generation > g0, 'the larger shape was meant to grow a workspace'
    n1 = r.stats['replayed']
    both(1)
    assert r.stats['dropped'] == d0 + 1 and r.stats['replayed'] == n1, r.stats       # dropped and walked, not replayed
    both(0)
    assert r.stats['dropped'] == d0 + 2 and r.stats['replayed'] == n1, r.stats
    # (3) tamper with a sealed plan: swap it for the plan of ANOTHER shape's walk (same cuts count or not: the diff decides)
    for _ in range(3):
        both(1)                       # recorded again, sealed (or twin-confirmed), replayed
    key1 = next(k for k, st in r.states.items() if k[1:3] == (224, 160))
    key0 = next(k for k, st in r.states.items() if k[1:3] == (256, 256))
    for _ in range(3):
        both(0)
    st1, st0 = r.states[key1], r.states[key0]
    assert st1.plan is not None and st0.plan is not None
    st1.plan, st1.audit_at = st0.plan, st1.replays      # the wrong launch list under shape 1's key; audit due now
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        both(1)                       # the audit walks (correct results) and finds the plan different
    assert r.stats.get('distrusted', 0) == 1 and st1.plan is None, r.stats
    assert any('failed its audit' in str(x.message) for x in w)
    both(1)
    both(1)
    assert r.states[key1].plan is not None      # recorded again (sealed, to be audited again) and replaying


def test_end_to_end_staging_feeds_the_same_step_as_the_general_path():
    """bench.py --end-to-end (and utils/data.py DevicePrefetcher): uint8 batch -> wesup_augment -> wesup_slic on a second stream,
    counts to pinned host memory, LabelMaps as the fourth element of the data tuple, the step through the runner (walked, recorded,
    replayed).  Against the trainer's general path (preprocess -> forward -> compute_loss -> backward -> step) on copies of the
    SAME augmented images, masks and label maps: loss, metrics and every parameter bit for bit, every iteration."""
    from oracle import wesup_oracle as orc
    from wesup_amd import ops, synth
    from wesup_amd.utils import data as D
    dev = torch.device('cuda:0')
    B, H, W = 2, 96, 112
    weights = orc.make_weights(13, feat_scale=0.05)
    a = _trainer(weights, native_step=False)          # the general path
    b = _trainer(weights)                              # the runner
    for t in (a, b):
        t.kwargs['max_superpixels'] = None             # rows = the label maps' own counts (known on the host)
    seg_fn = b.prefetch_segment_fn()
    side = torch.cuda.Stream(device=dev)
    rs = np.random.RandomState(3)
    raw = []
    for i in range(2):
        u8 = np.ascontiguousarray((np.stack([synth.synth_image(50 + 10 * i + k, H, W) for k in range(B)]).transpose(0, 2, 3, 1) * 255)
                                  .astype(np.uint8))
        msk = (rs.random_sample((B, H, W)) > 0.5).astype(np.uint8)
        par = np.stack([D.sample_params(rs, H, W, True)[0] for _ in range(B)])
        pts = torch.zeros(B, 2, H, W, dtype=torch.uint8, device=dev)
        idx = rs.randint(0, min(H, W), (B, 30, 2))
        for k in range(B):
            pts[k, rs.randint(0, 2, 30), idx[k, :, 0], idx[k, :, 1]] = 1
        raw.append((torch.from_numpy(u8).to(dev), torch.from_numpy(msk).to(dev), torch.from_numpy(par).to(dev), pts))

    def stage(i):
        d_img, d_mask, params, pts = raw[i % 2]
        with torch.cuda.stream(side):
            img, pm = ops.augment(d_img, d_mask, params)
            seg, n_dev = seg_fn(img)
            counts = torch.empty(n_dev.shape, dtype=n_dev.dtype).pin_memory()
            counts.copy_(n_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return img, pm, pts, seg, counts, ev

    nxt = stage(0)
    n = 7
    for i in range(n):
        img, pm, pts, seg, counts, ev = nxt
        nxt = stage(i + 1)                             # the next batch is staged beside this step, as in the bench
        torch.cuda.current_stream().wait_event(ev)
        ev.synchronize()
        cnt = [int(v) for v in counts]
        twin = (img.clone(), pm.clone(), pts.clone(), D.LabelMaps(seg.clone(), list(cnt)))
        b.train_one_iteration('train', img, pm, pts, D.LabelMaps(seg, cnt))
        a.train_one_iteration('train', *twin)
        for k in ('loss', 'labeled_sp_ratio', 'propagated_labels', 'propagate_loss', 'accuracy', 'dice'):
            assert a.tracker.history[k][-1] == b.tracker.history[k][-1], (i, k)
        assert torch.equal(_flat(a), _flat(b)), f'parameters differ after step {i}'
        assert torch.equal(a.model._flat_grad, b.model._flat_grad), i
    st = b.step_runner().stats
    # SLIC's superpixel count of an image is the same every time it comes round: two batches, at most two shapes (Kmax), each
    # replayed from its third occurrence
    assert st['eager'] + st['recorded'] + st['replayed'] == n and st['replayed'] >= 1, st
    assert a.step_runner() is None
