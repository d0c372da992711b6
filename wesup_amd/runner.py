"""The training iteration of the standard case as ONE walk over the C ABI -- recordable and replayable.

``BaseTrainer.train_one_iteration`` (models/base.py:184-211 of the reference: preprocess -> zero_grad -> forward -> loss ->
NaN check -> backward -> step -> evaluate -> tracker) keeps its interface; for the case every benchmark configuration and
the GlaS / CRAG training loops are in -- train phase, label maps given with the batch (tensor or ``LabelMaps``), point /
pixel masks as (B,C,H,W) tensors, accuracy / dice as metrics, the fused SGD -- it hands the iteration to this runner:

* every buffer the iteration touches lives in a per-shape state (inputs are COPIED into it), nothing is allocated per step and
  nothing goes through autograd: the loss gradient is ``wesup_loss_bwd`` with an upstream gradient of one, the engine's
  backward is called directly;
* loss, per-image loss terms, segmentation sums and the superpixel counts sit in one device buffer that ONE copy brings to
  pinned host memory; the host waits for it behind the queued backward, checks for NaN, and only then queues the optimiser
  (``ValueError('Loss is nan!')`` with the weights untouched, models/base.py:202-203);
* the FIRST occurrence of a shape records the walk into a step plan (csrc/plan.hip), the second records it again and the
  two recordings are compared node by node; when they are identical, the third and later iterations of that shape REPLAY the
  plan from C (round 4 walked twice before the first recording: with the reference's ~100 shapes x 85 images per epoch,
  utils/data.py:98-101, the first epochs hardly replayed.  A first walk that allocates -- the shape's buffer set, a workspace
  that grows -- records addresses that are final when the walk ends; the twin either confirms them or replaces the candidate)
  (``wesup_plan_replay``): ~330 launches without Python or ctypes in between.  Host work inside the iteration (the NaN check, a
  gradient bucket handed to RCCL) splits the replay into segments.  Anything that moves a buffer (a workspace that grew, the
  engine's buffer cache evicting the shape) or changes the walk (an engine switch, the learning rate, frozen parameters)
  drops the plan; the Python walk is always there and computes the same thing.

Results are bit-identical to the trainer's general path (tests/test_runner_gpu.py): the same kernels on the same operands in
the same stream order.
"""
import ctypes
import gc
import math
import os

import numpy as np
import torch

from . import _lib, ops
from .utils import is_empty_tensor
from .utils import metrics as M

RB_SLOT = 40          # event slot of the read-back copy (the engine uses 0 .. 27 and 64 ..)
RECORD_AT = 0         # eager iterations of a shape before the first recording
MAX_RECORD_TRIES = 4
TRUST_AFTER = None    # twin-confirmed shapes after which a clean first recording is sealed at once.  None (the default since round 6):
                      # never -- every shape's first recording waits for its twin.  Multi-scale training opts in with the trainer kwarg
                      # trust_first_recording_after=1 (2 left 37 of 94 shapes of a cold run to their twins).
AUDIT_AFTER = 1       # replays of a SEALED plan (no twin ever compared with it) after which the shape is walked and recorded once more
                      # and the recording compared with the sealed plan node by node: equal -> the plan is as good as a twin-confirmed
                      # one; different -> it is dropped, counted in stats['distrusted'] and a warning is raised
MAX_STATES = 256      # shapes with a state (inputs, read-back block, plan) of their own, least recently used first out


_frozen = False


def freeze_startup_objects():
    """gc.collect(); gc.freeze() -- once per process, logged."""
    global _frozen
    if not _frozen:
        _frozen = True
        gc.collect()
        gc.freeze()
        if os.environ.get('WESUP_PLAN_DEBUG'):
            print(f'[step runner] gc.freeze(): {gc.get_freeze_count()} startup objects moved out of the collector')


class _Plan:
    def __init__(self):
        self.h = ctypes.c_void_p()
        _lib.call('wesup_plan_create', ctypes.byref(self.h))
        self.cuts = []        # (node index, callback) in walk order

    def size(self):
        return _lib.load().wesup_plan_size(self.h)

    def __del__(self):
        try:
            if self.h:
                _lib.load().wesup_plan_destroy(self.h)
        except Exception:
            pass


class _State:
    """Everything one shape's iteration reads or writes outside the engine's own buffers."""

    def __init__(self, B, H, W, C, Kmax, dev, has_mask, has_gt):
        f32 = dict(dtype=torch.float32, device=dev)
        self.img = torch.empty(B, 3, H, W, **f32)
        self.labels = torch.empty(B, H, W, dtype=torch.int32, device=dev)
        self.mask = torch.empty(B, C, H, W, dtype=torch.uint8, device=dev) if has_mask else None
        self.gt = torch.empty(B, C, H, W, dtype=torch.uint8, device=dev) if has_gt else None
        self.meta = None
        # read-back block: loss | terms (B,8) | seg (B,4) | NaN flag of the other ranks as floats, then n_sp | n_l | status as
        # int32 (the superpixel preprocessing writes them there): ONE copy brings the block to the host
        self.n_f = 1 + 8 * B + 4 * B + 1
        self.rb = torch.zeros(self.n_f + 3 * B, **f32)
        self.loss = self.rb[0:1]
        self.terms = self.rb[1:1 + 8 * B].view(B, 8)
        self.seg = self.rb[1 + 8 * B:1 + 12 * B].view(B, 4)
        self.rb_flag = self.rb[self.n_f - 1:self.n_f]
        self.rb_counts = self.rb[self.n_f:].view(torch.int32).view(3, B)
        self.host = torch.empty(self.n_f + 3 * B, dtype=torch.float32).pin_memory()
        self.rb_event = None      # data parallel: the copy runs on the NaN flag's stream, marked by a torch event
        self.y_all = torch.empty(B, Kmax, C, **f32)
        self.src = torch.empty(B, Kmax, dtype=torch.int32, device=dev)
        self.sim = torch.empty(B, Kmax, **f32)
        self.dpred = torch.empty(B, Kmax, C, **f32)
        self.one = torch.ones(1, **f32)
        self.count = 0
        self.plan = None          # the plan being replayed
        self.cand = None          # first recording, waiting for its twin
        self.replays = 0          # replays of self.plan so far
        self.audit_at = None      # replay count at which self.plan is walked, recorded and compared again (None: never)
        self.tries = 0
        self.sig = None


class StepRunner:
    def __init__(self, trainer):
        self.t = trainer
        self.states = {}
        self.replay = os.environ.get('WESUP_STEP_PLAN', '1') != '0' and trainer.kwargs.get('step_plan', True)
        self.stats = {'eager': 0, 'recorded': 0, 'replayed': 0, 'dropped': 0}
        self.confirmed = 0                # shapes whose first recording its twin has confirmed
        ta = trainer.kwargs.get('trust_first_recording_after', TRUST_AFTER)
        self.trust_after = None if (ta is None or os.environ.get('WESUP_PLAN_TRUST', '1') == '0') else int(ta)
        # Audits: a sealed plan after AUDIT_AFTER replays (kwarg plan_audit_after), and -- for soak runs / CI, kwarg plan_audit_every or
        # WESUP_PLAN_AUDIT_EVERY=N -- EVERY plan again after each N replays: the generic check behind the hand-kept validity key
        # (_signature, ws_generation, bufs_gen); anything the key misses shows up as a recording that differs from the plan.
        self.audit_after = max(0, int(trainer.kwargs.get('plan_audit_after', AUDIT_AFTER)))
        ae = trainer.kwargs.get('plan_audit_every', os.environ.get('WESUP_PLAN_AUDIT_EVERY'))
        self.audit_every = int(ae) if ae not in (None, '', 0, '0') else None
        self.fuse_head = bool(trainer.kwargs.get('fuse_head', True))      # (A/B: the six head launches of round 4)
        self.split_sgd = bool(trainer.kwargs.get('split_sgd', True))      # (A/B: one optimiser launch behind the whole backward)
        # A shape's first walk allocates its buffer set, state and plan -- a few hundred long-lived Python objects -- and a
        # multi-scale epoch brings a new shape every other step: CPython's collector then runs full collections over everything
        # the process holds (268 K tracked objects with torch imported: 67 ms each, 2 ms per step averaged over 40 new shapes,
        # tools/cold_walk.py).  Everything alive when training starts (modules, the model, the loaders) is moved to the permanent
        # generation once; later collections only look at what the iterations create.  A process-wide side effect, so OPT-IN
        # (``gc_freeze=True``: bench.py's multi-scale mode and the ``python -m wesup_amd.train`` script pass it) and done at most once per process.
        if trainer.kwargs.get('gc_freeze', False):
            freeze_startup_objects()
        self._cuts = None
        self._routes = {}

    # ------------------------------------------------------------------ which iterations take this path
    def parse(self, phase, data):
        """The data tuple of an iteration this runner covers as (img, pixel_mask, point_mask, labels, counts), else None."""
        t = self.t
        if phase != 'train' or len(data) != 4 or t.kwargs.get('check_label_maps', False):
            return None
        from .optim import FusedSGD
        if not isinstance(t.optimizer, FusedSGD) or type(t.model).__name__ != 'WESUP':
            return None
        names = [f.__name__ for f in (t.metric_funcs or [])]
        if not all(n in ('accuracy', 'dice') for n in names):
            return None
        img, pixel_mask, point_mask, seg = data
        counts = None
        if hasattr(seg, 'labels') and hasattr(seg, 'counts'):
            counts, seg = list(seg.counts), seg.labels
        if not (torch.is_tensor(img) and torch.is_tensor(seg) and img.dim() == 4 and seg.dim() == 3):
            return None
        B, _, H, W = img.shape
        if seg.shape != (B, H, W) or B > 256:
            return None
        for m in (pixel_mask, point_mask):
            if not torch.is_tensor(m) or not (is_empty_tensor(m) or (m.dim() == 4 and m.shape[0] == B and m.shape[2:] == (H, W))):
                return None
        if counts is None and not seg.is_cuda:
            counts = [int(seg[b].max()) + 1 for b in range(B)]
        return img, pixel_mask, point_mask, seg, counts

    def _signature(self, eng, B, H, W):
        """Everything a recorded plan depends on besides its own buffer set (bufs_gen) and the shared workspaces (ws_generation):
        the settings that shape the launch list AND the identity of every long-lived allocation whose address the launches carry
        (a plan holds raw device addresses: replaying it over a freed or re-made buffer corrupts silently)."""
        t, o = self.t, self.t.optimizer
        g = o.param_groups[0]
        sw = tuple(sorted((k, v) for k, v in vars(eng).items() if isinstance(v, (bool, int, type(None))) and not k.startswith('_')
                          and k not in ('buf_generation', 'max_cached_shapes', 'max_cached_pixels')))
        red = t.reducer
        m = t.model
        pk = eng._packed
        panels = None if pk is None else tuple(0 if u is None else u.data_ptr() for u in list(pk.uf) + list(pk.ud))
        return (sw, eng.route_fn, type(eng).WINOGRAD_CONV_MIN_CI, type(eng).WINOGRAD_TILE, tuple(sorted(eng._diag_skip)),
                ops.STREAMK, tuple(sorted(ops.DIAG)),
                g['lr'], g['momentum'], g['weight_decay'], o.grad_scale, o._first,
                tuple(p.requires_grad for _, p in t.model._named),
                float(t.kwargs.get('propagate_threshold')), float(t.kwargs.get('propagate_weight')),
                bool(t.kwargs.get('enable_propagation')), float(t.kwargs.get('epsilon')), self.fuse_head, self.split_sgd,
                None if red is None else (id(red), t.world_size, red.bucket_elems, red.force),
                ops._stream().value,
                # the engine object itself (model.to(device) re-makes it and restarts buf_generation), the per-shape routing result
                # and the process-wide rule behind it, the flat parameter / gradient / momentum buffers, the Winograd filter panels
                id(eng), self._route_of(eng, B, H, W), ops.winograd_fused_min_blocks(),
                m._flat.data_ptr(), m._flat_grad.data_ptr(), o._vflat.data_ptr(), panels)

    def _route_of(self, eng, B, H, W):
        """eng.route(B, H, W) as a tuple, remembered per (routing rule, shape): thirteen calls of the rule per iteration otherwise."""
        key = (id(eng), eng.route_fn, eng.conv_winograd, ops.winograd_fused_min_blocks(), B, H, W)
        r = self._routes.get(key)
        if r is None:
            if len(self._routes) > 4 * MAX_STATES:
                self._routes.clear()
            r = self._routes[key] = tuple(eng.route(B, H, W))
        return r

    # ------------------------------------------------------------------ the iteration
    def run(self, parsed):
        t = self.t
        img, pixel_mask, point_mask, seg, counts = parsed
        model = t.model
        model._ensure_engine()
        eng = model.engine
        dev = eng.device
        B, _, H, W = img.shape
        mask = point_mask if not is_empty_tensor(point_mask) else (pixel_mask if not is_empty_tensor(pixel_mask) else None)
        has_gt = not is_empty_tensor(pixel_mask)
        names = [f.__name__ for f in (t.metric_funcs or [])]
        want_seg = has_gt and bool(names)
        C = mask.shape[1] if mask is not None else (pixel_mask.shape[1] if has_gt else 2)
        Kmax = t._kmax(counts, None, H, W)
        key = (B, H, W, C, Kmax, mask is not None, want_seg)
        st = self.states.pop(key, None)
        if st is None:
            while len(self.states) >= MAX_STATES:
                self.states.pop(next(iter(self.states)))
            st = _State(B, H, W, C, Kmax, dev, mask is not None, want_seg)
            if self.states:                           # a second shape: shared workspaces get headroom from now on (ops.workspace)
                ops.set_workspace_headroom(2)
        self.states[key] = st                         # (most recently used last)
        if t.reducer is not None:
            t.reducer.reset()                         # nothing may be left over from an iteration that raised
        # ---- inputs into the state's buffers (dtype conversions ride in the copies)
        # The conv chain needs the image only; label map and masks are read by the side stream (superpixel preprocessing, metrics):
        # their copies go there, behind whatever the caller's stream has queued so far (it may have produced them).
        st.img.copy_(img, non_blocking=True)
        with eng.side_stream():
            st.labels.copy_(seg, non_blocking=True)
            if mask is not None:
                st.mask.copy_(mask, non_blocking=True)
            if want_seg:
                st.gt.copy_(pixel_mask, non_blocking=True)
        st.n_sp_host = counts

        timing = eng.timer.enabled or (t.reducer is not None and t.reducer.profile)
        sig = self._signature(eng, B, H, W) if self.replay else None
        gens = (ops.ws_generation, eng.bufs_gen(B, H, W, st.y_all.shape[1]))
        if st.plan is not None and (st.sig != sig or st.gens != gens):
            st.plan = st.cand = None
            st.count = 0
            self.stats['dropped'] += 1
        metrics = {}
        if st.plan is not None and not timing and st.audit_at is not None and st.replays >= st.audit_at:
            host = self._audit(st, metrics, want_seg, eng, B, H, W, sig, gens)
        elif st.plan is not None and not timing:
            if st.meta is not None:                   # the replay does not re-enter sp_preprocess: this batch's host-side counts
                st.meta.n_sp_host = counts
            host = self._replay(st, metrics)
            st.replays += 1
            self.stats['replayed'] += 1
        else:
            if st.cand is not None and st.sig != sig:       # settings changed since the first recording: it has no twin to wait for
                if os.environ.get('WESUP_PLAN_DEBUG'):
                    print('[step plan] candidate dropped, signature fields that differ:',
                          [i for i, (x, y) in enumerate(zip(st.sig, sig)) if x != y])
                st.cand = None
            # (the very first optimiser step of a run differs from every later one -- buf = g --: not worth recording)
            record = self.replay and not timing and st.count >= RECORD_AT and st.tries < MAX_RECORD_TRIES and not t.optimizer._first
            plan = _Plan() if record else None
            ws_gen0 = ops.ws_generation
            host = self._walk(st, metrics, want_seg, plan)
            st.count += 1
            if plan is not None and ops.ws_generation != ws_gen0:
                # a workspace grew DURING this walk: the launches before that hold the old address, a recording nobody can match
                plan, st.cand = None, None
                self.stats['recorded'] += 1
            elif plan is not None:
                self.stats['recorded'] += 1
                gens = (ops.ws_generation, eng.bufs_gen(B, H, W, st.y_all.shape[1]))
                sig0 = sig
                sig = self._signature(eng, B, H, W)           # the state the recording ENDED in (a first walk allocates the filter
                                                              # panels and clears the optimiser's first-step flag: the twin decides)
                if st.cand is not None and st.gens == gens and _lib.load().wesup_plan_diff(st.cand.h, plan.h) == 0:
                    st.plan, st.cand = plan, None
                    st.replays, st.audit_at = 0, self.audit_every
                    self.confirmed += 1
                elif (self.trust_after is not None and self.confirmed >= self.trust_after and st.cand is None and sig == sig0):
                    # A first recording is sealed without its twin once the run has confirmed TRUST_AFTER shape(s) twin by twin:
                    # what made first recordings unrepeatable were addresses that moved during the walk -- a workspace that grew
                    # (such a walk is discarded above), filter panels or optimiser state made on the way (the signature before and
                    # behind the walk then differ) -- and a shape's own buffer set is allocated before the launches that use it.
                    # Under multi-scale training (a new shape every other step in the first epochs) every shape's second
                    # occurrence replays instead of walking again.
                    st.plan = plan
                    st.replays, st.audit_at = 0, self.audit_after
                    self.stats['trusted'] = self.stats.get('trusted', 0) + 1
                else:
                    if os.environ.get('WESUP_PLAN_DEBUG') and st.cand is None and self.trust_after is not None:
                        print(f'[step plan] first recording not sealed: confirmed {self.confirmed}, signature fields that moved '
                              f'{[i for i, (x, y) in enumerate(zip(sig0 or (), sig)) if x != y]}')
                    if st.cand is not None and st.gens != gens and os.environ.get('WESUP_PLAN_DEBUG'):
                        print(f'[step plan] twin recorded under other generations {st.gens} -> {gens}')
                    if st.cand is not None and st.gens == gens:      # (a workspace that grew in between is nobody's failure)
                        st.tries += 1
                        if os.environ.get('WESUP_PLAN_DEBUG'):
                            k = _lib.load().wesup_plan_diff(st.cand.h, plan.h)
                            print(f'[step plan] recordings differ at node {k - 1}: '
                                  f'{_lib.load().wesup_plan_node_name(plan.h, k - 1)} (sizes {st.cand.size()} / {plan.size()})')
                    st.cand = plan
                st.sig, st.gens = sig, gens
            else:
                self.stats['eager'] += 1
        return self._finish(st, host, metrics, want_seg, names, H, W)

    def _audit(self, st, metrics, want_seg, eng, B, H, W, sig, gens):
        """This occurrence of a shape with a plan is WALKED and recorded instead of replayed, and the recording compared with the
        plan node by node (kernel, grid, stream, every argument byte).  The walk computes the step either way."""
        import warnings
        plan2 = _Plan()
        ws_gen0 = ops.ws_generation
        host = self._walk(st, metrics, want_seg, plan2)
        st.count += 1
        self.stats['audited'] = self.stats.get('audited', 0) + 1
        same = (ops.ws_generation == ws_gen0 and (ops.ws_generation, eng.bufs_gen(B, H, W, st.y_all.shape[1])) == gens
                and self._signature(eng, B, H, W) == sig and len(plan2.cuts) == len(st.plan.cuts)
                and all(a[0] == b[0] for a, b in zip(plan2.cuts, st.plan.cuts))
                and _lib.load().wesup_plan_diff(st.plan.h, plan2.h) == 0)
        if same:
            st.audit_at = None if self.audit_every is None else st.replays + self.audit_every
        else:
            k = _lib.load().wesup_plan_diff(st.plan.h, plan2.h)
            where = 'validity key moved during the walk' if k == 0 else (
                f'node {k - 1} ({_lib.load().wesup_plan_node_name(plan2.h, k - 1)}), sizes {st.plan.size()} / {plan2.size()}')
            warnings.warn(f'wesup step plan of shape {(B, H, W)} failed its audit after {st.replays} replay(s): {where}; the plan is '
                          'dropped and the shape recorded again -- the iterations it replayed may have read stale addresses',
                          RuntimeWarning)
            self.stats['distrusted'] = self.stats.get('distrusted', 0) + 1
            st.plan, st.cand, st.replays, st.audit_at = None, None, 0, None
            st.tries += 1
        return host

    def _cut(self, plan, fn):
        """Host work inside the iteration: run it now, and when recording, remember where the replay has to stop for it."""
        if plan is not None:
            plan.cuts.append((plan.size(), fn))
        return fn()

    def _walk(self, st, metrics, want_seg, plan):
        t = self.t
        model, eng = t.model, t.model.engine
        red = t.reducer
        kw = t.kwargs
        eng._rot = 0
        if plan is not None:
            _lib.call('wesup_plan_begin', plan.h)
        saved_ready = eng.on_grads_ready
        try:
            if red is not None:
                # a gradient range handed to the reducer is host work on the stream that produced it
                def ready(names, _red=red):
                    cur = torch.cuda.current_stream()
                    def go(names=list(names), cur=cur):
                        with eng._On(cur):
                            _red.ready(names)
                    self._cut(plan, go)
                eng.on_grads_ready = ready
            model.prefetch_weights(train=True)                # side stream: conv1_1 only waits for its own panel
            with eng.side_stream():
                st.meta = ops.sp_preprocess(st.labels, st.mask, st.y_all.shape[1], n_classes=st.y_all.shape[2],
                                            n_sp_host=st.n_sp_host, into=st.meta, counts=st.rb_counts)
            eng.frozen = {n for n, p in model._named if not p.requires_grad}
            multi = red is not None and t.world_size > 1
            # One rank: the head of the step -- classifier, label propagation, loss, its gradient, the classifier's backward: six
            # launches on the chain between the fc layers' forward and backward -- in two (ops.head_fwd / head_bwd, bit-identical)
            fuse = self.fuse_head and not multi and ops.head_bwd_supported(st.y_all.shape[1], st.y_all.shape[2])
            feats, sp_pred, pred = eng.forward(st.img, st.meta, train=True, need_paint=multi, head=not fuse)
            if want_seg and multi:
                ops.seg_metrics(pred, st.gt, out=st.seg)
            if fuse:
                b, P = eng._last, eng.p
                ops.head_fwd(feats, P['classifier.0.weight'], P['classifier.0.bias'], b.sp_pred, st.meta,
                             float(kw.get('propagate_threshold')), enable=bool(kw.get('enable_propagation')), out=(st.y_all, st.src, st.sim))
                with eng.side_stream():                                   # (idle between the forward's last pooling and the backward)
                    ops.paint_fwd(sp_pred, st.meta, 1, out=pred)          # (the pixel-wise prediction: metrics and callers only)
                    if want_seg:
                        ops.seg_metrics(pred, st.gt, out=st.seg)
                ops.head_bwd(b.feats, P['classifier.0.weight'], b.sp_pred, st.y_all, st.meta, st.one, float(kw.get('epsilon')),
                             float(kw.get('propagate_weight')), st.terms, st.dpred, b.dfeat, b.cls_part)
                st.rb_event = None
                with eng.side_stream():                                   # the read-back, behind the loss terms
                    _lib.call('wesup_copy_to_host', ctypes.c_void_p(st.host.data_ptr()), ops._p(st.rb), st.rb.numel() * 4, ops._stream())
                    ops.sync_record(RB_SLOT)
            else:
                ops.propagate(feats, st.meta, float(kw.get('propagate_threshold')), enable=bool(kw.get('enable_propagation')),
                              out=(st.y_all, st.src, st.sim))
                # (one rank: the mean over the images is formed on the host from the per-image terms of the read-back block, in the
                # kernel's own order and precision -- no wesup_loss_mean launch on the chain; several ranks need the loss on the
                # device for the NaN flag)
                ops.loss_fwd(sp_pred, st.y_all, st.meta, float(kw.get('epsilon')), float(kw.get('propagate_weight')),
                             out=(st.loss if multi else None, st.terms))
                if multi:
                    # A NaN loss on one rank must stop every rank (models/base.py _loss_flag): a MAX all-reduce of a flag on a
                    # stream of its own, the read-back copy behind it.  Host work (torch collectives), i.e. a cut of the plan.
                    def nan_flag():
                        f, aux = t._loss_flag(st.loss)
                        with torch.cuda.stream(aux):
                            st.rb_flag.copy_(f)
                            st.host.copy_(st.rb, non_blocking=True)
                            st.rb_event = torch.cuda.Event()
                            st.rb_event.record()
                    self._cut(plan, nan_flag)
                else:
                    # The segmentation metrics and the read-back are not on the way to the loss gradient: on the side stream (idle
                    # between the last pooling of the forward and the first side-branch gradient), behind the loss.
                    st.rb_event = None
                    with eng.side_stream():
                        ops.paint_fwd(sp_pred, st.meta, 1, out=pred)      # (the pixel-wise prediction: metrics and callers only)
                        if want_seg:
                            ops.seg_metrics(pred, st.gt, out=st.seg)
                        _lib.call('wesup_copy_to_host', ctypes.c_void_p(st.host.data_ptr()), ops._p(st.rb), st.rb.numel() * 4, ops._stream())
                        ops.sync_record(RB_SLOT)
            if red is not None and red.profile:
                red.t_backward = torch.cuda.Event(enable_timing=True)
                red.t_backward.record()
            if not fuse:
                ops.loss_bwd(sp_pred, st.y_all, st.meta, st.terms, st.one, float(kw.get('epsilon')),
                             float(kw.get('propagate_weight')), out=st.dpred)
            # One rank: the optimiser step in two launches -- everything but the lowest layer's parameters on the weight-gradient
            # stream as soon as those gradients are queued (beside conv1_2's input gradient, the last long kernel of the chain),
            # conv1_1's 1 792 values behind its weight gradient: 60 us less at the end of the step.  The NaN check moves with it.
            early = {}
            if self.split_sgd and red is None:
                def tail(wg, late):
                    eng._edge(eng._side(), wg)
                    early['host'] = self._cut(plan, lambda: self._wait_and_check(st))
                    model._publish_grads()
                    with eng._On(wg):
                        t.optimizer.step_early(late)
                eng.on_tail = tail
            try:
                eng.backward(None, st.dpred, head_done=fuse)
            except BaseException:
                # (the NaN check inside on_tail raised, or anything else did: the weight-gradient and side streams may still run
                # this step's kernels on the set's buffers -- the caller's stream waits for both before anything else is queued,
                # and the engine forgets the half-walked backward.  Gradients are undefined after 'Loss is nan!'.)
                eng.abort_backward()
                raise
            finally:
                eng.on_tail = None
            if red is not None:
                self._cut(plan, red.finish)
            if early:
                host = early['host']
                t.optimizer.step_late()
            else:
                host = self._cut(plan, lambda: self._wait_and_check(st))
                model._publish_grads()                        # p.grad = views of the flat gradient buffer (what step() looks at)
                t.optimizer.step()
        finally:
            eng.on_grads_ready = saved_ready
            if plan is not None:
                _lib.load().wesup_plan_end(plan.h)
        st.feats, st.sp_pred = feats, sp_pred
        self._publish(st)
        return host

    def _replay(self, st, metrics):
        plan = st.plan
        lib = _lib.load()
        pos, host = 0, None
        try:
            for node, fn in plan.cuts:
                if node > pos:
                    _lib.check(lib.wesup_plan_replay(plan.h, pos, node), 'wesup_plan_replay')
                pos = node
                r = fn()
                if isinstance(r, dict):
                    host = r
            _lib.check(lib.wesup_plan_replay(plan.h, pos, plan.size()), 'wesup_plan_replay')
        except BaseException:
            # ('Loss is nan!' at the plan's cut, or a failed replay: the rest of the plan is not queued -- the caller's stream waits
            # for what the side and weight-gradient streams already hold, as the walk does, engine.abort_backward)
            self.t.model.engine.abort_backward()
            raise
        self.t.model._publish_grads()
        self._publish(st)
        return host

    def _wait_and_check(self, st):
        if st.rb_event is not None:
            st.rb_event.synchronize()
        else:
            _lib.call('wesup_sync_synchronize', RB_SLOT)             # <- the host sync of the iteration
        B = st.terms.shape[0]
        raw = st.host.numpy()
        f = raw[:st.n_f].astype(np.float64)
        cnt = raw[st.n_f:].view(np.int32).reshape(3, B)
        if st.rb_event is not None:
            loss = float(f[0])
        else:                               # loss[0] = (sum of terms[b][5] over b ascending, fp32) / B, as loss_mean_kernel forms it
            t32 = raw[1:1 + 8 * B].reshape(B, 8)
            s32 = np.float32(0.0)
            for b in range(B):
                s32 = np.float32(s32 + t32[b, 5])
            loss = float(np.float32(s32 / np.float32(B)))
        host = {'loss': loss, 'terms': f[1:1 + 8 * B].reshape(B, 8), 'seg': f[1 + 8 * B:1 + 12 * B].reshape(B, 4),
                'n_sp': cnt[0].astype(np.float64), 'n_l': cnt[1].astype(np.float64)}
        nan_anywhere = f[st.n_f - 1] if st.rb_event is not None else 0.0
        if math.isnan(host['loss']) or nan_anywhere > 0:
            raise ValueError('Loss is nan!')
        return host

    def _publish(self, st):
        """The attributes the reference's forward leaves on the module (models/wesup.py:287-292); compute_loss clears sp_pred."""
        m = self.t.model
        m.fm_size = (st.img.shape[2], st.img.shape[3])
        m._last_meta = st.meta
        m._padded = (st.feats, st.sp_pred)
        # (models/wesup.py:287-292: one image -> (n_sp, D), as the general path publishes it)
        m.sp_features = st.feats[0, :int(st.n_sp_host[0])] if st.feats.shape[0] == 1 and st.n_sp_host is not None else st.feats
        m.sp_pred = None

    def _finish(self, st, host, metrics, want_seg, names, H, W):
        t = self.t
        t._metrics_from_terms(host, metrics)
        metrics['loss'] = host['loss']
        ev = {}
        if want_seg:
            if 'accuracy' in names:
                ev['accuracy'] = M.accuracy_from_sums(host['seg'], H * W)
            if 'dice' in names:
                ev['dice'] = M.dice_from_sums(host['seg'])
        t.tracker.step({**metrics, **ev})
