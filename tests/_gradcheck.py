"""Gradient yardstick for the GPU tests: an evaluation of the oracle's training step (any dtype) in which every
DISCONTINUOUS decision of the network -- the sign of each pre-activation (ReLU: the 13 conv layers and the three fc
layers) and the arg-max of each 2x2 pooling window -- is taken from the GPU run instead of being re-decided.

Why: a pre-activation that fp64 puts within fp32 rounding of zero can land on the other side of the ReLU in fp32 (seen:
+4.4e-7 vs -1.0e-7 on activations of O(1)); that unit's gradient is then dropped and every layer below moves by up to
~5e-3 of its max.  That is a property of fp32 + ReLU, not of a kernel (torch's own fp32 CPU path flips on other inputs).
Round 1 met it with a loose 2e-2 bound whenever a flip was detected, which could hide a real regression.  Here the
two questions are separated and both are answered tightly:

  1. every decision on which the GPU and the fp64 evaluation disagree is NAMED and must be a genuine near-tie
     (|pre-activation| resp. the gap between the two window candidates within ``tie_tol`` x the layer's max);
  2. GIVEN the same decisions, every parameter gradient must agree with fp64 to 1e-4 of the tensor's max -- a fixed bar
     (round 4 also accepted "no worse than twice the error of torch's CPU fp32 evaluation under the same decisions", which
     reached 2.9e-3 on one tensor while the worst GPU error ever observed is 1.7e-6: a bar that pinned nothing; the torch
     fp32 evaluation is now an optional yardstick that is recorded, never a bar).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import wesup_oracle as orc

POOLED = [l for l in range(12) if orc.POOL_AFTER[l]]

# Budget of near-tie decisions (ReLU signs, pooling arg-max, fc ReLUs, propagation rows) that the GPU may take differently from the
# fp64 evaluation, per million pre-activations of the case, and the floor for small cases.  Round 6: 10 x the worst rate observed on
# the round's green run (profiles/r06_tolerances.json, class 'near-tie decisions differing from fp64'), instead of n_units // 20000
# (= 50 per million: 12 000 at configs[1], where the observed count is a handful).  Every count is recorded through tests/_tol.py.
NEAR_TIE_BUDGET_PER_M = 3.0       # observed: 0.27 per million at 480^2 (17 per image, 68 at configs[1]), 0.19 - 0.22 at 800^2 / 1024^2
NEAR_TIE_FLOOR = 20               # small cases (< 7 M pre-activations): 0 - 2 observed


def _windows(t):
    """(1,C,H,W) -> (1,C,H//2,W//2,4) the 2x2 windows in torch's scan order (0,0),(0,1),(1,0),(1,1)."""
    _, C, H, W = t.shape
    hh, ww = H // 2 * 2, W // 2 * 2
    return t[:, :, :hh, :ww].reshape(1, C, hh // 2, 2, ww // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(1, C, hh // 2, ww // 2, 4)


class _Named(list):
    """The named disagreements: at most 64 examples per (kind, layer, image) are listed, but EVERY disagreeing unit is
    tested for being a near-tie (``total`` counts them all, and every unit that is not a near-tie is listed)."""
    total = 0


def _examples(named, idx_all, near_all, make):
    """idx_all: (n, k) indices of all disagreeing units, near_all: (n,) bool.  Lists the first 64 and, beyond those, every
    unit that is NOT a near-tie (up to 16), so that the assertion in check_gradients sees the full set."""
    named.total += int(idx_all.shape[0])
    rows = list(range(min(64, idx_all.shape[0])))
    rows += [r for r in (~near_all).nonzero().flatten().tolist() if r >= 64][:16]
    for r in rows:
        named.append(make(idx_all[r].tolist(), bool(near_all[r])))


def forced_step(weights, imgs, segs, masks, y_gpu, dtype=torch.float64, tie_tol=2e-5, mlp_gpu=None, pseudo_gpu=None,
                **loss_kw):
    """y_gpu: 13 tensors (B,C,h,w), the GPU's pre-ReLU conv outputs; mlp_gpu: the GPU's three fc-layer outputs
    (post-ReLU: only their sign pattern is used), each (B, Kmax, width); pseudo_gpu: the label rows (B, Kmax, C) the
    GPU's propagation produced (labelled rows first, then the propagated pseudo labels): the third discrete decision
    (``max_sim > threshold`` and the arg-max among the labelled rows).  Returns (loss, grads, disagreements) where
    disagreements is a list of dicts naming every unit whose decision differs from this evaluation's own."""
    B = imgs.shape[0]
    w = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in weights.items()}
    named = _Named()
    cur = {'b': 0}

    def backbone(wd, x):
        b = cur['b']
        outs, h = [], x
        for li, (idx, off) in enumerate(zip(orc.CONV_IDX, orc.SIDE_OFF)):
            y = F.conv2d(h, wd[f'backbone.{idx}.weight'], wd[f'backbone.{idx}.bias'], padding=1)
            outs.append(F.conv2d(y, wd[f'side_conv{off}.weight'], wd[f'side_conv{off}.bias']))
            yg = y_gpu[li][b:b + 1].to(dtype)
            yd = y.detach()
            scale = float(yd.abs().max())
            on = yg > 0
            diff = (yd > 0) != on
            if bool(diff.any()):
                _examples(named, diff.nonzero(), yd[diff].abs() <= tie_tol * scale,
                          lambda ix, near: dict(kind='relu', layer=li, image=b, c=ix[1], h=ix[2], w=ix[3],
                                                ref=float(yd[0, ix[1], ix[2], ix[3]]), gpu=float(yg[0, ix[1], ix[2], ix[3]]),
                                                near_tie=near))
            h = y * on.to(dtype)                                   # ReLU with the GPU's signs (gradient: the same mask)
            if li in POOLED:
                pick = _windows(yg).argmax(dim=-1, keepdim=True)   # first maximum, as torch and the kernel scan
                own = _windows(yd).argmax(dim=-1, keepdim=True)
                win = _windows(h)
                d2 = (pick != own)
                if bool(d2.any()):
                    wv = _windows(yd)
                    gap_all = (wv.gather(-1, pick) - wv.gather(-1, own)).abs()[d2]
                    # a window whose maximum is not positive passes no gradient either way (ReLU)
                    dead_all = (wv.max(dim=-1, keepdim=True).values <= 0)[d2]
                    ix_all = d2.nonzero()
                    _examples(named, ix_all, dead_all | (gap_all <= tie_tol * scale),
                              lambda ix, near: dict(kind='pool', layer=li, image=b, c=ix[1], h=ix[2], w=ix[3],
                                                    ref=int(own[0, ix[1], ix[2], ix[3], 0]), gpu=int(pick[0, ix[1], ix[2], ix[3], 0]),
                                                    gap=float((wv[0, ix[1], ix[2], ix[3], int(pick[0, ix[1], ix[2], ix[3], 0])]
                                                               - wv[0, ix[1], ix[2], ix[3], int(own[0, ix[1], ix[2], ix[3], 0])]).abs()),
                                                    near_tie=near))
                h = win.gather(-1, pick).squeeze(-1)
        return outs

    def mlp(wd, sp_feat):
        """orc.mlp_head with the GPU's ReLU pattern of the three fc layers."""
        b = cur['b']
        h = sp_feat
        for li, k in enumerate((0, 2, 4)):
            pre = F.linear(h, wd[f'fc_layers.{k}.weight'], wd[f'fc_layers.{k}.bias'])
            on = (mlp_gpu[li][b, :pre.shape[0]] > 0).to(pre.device)
            pd = pre.detach()
            diff = (pd > 0) != on
            if bool(diff.any()):
                scale = float(pd.abs().max())
                _examples(named, diff.nonzero(), pd[diff].abs() <= tie_tol * scale,
                          lambda ix, near: dict(kind='fc-relu', layer=k, image=b, c=ix[1], h=ix[0], w=0, ref=float(pd[ix[0], ix[1]]),
                                                gpu=float(mlp_gpu[li][b, ix[0], ix[1]]), near_tie=near))
            h = pre * on.to(dtype)
        pred = F.softmax(F.linear(h, wd['classifier.0.weight'], wd['classifier.0.bias']), dim=1)
        return h, pred

    saved, saved_mlp = orc.backbone_side_outputs, orc.mlp_head
    orc.backbone_side_outputs = backbone
    if mlp_gpu is not None:
        orc.mlp_head = mlp
    total = 0.0
    try:
        for b in range(B):                                         # one image at a time: the graph of one image is
            cur['b'] = b                                           # several GB at 480x480 in fp64
            o = orc.forward_image(w, torch.from_numpy(imgs[b]).to(dtype), torch.from_numpy(segs[b].astype(np.int64)),
                                  None if masks is None else torch.from_numpy(masks[b].astype(np.int64)))
            pp = o['pp']
            if pseudo_gpu is not None and pp['n_l'] < pp['K'] and loss_kw.get('enable_propagation', True):
                thr = loss_kw.get('propagate_threshold', 0.8)
                n, n_l = pp['K'], pp['n_l']
                y_own, W_ul, max_sim, _ = orc.label_propagate(o['sp_features'], pp['sp_labels'], thr, return_aux=True)
                y_gpu_b = pseudo_gpu[b, n_l:n].to(y_own.dtype)
                differ = (y_gpu_b != y_own).any(dim=1)
                if bool(differ.any()):
                    top2 = W_ul.topk(min(2, W_ul.shape[1]), dim=1).values
                    close = ((max_sim - thr).abs() < 1e-5) | ((top2[:, 0] - top2[:, -1]).abs() < 1e-5)
                    _examples(named, differ.nonzero(), close[differ],
                              lambda ix, near: dict(kind='propagate', layer=-1, image=b, c=0, h=ix[0], w=0,
                                                    ref=float(max_sim[ix[0]]), gpu=float(y_gpu_b[ix[0]].sum()), near_tie=near))
                loss_b = orc.cross_entropy(o['sp_pred'][:n_l], pp['sp_labels'])
                loss_b = loss_b + loss_kw.get('propagate_weight', 0.5) * orc.cross_entropy(o['sp_pred'][n_l:], y_gpu_b)
            else:
                loss_b = orc.compute_loss(o['sp_pred'], o['sp_features'], pp['sp_labels'], **loss_kw)
            if loss_b.requires_grad:
                (loss_b / B).backward()
            total += float(loss_b.detach()) / B
            del o, loss_b
    finally:
        orc.backbone_side_outputs, orc.mlp_head = saved, saved_mlp
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in w.items()}
    return total, grads, named


def gpu_preactivations(engine):
    """The 13 pre-ReLU conv outputs of the engine's last forward as (B,C,h,w) CPU tensors."""
    b = engine._last
    return [y.detach().permute(0, 3, 1, 2).float().cpu() for y in b.y]


def gpu_mlp_outputs(engine):
    """The three fc-layer outputs (post-ReLU) of the engine's last forward as (B, Kmax, width) CPU tensors."""
    b = engine._last
    B = b.shape[0]
    return [t.detach().float().cpu().view(B, -1, t.shape[-1]) for t in (b.h1, b.h2, b.feats)]


def check_gradients(model, weights, imgs, segs, masks, names=None, tol=1e-4, tie_tol=2e-5, case=None, yardstick=False, **loss_kw):
    """Assert points 1 and 2 of the module docstring for the gradients the model holds after a backward pass.
    Returns (worst relative error, number of named disagreements)."""
    ys = gpu_preactivations(model.engine)
    hs = gpu_mlp_outputs(model.engine)
    pseudo = None
    meta = getattr(model, '_last_meta', None)
    if masks is not None and meta is not None and loss_kw.get('enable_propagation', True):
        from wesup_amd import ops
        b = model.engine._last
        y_all, _, _ = ops.propagate(b.feats.view(meta.B, meta.Kmax, -1).contiguous(), meta,
                                    loss_kw.get('propagate_threshold', 0.8))
        pseudo = y_all.cpu()
    _, g64, named = forced_step(weights, imgs, segs, masks, ys, torch.float64, tie_tol, mlp_gpu=hs, pseudo_gpu=pseudo, **loss_kw)
    bad = [n for n in named if not n['near_tie']]
    assert not bad, f'decisions that differ from fp64 without being near-ties: {bad[:5]}'
    # a handful of near-tie decisions per million units is what fp32 rounding produces; thousands would be a kernel
    # regression hiding behind the tie tolerance
    n_units = sum(int(y.numel()) for y in ys)
    budget = max(NEAR_TIE_FLOOR, int(np.ceil(NEAR_TIE_BUDGET_PER_M * n_units / 1e6)))
    import _tol
    ok = _tol.within(case or 'unnamed', 'near-tie decisions differing from fp64 (count)', named.total, budget,
                     f'units whose discrete decision (ReLU sign, pooling arg-max, fc ReLU, propagation row) the GPU takes differently from '
                     f'the fp64 evaluation, every one verified to be a near-tie; budget = max({NEAR_TIE_FLOOR}, {NEAR_TIE_BUDGET_PER_M} per '
                     f'million pre-activations)')
    _tol.within(case or 'unnamed', 'near-tie decisions differing from fp64 (per million pre-activations)', named.total / (n_units / 1e6),
                max(NEAR_TIE_BUDGET_PER_M, NEAR_TIE_FLOOR / (n_units / 1e6)), 'the same count as a rate')
    assert ok, (named.total, budget, n_units)
    g32 = None
    if yardstick:
        _, g32, _ = forced_step(weights, imgs, segs, masks, ys, torch.float32, tie_tol, mlp_gpu=hs, pseudo_gpu=pseudo, **loss_kw)
    worst = 0.0
    for k in (names or list(g64)):
        ref = g64[k]
        scale = float(ref.abs().max())
        got = model._grad_views[k].double().cpu()
        if scale == 0:
            assert float(got.abs().max()) == 0.0, k
            continue
        e_gpu = float((got - ref).abs().max()) / scale
        assert e_gpu < tol, (k, e_gpu, [(n['kind'], n['layer']) for n in named][:8])
        worst = max(worst, e_gpu)
        if case is not None:
            import _tol
            _tol.within(case, 'gradients vs fp64 (GPU decisions)', e_gpu, tol, 'max |g - g64| / max |g64| per parameter tensor; fixed bar 1e-4')
            if g32 is not None:
                e_cpu = float((g32[k].double() - ref).abs().max()) / scale
                _tol.within(case, 'torch CPU fp32 gradients vs fp64 (yardstick, no bar)', e_cpu, 1.0, 'the same measure for torch\'s own fp32 path')
    check_gradients.last_named = named
    check_gradients.last_g64 = g64          # (callers that hold the reference's own fp32 gradients measure ITS distance from these)
    return worst, named.total
