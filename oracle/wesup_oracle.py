"""CPU oracle for the WESUP training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU with plain torch fp32 ops, the algorithm of the
reference's ``models/wesup.py`` + ``models/base.py:184-211`` so that the HIP
path can be checked against it.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
``wesup_amd`` never does (and fails loudly without its HIP library).

Parity status: PINNED.  The reference has no tests/golden vectors of its own
(SURVEY.md 4), so the pins are outputs of the reference itself, generated in
the build container by ``oracle/make_golden.py`` (which imports
``/root/reference/models/wesup.py`` unmodified) and committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against
them.  Third-party pieces not under /root/reference: torchvision ``vgg16``
(unpinned in requirements.txt:13-14; only its layer list, cfg "D", is restated
here) and skimage ``slic`` (unpinned, outside the path: label maps are inputs).

Every function cites the reference lines it follows.
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F

# torchvision vgg16 cfg "D" (models/wesup.py:199 -> vgg16(...).features):
# conv indices inside nn.Sequential are 0,2,5,7,10,12,14,17,19,21,24,26,28.
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
CONV_CH = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
           (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]
POOL_AFTER = [False, True, False, True, False, False, True, False, False, True, False, False, True]
SIDE_OFF = [0, 32, 64, 128, 192, 320, 448, 576, 832, 1088, 1344, 1600, 1856]
FM_CHANNELS = 2112
EPS = 1e-7          # models/base.py:26


def param_names(D=32):
    """state_dict key order (probed)."""
    names = []
    for i in CONV_IDX:
        names += [f'backbone.{i}.weight', f'backbone.{i}.bias']
    for o in SIDE_OFF:
        names += [f'side_conv{o}.weight', f'side_conv{o}.bias']
    for i in (0, 2, 4):
        names += [f'fc_layers.{i}.weight', f'fc_layers.{i}.bias']
    names += ['classifier.0.weight', 'classifier.0.bias']
    return names


def param_shapes(D=32):
    shp = {}
    for i, (ci, co) in zip(CONV_IDX, CONV_CH):
        shp[f'backbone.{i}.weight'] = (co, ci, 3, 3)
        shp[f'backbone.{i}.bias'] = (co,)
    for o, (ci, co) in zip(SIDE_OFF, CONV_CH):
        shp[f'side_conv{o}.weight'] = (co // 2, co, 1, 1)
        shp[f'side_conv{o}.bias'] = (co // 2,)
    dims = [(FM_CHANNELS, 1024), (1024, 1024), (1024, D)]
    for i, (a, b) in zip((0, 2, 4), dims):
        shp[f'fc_layers.{i}.weight'] = (b, a)
        shp[f'fc_layers.{i}.bias'] = (b,)
    shp['classifier.0.weight'] = (2, D)
    shp['classifier.0.bias'] = (2,)
    return shp


def make_weights(seed, D=32, feat_scale=1.0):
    """Seeded Kaiming-scaled weights as a dict name -> float32 numpy array.

    Pretrained ImageNet weights need network (models/wesup.py:199), parity is
    weight-agnostic, so fixtures/tests/bench use this frozen RandomState stream.
    ``feat_scale`` shrinks fc_layers.4 so that superpixel features sit close
    together and label propagation (threshold on exp(-d^2)) actually fires.
    """
    rs = np.random.RandomState(seed)
    shp = param_shapes(D)
    w = {}
    for name in param_names(D):
        s = shp[name]
        if name.endswith('.bias'):
            w[name] = (rs.randn(*s) * 0.05).astype(np.float32)
            continue
        fan_in = int(np.prod(s[1:]))
        if name.startswith('classifier'):
            std = math.sqrt(1.0 / fan_in)
        else:
            std = math.sqrt(2.0 / fan_in)
        a = (rs.randn(*s) * std).astype(np.float32)
        if name == 'fc_layers.4.weight':
            a *= np.float32(feat_scale)
        w[name] = a
    if feat_scale != 1.0:
        w['fc_layers.4.bias'] = (w['fc_layers.4.bias'] * np.float32(feat_scale)).astype(np.float32)
    return w


def to_torch(weights, requires_grad=False):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad)
            for k, v in weights.items()}


# ----------------------------------------------------------------------------
# a2  _preprocess_superpixels  (models/wesup.py:18-63)
# ----------------------------------------------------------------------------
def preprocess_superpixels_dense(segments, mask=None, epsilon=EPS):
    """Faithful restatement incl. the dense (N,H,W) maps (models/wesup.py:18-63).

    segments: (H,W) int64 ids 0..K-1; mask: (C,H,W) int64 in {0,1} or None.
    Returns (sp_maps (N,H,W) f32, sp_labels (N_l,C) f32 | 0-dim tensor(0)).
    """
    sp_idx_list = segments.unique()                                   # :31
    if mask is not None and mask.dim() != 0:                          # :33
        K = int(segments.max()) + 1
        rows = []
        for i in range(K):                                            # :39-42
            sp_mask = (mask * (segments == i).long()).float()         # :35
            rows.append((sp_mask.sum(dim=(1, 2)) / (sp_mask.sum() + epsilon)).unsqueeze(0))  # :36
        sp_labels = torch.cat(rows)
        labeled = (sp_labels.sum(dim=-1) > 0).nonzero().flatten()     # :45
        unlabeled = (sp_labels.sum(dim=-1) == 0).nonzero().flatten()  # :46
        sp_idx_list = torch.cat([labeled, unlabeled])                 # :47
        sp_labels = sp_labels[labeled]                                # :50
        sp_labels = (sp_labels == sp_labels.max(dim=-1, keepdim=True)[0]).float()  # :51-52
    else:
        sp_labels = torch.tensor(0)                                   # :54 (utils/__init__.py:10-13)
    sp_maps = (segments == sp_idx_list[:, None, None]).squeeze().float()   # :57-58
    sp_maps = sp_maps / sp_maps.sum(dim=(1, 2), keepdim=True)         # :61
    return sp_maps, sp_labels


def preprocess_superpixels(segments, mask=None):
    """Label-map restatement of models/wesup.py:18-63 (no dense maps).

    Returns dict(perm (N,) new row -> old id, inv_perm (K,), area (K,) int64 by
    old id, n_l, sp_labels (n_l, C) f32).  Label fractions are compared through
    integer counts: sum_c/(sum_all+eps) is monotone in sum_c for a fixed
    superpixel, so ``== rowmax`` (:51-52) is the same as ``count == maxcount``.
    """
    seg = segments.reshape(-1).long()
    K = int(seg.max()) + 1
    area = torch.bincount(seg, minlength=K)
    if (area == 0).any():
        raise ValueError('label ids must be contiguous 0..K-1 (empty id gives a NaN map row '
                         'in the reference, models/wesup.py:57-61)')
    if mask is not None and mask.dim() != 0:
        C = mask.shape[0]
        m = mask.reshape(C, -1).long()
        cnt = torch.zeros(K, C, dtype=torch.long)
        for c in range(C):
            cnt[:, c] = torch.bincount(seg, weights=m[c].double(), minlength=K).long()
        is_l = cnt.sum(dim=1) > 0
        labeled = is_l.nonzero().flatten()
        unlabeled = (~is_l).nonzero().flatten()
        perm = torch.cat([labeled, unlabeled])
        cl = cnt[labeled]
        sp_labels = (cl == cl.max(dim=1, keepdim=True)[0]).float()
        n_l = int(labeled.numel())
    else:
        perm = torch.arange(K)
        sp_labels = torch.zeros(0, 2)
        n_l = 0
    inv = torch.empty(K, dtype=torch.long)
    inv[perm] = torch.arange(K)
    return dict(perm=perm, inv_perm=inv, area=area, n_l=n_l, sp_labels=sp_labels, K=K)


# ----------------------------------------------------------------------------
# a4/a5  backbone + side outputs  (models/wesup.py:199-210,246-261,278-281)
# ----------------------------------------------------------------------------
def backbone_side_outputs(w, x):
    """x: (B,3,H,W).  Returns list of 13 side-conv outputs at native scale.

    The hook taps the Conv2d output BEFORE the in-place ReLU and clones it
    (models/wesup.py:253); side conv is 1x1 C -> C/2 (:208-209).
    """
    outs = []
    h = x
    for li, (idx, off) in enumerate(zip(CONV_IDX, SIDE_OFF)):
        y = F.conv2d(h, w[f'backbone.{idx}.weight'], w[f'backbone.{idx}.bias'], padding=1)
        outs.append(F.conv2d(y, w[f'side_conv{off}.weight'], w[f'side_conv{off}.bias']))
        h = F.relu(y)
        if POOL_AFTER[li] and li != 12:          # 13th pool output unused (:279 `_ =`)
            h = F.max_pool2d(h, 2, 2)
    return outs


def feature_maps(w, x):
    """(B,2112,H,W): bilinear align_corners=True upsample + concat (:254-261)."""
    H, W = x.shape[-2:]
    outs = backbone_side_outputs(w, x)
    ups = [F.interpolate(o, (H, W), mode='bilinear', align_corners=True) for o in outs]
    return torch.cat(ups, dim=1)


def pool_labelmap(fm, seg_new, area_new, N):
    """sp_feat[n,c] = sum_{p: row(p)=n} fm[c,p] * (1/area_n)  (models/wesup.py:283-285).

    fm (2112,HW); seg_new (HW,) new-row index per pixel; area_new (N,) float.
    """
    inv_area = (1.0 / area_new.float())
    out = torch.zeros(N, fm.shape[0], dtype=fm.dtype)
    out = out.index_add(0, seg_new, (fm * inv_area[seg_new][None, :]).t())
    return out


def mlp_head(w, sp_feat):
    """fc_layers + classifier (models/wesup.py:213-232,287-292)."""
    h = F.relu(F.linear(sp_feat, w['fc_layers.0.weight'], w['fc_layers.0.bias']))
    h = F.relu(F.linear(h, w['fc_layers.2.weight'], w['fc_layers.2.bias']))
    feats = F.relu(F.linear(h, w['fc_layers.4.weight'], w['fc_layers.4.bias']))
    pred = F.softmax(F.linear(feats, w['classifier.0.weight'], w['classifier.0.bias']), dim=1)
    return feats, pred


# ----------------------------------------------------------------------------
# a10  _cross_entropy  (models/wesup.py:66-96)
# ----------------------------------------------------------------------------
def cross_entropy(y_hat, y_true, epsilon=EPS):
    y_hat = torch.clamp(y_hat, min=epsilon, max=(1 - epsilon))        # :83
    n = torch.sum(y_true.sum(dim=1) > 0).float()                      # :86
    if n.item() == 0:                                                 # :88-89
        return torch.tensor(0.)
    ce = -y_true * torch.log(y_hat)                                   # :91
    return torch.sum(ce) / n                                          # :96


# ----------------------------------------------------------------------------
# a11  _label_propagate  (models/wesup.py:99-139)
# ----------------------------------------------------------------------------
def label_propagate(features, y_l, threshold=0.95, return_aux=False):
    features = features.detach()
    y_l = y_l.detach()
    n_l = y_l.size(0)
    n_u = features.size(0) - n_l
    # :121-126 computes the full NxN matrix; only W[n_l:, :n_l] is used.  The
    # squared distance is restated as a direct difference, summed over k.
    fu = features[n_l:]
    fl = features[:n_l]
    diff = fl.unsqueeze(0) - fu.unsqueeze(1)                          # (n_u, n_l, D): f_j - f_i
    W_ul = torch.exp(-(diff * diff).sum(dim=2))
    y_u = torch.zeros(n_u, y_l.size(1))
    if n_l == 0 or n_u == 0:
        if return_aux:
            return y_u, W_ul, torch.zeros(n_u), torch.zeros(n_u, dtype=torch.long)
        return y_u
    max_sim, src = W_ul.max(dim=1)                                    # :130 (tie -> first index)
    prop = max_sim > threshold                                        # :136 strict
    y_u[prop] = y_l[src[prop]]                                        # :137
    if return_aux:
        return y_u, W_ul, max_sim, src
    return y_u


# ----------------------------------------------------------------------------
# a9  compute_loss  (models/wesup.py:492-531)
# ----------------------------------------------------------------------------
def compute_loss(sp_pred, sp_features, sp_labels, enable_propagation=True,
                 propagate_threshold=0.8, propagate_weight=0.5, metrics=None):
    total_num = sp_pred.size(0)
    labeled_num = sp_labels.size(0)
    if labeled_num < total_num:
        loss = cross_entropy(sp_pred[:labeled_num], sp_labels)                      # :510
        if enable_propagation:
            prop = label_propagate(sp_features, sp_labels, threshold=propagate_threshold)   # :513
            ploss = cross_entropy(sp_pred[labeled_num:], prop)                      # :516
            loss = loss + propagate_weight * ploss                                  # :518
        if metrics is not None:
            metrics['labeled_sp_ratio'] = labeled_num / total_num                   # :521
            if enable_propagation:
                metrics['propagated_labels'] = prop.sum().item()                    # :523
                metrics['propagate_loss'] = ploss.item()                          # :524
    else:
        loss = cross_entropy(sp_pred, sp_labels)                                    # :526
    return loss


# ----------------------------------------------------------------------------
# full forward for one image on a label map (models/wesup.py:263-304)
# ----------------------------------------------------------------------------
def forward_image(w, img, segments, mask=None):
    """img (3,H,W) f32, segments (H,W) int64, mask (C,H,W) int64|None.

    Returns dict with sp_features (N,D), sp_pred (N,2), pred (H,W) class-1
    probability painted back (:294-304 == gather sp_pred[inv_perm[label],1]),
    plus the preprocess dict.
    """
    pp = preprocess_superpixels(segments, mask)
    fm = feature_maps(w, img.unsqueeze(0))[0]
    H, W = img.shape[-2:]
    seg_new = pp['inv_perm'][segments.reshape(-1).long()]
    area_new = pp['area'][pp['perm']]
    sp_feat = pool_labelmap(fm.reshape(FM_CHANNELS, H * W), seg_new, area_new, pp['K'])
    feats, sp_pred = mlp_head(w, sp_feat)
    pred = sp_pred[seg_new, 1].reshape(H, W)
    return dict(pp=pp, fm=fm, sp_in=sp_feat, sp_features=feats, sp_pred=sp_pred, pred=pred)


def batch_loss(w, imgs, segs, masks, **loss_kw):
    """Batch semantics (SURVEY.md 7 'Batching'): B independent images, each
    with the loss the reference computes for it alone; batch loss = mean."""
    losses, outs, mets = [], [], []
    for b in range(imgs.shape[0]):
        o = forward_image(w, imgs[b], segs[b], None if masks is None else masks[b])
        m = {}
        losses.append(compute_loss(o['sp_pred'], o['sp_features'], o['pp']['sp_labels'], metrics=m, **loss_kw))
        outs.append(o)
        mets.append(m)
    return torch.stack([l.reshape(()) for l in losses]).mean(), outs, mets


def sgd_step(params, grads, bufs, lr=5e-5, momentum=0.9, weight_decay=1e-3):
    """torch.optim.SGD semantics (models/wesup.py:445-451): g += wd*p;
    buf = mu*buf + g (first step buf = g); p -= lr*buf."""
    for k in params:
        g = grads[k] + weight_decay * params[k]
        if bufs.get(k) is None:
            bufs[k] = g.clone()
        else:
            bufs[k] = momentum * bufs[k] + g
        params[k] = params[k] - lr * bufs[k]
    return params, bufs


def train_step(weights, imgs, segs, masks, lr=5e-5, momentum=0.9, weight_decay=1e-3, bufs=None, **loss_kw):
    """One training iteration (models/base.py:192-208) on CPU; returns
    (loss, grads dict, new weights dict of numpy, bufs, outs, metrics)."""
    w = to_torch(weights, requires_grad=True)
    loss, outs, mets = batch_loss(w, torch.as_tensor(imgs), torch.as_tensor(segs).long(),
                                  None if masks is None else torch.as_tensor(masks).long(), **loss_kw)
    if loss.requires_grad:
        loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in w.items()}
    p = {k: v.detach() for k, v in w.items()}
    bufs = bufs if bufs is not None else {}
    p, bufs = sgd_step(p, grads, bufs, lr, momentum, weight_decay)
    return float(loss), grads, {k: v.numpy() for k, v in p.items()}, bufs, outs, mets


# ----------------------------------------------------------------------------
# metrics (utils/metrics.py:31-45,112-135) and postprocess (models/wesup.py:533-537)
# ----------------------------------------------------------------------------
def accuracy(P, G):
    return (P == G).float().mean().item()


def dice(S, G, epsilon=EPS):
    S = S.unsqueeze(0) if S.dim() == 2 else S
    G = G.unsqueeze(0) if G.dim() == 2 else G
    S, G = S.float(), G.float()
    d = 2 * (G * S).sum(dim=(1, 2)) / (G.sum(dim=(1, 2)) + S.sum(dim=(1, 2)) + epsilon)
    return d.mean().item()


def forward_image_faithful(w, img, sp_maps):
    """The reference's forward AS IT EXECUTES IT (models/wesup.py:246-304): dense row-normalised (N,H,W) superpixel
    maps, the (C,H,W) feature map grown by ``torch.cat`` after every conv layer (:258-261, the quadratic re-copy),
    pooling as the dense ``torch.mm`` (:283-285), paint-back by ``argmax`` over the maps and a loop of N masked writes
    (:295-302).  img (3,H,W); sp_maps (N,H,W).  Used as the "faithful" CPU baseline variant."""
    H, W = img.shape[-2:]
    fm = None
    h = img.unsqueeze(0)
    for li, (idx, off) in enumerate(zip(CONV_IDX, SIDE_OFF)):
        y = F.conv2d(h, w[f'backbone.{idx}.weight'], w[f'backbone.{idx}.bias'], padding=1)
        o = F.conv2d(y.clone(), w[f'side_conv{off}.weight'], w[f'side_conv{off}.bias'])            # :253
        o = F.interpolate(o, (H, W), mode='bilinear', align_corners=True)                           # :254-255
        fm = o.squeeze() if fm is None else torch.cat((fm, o.squeeze()))                            # :257-261
        h = F.relu(y)
        if POOL_AFTER[li]:
            h = F.max_pool2d(h, 2, 2)          # the 13th pool runs too, its output is dropped (:279)
    n = sp_maps.size(0)
    x = torch.mm(sp_maps.view(n, -1), fm.view(fm.size(0), -1).t())                                  # :281-285
    feats, sp_pred = mlp_head(w, x)                                                                 # :288-292
    ids = sp_maps.view(n, H, W).argmax(dim=0)                                                       # :295
    pred = torch.zeros(H, W, sp_pred.size(1))
    for sp_idx in range(int(ids.max()) + 1):                                                        # :301-302
        pred[ids == sp_idx] = sp_pred[sp_idx]
    return dict(fm=fm, sp_features=feats, sp_pred=sp_pred, pred=pred[..., 1])


def time_cpu_baseline(H=480, W=480, g=14, iters=5, warmup=3, seed=0, threads=None, variant='label_map'):
    """Timed CPU leg for bench.py: config c1 (1 image 480x480, ~200 SP), forward + loss + backward with this oracle.
    variant 'label_map': the label-map restatement (scatter-mean, no dense maps);
    variant 'faithful': the reference's own algorithm (_preprocess_superpixels with its Python loop and dense maps,
    dense mm pooling, incremental cat, argmax paint-back), i.e. what ``train_one_iteration`` of the reference costs on
    this host (models/base.py:192-207 without SLIC and the optimiser step).
    Returns (img/s, threads, sample description)."""
    import os
    from wesup_amd import synth
    if threads is None:
        # (bench.py sweeps the thread count with sweep_cpu_threads and passes the best; 16 is the stand-alone default)
        threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    weights = make_weights(seed, feat_scale=0.05)
    imgs, labs, pts, _ = synth.make_batch(seed, 1, H, W, g)
    w = to_torch(weights, requires_grad=True)
    t_img, t_seg, t_pts = torch.as_tensor(imgs), torch.as_tensor(labs).long(), torch.as_tensor(pts).long()
    ts = []
    for it in range(warmup + iters):
        for v in w.values():
            v.grad = None
        t0 = time.perf_counter()
        if variant == 'faithful':
            sp_maps, sp_labels = preprocess_superpixels_dense(t_seg[0], t_pts[0])
            o = forward_image_faithful(w, t_img[0], sp_maps)
            loss = compute_loss(o['sp_pred'], o['sp_features'], sp_labels)
        else:
            loss, _, _ = batch_loss(w, t_img, t_seg, t_pts)
        loss.backward()
        ts.append(time.perf_counter() - t0)
    ts = ts[warmup:]
    if not ts:                                # (warm-up only: sweep_cpu_threads times the call itself)
        return 0.0, threads, 'warm-up only'
    med = sorted(ts)[len(ts) // 2]
    what = ('preprocess (dense maps) + fwd (cat, dense mm) + loss + bwd' if variant == 'faithful'
            else 'label-map preprocess + fwd (scatter-mean) + loss + bwd')
    sample = (f'{iters} timed steps after {warmup} warm-up, {what}, 1 image {H}x{W}, {g*g} superpixels (config c1), '
              f'median {med:.2f} s')
    return 1.0 / med, threads, sample


def sweep_cpu_threads(candidates=(16, 32, 64), variant='faithful', give_up_s=25.0, **kw):
    """Which thread count is this host's best for the CPU leg?  One warm-up + two timed steps per candidate (enough to rank them);
    a candidate whose warm-up step alone takes longer than ``give_up_s`` is not timed further (torch CPU collapses when
    oversubscribed) and ends the sweep -- larger counts are not tried.  Returns [{threads, s_per_step | skipped}], best first."""
    import os
    pts, ncpu = [], os.cpu_count() or 1
    for th in sorted({min(int(c), ncpu) for c in candidates}):
        t0 = time.perf_counter()
        time_cpu_baseline(iters=0, warmup=1, threads=th, variant=variant, **kw)
        w = time.perf_counter() - t0
        if w > give_up_s:
            pts.append({'threads': th, 'skipped': f'warm-up step took {w:.1f} s'})
            break
        v, _, _ = time_cpu_baseline(iters=2, warmup=0, threads=th, variant=variant, **kw)
        pts.append({'threads': th, 's_per_step': round(1.0 / v, 3)})
    pts.sort(key=lambda d: d.get('s_per_step', float('inf')))
    return pts


# ----------------------------------------------------------------------------
# WESUPPixelInference.forward  (models/wesup.py:382-400) -- SURVEY.md 8(f) row 3
# ----------------------------------------------------------------------------
def pixel_inference(w, img):
    """img (1,3,H,W) -> (H,W,C): fc_layers + classifier applied to every pixel's 2112-vector."""
    H, W = img.shape[-2:]
    fm = feature_maps(w, img)[0]                        # (2112,H,W)
    x = fm.reshape(fm.shape[0], -1)                     # :397
    _, pred = mlp_head(w, x.t())                        # :398
    return pred.view(H, W, -1)                          # :400
