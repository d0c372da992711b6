"""Host-side mirror of the reference's ``models/wesup.py`` on the HIP kernels.

Same names, argument meaning and error behaviour as the reference so that the
training step is a drop-in; everything device-side goes
through libwesup_hip.so (wesup_amd.ops / wesup_amd.engine).  There is no CPU or
eager-PyTorch fallback: without the HIP library / a GPU these raise.

Additions over the reference surface (all optional):
  * ``SuperpixelMaps``: what ``_preprocess_superpixels`` returns instead of the
    dense (N,H,W) maps -- a label-map based description; ``WESUP.forward``
    accepts it as well as a dense tensor (models/wesup.py:263-275);
  * ``SuperpixelLabels``: what ``preprocess`` returns as ``sp_labels`` -- it behaves
    like the reference's (N_l, C) tensor when looked at and carries the batched
    device-side description so that ``compute_loss`` needs no host sync;
  * batches: B independent images, per-image loss as the reference computes
    it, batch loss = mean (the reference is batch-1 only, models/wesup.py:178).
"""
import os.path as osp
import warnings
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..engine import WesupEngine, CONV_CH, CONV_IDX, POOL_AFTER, SIDE_OFF, FM_CHANNELS
from ..utils import empty_tensor, is_empty_tensor
from .base import BaseConfig, BaseTrainer


class SuperpixelMaps:
    """Label-map replacement of the reference's dense, row-normalised (N,H,W) ``sp_maps``.

    Wraps the device-side preprocessing result for a batch of B label maps.
    ``dense()`` rebuilds the reference tensor for one image on demand (not used
    on the hot path)."""

    def __init__(self, meta):
        self.meta = meta

    def size(self, dim=None):
        n = self.meta.n_sp_host[0] if self.meta.n_sp_host is not None else int(self.meta.n_sp[0])
        s = (n, self.meta.H, self.meta.W)
        return s if dim is None else s[dim]

    def to(self, *a, **k):
        return self

    def dense(self, b=0):
        m = self.meta
        n = int(m.n_sp[b])
        rows = torch.arange(n, device=m.new_row.device, dtype=torch.int32)[:, None]
        maps = (m.new_row[b][None, :] == rows).float()
        maps = maps / m.area_new[b, :n, None].float()
        return maps.view(n, m.H, m.W)


class SuperpixelLabels:
    """``sp_labels`` as ``WESUPTrainer.preprocess`` returns it (models/wesup.py:487-490).

    To a caller it is the reference's (N_l, C) float tensor of the labelled superpixels (labelled rows first,
    multi-hot on ties): ``size``/``shape``/``len``/indexing/``to``/torch functions resolve it lazily (one host sync to
    learn N_l, which the reference pays in ``nonzero()``, models/wesup.py:45).  ``compute_loss`` never looks: it takes
    the batched device-side description from ``.meta`` and stays on the device.  Without any mask the reference's
    value is the 0-dim ``empty_tensor()`` (models/wesup.py:54), and so is this one's."""

    def __init__(self, meta, has_mask=True):
        self.meta = meta
        self.has_mask = has_mask
        self._t = {}

    def tensor(self, b=None):
        m = self.meta
        if b is None:
            if m.B != 1:
                raise ValueError(f'sp_labels of a batch of {m.B} images: ask for one image with .tensor(b)')
            b = 0
        if b not in self._t:
            if not self.has_mask:
                self._t[b] = empty_tensor().to(m.sp_labels.device)
            else:
                self._t[b] = m.sp_labels[b, :int(m.n_l[b])]               # host sync
        return self._t[b]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        conv = lambda a: a.tensor() if isinstance(a, cls) else a
        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in (kwargs or {}).items()})

    def size(self, dim=None):
        t = self.tensor()
        return t.size() if dim is None else t.size(dim)

    shape = property(lambda self: self.tensor().shape)
    dtype = property(lambda self: self.tensor().dtype)
    device = property(lambda self: self.meta.sp_labels.device)

    def dim(self):
        return self.tensor().dim()

    def __len__(self):
        return len(self.tensor())

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def to(self, *a, **k):
        return self                         # already on the device the step runs on

    def __getattr__(self, name):            # sum(), float(), cpu(), numpy(), ... of the resolved tensor
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.tensor(), name)


def _to_mask_u8(mask, B, H, W):
    if mask is None or is_empty_tensor(mask):
        return None
    m = mask
    if m.dim() == 3:
        m = m.unsqueeze(0)
    return m.to(torch.uint8).contiguous()


def preprocess_label_maps(segments, mask=None, Kmax=None, n_sp_host=None, n_classes=2):
    """Batched device-side _preprocess_superpixels (models/wesup.py:18-63) on label maps.

    segments (B,H,W) or (H,W) integer ids 0..K-1 on the GPU; mask (B,C,H,W) / (C,H,W) in {0,1} or None."""
    if segments.dim() == 2:
        segments = segments.unsqueeze(0)
    B, H, W = segments.shape
    labels = segments.to(torch.int32).contiguous()
    if Kmax is None:
        Kmax = int(labels.max().item()) + 1          # host sync, as models/wesup.py:41 `range(segments.max()+1)`
        n_sp_host = n_sp_host or None
    m8 = _to_mask_u8(mask, B, H, W)
    C = m8.shape[1] if m8 is not None else n_classes
    meta = ops.sp_preprocess(labels, m8, int(Kmax), n_classes=C, n_sp_host=n_sp_host)
    return meta


def _preprocess_superpixels(segments, mask=None, epsilon=1e-7):
    """Reference signature (models/wesup.py:18): (H,W) segments [+ (C,H,W) mask] ->
    (sp_maps, sp_labels).  sp_maps is a SuperpixelMaps; sp_labels is the (N_l, C) float tensor
    (labelled superpixels first, multi-hot on ties) or the 0-dim empty tensor without a mask."""
    meta = preprocess_label_maps(segments, mask)
    meta.check()
    if mask is None or is_empty_tensor(mask):
        sp_labels = empty_tensor().to(segments.device)                  # models/wesup.py:54
    else:
        n_l = int(meta.n_l[0])                                          # host sync (reference: nonzero(), :45)
        sp_labels = meta.sp_labels[0, :n_l].clone()
    meta.n_sp_host = [int(meta.n_sp[0])]
    return SuperpixelMaps(meta), sp_labels


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat, y_true, epsilon, class_weights):
        y_hat = y_hat.contiguous()
        y_true = y_true.contiguous().float()
        out = ops.cross_entropy_fwd(y_hat, y_true, epsilon, class_weights)
        ctx.save_for_backward(y_hat, y_true, out)
        ctx.eps, ctx.cw = epsilon, class_weights
        return out[2].clone()

    @staticmethod
    def backward(ctx, dloss):
        y_hat, y_true, out = ctx.saved_tensors
        return (ops.cross_entropy_bwd(y_hat, y_true, out, dloss.reshape(1).contiguous(), ctx.eps, ctx.cw),
                None, None, None)


def _cross_entropy(y_hat, y_true, class_weights=None, epsilon=1e-7):
    """Semi-supervised cross entropy (models/wesup.py:66-96): rows of ``y_true`` that are all zero do
    not count; returns 0 when no row is labelled; ``class_weights`` (C,) scales the per-class terms
    (models/wesup.py:93-94).  No host sync (the reference does one ``.item()``)."""
    if isinstance(y_true, SuperpixelLabels):
        y_true = y_true.tensor()
    if y_hat.size(0) == 0:
        return torch.zeros((), device=y_hat.device)
    if class_weights is not None:
        class_weights = torch.as_tensor(class_weights, dtype=torch.float32, device=y_hat.device).contiguous()
    return _CrossEntropyFn.apply(y_hat, y_true, float(epsilon), class_weights)


def _label_propagate(features, y_l, threshold=0.95):
    """Label propagation over the affinity exp(-|fi-fj|^2) (models/wesup.py:99-139).
    features (N,D) with the n_l labelled rows first, y_l (n_l,C) -> y_u (N-n_l, C)."""
    if isinstance(y_l, SuperpixelLabels):
        y_l = y_l.tensor()
    features = features.detach().contiguous()
    y_l = y_l.detach().float()
    N, D = features.shape
    n_l, C = y_l.shape
    dev = features.device
    if N - n_l <= 0:
        return torch.zeros(0, C, device=dev)
    meta = ops.SuperpixelMeta()
    meta.B, meta.Kmax, meta.C = 1, N, C
    meta.sp_labels = torch.zeros(1, N, C, device=dev)
    meta.sp_labels[0, :n_l] = y_l
    meta.n_sp = torch.tensor([N], dtype=torch.int32, device=dev)
    meta.n_l = torch.tensor([n_l], dtype=torch.int32, device=dev)
    y_all, _, _ = ops.propagate(features.view(1, N, D), meta, threshold, enable=True)
    return y_all[0, n_l:]


class WESUPConfig(BaseConfig):
    """Configuration for WESUP model (defaults of models/wesup.py:142-179)."""

    rescale_factor = 0.5
    multiscale_range = (0.3, 0.4)
    n_classes = 2
    class_weights = (3, 1)
    sp_area = 200
    sp_compactness = 40
    enable_propagation = True
    propagate_threshold = 0.8
    propagate_weight = 0.5
    momentum = 0.9
    weight_decay = 0.001
    freeze_backbone = False
    batch_size = 1
    epochs = 300


def load_backbone_weights(backbone, path):
    """Load ImageNet VGG16 convolution weights into ``backbone`` from a file saved with ``torch.save``: either
    torchvision's ``vgg16().state_dict()`` (keys ``features.N.weight``; the classifier entries are ignored) or the
    ``features`` sub-dict (keys ``N.weight``).  The reference gets the same tensors from
    ``vgg16(pretrained=True)`` (models/wesup.py:199), which needs network access."""
    sd = torch.load(path, map_location='cpu')
    sd = sd.get('state_dict', sd) if isinstance(sd, dict) else sd
    picked = {}
    for k, v in sd.items():
        k = k[len('features.'):] if k.startswith('features.') else k
        if k.split('.')[0].isdigit() and int(k.split('.')[0]) in CONV_IDX:
            picked[k] = v
    missing = [f'{i}.{t}' for i in CONV_IDX for t in ('weight', 'bias') if f'{i}.{t}' not in picked]
    if missing:
        raise ValueError(f'{path}: not a VGG16 "features" state_dict, missing {missing[:4]}...')
    backbone.load_state_dict(picked)


def _vgg16_features():
    """torchvision VGG16 cfg "D" layer list (models/wesup.py:199); random init unless ``backbone_weights`` names a
    file with the ImageNet weights the reference downloads (no network here)."""
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    layers, c = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


class _WesupFn(torch.autograd.Function):
    """Connects the hand-written forward/backward to ``loss.backward()``.  Parameter gradients do not
    travel through autograd: the engine writes them into the model's flat gradient buffer and the
    parameters' ``.grad`` are views of it."""

    @staticmethod
    def forward(ctx, anchor, model, img, meta):
        # grad mode is off inside Function.forward; the anchor needs grad iff apply() ran with grad enabled
        feats, sp_pred, pred = model.engine.forward(img, meta, train=bool(ctx.needs_input_grad[0]))
        ctx.model = model
        ctx.mark_non_differentiable(pred)
        return feats, sp_pred, pred

    @staticmethod
    def backward(ctx, dfeat, dpred, _dpaint):
        model = ctx.model
        if dpred is None:
            dpred = torch.zeros(model.engine.ctx[3], model.engine.ctx[6], 2, device=model.engine.device)
        model.engine.backward(None if dfeat is None else dfeat.contiguous(), dpred.contiguous())
        model._publish_grads()
        return None, None, None, None


class WESUP(nn.Module):
    """Weakly supervised histopathology image segmentation with sparse point annotations
    (mirror of models/wesup.py:182-304)."""

    def __init__(self, n_classes=2, D=32, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self.D = D
        self.backbone = _vgg16_features()
        # vgg16(pretrained=True) of the reference (models/wesup.py:199) downloads the ImageNet weights; here they come
        # from a file (``backbone_weights=<path>``); without one the backbone starts from random weights and the
        # trainer says so loudly before training
        self.pretrained_backbone = False
        if kwargs.get('backbone_weights'):
            load_backbone_weights(self.backbone, kwargs['backbone_weights'])
            self.pretrained_backbone = True
        self.fm_channels_sum = 0
        for layer in self.backbone:
            if isinstance(layer, nn.Conv2d):
                setattr(self, f'side_conv{self.fm_channels_sum}',
                        nn.Conv2d(layer.out_channels, layer.out_channels // 2, 1))
                self.fm_channels_sum += layer.out_channels // 2
        self.fc_layers = nn.Sequential(
            nn.Linear(self.fm_channels_sum, 1024), nn.ReLU(),
            nn.Linear(1024, 1024), nn.ReLU(),
            nn.Linear(1024, D), nn.ReLU())
        # `n_classes` is a named parameter, so kwargs never holds it: the classifier is always
        # 2-way in the reference (models/wesup.py:230).
        self.classifier = nn.Sequential(nn.Linear(D, self.kwargs.get('n_classes', 2)), nn.Softmax(dim=1))
        self.fm_size = None
        self.sp_features = None
        self.sp_pred = None
        self.engine = None
        self._flat = None
        self._flat_grad = None
        self._anchor = None
        self._last_meta = None

    # ------------------------------------------------------------------ flat parameter storage
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flat = None          # parameters were re-created: rebuild flat storage lazily
        self.engine = None
        self._named = None
        return r

    def _ensure_engine(self):
        # (the Parameter objects are fixed once they are views of the flat buffer: their (name, parameter) list is kept -- walking
        # the module tree for it costs 0.1 ms, and the step asked for it five times)
        named = getattr(self, '_named', None)
        if named and self.engine is not None and self._flat is not None and self._flat.device == named[0][1].device:
            return
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if self.engine is not None and self._flat is not None and self._flat.device == dev:
            self._named = list(params.items())
            return
        if dev.type != 'cuda':
            raise RuntimeError('WESUP runs on the HIP kernels only: move the model to a GPU (no CPU fallback)')
        offs, total = {}, 0
        for name, p in params.items():
            offs[name] = total
            total += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        pv, gv = {}, {}
        with torch.no_grad():
            for name, p in params.items():
                o, n = offs[name], p.numel()
                flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                pv[name] = p.data
                gv[name] = gflat[o:o + n].view(p.shape)
        self._flat, self._flat_grad, self._offs = flat, gflat, offs
        self._named = list(params.items())
        self._grad_views = gv
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self.engine = WesupEngine(pv, gv, D=self.D)
        # bounds of the per-shape buffer cache (engine.py _get_bufs): WESUP(..., max_cached_shapes=, max_cached_pixels=)
        for k in ('max_cached_shapes', 'max_cached_pixels'):
            if self.kwargs.get(k) is not None:
                setattr(self.engine, k, int(self.kwargs[k]))

    def _publish_grads(self):
        for name, p in self._named:
            if not p.requires_grad:
                continue
            gv = self._grad_views[name]
            if p.grad is None:
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                p.grad.add_(gv)

    # ------------------------------------------------------------------ forward
    @property
    def feature_maps(self):
        """(2112,H,W) side-output feature maps of the last forward (models/wesup.py:280); a permuted
        view of the engine's pixel-major (H,W,2112) tensor (batched: (B,2112,H,W))."""
        if self.engine is None or self.engine._last is None:
            return None
        fm = self.engine.feature_maps()
        return fm[0].permute(2, 0, 1) if fm.shape[0] == 1 else fm.permute(0, 3, 1, 2)

    def prefetch_weights(self, train=True):
        """Queue the repacking of the weights for the coming forward (see WesupEngine.prefetch_weights)."""
        self._ensure_engine()
        self.engine.prefetch_weights(train and self._anchor.requires_grad)

    def forward(self, x):
        """x = (img (B,3,H,W), sp_maps) with sp_maps a dense (N,H,W) tensor as in the reference (B = 1)
        or a SuperpixelMaps.  Returns the painted class-1 probability, (B,H,W)  (models/wesup.py:263-304)."""
        img, sp_maps = x
        self._ensure_engine()
        if isinstance(sp_maps, SuperpixelMaps):
            meta = sp_maps.meta
        else:
            if img.size(0) != 1:
                raise ValueError('dense sp_maps imply batch size 1 (models/wesup.py:178)')
            n = sp_maps.size(0)
            labels = ops.spmaps_to_labels(sp_maps.contiguous().float())      # models/wesup.py:295
            meta = ops.sp_preprocess(labels, None, n, n_sp_host=[n])        # rows keep the caller's order
        img = img.contiguous().float()
        self.fm_size = (img.size(2), img.size(3))
        # parameters with requires_grad=False (freeze_backbone, models/wesup.py:427-429): the engine skips the frozen
        # backbone layers' wgrad and every dgrad below the lowest trainable one
        self.engine.frozen = {n for n, p in self._named if not p.requires_grad}
        feats, sp_pred, pred = _WesupFn.apply(self._anchor, self, img, meta)
        self._last_meta = meta
        self._padded = (feats, sp_pred)        # (B,Kmax,D), (B,Kmax,2): what the batched loss consumes
        if meta.B == 1 and meta.n_sp_host is not None:
            n = meta.n_sp_host[0]
            self.sp_features, self.sp_pred = feats[0, :n], sp_pred[0, :n]
        else:
            self.sp_features, self.sp_pred = feats, sp_pred
        return pred


class WESUPPixelInference(WESUP):
    """Pixel-wise inference (mirror of models/wesup.py:307-400): the MLP head runs on every
    pixel's 2112-vector instead of on superpixel means.  Reuses the training kernels: backbone + side outputs
    (engine), bilinear upsample into the pixel-major feature map, three fused GEMMs, classifier + softmax."""

    @torch.no_grad()
    def forward(self, x):
        """x (1,3,H,W) -> (H,W,C) class probabilities (models/wesup.py:382-400)."""
        self._ensure_engine()
        if x.size(0) != 1:
            raise ValueError('pixel inference takes one image (models/wesup.py:386)')
        x = x.contiguous().float()
        H, W = x.size(2), x.size(3)
        labels = torch.zeros(1, H, W, dtype=torch.int32, device=x.device)        # one dummy superpixel
        meta = ops.sp_preprocess(labels, None, 1, n_sp_host=[1])
        self.engine.forward(x, meta, train=False, need_paint=False)
        fm = self.engine.feature_maps().view(H * W, FM_CHANNELS)                  # (HW, 2112), pixel-major
        self.fm_size = (H, W)
        p = self.engine.p
        h1 = ops.gemm_nt(fm, p['fc_layers.0.weight'], p['fc_layers.0.bias'], flags=ops.RELU_OUT)
        h2 = ops.gemm_nt(h1, p['fc_layers.2.weight'], p['fc_layers.2.bias'], flags=ops.RELU_OUT)
        feats = ops.gemm_nt(h2, p['fc_layers.4.weight'], p['fc_layers.4.bias'], flags=ops.RELU_OUT)
        pred = ops.classifier_fwd(feats, p['classifier.0.weight'], p['classifier.0.bias'])
        return pred.view(H, W, -1)


class _WesupLossFn(torch.autograd.Function):
    """compute_loss of models/wesup.py:492-531 for a padded batch, all on the device."""

    @staticmethod
    def forward(ctx, sp_pred, sp_features, meta, threshold, weight, enable, eps):
        sp_pred = sp_pred.contiguous()
        y_all, src, sim = ops.propagate(sp_features.detach().contiguous(), meta, threshold, enable=enable)
        loss, terms = ops.loss_fwd(sp_pred, y_all, meta, eps, weight)
        ctx.save_for_backward(sp_pred, y_all, terms)
        ctx.meta, ctx.eps, ctx.weight = meta, eps, weight
        ctx.mark_non_differentiable(terms)
        return loss.reshape(()), terms

    @staticmethod
    def backward(ctx, dloss, _dterms):
        sp_pred, y_all, terms = ctx.saved_tensors
        dpred = ops.loss_bwd(sp_pred, y_all, ctx.meta, terms, dloss.reshape(1).contiguous(), ctx.eps, ctx.weight)
        return dpred, None, None, None, None, None, None


class WESUPTrainer(BaseTrainer):
    """Trainer for WESUP (mirror of models/wesup.py:403-547)."""

    def __init__(self, model, **kwargs):
        config = WESUPConfig()
        # the reference reads the class default only (models/wesup.py:426-429); the kwarg is honoured as well
        if config.freeze_backbone or kwargs.get('freeze_backbone'):
            for param in model.backbone.parameters():
                param.requires_grad = False
        kwargs = {**config.to_dict(), **kwargs}
        super().__init__(model, **kwargs)
        self.xentropy = partial(_cross_entropy)

    def get_default_dataset(self, root_dir, train=True, proportion=1.0):
        from ..utils.data import get_dataset
        return get_dataset(root_dir, train=train, proportion=proportion,
                           multiscale_range=self.kwargs.get('multiscale_range'),
                           rescale_factor=self.kwargs.get('rescale_factor'))

    def get_default_optimizer(self):
        from ..optim import FusedSGD
        self.model._ensure_engine()
        optimizer = FusedSGD(self.model, lr=5e-5, momentum=self.kwargs.get('momentum'),
                             weight_decay=self.kwargs.get('weight_decay'))
        # the reference builds a ReduceLROnPlateau scheduler and discards it (models/wesup.py:452-455)
        return optimizer, None

    def slic(self, img):
        """Superpixel segmentation of a batch (B,3,H,W) -> (labels (B,H,W) int32 on the device, upper bound on ids).

        The reference calls skimage.segmentation.slic on the CPU for every iteration (models/wesup.py:471-476, with
        a device->host->device round trip).  Here the default is the GPU SLIC of libwesup_hip.so (``wesup_slic``,
        same n_segments / compactness, 10 iterations, connectivity enforced, contiguous 0-based ids); a CPU
        implementation can still be plugged in with ``slic_fn=callable(img_hwc_numpy, n_segments, compactness)``."""
        H, W = img.size(-2), img.size(-1)
        n_segments = int(H * W / self.kwargs.get('sp_area'))                    # models/wesup.py:473-474
        fn = self.kwargs.get('slic_fn')
        if fn is not None:
            segs = []
            for b in range(img.size(0)):
                seg = np.asarray(fn(img[b].cpu().numpy().transpose(1, 2, 0), n_segments, self.kwargs.get('sp_compactness')))
                segs.append(torch.as_tensor(seg - seg.min(), dtype=torch.int32))
            return torch.stack(segs).to(self.device), max(int(s.max()) + 1 for s in segs)
        labels, _ = ops.slic(img.contiguous().float(), n_segments, float(self.kwargs.get('sp_compactness')), 10)
        return labels, self._slic_bound(H, W)

    def _slic_bound(self, H, W):
        # every connected superpixel holds its first pixel alone -> at most one id per grid centre plus split-offs;
        # the number of initial centres bounds the count after small components were absorbed only loosely, so use
        # the safe bound 2 * n_segments (no host sync)
        return int(2 * int(H * W / self.kwargs.get('sp_area'))) + 8

    def _kmax(self, n_sp_host, kmax_bound, H, W):
        """Rows per image of the padded superpixel tables: the largest count when the counts are on the host, else a bound."""
        Kmax = max(n_sp_host) if n_sp_host is not None else (
            kmax_bound or self.kwargs.get('max_superpixels') or self._slic_bound(H, W))
        # padded rows are inert: rounding up keeps the set of buffer shapes small when the superpixel count changes
        # from batch to batch (every new (B,H,W,Kmax) is a new set of engine buffers) and keeps Kmax % 4 == 0, which
        # the matrix form of the deep layers' pooling needs
        return (int(Kmax) + 63) // 64 * 64

    def prefetch_segment_fn(self):
        """GPU SLIC of the next batch on the input pipeline's copy stream, beside the current training step
        (``slic_ahead``, default on; a CPU ``slic_fn`` stays inside preprocess)."""
        if self.kwargs.get('slic_fn') is not None or not self.kwargs.get('slic_ahead', True):
            return None
        def segment(img):
            H, W = img.size(-2), img.size(-1)
            return ops.slic(img.contiguous().float(), int(H * W / self.kwargs.get('sp_area')),
                            float(self.kwargs.get('sp_compactness')), 10)
        return segment

    def preprocess(self, *data):
        """(img, pixel_mask[, point_mask[, segments]]) -> ((img, sp_maps), (pixel_mask, sp_labels))
        (models/wesup.py:457-490).  ``segments`` (B,H,W) is an extension: precomputed SLIC label maps."""
        segments = None
        if len(data) == 4:
            *data, segments = data
        data = [datum.to(self.device) for datum in data]
        if len(data) == 3:
            img, pixel_mask, point_mask = data
        elif len(data) == 2:
            img, pixel_mask = data
            point_mask = empty_tensor()
        elif len(data) == 1:
            img, = data
            point_mask = empty_tensor()
            pixel_mask = empty_tensor()
        else:
            raise ValueError('Invalid input data for WESUP')

        B = img.size(0)
        n_sp_host = None
        kmax_bound = None
        if hasattr(segments, 'labels') and hasattr(segments, 'counts'):       # utils/data.py LabelMaps (SLIC ahead)
            n_sp_host = list(segments.counts)
            segments = segments.labels
        if segments is None:
            segments, kmax_bound = self.slic(img)
        elif not segments.is_cuda:
            n_sp_host = [int(segments[b].max()) + 1 for b in range(B)]
        segments = segments.to(self.device)

        if point_mask is not None and not is_empty_tensor(point_mask):
            mask = point_mask
        elif pixel_mask is not None and not is_empty_tensor(pixel_mask):
            mask = pixel_mask
        else:
            mask = None

        # label maps already on the GPU: an upper bound on the ids avoids a host sync (padded rows are inert)
        # (label maps that arrive on the GPU without counts -- the pipeline's SLIC-ahead -- get the SLIC bound unless
        #  max_superpixels says otherwise)
        Kmax = self._kmax(n_sp_host, kmax_bound, img.size(-2), img.size(-1))
        meta = preprocess_label_maps(segments, mask, Kmax=Kmax, n_sp_host=n_sp_host)
        if self.kwargs.get('check_label_maps', False):
            meta.check()
        return (img, SuperpixelMaps(meta)), (pixel_mask, SuperpixelLabels(meta, has_mask=mask is not None))

    def compute_loss(self, pred, target, metrics=None):
        """models/wesup.py:492-531.  ``target[1]`` is either the SuperpixelLabels from ``preprocess`` (device-side,
        batched, no host sync) or a plain (N_l, C) ``sp_labels`` tensor as in the reference."""
        _, sp_labels = target
        sp_features = self.model.sp_features
        sp_pred = self.model.sp_pred
        if sp_pred is None:
            raise RuntimeError('You must run a forward pass before computing loss.')

        if isinstance(sp_labels, (SuperpixelLabels, SuperpixelMaps)):
            meta = sp_labels.meta
            B, Kmax = meta.B, meta.Kmax
            feats_p, pred_p = self.model._padded
            loss, terms = _WesupLossFn.apply(pred_p, feats_p, meta,
                                             float(self.kwargs.get('propagate_threshold')),
                                             float(self.kwargs.get('propagate_weight')),
                                             bool(self.kwargs.get('enable_propagation')), float(self.kwargs.get('epsilon')))
            if metrics is not None and isinstance(metrics, dict):
                metrics['_device_terms'] = (terms, meta)        # resolved with ONE host sync by the trainer
        else:
            total_num = sp_pred.size(0)
            labeled_num = sp_labels.size(0)
            if labeled_num < total_num:
                loss = self.xentropy(sp_pred[:labeled_num], sp_labels)
                if self.kwargs.get('enable_propagation'):
                    propagated_labels = _label_propagate(sp_features, sp_labels,
                                                         threshold=self.kwargs.get('propagate_threshold'))
                    propagate_loss = self.xentropy(sp_pred[labeled_num:], propagated_labels)
                    loss = loss + self.kwargs.get('propagate_weight') * propagate_loss
                if metrics is not None and isinstance(metrics, dict):
                    metrics['labeled_sp_ratio'] = labeled_num / total_num
                    if self.kwargs.get('enable_propagation'):
                        metrics['propagated_labels'] = propagated_labels.sum().item()
                        metrics['propagate_loss'] = propagate_loss.item()
            else:
                loss = self.xentropy(sp_pred, sp_labels)

        self.model.sp_pred = None          # clear outdated superpixel prediction (models/wesup.py:529)
        return loss

    def postprocess(self, pred, target=None):
        pred = pred.round().long()
        if target is not None:
            return pred, target[0].argmax(dim=1)
        return pred

    def post_epoch_hook(self, epoch):
        if self.scheduler is not None:
            labeled_loss = np.mean(self.tracker.history['loss'])
            if 'propagate_loss' in self.tracker.history:
                labeled_loss -= np.mean(self.tracker.history['propagate_loss'])
            self.scheduler.step(labeled_loss)
