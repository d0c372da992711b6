"""Generate golden vectors from the REAL reference (run in the build container only).

Imports /root/reference/models/wesup.py unmodified, with sys.modules stand-ins
for the third-party imports that are absent from this image (torchvision,
skimage, cv2, albumentations, fire).  The only behaviour a
stand-in supplies is torchvision's VGG16 layer list (cfg "D"); everything
arithmetic is the reference's own code on torch CPU ops.

Writes small .npz fixtures to tests/golden/.  Weights are not stored: they are
regenerated from ``oracle.wesup_oracle.make_weights(seed, feat_scale)``.

Usage:  python oracle/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'


class _Anything(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        m = _Anything(self.__name__ + '.' + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return None


def _install_standins():
    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, c = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        obj = types.SimpleNamespace()
        obj.features = nn.Sequential(*layers)
        return obj

    tv = _Anything('torchvision')
    tvm = _Anything('torchvision.models')
    tvm.vgg16 = vgg16
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    for name in ['torchvision.transforms', 'torchvision.transforms.functional', 'skimage',
                 'skimage.segmentation', 'skimage.io', 'skimage.morphology', 'skimage.transform',
                 'skimage.measure', 'cv2', 'albumentations', 'fire']:
        sys.modules[name] = _Anything(name)


def _reference_case(ref, orc, synth, out_dir, name, H, W, g, mode, fs, seed, compact=False):
    """Run the real reference (models/wesup.py:18-139,263-304,492-531 + autograd backward) on one seeded synthetic
    image and store its outputs.  compact=True (full-size cases): the inputs are NOT stored (the test regenerates
    them from wesup_amd.synth with the same seeds; a checksum is kept) and the big dense outputs are reduced to
    bit-packed / strided samples, so that a 480x480 fixture stays at a few hundred KB."""
    weights = orc.make_weights(seed, feat_scale=fs)
    model = ref.WESUP()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.train()
    img = synth.synth_image(seed, H, W)
    seg = synth.voronoi_labels(seed, H, W, g)
    if mode == 'full':
        mask = synth.pixel_mask(seed, H, W)
    elif mode == 'point_tie':
        mask = synth.point_mask(seed, seg, 0.3, 2, tie_every=2)
    elif mode == 'point20':
        mask = synth.point_mask(seed, seg, 0.2, 2)               # the benchmark's 20 % point-labelled superpixels
    else:
        mask = synth.point_mask(seed, seg, 0.25, 2)
    t_img = torch.from_numpy(img).unsqueeze(0)
    t_seg = torch.from_numpy(seg).long()
    t_mask = torch.from_numpy(mask).long()

    sp_maps, sp_labels = ref._preprocess_superpixels(t_seg, t_mask, epsilon=1e-7)
    # ordering implied by the maps: argmax over N recovers new row per pixel
    new_row = sp_maps.argmax(dim=0)
    pred = model((t_img, sp_maps))
    sp_features = model.sp_features
    sp_pred = model.sp_pred
    fm = model.feature_maps.detach()

    trainer = ref.WESUPTrainer.__new__(ref.WESUPTrainer)      # compute_loss only needs these:
    trainer.model = model
    trainer.kwargs = {**ref.WESUPConfig().to_dict(), 'epsilon': 1e-7}
    trainer.xentropy = ref._cross_entropy
    metrics = {}
    pixel_mask = t_mask.unsqueeze(0)
    sp_feat_keep = sp_features.detach().clone()
    sp_pred_keep = sp_pred.detach().clone()
    loss = trainer.compute_loss(pred, (pixel_mask, sp_labels), metrics=metrics)
    n_l = sp_labels.size(0)
    n = sp_pred_keep.size(0)
    if n_l < n:
        y_u = ref._label_propagate(sp_feat_keep, sp_labels, threshold=0.8)
        f = sp_feat_keep
        Wfull = torch.exp(-torch.einsum('ijk,ijk->ij', f - f.unsqueeze(1), f - f.unsqueeze(1)))
        W_ul = Wfull[n_l:, :n_l]
        max_sim, src = W_ul.max(dim=1)
    else:
        y_u = torch.zeros(0, 2)
        W_ul = torch.zeros(0, n_l)
        max_sim = torch.zeros(0)
        src = torch.zeros(0, dtype=torch.long)
    model.zero_grad()
    if loss.requires_grad:
        loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    post_pred = trainer.postprocess(pred.detach(), None)

    fx = dict(
        H=H, W=W, g=g, seed=seed, feat_scale=fs, mode=mode,
        sp_labels=sp_labels.numpy(), new_row=new_row.numpy().astype(np.int16),
        sp_maps_rowsum=sp_maps.sum(dim=(1, 2)).numpy(),
        sp_maps_max=sp_maps.amax(dim=(1, 2)).numpy(),
        fm_chan_mean=fm.mean(dim=(1, 2)).numpy(),
        sp_features=sp_feat_keep.numpy(), sp_pred=sp_pred_keep.numpy(),
        max_sim=max_sim.numpy(), src=src.numpy().astype(np.int32),
        y_u=y_u.numpy(), loss=np.float32(loss.item()),
        labeled_sp_ratio=np.float64(metrics.get('labeled_sp_ratio', -1)),
        propagated_labels=np.float64(metrics.get('propagated_labels', -1)),
        propagate_loss=np.float64(metrics.get('propagate_loss', -1)),
    )
    if compact:
        # inputs come from wesup_amd.synth (seeded numpy); the checksums make a drifted generator fail loudly
        fx.update(img_sum=np.float64(img.astype(np.float64).sum()), seg_sum=np.int64(seg.astype(np.int64).sum()),
                  mask_sum=np.int64(mask.sum()),
                  fm_sample=fm[::97, ::23, ::29].numpy(), pred_sample=pred.detach()[0, ::7, ::11].numpy(),
                  post_pred_bits=np.packbits(post_pred.numpy().astype(np.uint8), axis=None),
                  W_ul_rowsum=W_ul.sum(dim=1).numpy())
        all_keys = list(grads)
    else:
        fx.update(img=img, seg=seg.astype(np.int16), mask=np.packbits(mask, axis=None), mask_shape=np.array(mask.shape),
                  fm_sample=fm[::37, ::5, ::7].numpy(), pred=pred.detach().numpy(),
                  post_pred=post_pred.numpy().astype(np.int8), W_ul=W_ul.numpy())
        all_keys = ['backbone.0.weight', 'backbone.0.bias', 'backbone.12.weight', 'backbone.28.weight',
                    'backbone.28.bias', 'side_conv0.weight', 'side_conv0.bias', 'side_conv576.weight',
                    'side_conv1856.weight', 'fc_layers.0.weight', 'fc_layers.2.bias', 'fc_layers.4.weight',
                    'classifier.0.weight', 'classifier.0.bias']
    for k in all_keys:
        gk = grads[k]
        fx['gnorm.' + k] = np.float64(gk.double().norm().item())
        fx['gsamp.' + k] = gk.flatten()[::max(1, gk.numel() // 64)][:64].numpy()
        if compact:
            fx['gmax.' + k] = np.float64(gk.double().abs().max().item())
    np.savez_compressed(os.path.join(out_dir, name + '.npz'), **fx)
    print(f'{name}: N={n} n_l={n_l} loss={loss.item():.6f} propagated={metrics.get("propagated_labels")}'
          f' ploss={metrics.get("propagate_loss")}')


def main():
    _install_standins()
    sys.path.insert(0, REF)
    import models.wesup as ref                                   # the real reference
    from oracle import wesup_oracle as orc
    from wesup_amd import synth

    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)

    cases = [
        # name, H, W, g, mode, feat_scale, seed
        ('c32_point', 32, 32, 4, 'point', 0.02, 1),
        ('c32_point_far', 32, 32, 4, 'point', 1.0, 2),
        ('c64_point_tie', 64, 64, 6, 'point_tie', 0.02, 3),
        ('c64_full', 64, 64, 6, 'full', 0.02, 4),
        ('c96x80_point', 96, 80, 7, 'point', 0.05, 5),
        ('c64_identical', 64, 64, 6, 'identical', 0.0, 6),
    ]
    for case in cases:
        _reference_case(ref, orc, synth, out_dir, *case)

    # pixel-wise inference (models/wesup.py:307-400), SURVEY.md 8(f) row 3
    weights = orc.make_weights(4, feat_scale=0.3)
    pmodel = ref.WESUPPixelInference()
    pmodel.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    pmodel.eval()
    img = synth.synth_image(8, 48, 80)
    with torch.no_grad():
        pout = pmodel(torch.from_numpy(img).unsqueeze(0))
    np.savez_compressed(os.path.join(out_dir, 'pixel_infer.npz'), img=img, seed=4, feat_scale=0.3,
                        out=pout.numpy().astype(np.float32))
    print('pixel_infer:', tuple(pout.shape), float(pout[..., 1].mean()))

    # reference behaviours the tests pin (SURVEY.md 8(c) last row)
    f = torch.zeros(5, 32)
    y_l = torch.tensor([[1., 0.], [0., 1.]])
    yu = ref._label_propagate(f, y_l, threshold=0.8)
    ce0 = ref._cross_entropy(torch.tensor([[0.3, 0.7]]), torch.zeros(1, 2))
    np.savez_compressed(os.path.join(out_dir, 'behaviours.npz'),
                        identical_yu=yu.numpy(), ce_zero=np.float32(ce0.item()))
    print('behaviours: identical_yu', yu.tolist(), 'ce_zero', ce0.item())


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def metrics_golden():
    """GlaS challenge metrics of the reference (utils/metrics.py:48-281) on small blob masks -> tests/golden/metrics.npz.
    The reference labels connected components with skimage.measure.label (absent here): the stand-in supplies
    scipy.ndimage.label with 8-connectivity, which is skimage's documented default for 2-D input; everything else
    (object matching, weights, Dice / F1 / Hausdorff formulas) is the reference's own code."""
    from scipy import ndimage
    _install_standins()
    sk = _Anything('skimage.measure')
    sk.label = lambda a: ndimage.label(np.asarray(a) != 0, structure=np.ones((3, 3), dtype=np.int32))[0]
    sys.modules['skimage.measure'] = sk
    sys.modules['skimage'].measure = sk
    sys.path.insert(0, REF)
    import importlib
    refm = importlib.import_module('utils.metrics')

    def blobs(seed, H=64, W=72, n=6):
        rs = np.random.RandomState(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.zeros((H, W), dtype=np.uint8)
        for _ in range(n):
            cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(3, 10)
            m |= ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.uint8)
        return m

    cases = [(blobs(1), blobs(2)), (blobs(3), blobs(3)), (blobs(4, n=2), blobs(5, n=9)),
             (np.zeros((64, 72), np.uint8), blobs(6)), (blobs(7), np.zeros((64, 72), np.uint8)),
             (np.zeros((64, 72), np.uint8), np.zeros((64, 72), np.uint8))]
    # one pair shifted by a few pixels: partial overlaps around the 50 % threshold
    b = blobs(8)
    cases.append((np.roll(b, 4, axis=1), b))
    out = dict(n=len(cases))
    for i, (S, G) in enumerate(cases):
        out[f'S{i}'], out[f'G{i}'] = S, G
        vals = [refm.detection_f1(S, G), refm.object_dice(S, G)]
        # object_hausdorff of an empty against a non-empty map is ill-defined in the reference (division by zero): skip
        vals.append(refm.object_hausdorff(S, G) if S.any() and G.any() else np.nan)
        vals.append(refm.hausdorff(S, G))
        out[f'v{i}'] = np.array(vals, dtype=np.float64)
        print('metrics case', i, out[f'v{i}'])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'metrics.npz'), **out)


if __name__ == '__main__' and 'metrics' in sys.argv[1:]:
    metrics_golden()


def tiles_golden():
    """Window strategy of the reference (infer_tile.py:23-91): top-left coordinates, division into patches and the
    running-average recombination, on seeded images whose sizes are not multiples of the patch size
    -> tests/golden/tiles.npz.  The three functions are numpy only; the module's third-party imports are stand-ins."""
    _install_standins()
    for name in ['tqdm', 'PIL', 'PIL.Image']:
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = _Anything(name)
    sys.path.insert(0, REF)
    import importlib
    reft = importlib.import_module('infer_tile')
    out = {}
    cases = [(50, 70, 32), (64, 64, 32), (33, 97, 16), (40, 40, 40), (45, 31, 24)]
    for i, (H, W, ps) in enumerate(cases):
        rs = np.random.RandomState(100 + i)
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        coords = np.array(list(reft._get_top_left_coordinates(H, W, ps)), dtype=np.int64)
        patches = reft.divide_image_to_patches(img, ps)
        preds = rs.rand(patches.shape[0], ps, ps).astype(np.float64)
        combined = reft.combine_patches_to_image(preds, H, W)
        combined_c = reft.combine_patches_to_image(patches.astype(np.float64), H, W)      # with a channel axis
        # inputs are regenerated in the test from RandomState(100 + i): only the reference's outputs are stored
        out[f'shape{i}'] = np.array([H, W, ps])
        out[f'coords{i}'] = coords
        out[f'patch_sums{i}'] = patches.reshape(patches.shape[0], -1).sum(1).astype(np.int64)
        out[f'combined{i}'] = combined.astype(np.float32)
        if i in (0, 4):
            out[f'patches{i}'] = patches
            out[f'combined_c{i}'] = combined_c.astype(np.float32)
        print('tiles case', i, (H, W, ps), 'patches', patches.shape, 'combined', combined.shape)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'tiles.npz'), **out)


if __name__ == '__main__' and 'tiles' in sys.argv[1:]:
    tiles_golden()


def full_size_golden():
    """BASELINE configs[0] at full size through the REAL reference: one 480x480 image, 196 superpixels (g = 14),
    20 % point-labelled -- and the same image at the benchmark's 576 superpixels (g = 24, one image of configs[1])
    -> tests/golden/c480_g14.npz, c480_g24.npz.  ~10 s and a few GB each on the build container's CPU
    (the reference materialises 184 / 531 MB of dense sp_maps and re-copies the growing (C,H,W) map 12 times)."""
    _install_standins()
    sys.path.insert(0, REF)
    import models.wesup as ref
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    torch.set_num_threads(8)
    _reference_case(ref, orc, synth, out_dir, 'c480_g14', 480, 480, 14, 'point20', 1.5, 21, compact=True)
    _reference_case(ref, orc, synth, out_dir, 'c480_g24', 480, 480, 24, 'point20', 1.5, 22, compact=True)


if __name__ == '__main__' and 'full' in sys.argv[1:]:
    full_size_golden()


def crag_size_golden():
    """One image of BASELINE configs[3] through the REAL reference: 800x800, 1521 superpixels (g = 39), 20 %
    point-labelled -> tests/golden/c800_g39.npz.  The reference materialises 3.9 GB of dense sp_maps and a 5.4 GB feature
    map (plus its gradient and the re-copies of the growing map): ~30 GB peak and a few minutes on the build container."""
    _install_standins()
    sys.path.insert(0, REF)
    import models.wesup as ref
    from oracle import wesup_oracle as orc
    from wesup_amd import synth
    torch.set_num_threads(8)
    _reference_case(ref, orc, synth, os.path.join(ROOT, 'tests', 'golden'), 'c800_g39', 800, 800, 39, 'point20', 1.5, 23,
                    compact=True)


if __name__ == '__main__' and 'crag' in sys.argv[1:]:
    crag_size_golden()


def postprocess_golden():
    """The small-region post-processing of the reference's evaluation script (scripts/evaluate_glas.py:29-43) on seeded
    blob masks -> tests/golden/postprocess.npz.  The script executes its whole evaluation at import, so only the
    ``postprocess`` function is taken from it: its definition is located with ``ast`` in the file where it lies and
    compiled from there (nothing of it is stored here); ``label`` is the same scipy stand-in for skimage.measure.label
    as in metrics_golden()."""
    import ast
    from scipy import ndimage
    path = os.path.join(REF, 'scripts', 'evaluate_glas.py')
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'postprocess')
    ns = {'label': lambda a: ndimage.label(np.asarray(a) != 0, structure=np.ones((3, 3), dtype=np.int32))[0], 'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    ref_postprocess = ns['postprocess']

    def blobs(seed, H, W, n, rmax):
        rs = np.random.RandomState(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.zeros((H, W), dtype=np.float64)
        for _ in range(n):
            cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(3, rmax)
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
        for _ in range(n):                                        # holes of assorted sizes
            cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(2, rmax // 2)
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
        return m
    out = {}
    cases = [(1, 200, 260, 14, 40), (2, 200, 260, 30, 30), (3, 150, 150, 3, 60), (4, 120, 300, 0, 10)]
    cases_full = [np.ones((100, 120)), np.zeros((100, 120))]
    n = 0
    for seed, H, W, k, rmax in cases:
        m = blobs(seed, H, W, k, rmax)
        out[f'in{n}'] = np.packbits(m.astype(np.uint8), axis=None)
        out[f'shape{n}'] = np.array(m.shape)
        out[f'out{n}'] = np.packbits(ref_postprocess(m.copy()).astype(np.uint8), axis=None)
        print('postprocess case', n, m.shape, int(m.sum()), '->', int(np.unpackbits(out[f'out{n}'])[:m.size].sum()))
        n += 1
    for m in cases_full:
        out[f'in{n}'] = np.packbits(m.astype(np.uint8), axis=None)
        out[f'shape{n}'] = np.array(m.shape)
        out[f'out{n}'] = np.packbits(ref_postprocess(m.astype(np.float64).copy()).astype(np.uint8), axis=None)
        n += 1
    out['n'] = n
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'postprocess.npz'), **out)


if __name__ == '__main__' and 'postprocess' in sys.argv[1:]:
    postprocess_golden()
