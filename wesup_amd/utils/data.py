"""Input pipeline of the training step (reference utils/data.py:33-165,279-375).

On-disk contract (reference README.md:17-59): ``root/images/*`` (PNG/BMP), optional ``root/masks/*`` (class index per
pixel), optional ``root/points/*.csv`` (rows ``x,y,class``).  The reference decodes, resizes AND augments every item
on the CPU in DataLoader workers (skimage + albumentations).  Here the host only decodes and resizes (PIL, in
DataLoader workers); ``DevicePrefetcher`` moves a collated uint8 batch to the GPU through pinned memory one batch
ahead and runs the augmentation there (``wesup_augment``: flips + shift/scale/rotate as one affine map, HSV shift,
brightness/contrast, ToTensor, one-hot masks); keypoints follow the same affine map on the host and are rasterised
as radius-0 dots (utils/data.py:352-362).  What comes out is the reference's item contract, batched:
``img f32 (B,3,H,W) in [0,1]``, ``pixel_mask (B,C,H,W)`` one-hot or the empty tensor, ``point_mask (B,C,H,W)``.

CLAHE and Blur run on the GPU as well (``wesup_appearance``, on the un-warped image and in the reference's order),
ElasticTransform -- with the albumentations defaults a random 3-point affine plus a displacement field of < 0.02 px, both modelled --
is folded into the affine map.  Not reproduced: exact skimage / OpenCV / albumentations numerics (all absent from the
build image: parity unpinned).  ``SyntheticGlasDataset`` keeps producing
GlaS-shaped items of the same contract for benchmarks ('synthetic:H:W:g:n')."""
import csv
from pathlib import Path

import numpy as np
import torch

from .. import synth
from . import empty_tensor

NO_CLASS = 255


class SyntheticGlasDataset(torch.utils.data.Dataset):
    def __init__(self, H=480, W=480, g=24, n=16, seed=0, frac=0.2, with_points=True):
        self.H, self.W, self.g, self.n, self.seed, self.frac = H, W, g, n, seed, frac
        self.with_points = with_points

    def __len__(self):
        return self.n

    def summary(self, logger=None):
        msg = f'SyntheticGlasDataset: {self.n} samples of {self.H}x{self.W}, {self.g * self.g} superpixels'
        (logger.info if logger else print)(msg)

    def __getitem__(self, i):
        s = self.seed * 100003 + i
        img = torch.from_numpy(synth.synth_image(s, self.H, self.W))
        seg = torch.from_numpy(synth.voronoi_labels(s, self.H, self.W, self.g))
        pix = torch.from_numpy(synth.pixel_mask(s, self.H, self.W)).long()
        if not self.with_points:
            return img, pix, torch.zeros(0), seg
        pts = torch.from_numpy(synth.point_mask(s, seg.numpy(), self.frac, 2)).long()
        return img, pix, pts, seg


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return im.copy()


class SegmentationDataset(torch.utils.data.Dataset):
    """Decode + resize on the host (utils/data.py:33-165).  Items are RAW: ``img uint8 (H,W,3)``, ``mask uint8 (H,W)``
    class index (255 everywhere when the dataset has no masks), ``points int32 (P_MAX,3)`` rows (x,y,class) padded
    with -1 (only PointSupervisionDataset fills them).  Augmentation, ToTensor and one-hot happen on the GPU in
    ``DevicePrefetcher``; ``to_reference_item`` gives the reference's un-augmented CPU item for a raw item."""
    P_MAX = 4096

    def __init__(self, root_dir, mode=None, target_size=None, rescale_factor=None, multiscale_range=None, train=True,
                 proportion=1, n_classes=2, seed=0):
        self.root_dir = Path(root_dir).expanduser()
        self.img_paths = sorted((self.root_dir / 'images').iterdir())
        self.mask_paths = None
        if (self.root_dir / 'masks').exists():
            self.mask_paths = sorted((self.root_dir / 'masks').iterdir())
        self.mode = mode or 'mask' if self.mask_paths is not None else None           # utils/data.py:68
        self.target_size, self.rescale_factor, self.multiscale_range = target_size, rescale_factor, multiscale_range
        self.train, self.proportion, self.n_classes = train, proportion, n_classes
        self.picked = np.arange(len(self.img_paths))
        if self.proportion < 1:                                                        # utils/data.py:84-88
            rs = np.random.RandomState(seed)
            rs.shuffle(self.picked)
            self.picked = np.sort(self.picked[:len(self)])

    def __len__(self):
        return int(self.proportion * len(self.img_paths))

    def summary(self, logger=None):
        msg = (f'{type(self).__name__} at {self.root_dir}: {len(self)} of {len(self.img_paths)} images, '
               f'masks: {self.mask_paths is not None}, mode: {self.mode}, train: {self.train}')
        (logger.info if logger else print)(msg)

    def _target_hw(self, height, width):
        if self.target_size is not None:
            return tuple(self.target_size), None
        f = self.rescale_factor
        if self.multiscale_range is not None:
            f = float(np.random.uniform(*self.multiscale_range))
        if f is not None:
            return (int(np.ceil(f * height)), int(np.ceil(f * width))), f               # utils/data.py:99-106
        return (height, width), None

    def _load(self, idx):
        from PIL import Image
        img = _imread(self.img_paths[idx]).convert('RGB')
        width, height = img.size
        (th, tw), factor = self._target_hw(height, width)
        if (th, tw) != (height, width):
            img = img.resize((tw, th), Image.BILINEAR)                                 # order 1, no anti-aliasing
        mask = None
        if self.mask_paths is not None:
            mask = _imread(self.mask_paths[idx])
            if (th, tw) != (height, width):
                mask = mask.resize((tw, th), Image.NEAREST)                            # order 0
            mask = np.asarray(mask)
            if mask.ndim == 3:
                mask = mask[..., 0]
            mask = mask.astype(np.uint8)
        return np.array(img, dtype=np.uint8), mask, (height, width), factor

    def __getitem__(self, i):
        idx = int(self.picked[i])
        img, mask, _, _ = self._load(idx)
        if mask is None:
            mask = np.full(img.shape[:2], NO_CLASS, dtype=np.uint8)
        pts = np.full((self.P_MAX, 3), -1, dtype=np.int32)
        return torch.from_numpy(img), torch.from_numpy(mask), torch.from_numpy(pts)

    def to_reference_item(self, raw):
        """The reference's CPU item for a raw item without augmentation (utils/data.py:135-152)."""
        img, mask, pts = raw
        out = [img.permute(2, 0, 1).float() / 255.0]
        if self.mask_paths is not None:
            out.append(torch.stack([(mask == k) for k in range(self.n_classes)]).long())
        else:
            out.append(empty_tensor())
        return tuple(out)


class PointSupervisionDataset(SegmentationDataset):
    """images + points/*.csv (+ masks) -> raw item with the rescaled keypoints (utils/data.py:279-375)."""

    def __init__(self, root_dir, target_size=None, rescale_factor=None, multiscale_range=None, radius=0, train=True,
                 proportion=1):
        super().__init__(root_dir, mode='point', target_size=target_size, rescale_factor=rescale_factor, train=train,
                         proportion=proportion, multiscale_range=multiscale_range)
        self.point_paths = sorted((self.root_dir / 'points').glob('*.csv'))
        if radius != 0:
            raise NotImplementedError('only radius-0 point labels (the reference default) are rasterised on the GPU')
        self.radius = radius

    def __getitem__(self, i):
        idx = int(self.picked[i])
        img, mask, (oh, ow), factor = self._load(idx)
        h, w = img.shape[:2]
        if factor is None:                                                              # utils/data.py:340-347
            rescaler = np.array([[w / ow, h / oh, 1]])
        else:
            rescaler = np.array([[factor, factor, 1]])
        with open(str(self.point_paths[idx])) as fp:
            rows = [[int(d) for d in r] for r in csv.reader(fp) if r]
        pts = np.full((self.P_MAX, 3), -1, dtype=np.int32)
        if rows:
            p = np.floor(np.array(rows) * rescaler).astype(np.int32)[:self.P_MAX]         # utils/data.py:350-353
            pts[:len(p)] = p
        if mask is None:
            mask = np.full((h, w), NO_CLASS, dtype=np.uint8)
        return torch.from_numpy(img), torch.from_numpy(mask), torch.from_numpy(pts)



class Digest2019PointDataset(PointSupervisionDataset):
    """PointSupervisionDataset plus the Digest-2019 rule for images whose file name starts with ``negative``
    (utils/data.py:409-512): such an image has no tumour anywhere, so instead of reading a csv its whole pixel mask IS
    its point annotation (``point_mask = pixel_mask``, :497-499).  The raw item says so with one sentinel point row
    (-2, -2, -2); ``DevicePrefetcher`` then copies the augmented one-hot pixel mask into the point mask."""
    NEGATIVE = -2

    def __getitem__(self, i):
        idx = int(self.picked[i])
        if not self.img_paths[idx].name.startswith('negative'):
            return super().__getitem__(i)
        img, mask, _, _ = self._load(idx)
        if mask is None:
            mask = np.zeros(img.shape[:2], dtype=np.uint8)              # "negative": background everywhere
        pts = np.full((self.P_MAX, 3), -1, dtype=np.int32)
        pts[0] = self.NEGATIVE
        return torch.from_numpy(img), torch.from_numpy(mask), torch.from_numpy(pts)


class AreaConstraintDataset(SegmentationDataset):
    """images + masks + ``area.csv`` (columns ``img,area``: foreground fraction per image) -> raw item with a fourth
    element, the (lower, upper) bound of the foreground area (utils/data.py:168-277).  ``area_type`` 'decimal' keeps
    the fraction, 'integer' counts the positive pixels of the (resized) mask; ``constraint`` 'equality' gives
    (a, a), 'individual' (a(1-margin), a(1+margin)) truncated to integers as the reference's ``.long()`` does,
    'common' the dataset-wide (min, max)."""

    def __init__(self, root_dir, target_size=None, rescale_factor=None, area_type='decimal', constraint='equality',
                 margin=0.1, train=True, proportion=1.0):
        super().__init__(root_dir, mode='area', target_size=target_size, rescale_factor=rescale_factor, train=train,
                         proportion=proportion)
        with open(self.root_dir / 'area.csv') as fp:
            rows = list(csv.DictReader(fp))
        self.area_info = np.array([float(r['area']) for r in rows], dtype=np.float64)      # row i <-> image i (:257)
        self.area_type, self.constraint, self.margin = area_type, constraint, margin

    def __getitem__(self, i):
        idx = int(self.picked[i])
        img, mask, pts = super().__getitem__(i)
        if self.area_type == 'decimal':
            area = float(self.area_info[idx])
        else:
            area = float((mask == 1).sum())
        if self.constraint == 'equality':
            bounds = torch.tensor([area, area], dtype=torch.float32)
        elif self.constraint == 'individual':
            bounds = torch.tensor([area * (1 - self.margin), area * (1 + self.margin)]).long()
        else:
            lower, upper = float(self.area_info.min()), float(self.area_info.max())
            if self.area_type == 'integer':
                size = self.target_size if self.target_size is not None else mask.shape
                lower, upper = int(lower * np.prod(size)), int(upper * np.prod(size))
            bounds = torch.tensor([lower, upper])
        return img, mask, pts, bounds


class WESUPV2Dataset(SegmentationDataset):
    """images + ``spl-masks/*.npy`` (per-pixel soft label maps (H, W, C) made by scripts/generate_spl_masks.py) -> raw
    item (img, mask (C, H, W) int64, coords (2, H, W) float32), utils/data.py:378-406.  ``coords`` is the reference's
    normalised position map (its ``_generate_coords`` tiles ``linspace(0, 1, H)`` along the fast axis and views the
    result as (2, H, W), which is what is reproduced here, quirk included)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.spl_paths = None
        if (self.root_dir / 'spl-masks').exists():
            self.spl_paths = sorted((self.root_dir / 'spl-masks').iterdir())

    @staticmethod
    def _generate_coords(shape):
        x = np.linspace(0, 1, shape[0])
        y = np.linspace(0, 1, shape[1])
        coords = torch.as_tensor(np.stack([np.tile(x, len(y)), np.repeat(y, len(x))]), dtype=torch.float32)
        return coords.view(2, shape[0], shape[1])

    def __getitem__(self, i):
        idx = int(self.picked[i])
        img, _, _, _ = self._load(idx)
        h, w = img.shape[:2]
        if self.spl_paths is None:
            raise FileNotFoundError(f'{self.root_dir}/spl-masks is missing')
        spl = np.load(self.spl_paths[idx])
        if spl.shape[:2] != (h, w):                                    # nearest resize, as for masks (utils/data.py:26-30)
            yy = (np.arange(h) * spl.shape[0] / h).astype(np.int64)
            xx = (np.arange(w) * spl.shape[1] / w).astype(np.int64)
            spl = spl[yy][:, xx]
        mask = torch.as_tensor(np.ascontiguousarray(spl.transpose(2, 0, 1)), dtype=torch.long)
        return torch.from_numpy(img), mask, self._generate_coords(img.shape)


class CompoundDataset(torch.utils.data.Dataset):
    """Several datasets read in lock step: item i is the tuple of every dataset's item i (utils/data.py:515-528)."""

    def __init__(self, *datasets):
        self.datasets = datasets

    def __len__(self):
        return len(self.datasets[0])

    def __getitem__(self, idx):
        return tuple(dataset[idx] for dataset in self.datasets)

    def summary(self, logger=None):
        for dataset in self.datasets:
            dataset.summary(logger=logger)


ELASTIC_CELL = 8       # pixels per cell of the coarse displacement grid (sigma = 50 px: the field is smooth on that scale)


def sample_params(rs, H, W, train, point_pipeline=True, elastic_out=None):
    """12 floats for wesup_augment + the forward 2x3 matrix for keypoints.  Parameter ranges are the albumentations
    defaults the reference's pipelines rely on (utils/data.py:116-133 for masks, :302-327 for points).
    elastic_out (optional list): when ElasticTransform is drawn, its displacement field and the 12 floats that place it are
    appended as ``(field (2, hc, wc) float32, params (12,) float32)`` (else ``None``) -- ``ops.augment(elastic=...)``."""
    M = np.eye(3)
    row = np.zeros(12, dtype=np.float32)
    row[6] = 1.0
    el = None
    if train:
        if not point_pipeline and rs.random_sample() < 0.5:      # A.ElasticTransform(p=0.5), mask pipelines only
            E = elastic_affine(rs, H, W)
            M = E @ M
            if elastic_out is not None:
                par = np.zeros(12, dtype=np.float32)
                par[0:3], par[3:6] = E[0], E[1]
                par[6:10] = np.linalg.inv(E)[:2, :2].reshape(-1)
                par[10] = 1.0
                el = (elastic_field(rs, H, W), par)
        if rs.random_sample() < 0.5:
            M = np.array([[-1, 0, W - 1], [0, 1, 0], [0, 0, 1.0]]) @ M
        if rs.random_sample() < 0.5:
            M = np.array([[1, 0, 0], [0, -1, H - 1], [0, 0, 1.0]]) @ M
        if point_pipeline or rs.random_sample() < 0.8:            # ShiftScaleRotate p=1 (points) / p=0.8 (masks)
            ang = np.deg2rad(rs.uniform(-45, 45))
            sc = 1.0 + rs.uniform(-0.1, 0.1)
            dx, dy = rs.uniform(-0.0625, 0.0625), rs.uniform(-0.0625, 0.0625)
            cx, cy = (W - 1) * 0.5, (H - 1) * 0.5
            a, b = sc * np.cos(ang), sc * np.sin(ang)
            M = np.array([[a, b, (1 - a) * cx - b * cy + dx * W], [-b, a, b * cx + (1 - a) * cy + dy * H], [0, 0, 1.0]]) @ M
        lim_bc, lim_h, lim_s, lim_v = (0.3, 20.0, 30.0, 20.0) if point_pipeline else (0.1, 10.0, 10.0, 10.0)
        row[6] = 1.0 + rs.uniform(-lim_bc, lim_bc)
        row[7] = rs.uniform(-lim_bc, lim_bc)
        row[8], row[9], row[10] = rs.uniform(-lim_h, lim_h), rs.uniform(-lim_s, lim_s), rs.uniform(-lim_v, lim_v)
    Minv = np.linalg.inv(M)
    row[0:3], row[3:6] = Minv[0], Minv[1]
    if elastic_out is not None:
        elastic_out.append(el)
    return row, M[:2]


def _gauss_matrix(n, sigma, truncate=4.0):
    """(n, n) matrix of a 1-D Gaussian filter with scipy.ndimage's conventions: taps at integer offsets up to
    int(truncate * sigma + 0.5), normalised, borders handled by reflection about the edge of the first / last sample
    (mode='reflect': d c b a | a b c d | d c b a)."""
    r = int(truncate * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    K = np.zeros((n, n))
    idx = np.arange(n)
    for o, wgt in zip(range(-r, r + 1), k):
        j = np.mod(idx + o, 2 * n)
        j = np.where(j < n, j, 2 * n - 1 - j)
        np.add.at(K, (idx, j), wgt)
    return K


def elastic_field(rs, H, W, alpha=1.0, sigma=50.0, cell=ELASTIC_CELL):
    """The displacement field of albumentations' ElasticTransform(alpha=1, sigma=50) (utils/data.py:124) on a coarse grid:
    (2, hc, wc) float32, plane 0 = dx, plane 1 = dy, hc = ceil(H / cell).  The reference's field is
    gaussian_filter(U(-1, 1) per pixel, sigma) * alpha; a Gaussian of width sigma over per-pixel noise equals, up to the
    discretisation of the kernel, a Gaussian of width sigma / cell over the cell-averaged noise, so the per-pixel noise is
    drawn as the reference draws it, averaged over cell x cell blocks (partial blocks at the border over the pixels they
    have) and smoothed on the coarse grid; the kernel interpolates bilinearly.  Against the per-pixel filter on the same
    noise the interpolated field differs by < 0.001 px (tests/test_data_cpu.py); the field itself has a standard deviation
    of ~0.0034 px and a maximum of ~0.01-0.02 px per image at alpha = 1."""
    hc, wc = -(-H // cell), -(-W // cell)
    out = np.empty((2, hc, wc), dtype=np.float32)
    Ky, Kx = _gauss_matrix(hc, sigma / cell), _gauss_matrix(wc, sigma / cell)
    cnt = np.add.reduceat(np.add.reduceat(np.ones((H, W)), np.arange(0, H, cell), 0), np.arange(0, W, cell), 1)
    for a in range(2):
        noise = rs.rand(H, W) * 2.0 - 1.0
        blk = np.add.reduceat(np.add.reduceat(noise, np.arange(0, H, cell), 0), np.arange(0, W, cell), 1) / cnt
        out[a] = (Ky @ blk @ Kx.T * alpha).astype(np.float32)
    return out


def elastic_affine(rs, H, W, alpha_affine=50.0):
    """The affine part of albumentations' ElasticTransform(alpha=1, sigma=50, alpha_affine=50) (utils/data.py:124):
    three corners of a centred square are moved by U(-alpha_affine, alpha_affine) pixels each and the affine map through
    the three pairs is applied (cv2.getAffineTransform / warpAffine).  Its second part, the displacement field
    gaussian_filter(U(-1, 1), sigma=50) * alpha (standard deviation 0.0034 px, up to ~0.02 px), is ``elastic_field``.
    (albumentations builds the points from (height, width) and hands them to OpenCV as (x, y); kept.)"""
    c = np.array([H // 2, W // 2], dtype=np.float64)
    sq = min(H, W) // 3
    pts1 = np.array([c + sq, [c[0] + sq, c[1] - sq], c - sq])
    pts2 = pts1 + rs.uniform(-alpha_affine, alpha_affine, size=pts1.shape)
    A = np.concatenate([pts1, np.ones((3, 1))], 1)
    sol = np.linalg.solve(A, pts2)                     # rows of sol: coefficients of x, y, 1 for (x', y')
    return np.array([[sol[0, 0], sol[1, 0], sol[2, 0]], [sol[0, 1], sol[1, 1], sol[2, 1]], [0, 0, 1.0]])


def sample_appearance(rs, train, point_pipeline=True):
    """(clahe clip limit or 0, blur 0/1): A.CLAHE(p=0.5) draws its clip limit from U(1, 4); A.Blur(blur_limit=3, p=0.5)
    is always the 3x3 box (utils/data.py:122,125,309-310).  Both pipelines of the reference carry the two."""
    if not train:
        return 0.0, 0.0
    clip = float(rs.uniform(1.0, 4.0)) if rs.random_sample() < 0.5 else 0.0
    blur = 1.0 if rs.random_sample() < 0.5 else 0.0
    return clip, blur


def transform_points(pts_xyc, M, H, W):
    """Keypoints (x,y,class) through the forward map; points leaving the image are dropped, the rest floored."""
    p = np.asarray(pts_xyc, dtype=np.float64).reshape(-1, 3)
    p = p[p[:, 2] >= 0]
    if len(p) == 0:
        return np.zeros((0, 3), dtype=np.int64)
    xy = p[:, :2] @ M[:, :2].T + M[:, 2]
    keep = (xy[:, 0] >= 0) & (xy[:, 0] < W) & (xy[:, 1] >= 0) & (xy[:, 1] < H)
    return np.concatenate([np.floor(xy[keep]), p[keep, 2:3]], 1).astype(np.int64)


class LabelMaps:
    """Label maps that were computed ahead of the step on the device, with their superpixel counts already on the host
    (no sync and no padding to a worst-case bound when the trainer preprocesses them)."""

    def __init__(self, labels, counts):
        self.labels, self.counts = labels, counts


class DevicePrefetcher:
    """Wraps a DataLoader over raw items: pinned staging + asynchronous H2D on a copy stream one batch ahead,
    augmentation / ToTensor / one-hot / point rasterisation on the GPU.  Yields the trainer's data tuple
    ``(img, pixel_mask, point_mask)`` (or ``(img, pixel_mask)`` for mask datasets)."""

    def __init__(self, loader, device, train=True, with_points=True, has_masks=True, n_classes=2, seed=0,
                 segment_fn=None):
        """``segment_fn(img (B,3,H,W) float on the device) -> ((B,H,W) int32 label maps, (B,) int32 counts)``: when
        given, the superpixel segmentation of the NEXT batch also runs on the copy stream, beside the training step of
        the current one; its counts are copied to pinned host memory behind it, so that when the batch is consumed they
        are plain integers (exact row count, no host sync in the step) and the label maps travel as a ``LabelMaps``
        fourth element of the data tuple (WESUPTrainer.preprocess takes it)."""
        self.loader, self.device, self.train = loader, torch.device(device), train
        self.with_points, self.has_masks, self.n_classes = with_points, has_masks, n_classes
        self.segment_fn = segment_fn
        self.rs = np.random.RandomState(seed)
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, raw):
        from .. import ops
        self._check_item(raw)
        img, mask, pts, *extra = raw             # extra: what a dataset adds to the item (area bounds, coordinate maps)
        B, H, W, _ = img.shape
        els = []
        rows, mats = zip(*[sample_params(self.rs, H, W, self.train, self.with_points, elastic_out=els) for _ in range(B)])
        rows = np.stack(rows)
        elastic = None
        if any(e is not None for e in els):      # ElasticTransform's displacement field for the images that drew it
            hc, wc = -(-H // ELASTIC_CELL), -(-W // ELASTIC_CELL)
            ef, ep = np.zeros((B, 2, hc, wc), dtype=np.float32), np.zeros((B, 12), dtype=np.float32)
            for b, e in enumerate(els):
                if e is not None:
                    ef[b], ep[b] = e
            elastic = (torch.from_numpy(ef), torch.from_numpy(ep))
        app = np.zeros((B, 8), dtype=np.float32)
        app[:, 0] = 1.0
        for b in range(B):
            app[b, 5], app[b, 6] = sample_appearance(self.rs, self.train, self.with_points)
        need_app = bool((app[:, 5:7] != 0).any()) and min(H, W) >= 8
        if need_app:           # colour first, then CLAHE / Blur, all on the un-warped image; the warp gets neutral colour
            app[:, 0:5] = rows[:, 6:11]
            rows[:, 6], rows[:, 7:11] = 1.0, 0.0
        params = torch.from_numpy(rows)
        negative = [b for b in range(B) if int(pts[b][0, 2]) == Digest2019PointDataset.NEGATIVE]
        bi, ci, yi, xi = [], [], [], []
        if self.with_points:                     # keypoints follow the forward map on the host (a few hundred per image)
            for b in range(B):
                q = transform_points(pts[b].numpy(), mats[b], H, W)
                bi += [b] * len(q); xi += q[:, 0].tolist(); yi += q[:, 1].tolist(); ci += q[:, 2].tolist()
        idx = torch.tensor([bi, ci, yi, xi], dtype=torch.int64).reshape(4, -1)
        with torch.cuda.stream(self.copy_stream):
            d_img = img.pin_memory().to(self.device, non_blocking=True)
            d_mask = mask.pin_memory().to(self.device, non_blocking=True) if self.has_masks else None
            d_par = params.pin_memory().to(self.device, non_blocking=True)
            if need_app:
                d_img = ops.appearance(d_img, torch.from_numpy(app).pin_memory().to(self.device, non_blocking=True))
            d_el = None
            if elastic is not None:
                d_el = (elastic[0].pin_memory().to(self.device, non_blocking=True),
                        elastic[1].pin_memory().to(self.device, non_blocking=True), ELASTIC_CELL)
            out_img, out_mask = ops.augment(d_img, d_mask, d_par, self.n_classes, elastic=d_el)
            point_mask = None
            if self.with_points:
                point_mask = torch.zeros(B, self.n_classes, H, W, dtype=torch.uint8, device=self.device)
                if idx.shape[1]:
                    d_idx = idx.pin_memory().to(self.device, non_blocking=True)
                    point_mask[d_idx[0], d_idx[1], d_idx[2], d_idx[3]] = 1
                for b in negative:                   # Digest-2019 "negative" images: the pixel mask IS the annotation
                    if out_mask is not None:
                        point_mask[b] = out_mask[b]
            segments = counts = None
            if self.segment_fn is not None:
                segments, n_dev = self.segment_fn(out_img)
                counts = torch.empty(n_dev.shape, dtype=n_dev.dtype).pin_memory()
                counts.copy_(n_dev, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        pixel_mask = out_mask if self.has_masks else empty_tensor()
        if segments is not None:
            return (out_img, pixel_mask, point_mask if self.with_points else empty_tensor(), LabelMaps(segments, counts)), done
        item = (out_img, pixel_mask, point_mask) if self.with_points else (out_img, pixel_mask)
        return item + tuple(e.to(self.device, non_blocking=True) if torch.is_tensor(e) else e for e in extra), done

    @staticmethod
    def _check_item(raw):
        """The raw-item layout this prefetcher stages: ``(img uint8 (B,H,W,3), mask uint8 (B,H,W), pts int (B,P,3), ...)``
        -- what the raw datasets of this module collate to.  Anything else (WESUPV2Dataset's ``(img, mask (C,H,W), coords
        (2,H,W))``, CompoundDataset's tuples of items) is refused here, by name, instead of being mis-read further down."""
        if not isinstance(raw, (tuple, list)) or len(raw) < 3 or not all(torch.is_tensor(t) for t in raw[:3]):
            raise TypeError('DevicePrefetcher: expected raw items (img, mask, points, ...) of tensors, got '
                            f'{type(raw).__name__} of {[type(t).__name__ for t in raw] if isinstance(raw, (tuple, list)) else "?"} '
                            '(CompoundDataset items are tuples of items: prefetch its member datasets separately)')
        img, mask, pts = raw[:3]
        if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[-1] != 3:
            raise TypeError(f'DevicePrefetcher: img must be uint8 (B,H,W,3) (a raw, un-normalised item), got {img.dtype} {tuple(img.shape)}')
        if mask.dtype != torch.uint8 or mask.dim() != 3:
            raise TypeError(f'DevicePrefetcher: mask must be a uint8 class-index map (B,H,W), got {mask.dtype} {tuple(mask.shape)} '
                            '(one-hot (C,H,W) int64 masks, e.g. WESUPV2Dataset items, are tensors for the trainer, not raw items)')
        if pts.dim() != 3 or pts.shape[-1] != 3 or pts.is_floating_point():
            raise TypeError(f'DevicePrefetcher: points must be integer (B,P,3) rows of (x, y, class), got {pts.dtype} {tuple(pts.shape)}')

    @staticmethod
    def _hand_over(item):
        """The tensors were allocated on the copy stream and are consumed on the caller's: tell the caching allocator."""
        cur = torch.cuda.current_stream()
        for t in item:
            if isinstance(t, LabelMaps):
                t.labels.record_stream(cur)
            elif torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)
        return item

    def __iter__(self):
        nxt = None
        for raw in self.loader:
            cur, nxt = nxt, self._stage(raw)
            if cur is not None:
                yield self._consume(cur)
        if nxt is not None:
            yield self._consume(nxt)

    def _consume(self, staged):
        item, done = staged
        torch.cuda.current_stream().wait_event(done)
        if isinstance(item[-1], LabelMaps):
            done.synchronize()                   # queued a whole step ago: the counts are on the host by now
            item[-1].counts = [int(v) for v in item[-1].counts]
        return self._hand_over(item)


def get_dataset(root_dir, train=True, proportion=1.0, multiscale_range=None, rescale_factor=None, target_size=None):
    """WESUPTrainer.get_default_dataset (models/wesup.py:436-443): training data with a ``points`` directory is a
    Digest2019PointDataset, everything else a SegmentationDataset ('synthetic:H:W:g:n' makes synthetic items)."""
    root = str(root_dir)
    if 'synthetic:' in root:
        spec = root[root.index('synthetic:'):].split('/')[0].split(':')[1:]
        H, W, g, n = (int(v) for v in (spec + ['480', '480', '24', '16'][len(spec):])[:4])
        return SyntheticGlasDataset(H, W, g, max(1, int(n * proportion)), seed=0 if train else 1)
    root_dir = Path(root_dir)
    if train and (root_dir / 'points').exists():                                       # models/wesup.py:436-443
        return Digest2019PointDataset(root_dir, target_size=target_size, rescale_factor=rescale_factor,
                                      multiscale_range=multiscale_range, train=train, proportion=proportion)
    return SegmentationDataset(root_dir, target_size=target_size, rescale_factor=rescale_factor, train=train,
                               proportion=proportion, multiscale_range=multiscale_range)
