"""Synthetic GlaS-shaped inputs for tests and bench (numpy only, no torch).

The training-step hot path takes, per image: an RGB patch in [0, 1], a SLIC
label map and a sparse point-annotation mask (reference data contract:
``utils/data.py:135-152,459-512``; SLIC call ``models/wesup.py:471-476``).
SLIC is outside the path, so the label map here is a
jittered-grid Voronoi tessellation with exactly ``g*g`` contiguous 0-based ids
(SURVEY.md 8(d)).  Everything is drawn from ``np.random.RandomState`` whose
stream is frozen, so the GPU box regenerates identical inputs.
"""
import numpy as np


def synth_image(seed, H, W):
    """(3, H, W) float32 H&E-like patch in [0, 1)."""
    rs = np.random.RandomState(seed)
    base = np.array([0.8, 0.5, 0.7], dtype=np.float32)[:, None, None]
    lo = rs.rand(3, (H + 15) // 16 + 1, (W + 15) // 16 + 1).astype(np.float32)
    lo = np.repeat(np.repeat(lo, 16, axis=1), 16, axis=2)[:, :H, :W]
    hi = rs.rand(3, H, W).astype(np.float32)
    img = 0.5 * base + 0.25 * lo + 0.25 * hi
    return np.clip(img, 0.0, 0.999).astype(np.float32)


def voronoi_labels(seed, H, W, g, jitter=0.3):
    """(H, W) int32 label map with exactly g*g ids 0..g*g-1, every id non-empty.

    Seeds sit on a g x g grid, each jittered by +-``jitter`` cell; a pixel takes
    the nearest seed among the 5x5 neighbouring cells (enough for jitter<=0.5).
    """
    rs = np.random.RandomState(seed)
    ch, cw = H / g, W / g
    jy = (rs.rand(g, g) * 2 - 1) * jitter
    jx = (rs.rand(g, g) * 2 - 1) * jitter
    sy = (np.arange(g)[:, None] + 0.5 + jy) * ch          # (g, g)
    sx = (np.arange(g)[None, :] + 0.5 + jx) * cw
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32) + 0.5,
                         np.arange(W, dtype=np.float32) + 0.5, indexing='ij')
    cy = np.minimum((yy / ch).astype(np.int64), g - 1)
    cx = np.minimum((xx / cw).astype(np.int64), g - 1)
    best = np.full((H, W), np.inf, dtype=np.float64)
    lab = np.zeros((H, W), dtype=np.int64)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            ny = np.clip(cy + dy, 0, g - 1)
            nx = np.clip(cx + dx, 0, g - 1)
            d = (yy - sy[ny, nx]) ** 2 + (xx - sx[ny, nx]) ** 2
            idx = ny * g + nx
            upd = (d < best) | ((d == best) & (idx < lab))
            best = np.where(upd, d, best)
            lab = np.where(upd, idx, lab)
    # make sure every id owns at least its own seed pixel
    py = np.clip(sy.astype(np.int64), 0, H - 1)
    px = np.clip(sx.astype(np.int64), 0, W - 1)
    lab[py, px] = np.arange(g * g).reshape(g, g)
    assert len(np.unique(lab)) == g * g
    return lab.astype(np.int32)


def skewed_labels(seed, H, W, g, big=4):
    """Adversarial label map: ``big`` superpixels ~50x the median area."""
    lab = voronoi_labels(seed, H, W, g)
    rs = np.random.RandomState(seed + 7919)
    n = g * g
    # merge a 7x7 patch of cells into one id, then renumber contiguously
    ids = np.arange(n).reshape(g, g)
    remap = np.arange(n)
    for k in range(big):
        y0 = rs.randint(0, max(1, g - 7))
        x0 = rs.randint(0, max(1, g - 7))
        blk = ids[y0:y0 + 7, x0:x0 + 7].ravel()
        remap[blk] = remap[blk[0]]
    uniq, inv = np.unique(remap, return_inverse=True)
    return inv[lab].astype(np.int32)


def point_mask(seed, labels, frac=0.2, n_classes=2, tie_every=0):
    """(C, H, W) uint8 one-hot point annotations, radius 0.

    ``round(frac*N)`` superpixels get one labelled pixel each; with
    ``tie_every`` > 0 every tie_every-th of them gets a second point of the
    other class (label-fraction tie -> multi-hot row, models/wesup.py:50-52).
    """
    rs = np.random.RandomState(seed)
    H, W = labels.shape
    n = int(labels.max()) + 1
    k = int(round(frac * n))
    chosen = rs.choice(n, size=k, replace=False)
    mask = np.zeros((n_classes, H, W), dtype=np.uint8)
    flat = labels.ravel()
    order = np.argsort(flat, kind='stable')
    starts = np.searchsorted(flat[order], np.arange(n))
    ends = np.searchsorted(flat[order], np.arange(n), side='right')
    for i, sp in enumerate(chosen):
        pix = order[starts[sp]:ends[sp]]
        p = pix[rs.randint(len(pix))]
        c = rs.randint(n_classes)
        mask[c, p // W, p % W] = 1
        if tie_every and i % tie_every == 0 and len(pix) > 1:
            q = pix[(np.where(pix == p)[0][0] + 1) % len(pix)]
            mask[(c + 1) % n_classes, q // W, q % W] = 1
    return mask


def pixel_mask(seed, H, W, n_classes=2):
    """(C, H, W) uint8 dense one-hot mask: blobby two-class ground truth."""
    rs = np.random.RandomState(seed)
    lo = rs.rand((H + 31) // 32 + 1, (W + 31) // 32 + 1)
    lo = np.repeat(np.repeat(lo, 32, axis=0), 32, axis=1)[:H, :W]
    cls = (lo * n_classes).astype(np.int64).clip(0, n_classes - 1)
    mask = np.zeros((n_classes, H, W), dtype=np.uint8)
    for c in range(n_classes):
        mask[c][cls == c] = 1
    return mask


def make_batch(seed, B, H, W, g, frac=0.2, tie_every=0):
    """Batch of B independent synthetic samples (imgs, labels, point masks, pixel masks)."""
    imgs = np.stack([synth_image(seed * 1000 + b, H, W) for b in range(B)])
    labs = np.stack([voronoi_labels(seed * 1000 + b, H, W, g) for b in range(B)])
    pts = np.stack([point_mask(seed * 1000 + b, labs[b], frac, 2, tie_every) for b in range(B)])
    pix = np.stack([pixel_mask(seed * 1000 + b, H, W) for b in range(B)])
    return imgs, labs, pts, pix
