"""CPU restatement of SLIC superpixels for checking wesup_slic.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference calls ``skimage.segmentation.slic`` (models/wesup.py:9,471-476), a third-party
dependency that is absent from /root/reference and unpinned (requirements.txt:9), and no reference test touches it.
This file restates the published algorithm (Achanta et al., "SLIC Superpixels Compared to State-of-the-art
Superpixel Methods", TPAMI 2012) with skimage's parameterisation (Lab colours divided by the compactness, grid step
round(sqrt(HW/n)) starting at floor(sqrt(HW/n)/2), 10 iterations, centre-wise 2S x 2S windows, connectivity
enforcement with min_size_factor 0.5).  The connectivity rule (absorb a small component into the component of the
pixel left of / above its first pixel; renumber in raster order) is this project's own deterministic choice.
"""
import numpy as np


def rgb2lab(img):
    """img (3,H,W) float in [0,1] -> (H,W,3) Lab (D65, as skimage.color.rgb2lab)."""
    rgb = np.moveaxis(img.astype(np.float32), 0, -1)
    lin = np.where(rgb > 0.04045, ((rgb + 0.055) / 1.055) ** 2.4, rgb / 12.92).astype(np.float32)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]],
                 dtype=np.float32)
    xyz = lin @ m.T
    xyz = xyz / np.array([0.95047, 1.0, 1.08883], dtype=np.float32)
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0).astype(np.float32)
    L = 116.0 * f[..., 1] - 16.0
    a = 500.0 * (f[..., 0] - f[..., 1])
    b = 200.0 * (f[..., 1] - f[..., 2])
    return np.stack([L, a, b], axis=-1).astype(np.float32)


def grid(H, W, n_segments):
    s = np.sqrt(H * W / n_segments)
    step = max(1, int(np.floor(s + 0.5)))
    start = min(int(np.floor(s / 2.0)), H - 1, W - 1)
    ys = np.arange(start, H, step)
    xs = np.arange(start, W, step)
    return ys, xs, step


def kmeans_labels(img, n_segments, compactness=40.0, max_iter=10):
    """Centre indices per pixel after max_iter rounds (no connectivity step)."""
    H, W = img.shape[1:]
    lab = rgb2lab(img) / np.float32(compactness)
    ys, xs, step = grid(H, W, n_segments)
    cy, cx = np.meshgrid(ys, xs, indexing='ij')
    cen = np.concatenate([cy.reshape(-1, 1).astype(np.float32), cx.reshape(-1, 1).astype(np.float32),
                          lab[cy.ravel(), cx.ravel()]], axis=1)          # (K, 5): y, x, L, a, b
    K = cen.shape[0]
    S = np.float32(step)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    labels = np.zeros((H, W), dtype=np.int64)
    for _ in range(max_iter):
        best = np.full((H, W), np.float32(3.0e38))
        labels[:] = -1
        for k in range(K):                                   # ascending k, strict <: ties keep the lower index
            y0 = int(max(0, np.ceil(cen[k, 0] - 2 * S)))
            y1 = int(min(H - 1, np.floor(cen[k, 0] + 2 * S)))
            x0 = int(max(0, np.ceil(cen[k, 1] - 2 * S)))
            x1 = int(min(W - 1, np.floor(cen[k, 1] + 2 * S)))
            if y1 < y0 or x1 < x0:
                continue
            dy = yy[y0:y1 + 1, x0:x1 + 1] - cen[k, 0]
            dx = xx[y0:y1 + 1, x0:x1 + 1] - cen[k, 1]
            dc = lab[y0:y1 + 1, x0:x1 + 1] - cen[k, 2:5]
            d = (dc[..., 0] ** 2 + dc[..., 1] ** 2 + dc[..., 2] ** 2) + (dy * dy + dx * dx) / (S * S)
            upd = d < best[y0:y1 + 1, x0:x1 + 1]
            best[y0:y1 + 1, x0:x1 + 1][upd] = d[upd]
            labels[y0:y1 + 1, x0:x1 + 1][upd] = k
        flat = labels.ravel()
        ok = flat >= 0
        cnt = np.bincount(flat[ok], minlength=K)
        for j, v in enumerate([yy.ravel(), xx.ravel(), lab[..., 0].ravel(), lab[..., 1].ravel(), lab[..., 2].ravel()]):
            s = np.bincount(flat[ok], weights=v[ok].astype(np.float64), minlength=K)
            cen[cnt > 0, j] = (s[cnt > 0] / cnt[cnt > 0]).astype(np.float32)
    return labels


def enforce_connectivity(labels, n_segments, min_size_factor=0.5):
    """4-connected components; components smaller than min_size are absorbed by the component of the pixel left of
    (else above) their first pixel (chains resolved towards earlier pixels); ids renumbered 0..K-1 in raster order
    of the first pixel of each surviving component."""
    from scipy import ndimage
    H, W = labels.shape
    HW = H * W
    min_size = int(np.float32(min_size_factor) * np.float32(HW) / np.float32(n_segments))
    comp = np.zeros((H, W), dtype=np.int64)
    ncomp = 0
    for v in np.unique(labels):
        c, n = ndimage.label(labels == v)                    # 4-connectivity is scipy's default structure
        comp[c > 0] = c[c > 0] + ncomp
        ncomp += n
    flat = comp.ravel() - 1
    first = np.full(ncomp, HW, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(HW))
    size = np.bincount(flat, minlength=ncomp)
    root = first[flat]                                       # first pixel of each pixel's component
    target = {}
    for c in range(ncomp):
        p = first[c]
        t = p
        if size[c] < min_size:
            if p % W > 0:
                t = root[p - 1]
            elif p >= W:
                t = root[p - W]
        target[p] = t
    final = np.empty(HW, dtype=np.int64)
    cache = {}
    for p in sorted(target):                                 # ascending: targets are already resolved
        t = target[p]
        cache[p] = p if t == p else cache[t]
    final = np.array([cache[r] for r in root])
    survivors = np.unique(final)                             # sorted = raster order of first pixels
    newid = {r: i for i, r in enumerate(survivors)}
    return np.array([newid[r] for r in final], dtype=np.int32).reshape(H, W), len(survivors)


def slic(img, n_segments, compactness=40.0, max_iter=10, min_size_factor=0.5):
    return enforce_connectivity(kmeans_labels(img, n_segments, compactness, max_iter), n_segments, min_size_factor)
