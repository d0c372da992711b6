"""CPU restatement (numpy) of the GPU input-pipeline kernel ``wesup_augment`` -- test infrastructure only.

PARITY UNPINNED.  The reference augments on the CPU with albumentations / OpenCV (utils/data.py:116-133,302-327);
neither library exists in the build image and the reference holds no test or golden vector for it, so there is
nothing to pin against.  This file restates, in numpy, the published definitions the kernel follows:

  * HorizontalFlip / VerticalFlip / ShiftScaleRotate(shift_limit=0.0625, scale_limit=0.1, rotate_limit=45,
    interpolation=linear, border=reflect_101) as ONE affine map about the image centre (albumentations
    ``functional.shift_scale_rotate``: cv2.getRotationMatrix2D(center, angle, scale) then translation dx*W, dy*H,
    applied with cv2.warpAffine, i.e. the kernel receives the INVERSE map output->source);
  * HueSaturationValue on OpenCV's 8-bit HSV (H in [0,180) wrapping, S and V saturating);
  * RandomBrightnessContrast: img*alpha + beta*255, clipped (brightness_by_max=True);
  * ToTensor: /255, HWC -> CHW; masks: nearest, one-hot (utils/data.py:136-142).
Only tests/ may import it."""
import numpy as np


def sample_params(rng, H, W, p_flip=0.5, shift=0.0625, scale=0.1, rotate=45.0, hue=20.0, sat=30.0, val=20.0,
                  brightness=0.3, contrast=0.3, appearance=True, geometry=True):
    """One parameter row (12 floats) + the forward 2x3 matrix (for keypoints).  albumentations defaults of the
    reference's PointSupervisionDataset pipeline (utils/data.py:302-327)."""
    M = np.eye(3)
    if geometry:
        if rng.random_sample() < p_flip:                      # HorizontalFlip
            M = np.array([[-1, 0, W - 1], [0, 1, 0], [0, 0, 1.0]]) @ M
        if rng.random_sample() < p_flip:                      # VerticalFlip
            M = np.array([[1, 0, 0], [0, -1, H - 1], [0, 0, 1.0]]) @ M
        ang = rng.uniform(-rotate, rotate)
        sc = 1.0 + rng.uniform(-scale, scale)
        dx, dy = rng.uniform(-shift, shift), rng.uniform(-shift, shift)
        cx, cy = (W - 1) * 0.5, (H - 1) * 0.5                  # albumentations >= 1.0 uses the pixel-centre convention
        a, b = sc * np.cos(np.deg2rad(ang)), sc * np.sin(np.deg2rad(ang))
        R = np.array([[a, b, (1 - a) * cx - b * cy + dx * W], [-b, a, b * cx + (1 - a) * cy + dy * H], [0, 0, 1.0]])
        M = R @ M
    Minv = np.linalg.inv(M)
    row = np.zeros(12, dtype=np.float32)
    row[0:3], row[3:6] = Minv[0], Minv[1]
    row[6], row[7] = 1.0, 0.0
    if appearance:
        row[6] = 1.0 + rng.uniform(-contrast, contrast)
        row[7] = rng.uniform(-brightness, brightness)
        row[8], row[9], row[10] = rng.uniform(-hue, hue), rng.uniform(-sat, sat), rng.uniform(-val, val)
    return row, M[:2].astype(np.float64)


def reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    period = 2 * n - 2
    i = np.mod(i, period)
    return np.where(i < n, i, period - i)


def rgb_to_hsv8(c):
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    mx, mn = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    d = mx - mn
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.where(mx > 0, np.float32(255.0) * d / mx, np.float32(0))
        hr = np.float32(60.0) * (g - b) / d
        hg = np.float32(120.0) + np.float32(60.0) * (b - r) / d
        hb = np.float32(240.0) + np.float32(60.0) * (r - g) / d
    hh = np.where(mx == r, hr, np.where(mx == g, hg, hb))
    hh = np.where(d > 0, hh, np.float32(0))
    hh = np.where(hh < 0, hh + np.float32(360.0), hh)
    return (np.float32(0.5) * hh).astype(np.float32), s.astype(np.float32), mx.astype(np.float32)


def hsv8_to_rgb(h, s, v):
    hh = h * np.float32(2.0) / np.float32(60.0)
    sf = s / np.float32(255.0)
    sec = np.floor(hh).astype(np.int64) % 6
    f = hh - np.floor(hh)
    p, q, t = v * (1 - sf), v * (1 - sf * f), v * (1 - sf * (1 - f))
    r = np.choose(sec, [v, q, p, p, t, v])
    g = np.choose(sec, [t, v, v, q, p, p])
    b = np.choose(sec, [p, p, t, v, v, q])
    return np.stack([r, g, b], -1).astype(np.float32)


def gaussian_filter_reflect(a, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(a, sigma) (mode='reflect', truncate=4) restated: separable, taps at integer offsets up to
    int(truncate * sigma + 0.5), normalised, reflection about the edge of the border samples."""
    r = int(truncate * sigma + 0.5)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    out = np.asarray(a, dtype=np.float64)
    for axis in (0, 1):
        n = out.shape[axis]
        idx = np.arange(-r, n + r)
        idx = np.mod(idx, 2 * n)
        idx = np.where(idx < n, idx, 2 * n - 1 - idx)
        padded = np.take(out, idx, axis=axis)
        out = sum(wgt * np.take(padded, np.arange(o, o + n), axis=axis) for o, wgt in enumerate(k))
    return out


def elastic_displacement(field, cell, qx, qy):
    """The coarse displacement grid (2, hc, wc) sampled bilinearly at pixel positions (qx, qy), as the kernel does: cell
    centres sit at (i + 1/2) * cell - 1/2, positions outside the grid take the border value."""
    _, hc, wc = field.shape
    f32 = np.float32
    inv = f32(1.0) / f32(cell)
    u = np.clip((qx + f32(0.5)) * inv - f32(0.5), 0, wc - 1).astype(f32)
    v = np.clip((qy + f32(0.5)) * inv - f32(0.5), 0, hc - 1).astype(f32)
    u0, v0 = u.astype(np.int64), v.astype(np.int64)
    u1, v1 = np.minimum(u0 + 1, wc - 1), np.minimum(v0 + 1, hc - 1)
    wu, wv = u - u0.astype(f32), v - v0.astype(f32)
    out = []
    for f in field.astype(f32):
        out.append((1 - wv) * ((1 - wu) * f[v0, u0] + wu * f[v0, u1]) + wv * ((1 - wu) * f[v1, u0] + wu * f[v1, u1]))
    return out[0].astype(f32), out[1].astype(f32)


def augment(img_u8, mask_u8, row, C=2, elastic=None):
    """img (H,W,3) uint8, mask (H,W) uint8 class index (or None), row: 12 floats -> (img f32 (3,H,W), mask u8 (C,H,W)).
    elastic = (field (2,hc,wc), params (12,), cell): ElasticTransform's displacement field (albumentations
    functional.elastic_transform: the affinely warped image is re-sampled at (x + dx, y + dy)), placed by its 12 floats
    {E (source grid -> field grid), L (linear part of E^-1), on}."""
    H, W = img_u8.shape[:2]
    a = row.astype(np.float32)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    sx = a[0] * x + a[1] * y + a[2]
    sy = a[3] * x + a[4] * y + a[5]
    if elastic is not None and elastic[1][10] != 0:
        field, e, cell = elastic
        e = e.astype(np.float32)
        qx, qy = e[0] * sx + e[1] * sy + e[2], e[3] * sx + e[4] * sy + e[5]
        dx, dy = elastic_displacement(field, cell, qx, qy)
        sx, sy = sx + (e[6] * dx + e[7] * dy), sy + (e[8] * dx + e[9] * dy)
    fx, fy = np.floor(sx), np.floor(sy)
    wx, wy = (sx - fx)[..., None], (sy - fy)[..., None]
    x0, x1 = reflect101(fx.astype(np.int64), W), reflect101(fx.astype(np.int64) + 1, W)
    y0, y1 = reflect101(fy.astype(np.int64), H), reflect101(fy.astype(np.int64) + 1, H)
    src = img_u8.astype(np.float32)
    c = (1 - wy) * ((1 - wx) * src[y0, x0] + wx * src[y0, x1]) + wy * ((1 - wx) * src[y1, x0] + wx * src[y1, x1])
    c = c.astype(np.float32)
    if a[8] != 0 or a[9] != 0 or a[10] != 0:
        h, s, v = rgb_to_hsv8(c)
        h = np.fmod(h + a[8] + np.float32(360.0), np.float32(180.0))
        s = np.clip(s + a[9], 0, 255)
        v = np.clip(v + a[10], 0, 255)
        c = hsv8_to_rgb(h.astype(np.float32), s.astype(np.float32), v.astype(np.float32))
    out = np.clip(c * a[6] + a[7] * np.float32(255.0), 0, 255) * np.float32(1.0 / 255.0)
    out_img = np.ascontiguousarray(out.transpose(2, 0, 1)).astype(np.float32)
    out_mask = None
    if mask_u8 is not None:
        mx = reflect101(np.floor(sx + np.float32(0.5)).astype(np.int64), W)
        my = reflect101(np.floor(sy + np.float32(0.5)).astype(np.int64), H)
        cls = mask_u8[my, mx]
        out_mask = np.stack([(cls == k) for k in range(C)]).astype(np.uint8)
    return out_img, out_mask


def transform_points(points_xyc, M, H, W):
    """Keypoints (x, y, class) through the forward map; points that leave the image are dropped, the rest floored
    (albumentations keypoint handling + the reference's integer rasterisation, utils/data.py:352-362)."""
    if len(points_xyc) == 0:
        return np.zeros((0, 3), dtype=np.int64)
    p = np.asarray(points_xyc, dtype=np.float64)
    xy = p[:, :2] @ M[:, :2].T + M[:, 2]
    keep = (xy[:, 0] >= 0) & (xy[:, 0] < W) & (xy[:, 1] >= 0) & (xy[:, 1] < H)
    out = np.concatenate([np.floor(xy[keep]), p[keep, 2:3]], 1)
    return out.astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# wesup_appearance: HueSaturationValue -> RandomBrightnessContrast -> CLAHE -> Blur(3) on uint8 images.
# Restated from the published algorithms (OpenCV's CLAHE: modules/imgproc/src/clahe.cpp; cv2.blur; 8-bit Lab), all in
# float32 like the kernel.  PARITY UNPINNED (no OpenCV / albumentations in the image).
# ---------------------------------------------------------------------------------------------------------------------
F32 = np.float32


def _sat_u8(v):
    return np.clip(np.rint(v), 0, 255).astype(np.float32)


def _srgb_to_linear(c):
    return np.where(c <= F32(0.04045), c * F32(1 / 12.92), np.power((c + F32(0.055)) * F32(1 / 1.055), F32(2.4))).astype(np.float32)


def _linear_to_srgb(c):
    return np.where(c <= F32(0.0031308), F32(12.92) * c, F32(1.055) * np.power(c, F32(1 / 2.4)) - F32(0.055)).astype(np.float32)


def _lab_fwd(t):
    return np.where(t > F32(0.008856), np.cbrt(t), F32(7.787) * t + F32(16.0 / 116.0)).astype(np.float32)


def rgb8_to_lab8(c):
    """(...,3) float32 RGB in 0..255 -> L8, a8, b8 as OpenCV's 8-bit Lab stores them (rounded)."""
    r, g, b = (_srgb_to_linear(c[..., k] * F32(1 / 255.0)) for k in range(3))
    X = (F32(0.412453) * r + F32(0.357580) * g + F32(0.180423) * b) * F32(1 / 0.950456)
    Y = F32(0.212671) * r + F32(0.715160) * g + F32(0.072169) * b
    Z = (F32(0.019334) * r + F32(0.119193) * g + F32(0.950227) * b) * F32(1 / 1.088754)
    fx, fy, fz = _lab_fwd(X), _lab_fwd(Y), _lab_fwd(Z)
    L = np.where(Y > F32(0.008856), F32(116.0) * fy - F32(16.0), F32(903.3) * Y)
    return (_sat_u8(L * F32(2.55)), _sat_u8(F32(500.0) * (fx - fy) + F32(128.0)), _sat_u8(F32(200.0) * (fy - fz) + F32(128.0)))


def lab8_to_rgb8(L8, a8, b8):
    L, a, b = L8 * F32(100.0 / 255.0), a8 - F32(128.0), b8 - F32(128.0)
    fy = (L + F32(16.0)) * F32(1 / 116.0)
    fx, fz = fy + a * F32(1 / 500.0), fy - b * F32(1 / 200.0)
    inv = lambda f: np.where(f > F32(0.206893), f * f * f, (f - F32(16.0 / 116.0)) * F32(1 / 7.787))
    Y = np.where(L > F32(7.9996), fy * fy * fy, L * F32(1 / 903.3))
    X, Z = inv(fx) * F32(0.950456), inv(fz) * F32(1.088754)
    r = F32(3.240479) * X - F32(1.537150) * Y - F32(0.498535) * Z
    g = F32(-0.969256) * X + F32(1.875991) * Y + F32(0.041556) * Z
    bl = F32(0.055648) * X - F32(0.204043) * Y + F32(1.057311) * Z
    return np.stack([_sat_u8(F32(255.0) * _linear_to_srgb(np.clip(ch, 0, 1).astype(np.float32))) for ch in (r, g, bl)], -1)


def clahe_l8(L8, clip_limit, tiles=8):
    """OpenCV CLAHE on one uint8-valued channel (H,W) float32 -> equalised channel."""
    H, W = L8.shape
    Hp = H + (tiles - H % tiles) % tiles
    Wp = W + (tiles - W % tiles) % tiles
    ys, xs = reflect101(np.arange(Hp), H), reflect101(np.arange(Wp), W)
    ext = L8[ys][:, xs].astype(np.int64)
    th, tw = Hp // tiles, Wp // tiles
    area = th * tw
    clip = max(int(F32(clip_limit) * F32(area) / F32(256.0)), 1)
    luts = np.zeros((tiles, tiles, 256), dtype=np.float32)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256)
            clipped = int(np.maximum(hist - clip, 0).sum())
            hist = np.minimum(hist, clip)
            batch = clipped // 256
            residual = clipped - batch * 256
            hist = hist + batch
            if residual > 0:
                step = max(256 // residual, 1)
                i = 0
                while i < 256 and residual > 0:
                    hist[i] += 1
                    i += step
                    residual -= 1
            luts[ty, tx] = _sat_u8(np.cumsum(hist).astype(np.float32) * (F32(255.0) / F32(area)))
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    tyf, txf = y * (F32(1.0) / F32(th)) - F32(0.5), x * (F32(1.0) / F32(tw)) - F32(0.5)
    ty1, tx1 = np.floor(tyf).astype(np.int64), np.floor(txf).astype(np.int64)
    ya, xa = (tyf - ty1).astype(np.float32), (txf - tx1).astype(np.float32)
    ty2, tx2 = np.minimum(ty1 + 1, tiles - 1), np.minimum(tx1 + 1, tiles - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = L8.astype(np.int64)
    res = (luts[ty1, tx1, v] * (1 - xa) + luts[ty1, tx2, v] * xa) * (1 - ya) + (luts[ty2, tx1, v] * (1 - xa) + luts[ty2, tx2, v] * xa) * ya
    return _sat_u8(res)


def blur3(img):
    """cv2.blur(img, (3,3)), BORDER_REFLECT_101, on (H,W,3) uint8-valued float32."""
    H, W = img.shape[:2]
    acc = np.zeros_like(img, dtype=np.float32)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            acc += img[reflect101(np.arange(H) + dy, H)][:, reflect101(np.arange(W) + dx, W)]
    return _sat_u8(acc * F32(1.0 / 9.0))


def appearance(img_u8, row):
    """img (H,W,3) uint8, row: 8 floats {alpha, beta, hue, sat, val, clahe_clip, blur, 0} -> (H,W,3) uint8."""
    a = np.asarray(row, dtype=np.float32)
    c = img_u8.astype(np.float32)
    if a[2] != 0 or a[3] != 0 or a[4] != 0:
        h, s, v = rgb_to_hsv8(c)
        h, s = np.rint(h), np.rint(s)
        h = np.fmod(h + a[2] + F32(360.0), F32(180.0))
        s = np.clip(s + a[3], 0, 255)
        v = np.clip(v + a[4], 0, 255)
        c = _sat_u8(hsv8_to_rgb(h.astype(np.float32), s.astype(np.float32), v.astype(np.float32)))
    c = _sat_u8(c * a[0] + a[1] * F32(255.0))
    if a[5] > 0:
        L8, a8, b8 = rgb8_to_lab8(c)
        c = lab8_to_rgb8(clahe_l8(L8, a[5]), a8, b8)
    if a[6] != 0:
        c = blur3(c)
    return c.astype(np.uint8)
