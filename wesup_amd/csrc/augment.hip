// Input pipeline on the GPU: the per-item augmentation the reference runs on the CPU in
// DataLoader workers (albumentations, utils/data.py:116-133,302-327) as one batched kernel over uint8 images that
// were only decoded and resized on the host.
//
//   geometry   : HorizontalFlip, VerticalFlip, ShiftScaleRotate folded into ONE inverse affine map per image
//                (output pixel -> source position), bilinear for the image, nearest for the label mask, borders
//                mirrored without repeating the edge pixel (cv2.BORDER_REFLECT_101, the albumentations default)
//   appearance : HueSaturationValue (OpenCV 8-bit HSV conventions: H in [0,180), S,V in [0,255]) and
//                RandomBrightnessContrast (img*alpha + beta*255), applied to the interpolated colour
//   output     : img fp32 NCHW in [0,1] (TF.to_tensor, utils/data.py:136) and the one-hot uint8 mask (C,H,W)
//                (utils/data.py:140-142) the superpixel preprocessing reads
// wesup_appearance (below) runs the appearance transforms that need a neighbourhood -- CLAHE and the 3x3 Blur -- on
// the un-warped uint8 image, after HueSaturationValue / RandomBrightnessContrast in the reference's order
// (utils/data.py:119-129, 306-312); ElasticTransform with the albumentations defaults (alpha 1, sigma 50,
// alpha_affine 50) is its random 3-point affine (the displacement field of alpha 1 under a sigma-50 blur is < 0.05 px)
// and is folded into the inverse affine map on the host (utils/data.py sample_params).
// Parity with albumentations/OpenCV is unpinned (both absent from the build image): oracle/augment_oracle.py restates
// these formulas in numpy and the GPU test compares against it.
#include "common.hpp"

struct AugParams {          // 12 floats per image
    float a00, a01, a02, a10, a11, a12;     // source = A * (x, y, 1)
    float alpha, beta;                      // contrast gain, brightness offset (fraction of 255)
    float hue, sat, val;                    // additive shifts in OpenCV 8-bit HSV units
    float pad;
};

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i = i % period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__device__ __forceinline__ void rgb_to_hsv8(float r, float g, float b, float& h, float& s, float& v) {
    // OpenCV COLOR_RGB2HSV on 8-bit data, kept in float: V = max, S = 255*(V-min)/V, H = 30*sector angle (0..180)
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    v = mx;
    s = mx > 0.f ? 255.f * d / mx : 0.f;
    float hh = 0.f;
    if (d > 0.f) {
        if (mx == r) hh = 60.f * (g - b) / d;
        else if (mx == g) hh = 120.f + 60.f * (b - r) / d;
        else hh = 240.f + 60.f * (r - g) / d;
        if (hh < 0.f) hh += 360.f;
    }
    h = 0.5f * hh;
}

__device__ __forceinline__ void hsv8_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float hh = h * 2.f / 60.f;                 // sector 0..6
    const float sf = s / 255.f;
    const int sec = ((int)floorf(hh)) % 6;
    const float f = hh - floorf(hh);
    const float p = v * (1.f - sf), q = v * (1.f - sf * f), t = v * (1.f - sf * (1.f - f));
    switch (sec) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// ElasticTransform's displacement field (albumentations ElasticTransform(alpha=1, sigma=50), utils/data.py:124:
// dx, dy = gaussian_filter(U(-1, 1) per pixel, sigma) * alpha, the image is re-sampled at (x + dx, y + dy) AFTER the
// transform's own random affine).  The field is smooth on the scale of sigma, so it arrives as a coarse grid (one value
// per cell x cell pixels, already smoothed: host side, utils/data.py elastic_field) and is interpolated bilinearly here.
// Per image 12 floats: E (2x3: source grid -> the grid the field lives on, i.e. the forward affine of the transform),
// L (2x2: linear part of E^-1, which takes the displacement back to the source grid), on (0/1), pad.
struct ElasticParams {
    float e00, e01, e02, e10, e11, e12;
    float l00, l01, l10, l11;
    float on, pad;
};
struct ElasticField {
    const float* field;          // [B][2][hc][wc] (dx plane, dy plane), or NULL
    const ElasticParams* par;    // [B]
    int hc, wc, cell;
    int pad_;                    // (explicit padding, zero: launch.hpp)
};
__device__ __forceinline__ float elastic_sample(const float* f, int hc, int wc, float u, float v) {
    u = fminf(fmaxf(u, 0.f), (float)(wc - 1));
    v = fminf(fmaxf(v, 0.f), (float)(hc - 1));
    const int u0 = (int)u, v0 = (int)v;
    const int u1 = min(u0 + 1, wc - 1), v1 = min(v0 + 1, hc - 1);
    const float wu = u - (float)u0, wv = v - (float)v0;
    return (1.f - wv) * ((1.f - wu) * f[v0 * wc + u0] + wu * f[v0 * wc + u1]) +
           wv * ((1.f - wu) * f[v1 * wc + u0] + wu * f[v1 * wc + u1]);
}

__global__ void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                               const AugParams* __restrict__ params, float* __restrict__ out_img,
                               uint8_t* __restrict__ out_mask, int H, int W, int C, const ElasticField el) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - (long)y * W;
    const AugParams a = params[b];
    float sx = a.a00 * x + a.a01 * y + a.a02, sy = a.a10 * x + a.a11 * y + a.a12;
    if (el.field) {
        const ElasticParams e = el.par[b];
        if (e.on != 0.f) {
            // q: this output pixel on the grid of the affinely warped image, where the field is defined
            const float qx = e.e00 * sx + e.e01 * sy + e.e02, qy = e.e10 * sx + e.e11 * sy + e.e12;
            const float inv = 1.f / (float)el.cell;
            const float u = (qx + 0.5f) * inv - 0.5f, v = (qy + 0.5f) * inv - 0.5f;      // cell centres sit at (i + 1/2) cell - 1/2
            const float* f = el.field + (long)b * 2 * el.hc * el.wc;
            const float dx = elastic_sample(f, el.hc, el.wc, u, v);
            const float dy = elastic_sample(f + el.hc * el.wc, el.hc, el.wc, u, v);
            sx += e.l00 * dx + e.l01 * dy;
            sy += e.l10 * dx + e.l11 * dy;
        }
    }
    const float fx = floorf(sx), fy = floorf(sy);
    const float wx = sx - fx, wy = sy - fy;
    const int x0 = reflect101((int)fx, W), x1 = reflect101((int)fx + 1, W);
    const int y0 = reflect101((int)fy, H), y1 = reflect101((int)fy + 1, H);
    const uint8_t* src = img + b * HW * 3;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v00 = src[((long)y0 * W + x0) * 3 + k], v01 = src[((long)y0 * W + x1) * 3 + k];
        const float v10 = src[((long)y1 * W + x0) * 3 + k], v11 = src[((long)y1 * W + x1) * 3 + k];
        c[k] = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    }
    // HueSaturationValue on 8-bit conventions: hue wraps modulo 180, saturation and value saturate at [0, 255]
    if (a.hue != 0.f || a.sat != 0.f || a.val != 0.f) {
        float h, s, v;
        rgb_to_hsv8(c[0], c[1], c[2], h, s, v);
        h = fmodf(h + a.hue + 360.f, 180.f);
        s = fminf(fmaxf(s + a.sat, 0.f), 255.f);
        v = fminf(fmaxf(v + a.val, 0.f), 255.f);
        hsv8_to_rgb(h, s, v, c[0], c[1], c[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = fminf(fmaxf(c[k] * a.alpha + a.beta * 255.f, 0.f), 255.f);
        out_img[(b * 3 + k) * HW + p] = v * (1.f / 255.f);
    }
    if (mask) {
        // nearest neighbour (cv2.INTER_NEAREST): round half away from zero on the source position
        const int mx = reflect101((int)floorf(sx + 0.5f), W), my = reflect101((int)floorf(sy + 0.5f), H);
        const int cls = mask[b * HW + (long)my * W + mx];
        for (int k = 0; k < C; ++k) out_mask[(b * C + k) * HW + p] = (cls == k) ? 1 : 0;
    }
}

extern "C" int wesup_augment(const uint8_t* img_hwc, const uint8_t* mask_hw, const float* params, const float* elastic_field,
                             const float* elastic_params, int hc, int wc, int cell, float* out_img_nchw,
                             uint8_t* out_mask_chw, int B, int H, int W, int C, void* stream) {
    if (!img_hwc || !params || !out_img_nchw || B <= 0 || H <= 0 || W <= 0 || C <= 0 || B > 65535) return WESUP_ERR_INVALID;
    if (mask_hw && !out_mask_chw) return WESUP_ERR_INVALID;
    if (elastic_field && (!elastic_params || hc <= 0 || wc <= 0 || cell <= 0)) return WESUP_ERR_INVALID;
    const long HW = (long)H * W;
    const ElasticField el = {elastic_field, reinterpret_cast<const ElasticParams*>(elastic_params), hc, wc, cell, 0};
    WESUP_LAUNCH(augment_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, img_hwc,
                       mask_hw, reinterpret_cast<const AugParams*>(params), out_img_nchw, out_mask_chw, H, W, C, el);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// wesup_appearance: HueSaturationValue -> RandomBrightnessContrast -> CLAHE -> Blur(3) on uint8 RGB images, each stage
// rounding to uint8 as the albumentations / OpenCV 8-bit pipeline does (utils/data.py:119-125, 306-312).
//   CLAHE (cv2.createCLAHE(clipLimit, tileGridSize=(8, 8)) on the L channel of 8-bit Lab, albumentations.CLAHE): the image
//   is cut into 8 x 8 tiles (bottom / right padded by reflection to a multiple of 8), per tile a 256-bin histogram of L
//   is clipped at max(1, int(clip * tile_area / 256)), the excess spread evenly (the remainder one count every
//   256/remainder bins), its scaled cumulative sum is the tile's look-up table, and a pixel takes the bilinear blend of
//   the tables of the four nearest tile centres.
//   Blur: cv2.blur 3x3 box, BORDER_REFLECT_101, rounded.
// Parity with OpenCV is unpinned (absent from the image); oracle/augment_oracle.py restates the same formulas in numpy.
// ------------------------------------------------------------------------------------------------------------------
struct AppParams {          // 8 floats per image
    float alpha, beta;      // contrast gain, brightness offset (fraction of 255)
    float hue, sat, val;    // OpenCV 8-bit HSV shifts
    float clahe_clip;       // 0: no CLAHE; else the clip limit (albumentations draws it from U(1, 4))
    float blur;             // 0 / 1: 3x3 box blur
    float pad;
};
#define CLAHE_TILES 8

__device__ __forceinline__ uint8_t sat_u8(float v) { return (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f); }

__device__ __forceinline__ float srgb_to_linear(float c) { return c <= 0.04045f ? c * (1.f / 12.92f) : powf((c + 0.055f) * (1.f / 1.055f), 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f; }
__device__ __forceinline__ float lab_fwd(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; }
// 8-bit Lab as OpenCV stores it: L * 255/100, a + 128, b + 128, each rounded
__device__ __forceinline__ void rgb8_to_lab8(float r8, float g8, float b8, float& L8, float& a8, float& bb8) {
    const float r = srgb_to_linear(r8 * (1.f / 255.f)), g = srgb_to_linear(g8 * (1.f / 255.f)), b = srgb_to_linear(b8 * (1.f / 255.f));
    const float X = (0.412453f * r + 0.357580f * g + 0.180423f * b) * (1.f / 0.950456f);
    const float Y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    const float Z = (0.019334f * r + 0.119193f * g + 0.950227f * b) * (1.f / 1.088754f);
    const float fx = lab_fwd(X), fy = lab_fwd(Y), fz = lab_fwd(Z);
    const float L = Y > 0.008856f ? 116.f * fy - 16.f : 903.3f * Y;
    L8 = fminf(fmaxf(rintf(L * 2.55f), 0.f), 255.f);
    a8 = fminf(fmaxf(rintf(500.f * (fx - fy) + 128.f), 0.f), 255.f);
    bb8 = fminf(fmaxf(rintf(200.f * (fy - fz) + 128.f), 0.f), 255.f);
}
__device__ __forceinline__ float lab_inv(float f) { return f > 0.206893f ? f * f * f : (f - 16.f / 116.f) * (1.f / 7.787f); }
__device__ __forceinline__ void lab8_to_rgb8(float L8, float a8, float b8, uint8_t& r8, uint8_t& g8, uint8_t& bb8) {
    const float L = L8 * (100.f / 255.f), a = a8 - 128.f, b = b8 - 128.f;
    const float fy = (L + 16.f) * (1.f / 116.f), fx = fy + a * (1.f / 500.f), fz = fy - b * (1.f / 200.f);
    const float Y = L > 7.9996f ? fy * fy * fy : L * (1.f / 903.3f);
    const float X = lab_inv(fx) * 0.950456f, Z = lab_inv(fz) * 1.088754f;
    const float r = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
    const float g = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    const float bl = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    r8 = sat_u8(255.f * linear_to_srgb(fminf(fmaxf(r, 0.f), 1.f)));
    g8 = sat_u8(255.f * linear_to_srgb(fminf(fmaxf(g, 0.f), 1.f)));
    bb8 = sat_u8(255.f * linear_to_srgb(fminf(fmaxf(bl, 0.f), 1.f)));
}

// stage 1: HSV shift + brightness / contrast on the source grid, uint8 in -> uint8 out
__global__ void app_color_kernel(const uint8_t* __restrict__ img, const AppParams* __restrict__ params,
                                 uint8_t* __restrict__ out, long HW) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const AppParams a = params[b];
    const uint8_t* s = img + (b * HW + p) * 3;
    float c[3] = {(float)s[0], (float)s[1], (float)s[2]};
    if (a.hue != 0.f || a.sat != 0.f || a.val != 0.f) {
        float h, sa, v;
        rgb_to_hsv8(c[0], c[1], c[2], h, sa, v);
        h = rintf(h); sa = rintf(sa);                       // the 8-bit HSV image OpenCV hands to albumentations
        h = fmodf(h + a.hue + 360.f, 180.f);
        sa = fminf(fmaxf(sa + a.sat, 0.f), 255.f);
        v = fminf(fmaxf(v + a.val, 0.f), 255.f);
        hsv8_to_rgb(h, sa, v, c[0], c[1], c[2]);
        c[0] = (float)sat_u8(c[0]); c[1] = (float)sat_u8(c[1]); c[2] = (float)sat_u8(c[2]);
    }
    uint8_t* o = out + (b * HW + p) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = sat_u8(c[k] * a.alpha + a.beta * 255.f);
}

// stage 2a: one block per (tile, image): histogram of L8 over the tile (reflected padding), clip, redistribute, LUT
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ img, const AppParams* __restrict__ params,
                                                        uint8_t* __restrict__ lut, int H, int W) {
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const float clip_f = params[b].clahe_clip;
    if (clip_f <= 0.f) return;
    __shared__ int hist[256];
    __shared__ int scan[256];
    hist[tid] = 0;
    __syncthreads();
    const int Hp = (H % CLAHE_TILES) ? H + CLAHE_TILES - H % CLAHE_TILES : H;
    const int Wp = (W % CLAHE_TILES) ? W + CLAHE_TILES - W % CLAHE_TILES : W;
    const int th = Hp / CLAHE_TILES, tw = Wp / CLAHE_TILES;
    const int ty = tile / CLAHE_TILES, tx = tile % CLAHE_TILES;
    const long HW = (long)H * W;
    for (int i = tid; i < th * tw; i += 256) {
        const int y = reflect101(ty * th + i / tw, H), x = reflect101(tx * tw + i % tw, W);    // copyMakeBorder(REFLECT_101)
        const uint8_t* s = img + (b * HW + (long)y * W + x) * 3;
        float L, a8, b8;
        rgb8_to_lab8(s[0], s[1], s[2], L, a8, b8);
        atomicAdd(&hist[(int)L], 1);                         // integer counts: order-independent
    }
    __syncthreads();
    const int area = th * tw;
    int clip = (int)(clip_f * area / 256.f);
    clip = clip < 1 ? 1 : clip;
    // clipped excess (block reduction through the scan buffer)
    const int h0 = hist[tid];
    scan[tid] = h0 > clip ? h0 - clip : 0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) scan[tid] += scan[tid + off];
        __syncthreads();
    }
    const int clipped = scan[0];
    __syncthreads();
    const int batch = clipped / 256;
    int residual = clipped - batch * 256;
    int h = (h0 > clip ? clip : h0) + batch;
    if (residual > 0) {
        const int step = 256 / residual > 1 ? 256 / residual : 1;
        // OpenCV: for (i = 0; i < 256 && residual > 0; i += step, --residual) ++hist[i]
        if (tid % step == 0 && tid / step < residual) ++h;
    }
    // inclusive scan of the 256 bins
    scan[tid] = h;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const float scale = 255.f / (float)area;
    lut[((long)b * CLAHE_TILES * CLAHE_TILES + tile) * 256 + tid] = sat_u8((float)scan[tid] * scale);
}

// stage 2b: blend the four nearest tiles' tables for the pixel's L, back to RGB
__global__ void clahe_apply_kernel(const uint8_t* __restrict__ img, const AppParams* __restrict__ params,
                                   const uint8_t* __restrict__ lut, uint8_t* __restrict__ out, int H, int W) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const uint8_t* s = img + (b * HW + p) * 3;
    uint8_t* o = out + (b * HW + p) * 3;
    if (params[b].clahe_clip <= 0.f) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; return; }
    const int y = p / W, x = p - (long)y * W;
    const int Hp = (H % CLAHE_TILES) ? H + CLAHE_TILES - H % CLAHE_TILES : H;
    const int Wp = (W % CLAHE_TILES) ? W + CLAHE_TILES - W % CLAHE_TILES : W;
    const float inv_th = 1.f / (float)(Hp / CLAHE_TILES), inv_tw = 1.f / (float)(Wp / CLAHE_TILES);
    float L, a8, b8;
    rgb8_to_lab8(s[0], s[1], s[2], L, a8, b8);
    const float tyf = y * inv_th - 0.5f, txf = x * inv_tw - 0.5f;
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    const float ya = tyf - ty1, xa = txf - tx1;
    int ty2 = ty1 + 1, tx2 = tx1 + 1;
    ty1 = ty1 < 0 ? 0 : ty1; tx1 = tx1 < 0 ? 0 : tx1;
    ty2 = ty2 > CLAHE_TILES - 1 ? CLAHE_TILES - 1 : ty2; tx2 = tx2 > CLAHE_TILES - 1 ? CLAHE_TILES - 1 : tx2;
    const uint8_t* lb = lut + (long)b * CLAHE_TILES * CLAHE_TILES * 256 + (int)L;
    const float v11 = lb[(ty1 * CLAHE_TILES + tx1) * 256], v12 = lb[(ty1 * CLAHE_TILES + tx2) * 256];
    const float v21 = lb[(ty2 * CLAHE_TILES + tx1) * 256], v22 = lb[(ty2 * CLAHE_TILES + tx2) * 256];
    const float res = (v11 * (1.f - xa) + v12 * xa) * (1.f - ya) + (v21 * (1.f - xa) + v22 * xa) * ya;
    lab8_to_rgb8((float)sat_u8(res), a8, b8, o[0], o[1], o[2]);
}

// stage 3: 3x3 box blur, reflect-101 borders, rounded
__global__ void blur3_kernel(const uint8_t* __restrict__ img, const AppParams* __restrict__ params, uint8_t* __restrict__ out,
                             int H, int W) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const uint8_t* base = img + b * HW * 3;
    uint8_t* o = out + (b * HW + p) * 3;
    if (params[b].blur == 0.f) { o[0] = base[p * 3]; o[1] = base[p * 3 + 1]; o[2] = base[p * 3 + 2]; return; }
    const int y = p / W, x = p - (long)y * W;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const uint8_t* s = base + ((long)reflect101(y + dy, H) * W + reflect101(x + dx, W)) * 3;
            acc[0] += s[0]; acc[1] += s[1]; acc[2] += s[2];
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = sat_u8((float)acc[k] * (1.f / 9.f));
}

extern "C" size_t wesup_appearance_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return align_up((size_t)B * H * W * 3, 256) + (size_t)B * CLAHE_TILES * CLAHE_TILES * 256;
}

extern "C" int wesup_appearance(const uint8_t* img_hwc, const float* params, uint8_t* out_hwc, int B, int H, int W,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!img_hwc || !params || !out_hwc || !ws || B <= 0 || H < CLAHE_TILES || W < CLAHE_TILES || B > 65535) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_appearance_workspace_bytes(B, H, W)) return WESUP_ERR_WORKSPACE;
    const long HW = (long)H * W;
    hipStream_t st = (hipStream_t)stream;
    const AppParams* pr = reinterpret_cast<const AppParams*>(params);
    uint8_t* tmp = (uint8_t*)ws;
    uint8_t* lut = tmp + align_up((size_t)B * HW * 3, 256);
    const dim3 grid((unsigned)((HW + 255) / 256), B);
    WESUP_LAUNCH(app_color_kernel, grid, dim3(256), 0, st, img_hwc, pr, out_hwc, HW);              // src -> out
    WESUP_LAUNCH(clahe_lut_kernel, dim3(CLAHE_TILES * CLAHE_TILES, B), dim3(256), 0, st, out_hwc, pr, lut, H, W);
    WESUP_LAUNCH(clahe_apply_kernel, grid, dim3(256), 0, st, out_hwc, pr, lut, tmp, H, W);         // out -> tmp
    WESUP_LAUNCH(blur3_kernel, grid, dim3(256), 0, st, tmp, pr, out_hwc, H, W);                    // tmp -> out
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
