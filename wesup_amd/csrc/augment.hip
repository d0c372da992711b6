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
// CLAHE, Blur and ElasticTransform of the reference's pipelines are not implemented (documented in DESIGN.md).
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

__global__ void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                               const AugParams* __restrict__ params, float* __restrict__ out_img,
                               uint8_t* __restrict__ out_mask, int H, int W, int C) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - (long)y * W;
    const AugParams a = params[b];
    const float sx = a.a00 * x + a.a01 * y + a.a02, sy = a.a10 * x + a.a11 * y + a.a12;
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

extern "C" int wesup_augment(const uint8_t* img_hwc, const uint8_t* mask_hw, const float* params, float* out_img_nchw,
                             uint8_t* out_mask_chw, int B, int H, int W, int C, void* stream) {
    if (!img_hwc || !params || !out_img_nchw || B <= 0 || H <= 0 || W <= 0 || C <= 0 || B > 65535) return WESUP_ERR_INVALID;
    if (mask_hw && !out_mask_chw) return WESUP_ERR_INVALID;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, img_hwc,
                       mask_hw, reinterpret_cast<const AugParams*>(params), out_img_nchw, out_mask_chw, H, W, C);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
