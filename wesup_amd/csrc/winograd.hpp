// Shared by gemm.hip and winograd.hip: tile counting and the shape limits of the Winograd-domain entries.
// m = output tile edge of F(m x m, 3x3): 2 (4x4 patches, 16 positions) or 4 (6x6 patches, 36 positions).
#pragma once
#include "common.hpp"

static inline bool wino_m_ok(int m) { return m == 2 || m == 4; }
static inline int wino_positions(int m) { return (m + 2) * (m + 2); }
static inline long wino_tiles(int B, int H, int W, int m) { return (long)B * ((H + m - 1) / m) * ((W + m - 1) / m); }
static inline bool wino_shape_ok(int B, int H, int W, int Ci, int Cout, int m) {
    if (!wino_m_ok(m) || B <= 0 || H <= 0 || W <= 0 || Ci < 32 || Cout < 32 || (Ci % 4) || (Cout % 4)) return false;
    const long T = wino_tiles(B, H, W, m);
    const long cmax = Ci > Cout ? Ci : Cout;
    // thread index and the FastDiv range (n * d < 2^40, quotient < 2^24)
    return T < (1l << 24) && T * (cmax / 4) < (1l << 31) && T * (cmax / 4) * (cmax / 4) < (1l << 40);
}

#ifdef __HIPCC__
// The max-pool backward that follows an input gradient at pooled resolution (autograd of ReLU -> MaxPool2d(2,2),
// models/wesup.py:199): the gradient v of pooled pixel (h, w) goes to the FIRST maximum of its 2x2 window of the pre-pool
// activations (torch's scan order) if that maximum is positive, and is ADDED to what dst holds there (the side-branch
// gradient).  Only the chosen position is touched: the separate kernel read and re-wrote all four.
struct WinoUnpool {
    const float* src;    // pre-pool activations (B, Hu, Wu, C), or NULL: no unpooling
    float* dst;          // gradient w.r.t. them, same shape, accumulated into
    int Hu, Wu;
};
// The decisions of one window and channel quad: position 0..3 of the first positive maximum per channel, -1 if none.
struct WinoPicks { int kx, ky, kz, kw; };
__device__ __forceinline__ int wino_pick(float p0, float p1, float p2, float p3) {
    int best = 0;
    float m = p0;
    if (p1 > m) { m = p1; best = 1; }
    if (p2 > m) { m = p2; best = 2; }
    if (p3 > m) { m = p3; best = 3; }
    return m > 0.f ? best : -1;
}
__device__ __forceinline__ WinoPicks wino_picks(float4 a0, float4 a1, float4 a2, float4 a3) {
    return WinoPicks{wino_pick(a0.x, a1.x, a2.x, a3.x), wino_pick(a0.y, a1.y, a2.y, a3.y), wino_pick(a0.z, a1.z, a2.z, a3.z),
                     wino_pick(a0.w, a1.w, a2.w, a3.w)};
}
// ... as a 12-bit code (3 bits per channel: position + 1, 0 = none): what the forward's pooling epilogue leaves for the
// backward instead of the four pre-pool values (pool_code [B][H/2][W/2][C/4] uint16)
__device__ __forceinline__ unsigned short wino_picks_code(const WinoPicks& k) {
    return (unsigned short)((k.kx + 1) | ((k.ky + 1) << 3) | ((k.kz + 1) << 6) | ((k.kw + 1) << 9));
}
__device__ __forceinline__ WinoPicks wino_code_picks(unsigned c) {
    return WinoPicks{(int)(c & 7) - 1, (int)((c >> 3) & 7) - 1, (int)((c >> 6) & 7) - 1, (int)((c >> 9) & 7) - 1};
}
__device__ __forceinline__ WinoPicks wino_unpool_picks(const WinoUnpool& u, const unsigned short* code, int b, int h, int w, int C,
                                                       int c0) {
    if (code) return wino_code_picks(code[(((long)b * (u.Hu >> 1) + h) * (u.Wu >> 1) + w) * (C >> 2) + (c0 >> 2)]);
    const long rs = (long)u.Wu * C;
    const long o00 = (((long)b * u.Hu + 2 * h) * u.Wu + 2 * w) * C + c0;
    return wino_picks(ld4(u.src + o00), ld4(u.src + o00 + C), ld4(u.src + o00 + rs), ld4(u.src + o00 + rs + C));
}
__device__ __forceinline__ void wino_unpool_add(const WinoUnpool& u, const unsigned short* code, int b, int h, int w, int C, int c0,
                                                float4 v) {
    const long rs = (long)u.Wu * C;
    const long o00 = (((long)b * u.Hu + 2 * h) * u.Wu + 2 * w) * C + c0;
    const WinoPicks pk = wino_unpool_picks(u, code, b, h, w, C, c0);
    const int kx = pk.kx, ky = pk.ky, kz = pk.kz, kw = pk.kw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (kx != k && ky != k && kz != k && kw != k) continue;
        float* d = u.dst + o00 + (k >> 1) * rs + (k & 1) * C;
        float4 old = ld4(d);
        old.x += kx == k ? v.x : 0.f; old.y += ky == k ? v.y : 0.f; old.z += kz == k ? v.z : 0.f; old.w += kw == k ? v.w : 0.f;
        st4(d, old);
    }
}

// The side-branch gradient of a native-resolution layer as a gather (commuted side branch, engine.py: the side conv sits
// behind the superpixel mean, so that gradient is constant over a superpixel): side [B][Kmax][C] rows per superpixel,
// row [B][H*W] the pixel's superpixel row, area [B][Kmax].  value(b, pix) = side[b][row[b][pix]][.] / area[b][row[b][pix]];
// area == NULL: the rows of side are divided already (wesup_scale_rows_by_area) -- one dependent load and a reciprocal less per pixel
// -- what wesup_upsample_bwd writes for such a layer, read here by the epilogue that would otherwise accumulate into it.
struct WinoGather {
    const float* src;    // NULL: no gather
    const int32_t* row;
    const int32_t* area;
    int Kmax;
    int pad_;            // (explicit padding, zero: by-value kernel parameters carry no indeterminate bytes, launch.hpp)
    long HW;             // pixels per image at the resolution of the epilogue's destination
};
__device__ __forceinline__ float4 wino_gather(const WinoGather& g, int b, long pix, int C, int c0) {
    const int r = g.row[(long)b * g.HW + pix];
    const float coef = g.area ? 1.f / (float)g.area[(long)b * g.Kmax + r] : 1.f;
    const float4 t = ld4(g.src + ((long)b * g.Kmax + r) * C + c0);
    return make_float4(coef * t.x, coef * t.y, coef * t.z, coef * t.w);
}
// wino_unpool_add with the destination's old content replaced by the gather: every position of the window is WRITTEN
// (gathered side gradient, plus v at the first positive maximum); Hu, Wu even (every pre-pool pixel sits in a window).
__device__ __forceinline__ void wino_unpool_gather(const WinoUnpool& u, const unsigned short* code, const WinoGather& g, int b,
                                                   int h, int w, int C, int c0, float4 v, int nt = 0) {
    const long rs = (long)u.Wu * C;
    const long o00 = (((long)b * u.Hu + 2 * h) * u.Wu + 2 * w) * C + c0;
    const WinoPicks pk = wino_unpool_picks(u, code, b, h, w, C, c0);
    const int kx = pk.kx, ky = pk.ky, kz = pk.kz, kw = pk.kw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4 o = wino_gather(g, b, (long)(2 * h + (k >> 1)) * u.Wu + 2 * w + (k & 1), C, c0);
        o.x += kx == k ? v.x : 0.f; o.y += ky == k ? v.y : 0.f; o.z += kz == k ? v.z : 0.f; o.w += kw == k ? v.w : 0.f;
        st4s(u.dst + o00 + (k >> 1) * rs + (k & 1) * C, o, nt);
    }
}

#endif

// internal entry points of winograd.hip used by the weight-gradient pass in gemm.hip (not part of the C ABI): the F(4x4)
// bias gradient travels as per-block rows from the outgrad transform to the filter-gradient reduce, with no launch of its own
long wino4_bias_rows(int B, int H, int W, int C);          // 0: C / 4 does not divide 256 (use wesup_colsum on dy instead)
int wino_outgrad_launch(const float* dy, float* dM, float* bias_part, int B, int H, int W, int C, int m, void* stream);
int wino_filter_grad_launch(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs, float* db, int Cout,
                            int Cin, int m, const float* bias_part, int bias_rows, void* stream);
// wino_fused.hip: the batched products and the output transform in one kernel (K = 64 / 128, N % 64 == 0, m = 4)
int wino_fused_supported(int K, int N, int m);      // 0 no, 1 forward epilogue only, 2 every epilogue
int wino_fused_route(int K, int N, int m, long tiles);      // the same, 0 when the grid of the one-kernel route would be too small
