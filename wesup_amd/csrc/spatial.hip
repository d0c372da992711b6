// Memory-bound NHWC helpers: input / weight packing, 2x2 max-pool fwd/bwd, bilinear (align_corners)
// upsample fwd and its gather-form backward (optionally fused with the superpixel-pooling backward).
// All of them move 16 B per lane along the channel axis (coalesced), no atomics, deterministic.
#include "common.hpp"

// ------------------------------------------------------------------ input packing
__global__ void pack_input_kernel(const float* __restrict__ img, float* __restrict__ out, int B, long HW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW, p = idx - b * HW;
    const float* s = img + b * 3 * HW + p;
    st4(out + idx * 4, make_float4(s[0], s[HW], s[2 * HW], 0.f));
}
extern "C" int wesup_pack_input(const float* img, float* out, int B, int H, int W, void* stream) {
    if (!img || !out || B <= 0 || H <= 0 || W <= 0) return WESUP_ERR_INVALID;
    const long tot = (long)B * H * W;
    WESUP_LAUNCH(pack_input_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img,
                       out, B, (long)H * W);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ conv weight packing
// w_fwd[co][k], k = ((ci/32)*9 + t)*32 + ci%32 for Ci >= 32 (the NT kernel's K order: chunk, tap, channel);
// image layer (Cip = 4): k = t*4 + ci  (lanes along ci)
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ wf, int Co, int Ci, int Cip, int Kf) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Co * Cip) return;
    const int co = idx / Cip, ci = idx - (long)co * Cip;
    const bool chunked = Cip >= 32;
    float* d = wf + (long)co * Kf + (chunked ? (ci >> 5) * 288 + (ci & 31) : ci);
    const int ts = chunked ? 32 : Cip;               // K distance between taps
    if (ci < Ci) {
        const float* s = w + ((long)co * Ci + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) d[t * ts] = s[t];
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) d[t * ts] = 0.f;
    }
    // zero the K padding (only exists when 9*Cip is not a multiple of 32, i.e. the image layer)
    if (ci == 0)
        for (int k = 9 * Cip; k < Kf; ++k) wf[(long)co * Kf + k] = 0.f;
}
// w_dgrad[ci][k], k = ((co/32)*9 + (8-t))*32 + co%32  (same K order with the roles of the channels swapped; lanes along co)
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wd, int Co, int Ci) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Co * Ci) return;
    const int ci = idx / Co, co = idx - (long)ci * Co;
    const float* s = w + ((long)co * Ci + ci) * 9;
    float* d = wd + (long)ci * 9 * Co + (co >> 5) * 288 + (co & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t) d[(8 - t) * 32] = s[t];
}
extern "C" int wesup_pack_conv3x3_weight(const float* w, float* w_fwd, float* w_dgrad, int Co, int Ci, void* stream) {
    if (!w || (!w_fwd && !w_dgrad) || Co <= 0 || Ci <= 0) return WESUP_ERR_INVALID;     // either panel may be skipped
    const int Cip = Ci < 4 ? 4 : Ci;
    const int Kf = wesup_conv3x3_kpad(Ci);
    hipStream_t st = (hipStream_t)stream;
    long tot = (long)Co * Cip;
    if (w_fwd)
        WESUP_LAUNCH(pack_w_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, w_fwd, Co, Ci, Cip, Kf);
    if (w_dgrad) {
        tot = (long)Co * Ci;
        WESUP_LAUNCH(pack_w_dgrad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, w_dgrad, Co, Ci);
    }
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ transpose (weights for dgrad GEMMs)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty 0..7
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[(long)c * rows + r] = t[tx][k];
    }
}
extern "C" int wesup_transpose(const float* in, float* out, int rows, int cols, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0,
                       (hipStream_t)stream, in, out, rows, cols);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// Several transposes in one launch (the step's 13 side-conv and 3 fc weight panels for the input-gradient GEMMs, the
// interpolation-pooling matrices of a batch): a block finds its item by a scan of the tile ranges.
#define TRANSPOSE_MAX 40
struct TransposeBatch {
    const float* in[TRANSPOSE_MAX];
    float* out[TRANSPOSE_MAX];
    int rows[TRANSPOSE_MAX], cols[TRANSPOSE_MAX];
    int first[TRANSPOSE_MAX + 1];
    int n;
};
__global__ void transpose_batched_kernel(const TransposeBatch p) {
    __shared__ float t[32][33];
    int j = 0;
    while (j + 1 < p.n && (int)blockIdx.x >= p.first[j + 1]) ++j;
    const int rows = p.rows[j], cols = p.cols[j];
    const int tcols = (cols + 31) / 32, tile = blockIdx.x - p.first[j];
    const int c0 = (tile % tcols) * 32, r0 = (tile / tcols) * 32;
    const float* __restrict__ in = p.in[j];
    float* __restrict__ out = p.out[j];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[(long)c * rows + r] = t[tx][k];
    }
}
extern "C" int wesup_transpose_batched(const WesupTransposeItem* items /* host */, int n, void* stream) {
    if (!items || n <= 0 || n > TRANSPOSE_MAX) return WESUP_ERR_INVALID;
    TransposeBatch p = {};
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].in || !items[i].out || items[i].rows <= 0 || items[i].cols <= 0) return WESUP_ERR_INVALID;
        p.in[i] = items[i].in; p.out[i] = items[i].out; p.rows[i] = items[i].rows; p.cols[i] = items[i].cols;
        p.first[i] = (int)blocks;
        blocks += (long)ceil_div(items[i].rows, 32) * ceil_div(items[i].cols, 32);
        if (blocks >= (1l << 30)) return WESUP_ERR_INVALID;
    }
    p.first[n] = (int)blocks;
    p.n = n;
    WESUP_LAUNCH(transpose_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ 2x2 max pool on pre-ReLU values
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__global__ void maxpool_fwd_kernel(const float* __restrict__ y, float* __restrict__ yp, int B, int H, int W, int C4,
                                   int relu_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = H >> 1, Wo = W >> 1;
    if (idx >= (long)B * Ho * Wo * C4) return;
    const int c = idx % C4;
    long t = idx / C4;
    const int xo = t % Wo;
    t /= Wo;
    const int yo = t % Ho;
    const int b = t / Ho;
    const float* s = y + (((long)b * H + 2 * yo) * W + 2 * xo) * C4 * 4 + 4 * c;
    const long rs = (long)W * C4 * 4;
    float4 v = max4(max4(ld4(s), ld4(s + C4 * 4)), max4(ld4(s + rs), ld4(s + rs + C4 * 4)));
    if (relu_out) v = relu4(v);
    st4(yp + idx * 4, v);
}
extern "C" int wesup_maxpool2_fwd(const float* y, float* yp, int B, int H, int W, int C, int relu_out, void* stream) {
    if (!y || !yp || B <= 0 || H < 2 || W < 2 || (C % 4)) return WESUP_ERR_INVALID;
    const long tot = (long)B * (H / 2) * (W / 2) * (C / 4);
    WESUP_LAUNCH(maxpool_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, yp,
                       B, H, W, C / 4, relu_out);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// backward through ReLU -> MaxPool(2,2): the gradient goes to the FIRST maximum of the window (torch scan
// order: (0,0),(0,1),(1,0),(1,1)) and only if that maximum is positive (ReLU).  Pixels of an odd trailing
// row/column belong to no window and get 0.
__global__ void maxpool_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dyp, float* __restrict__ dy,
                                   int B, int H, int W, int C4, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * W * C4) return;
    const int c = idx % C4;
    long t = idx / C4;
    const int x = t % W;
    t /= W;
    const int yy = t % H;
    const int b = t / H;
    const int Ho = H >> 1, Wo = W >> 1;
    const int yo = yy >> 1, xo = x >> 1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yo < Ho && xo < Wo) {
        const float* s = y + (((long)b * H + 2 * yo) * W + 2 * xo) * C4 * 4 + 4 * c;
        const long rs = (long)W * C4 * 4;
        const float4 v0 = ld4(s), v1 = ld4(s + C4 * 4), v2 = ld4(s + rs), v3 = ld4(s + rs + C4 * 4);
        const float4 d = ld4(dyp + ((((long)b * Ho + yo) * Wo + xo) * C4 + c) * 4);
        const int me = (yy & 1) * 2 + (x & 1);
        const float a0[4] = {v0.x, v1.x, v2.x, v3.x}, a1[4] = {v0.y, v1.y, v2.y, v3.y};
        const float a2[4] = {v0.z, v1.z, v2.z, v3.z}, a3[4] = {v0.w, v1.w, v2.w, v3.w};
        auto pick = [&](const float* a, float dv) {
            int best = 0;
            float m = a[0];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (a[k] > m) { m = a[k]; best = k; }
            return (best == me && m > 0.f) ? dv : 0.f;
        };
        g = make_float4(pick(a0, d.x), pick(a1, d.y), pick(a2, d.z), pick(a3, d.w));
    }
    float* o = dy + idx * 4;
    if (accumulate) {
        const float4 old = ld4(o);
        g = make_float4(g.x + old.x, g.y + old.y, g.z + old.z, g.w + old.w);
    }
    st4(o, g);
}
extern "C" int wesup_maxpool2_bwd(const float* y, const float* dyp, float* dy, int B, int H, int W, int C,
                                  int accumulate, void* stream) {
    if (!y || !dyp || !dy || B <= 0 || H < 2 || W < 2 || (C % 4)) return WESUP_ERR_INVALID;
    const long tot = (long)B * H * W * (C / 4);
    WESUP_LAUNCH(maxpool_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dyp,
                       dy, B, H, W, C / 4, accumulate);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ bilinear, align_corners=True
// torch: scale = (in-1)/(out-1) (float); src = scale*dst; i0 = min(int(src), in-1); i1 = i0 + (i0 < in-1);
// l1 = clamp(src - i0, 0, 1); l0 = 1 - l1.
struct Lerp {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in) {
    Lerp r;
    const float src = scale * (float)dst;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
    r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
    r.l0 = 1.f - r.l1;
    return r;
}
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

__global__ void upsample_fwd_kernel(const float* __restrict__ s, float* __restrict__ fm, int B, int h, int w, int H,
                                    int W, int C4, int ldf, int coff, float sh, float sw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * W * C4) return;
    const int c = idx % C4;
    long t = idx / C4;
    const int X = t % W;
    t /= W;
    const int Y = t % H;
    const int b = t / H;
    const Lerp ly = lerp_of(Y, sh, h), lx = lerp_of(X, sw, w);
    const float* base = s + (long)b * h * w * C4 * 4 + 4 * c;
    const float4 v00 = ld4(base + ((long)ly.i0 * w + lx.i0) * C4 * 4);
    const float4 v01 = ld4(base + ((long)ly.i0 * w + lx.i1) * C4 * 4);
    const float4 v10 = ld4(base + ((long)ly.i1 * w + lx.i0) * C4 * 4);
    const float4 v11 = ld4(base + ((long)ly.i1 * w + lx.i1) * C4 * 4);
    float4 o;
    o.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
    o.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
    o.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
    o.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    st4(fm + (((long)b * H + Y) * W + X) * ldf + coff + 4 * c, o);
}
extern "C" int wesup_upsample_fwd(const float* s, float* fm, int B, int h, int w, int H, int W, int C, int ldf,
                                  int coff, void* stream) {
    if (!s || !fm || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (C % 4) || (ldf % 4) || (coff % 4))
        return WESUP_ERR_INVALID;
    const long tot = (long)B * H * W * (C / 4);
    WESUP_LAUNCH(upsample_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, fm,
                       B, h, w, H, W, C / 4, ldf, coff, ac_scale(h, H), ac_scale(w, W));
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// Gather-form backward: one thread per (low-res pixel q, channel quad) sums over the full-res pixels whose
// interpolation touches q, in raster order (deterministic).  The candidate window is a safe superset; the
// weight is recomputed with the exact forward formula, so a candidate that does not touch q contributes 0.
template <bool FUSED>
__global__ void upsample_bwd_kernel(const float* __restrict__ src, const int32_t* __restrict__ new_row,
                                    const int32_t* __restrict__ area, float* __restrict__ ds, int B, int h, int w,
                                    int H, int W, int C4, int ldf, int coff, int Kmax, float sh, float sw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * h * w * C4) return;
    const int c = idx % C4;
    long t = idx / C4;
    const int qx = t % w;
    t /= w;
    const int qy = t % h;
    const int b = t / h;
    int Ylo = 0, Yhi = H - 1, Xlo = 0, Xhi = W - 1;
    if (sh > 0.f) {
        Ylo = max(0, (int)floorf((float)(qy - 1) / sh) - 1);
        Yhi = min(H - 1, (int)ceilf((float)(qy + 1) / sh) + 1);
    }
    if (sw > 0.f) {
        Xlo = max(0, (int)floorf((float)(qx - 1) / sw) - 1);
        Xhi = min(W - 1, (int)ceilf((float)(qx + 1) / sw) + 1);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // FUSED: neighbouring pixels mostly belong to the same superpixel, so the weights of a run of equal rows are summed
    // first and the row of g is fetched once per run (raster order kept: deterministic).  At 120x120 under a 480x480
    // image a cell has 64 candidate pixels and ~10 runs.
    int run_r = -1;
    float run_w = 0.f;
    auto flush = [&]() {
        if (run_r < 0) return;
        const float wgt = run_w * (1.f / (float)area[(long)b * Kmax + run_r]);
        const float4 v = ld4(src + ((long)b * Kmax + run_r) * ldf + coff + 4 * c);
        acc.x += wgt * v.x;
        acc.y += wgt * v.y;
        acc.z += wgt * v.z;
        acc.w += wgt * v.w;
    };
    for (int Y = Ylo; Y <= Yhi; ++Y) {
        const Lerp ly = lerp_of(Y, sh, h);
        const float wy = (ly.i0 == qy ? ly.l0 : 0.f) + (ly.i1 == qy ? ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int X = Xlo; X <= Xhi; ++X) {
            const Lerp lx = lerp_of(X, sw, w);
            const float wx = (lx.i0 == qx ? lx.l0 : 0.f) + (lx.i1 == qx ? lx.l1 : 0.f);
            if (wx == 0.f) continue;
            const long p = (long)Y * W + X;
            const float wgt = wy * wx;
            if (FUSED) {
                const int r = new_row[(long)b * H * W + p];
                if (r != run_r) {
                    flush();
                    run_r = r;
                    run_w = 0.f;
                }
                run_w += wgt;
            } else {
                const float4 v = ld4(src + ((long)b * H * W + p) * ldf + coff + 4 * c);
                acc.x += wgt * v.x;
                acc.y += wgt * v.y;
                acc.z += wgt * v.z;
                acc.w += wgt * v.w;
            }
        }
    }
    if (FUSED) flush();
    st4(ds + idx * 4, acc);
}
// Fused backward on a coarse map, one WAVE per coarse cell.  Phase 1: the lanes share out the candidate window (one
// full-resolution pixel each), compute its bilinear weight on this cell and look up its superpixel row.  The distinct
// rows among them (2-4 under a 120x120 map) are then peeled off one at a time -- first pending lane's row, ballot of
// the lanes that share it, fixed butterfly sum of their weights -- and phase 2 fetches that row of g ONCE with the lanes
// as channel quads.  The thread-per-(cell, quad) form above repeats the window scan in every channel thread and fetches
// a row per run of pixels (~10 per cell).  Order: rows by first occurrence in raster order, fixed reduction tree.
__global__ __launch_bounds__(256) void upsample_bwd_cell_kernel(const float* __restrict__ g, const int32_t* __restrict__ new_row,
                                                                const int32_t* __restrict__ area, float* __restrict__ ds,
                                                                int B, int h, int w, int H, int W, int C4, int ldf, int coff,
                                                                int Kmax, float sh, float sw) {
    const long cellid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cellid >= (long)B * h * w) return;          // wave-uniform
    const int lane = threadIdx.x & 63;
    long t = cellid;
    const int qx = t % w;
    t /= w;
    const int qy = t % h;
    const int b = t / h;
    int Ylo = 0, Yhi = H - 1, Xlo = 0, Xhi = W - 1;
    if (sh > 0.f) {
        Ylo = max(0, (int)floorf((float)(qy - 1) / sh) - 1);
        Yhi = min(H - 1, (int)ceilf((float)(qy + 1) / sh) + 1);
    }
    if (sw > 0.f) {
        Xlo = max(0, (int)floorf((float)(qx - 1) / sw) - 1);
        Xhi = min(W - 1, (int)ceilf((float)(qx + 1) / sw) + 1);
    }
    const int nwx = Xhi - Xlo + 1, ncand = nwx * (Yhi - Ylo + 1);
    const int32_t* rows = new_row + (long)b * H * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < ncand; base += 64) {
        const int i = base + lane;
        float wgt = 0.f;
        int r = -1;
        if (i < ncand) {
            const int dy = i / nwx, Y = Ylo + dy, X = Xlo + i - dy * nwx;
            const Lerp ly = lerp_of(Y, sh, h), lx = lerp_of(X, sw, w);
            const float wy = (ly.i0 == qy ? ly.l0 : 0.f) + (ly.i1 == qy ? ly.l1 : 0.f);
            const float wx = (lx.i0 == qx ? lx.l0 : 0.f) + (lx.i1 == qx ? lx.l1 : 0.f);
            wgt = wy * wx;
            if (wgt != 0.f) r = rows[(long)Y * W + X];
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const int rr = __builtin_amdgcn_readlane(r, first);
            const bool mine = (r == rr);
            float wsum = mine ? wgt : 0.f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) wsum += __shfl_xor(wsum, off);
            todo &= ~__ballot(mine);
            if (lane < C4) {
                const float coef = wsum * (1.f / (float)area[(long)b * Kmax + rr]);
                const float4 v = ld4(g + ((long)b * Kmax + rr) * ldf + coff + 4 * lane);
                acc.x += coef * v.x;
                acc.y += coef * v.y;
                acc.z += coef * v.z;
                acc.w += coef * v.w;
            }
        }
    }
    if (lane < C4) st4(ds + (cellid * C4 + lane) * 4, acc);
}
// Fused backward at native resolution (h == H, w == W): the upsample is the identity, a pixel's gradient is its
// superpixel's row of g over the area -- one gather per (pixel, channel quad), the value the window scan above produces
// for its single candidate (weight 1).
__global__ __launch_bounds__(256) void upsample_bwd_ident_kernel(const float* __restrict__ g, const int32_t* __restrict__ new_row,
                                                                 const int32_t* __restrict__ area, float* __restrict__ ds,
                                                                 long npix, int HW, int C4, int ldf, int coff, int Kmax) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * C4) return;
    const long p = idx / C4;
    const int c = idx - p * C4;
    const int b = p / HW;
    const int r = new_row[p];
    const float coef = 1.f * (1.f / (float)area[(long)b * Kmax + r]);
    const float4 v = ld4(g + ((long)b * Kmax + r) * ldf + coff + 4 * c);
    st4(ds + idx * 4, make_float4(coef * v.x, coef * v.y, coef * v.z, coef * v.w));
}

// The cell kernel for up to three layers that share a coarse resolution (commuted side branch: all their g's exist when
// backward starts): the window scan and the peeling of distinct rows -- most of the work of a cell -- are done once, the
// lanes then walk the layers' channel quads as one concatenated range, 64 quads per pass.
struct UpGroup {
    const float* g[3];
    float* ds[3];
    int c4[3];               // channel quads per layer (0: unused slot); rows of g[i] are c4[i]*4 floats apart
    int pad_;                // (explicit padding, zero: launch.hpp)
};
template <int NPASS>
__global__ __launch_bounds__(256) void upsample_bwd_cell_group_kernel(UpGroup G, const int32_t* __restrict__ new_row,
                                                                      const int32_t* __restrict__ area, int B, int h, int w,
                                                                      int H, int W, int Kmax, float sh, float sw) {
    const long cellid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cellid >= (long)B * h * w) return;          // wave-uniform
    const int lane = threadIdx.x & 63;
    long t = cellid;
    const int qx = t % w;
    t /= w;
    const int qy = t % h;
    const int b = t / h;
    int Ylo = 0, Yhi = H - 1, Xlo = 0, Xhi = W - 1;
    if (sh > 0.f) {
        Ylo = max(0, (int)floorf((float)(qy - 1) / sh) - 1);
        Yhi = min(H - 1, (int)ceilf((float)(qy + 1) / sh) + 1);
    }
    if (sw > 0.f) {
        Xlo = max(0, (int)floorf((float)(qx - 1) / sw) - 1);
        Xhi = min(W - 1, (int)ceilf((float)(qx + 1) / sw) + 1);
    }
    const int nwx = Xhi - Xlo + 1, ncand = nwx * (Yhi - Ylo + 1);
    const int32_t* rows = new_row + (long)b * H * W;
    // this lane's quad of pass k: layer, row stride and offset inside the row
    const float* src[NPASS];
    float* dst[NPASS];
    int ld[NPASS];
    float4 acc[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        int q = lane + 64 * k, i = 0;
        while (i < 3 && q >= G.c4[i]) { q -= G.c4[i]; ++i; }
        const bool on = i < 3;
        const int c4 = on ? G.c4[i] : 0;
        ld[k] = c4 * 4;
        src[k] = on ? G.g[i] + (long)b * Kmax * c4 * 4 + 4 * q : nullptr;
        dst[k] = on ? G.ds[i] + (cellid * c4 + q) * 4 : nullptr;
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int base = 0; base < ncand; base += 64) {
        const int i = base + lane;
        float wgt = 0.f;
        int r = -1;
        if (i < ncand) {
            const int dy = i / nwx, Y = Ylo + dy, X = Xlo + i - dy * nwx;
            const Lerp ly = lerp_of(Y, sh, h), lx = lerp_of(X, sw, w);
            const float wy = (ly.i0 == qy ? ly.l0 : 0.f) + (ly.i1 == qy ? ly.l1 : 0.f);
            const float wx = (lx.i0 == qx ? lx.l0 : 0.f) + (lx.i1 == qx ? lx.l1 : 0.f);
            wgt = wy * wx;
            if (wgt != 0.f) r = rows[(long)Y * W + X];
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const int rr = __builtin_amdgcn_readlane(r, first);
            const bool mine = (r == rr);
            float wsum = mine ? wgt : 0.f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) wsum += __shfl_xor(wsum, off);
            todo &= ~__ballot(mine);
            const float coef = wsum * (1.f / (float)area[(long)b * Kmax + rr]);
#pragma unroll
            for (int k = 0; k < NPASS; ++k)
                if (src[k]) {
                    const float4 v = ld4(src[k] + (long)rr * ld[k]);
                    acc[k].x += coef * v.x;
                    acc[k].y += coef * v.y;
                    acc[k].z += coef * v.z;
                    acc[k].w += coef * v.w;
                }
        }
    }
#pragma unroll
    for (int k = 0; k < NPASS; ++k)
        if (dst[k]) st4(dst[k], acc[k]);
}
/* Fused upsample + scatter-mean backward of up to three layers of one coarse resolution in one launch:
   ds_i (B,h,w,C_i) from g_i (B,Kmax,C_i), i < n.  Same numbers as n calls of wesup_upsample_bwd (same order of additions). */
extern "C" int wesup_upsample_bwd_group(const float* g0, const float* g1, const float* g2, float* ds0, float* ds1, float* ds2,
                                        int C0, int C1, int C2, int n, const int32_t* new_row, const int32_t* area_new, int B,
                                        int h, int w, int H, int W, int Kmax, void* stream) {
    if (n < 1 || n > 3 || !new_row || !area_new || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || Kmax <= 0 ||
        (h == H && w == W))
        return WESUP_ERR_INVALID;
    const float* gs[3] = {g0, g1, g2};
    float* dss[3] = {ds0, ds1, ds2};
    const int cs[3] = {C0, C1, C2};
    UpGroup G = {};
    int quads = 0;
    for (int i = 0; i < 3; ++i) {
        const bool on = i < n;
        if (on && (!gs[i] || !dss[i] || cs[i] <= 0 || (cs[i] % 4))) return WESUP_ERR_INVALID;
        G.g[i] = on ? gs[i] : nullptr;
        G.ds[i] = on ? dss[i] : nullptr;
        G.c4[i] = on ? cs[i] / 4 : 0;
        quads += G.c4[i];
    }
    const int npass = (quads + 63) / 64;
    if (npass > 3) return WESUP_ERR_INVALID;
    const dim3 grid((unsigned)(((long)B * h * w + 3) / 4));
    const float sh = ac_scale(h, H), sw = ac_scale(w, W);
    hipStream_t st = (hipStream_t)stream;
    if (npass == 1)
        WESUP_LAUNCH(upsample_bwd_cell_group_kernel<1>, grid, dim3(256), 0, st, G, new_row, area_new, B, h, w, H, W, Kmax, sh, sw);
    else if (npass == 2)
        WESUP_LAUNCH(upsample_bwd_cell_group_kernel<2>, grid, dim3(256), 0, st, G, new_row, area_new, B, h, w, H, W, Kmax, sh, sw);
    else
        WESUP_LAUNCH(upsample_bwd_cell_group_kernel<3>, grid, dim3(256), 0, st, G, new_row, area_new, B, h, w, H, W, Kmax, sh, sw);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

extern "C" int wesup_upsample_bwd(const float* dfm_or_g, const int32_t* new_row, const int32_t* area_new, float* ds,
                                  int B, int h, int w, int H, int W, int C, int ldf, int coff, int Kmax, void* stream) {
    if (!dfm_or_g || !ds || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (C % 4) || (ldf % 4) || (coff % 4))
        return WESUP_ERR_INVALID;
    if (new_row && (!area_new || Kmax <= 0)) return WESUP_ERR_INVALID;
    const long tot = (long)B * h * w * (C / 4);
    const dim3 grid((unsigned)((tot + 255) / 256));
    if (new_row && h == H && w == W)
        WESUP_LAUNCH(upsample_bwd_ident_kernel, grid, dim3(256), 0, (hipStream_t)stream, dfm_or_g, new_row, area_new, ds,
                           (long)B * H * W, H * W, C / 4, ldf, coff, Kmax);
    else if (new_row && C / 4 > 64 && C <= 768 && ldf == C && coff == 0)      // wide dense rows: the cell kernel in passes of 64 quads
        return wesup_upsample_bwd_group(dfm_or_g, nullptr, nullptr, ds, nullptr, nullptr, C, 0, 0, 1, new_row, area_new, B, h, w, H,
                                        W, Kmax, stream);
    else if (new_row && C / 4 <= 64)
        WESUP_LAUNCH(upsample_bwd_cell_kernel, dim3((unsigned)(((long)B * h * w + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, dfm_or_g, new_row, area_new, ds, B, h, w, H, W, C / 4, ldf, coff, Kmax,
                           ac_scale(h, H), ac_scale(w, W));
    else if (new_row)
        WESUP_LAUNCH(upsample_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dfm_or_g, new_row, area_new,
                           ds, B, h, w, H, W, C / 4, ldf, coff, Kmax, ac_scale(h, H), ac_scale(w, W));
    else
        WESUP_LAUNCH(upsample_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dfm_or_g, new_row, area_new,
                           ds, B, h, w, H, W, C / 4, ldf, coff, Kmax, ac_scale(h, H), ac_scale(w, W));
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// x [rows][C] *= 1 / area[row] (0 for a row of area 0: padded superpixel rows, which no pixel refers to).  The side-branch
// gradient rows a gather epilogue reads (wesup_conv3x3_dgrad_winograd_gather with area_new == NULL) are scaled here once per
// row instead of once per pixel and channel quad there -- with the same coefficient, 1.f / (float)area.
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ x, const int32_t* __restrict__ area, long total, int Q) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int a = area[i / Q];
    const float coef = a > 0 ? 1.f / (float)a : 0.f;
    float4 v = ld4(x + 4 * i);
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    st4(x + 4 * i, v);
}
extern "C" int wesup_scale_rows_by_area(float* x, const int32_t* area, long rows, int C, void* stream) {
    if (!x || !area || rows <= 0 || C <= 0 || (C % 4) || (((uintptr_t)x) & 15)) return WESUP_ERR_INVALID;
    const long total = rows * (C / 4);
    WESUP_LAUNCH(scale_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, area, total, C / 4);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
