// Winograd F(2x2, 3x3) transforms for gfx950: everything of the Winograd-domain convolutions that is NOT a GEMM --
// the memory-bound passes between the activations / filters and the [16 positions][tiles][C] operands of the batched
// GEMMs in gemm.hip (wesup_gemm_nt_batched, the TN launch inside wesup_conv3x3_wgrad_winograd).  DESIGN.md 3.1.1.
//
//   forward / input gradient:   V = B^T d B  ->  M_p = V_p . U_p^T  (gemm.hip)  ->  Y = A^T M A (+ epilogue)
//   weight gradient:            dM = A dY A^T, V as above  ->  dU_p = dM_p^T . V_p  (gemm.hip)  ->  dg = G^T dU G
//   filters:                    U = G g G^T once per step (forward), and of the rotated filter (input gradient)
//
// B^T, G, A^T are the F(2x2,3x3) matrices (entries 0, +-1, +-1/2); a transformed tensor is position-major so that each
// of the 16 GEMM operands is a plain row-major matrix.  Every kernel: one thread per (tile, 4 channels), channels
// fastest across lanes, 16-byte loads and stores of contiguous channel rows; tiles that hang over an odd border read
// zeros and skip the stores.
#include "winograd.hpp"

struct WinoGeom {
    int H, W, C, Th, Tw;
    long ps;             // elements between two position planes of the transformed tensor (>= T * C: a sub-batch may
                         // write its rows into the planes of the whole batch)
    long T;              // tiles = B * Th * Tw
    FastDiv dTw, dTh, dQ;
};

// thread = (tile, 4 channels): 16 float4 loads, B^T d B, 16 float4 stores (c fastest across lanes: coalesced both ways)
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                   const WinoGeom g, int relu) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int Q = g.C >> 2;
    if (idx >= g.T * Q) return;
    const int t = fast_div((int)idx, g.dQ);
    const int cq = (int)idx - t * Q;
    const int bi = fast_div(t, g.dTw);
    const int j = t - bi * g.Tw;
    const int b = fast_div(bi, g.dTh);
    const int i = bi - b * g.Th;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = 2 * i - 1 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int w = 2 * j - 1 + c;
            const bool in = (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
            float4 v = in ? ld4(x + (((long)b * g.H + h) * g.W + w) * g.C + 4 * cq) : z;
            d[r][c] = relu ? relu4(v) : v;
        }
    }
#define F4(op, a, b) make_float4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w)
    float4 m[4][4];      // rows: B^T d
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m[0][c] = F4(-, d[0][c], d[2][c]);
        m[1][c] = F4(+, d[1][c], d[2][c]);
        m[2][c] = F4(-, d[2][c], d[1][c]);
        m[3][c] = F4(-, d[1][c], d[3][c]);
    }
    float* out = V + (long)t * g.C + 4 * cq;
    const long ps = g.ps;
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // columns: (.) B
        st4(out + (4 * r + 0) * ps, F4(-, m[r][0], m[r][2]));
        st4(out + (4 * r + 1) * ps, F4(+, m[r][1], m[r][2]));
        st4(out + (4 * r + 2) * ps, F4(-, m[r][2], m[r][1]));
        st4(out + (4 * r + 3) * ps, F4(-, m[r][1], m[r][3]));
    }
}

// thread = (tile, 4 channels): the tile's 2x2 gradients -> A dY A^T with A = [[1,0],[1,1],[1,-1],[0,-1]]
__global__ __launch_bounds__(256) void wino_outgrad_transform_kernel(const float* __restrict__ dy, float* __restrict__ dM,
                                                                     const WinoGeom g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int Q = g.C >> 2;
    if (idx >= g.T * Q) return;
    const int t = fast_div((int)idx, g.dQ);
    const int cq = (int)idx - t * Q;
    const int bi = fast_div(t, g.dTw);
    const int j = t - bi * g.Tw;
    const int b = fast_div(bi, g.dTh);
    const int i = bi - b * g.Th;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 y[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int h = 2 * i + r, w = 2 * j + c;
            y[r][c] = (h < g.H && w < g.W) ? ld4(dy + (((long)b * g.H + h) * g.W + w) * g.C + 4 * cq) : z;
        }
    float4 m[4][2];      // rows: A dY
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        m[0][c] = y[0][c];
        m[1][c] = F4(+, y[0][c], y[1][c]);
        m[2][c] = F4(-, y[0][c], y[1][c]);
        m[3][c] = F4(-, z, y[1][c]);
    }
    float* out = dM + (long)t * g.C + 4 * cq;
    const long ps = g.ps;
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // columns: (.) A^T
        st4(out + (4 * r + 0) * ps, m[r][0]);
        st4(out + (4 * r + 1) * ps, F4(+, m[r][0], m[r][1]));
        st4(out + (4 * r + 2) * ps, F4(-, m[r][0], m[r][1]));
        st4(out + (4 * r + 3) * ps, F4(-, z, m[r][1]));
    }
}
#undef F4

// dw[co][ci][3][3] = G^T (sum_s slab[p][s][co][ci]) G ;  db[co] = sum_s (column sums of dM_5)
// block = 16 (co, ci) pairs x 16 positions: a thread adds the S split-K slabs of ONE position (a thread per pair walked
// 16 x S dependent loads -- 512 at conv2_2 -- with only Co*Ci/256 blocks on the chip: 113 us per launch on average,
// 1.1 ms per step); the 16 sums of a pair meet in LDS and one thread per pair applies G^T (.) G.  Fixed order.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, long stride, long batch_slab,
                                                                float* __restrict__ dw, int Co, int Ci, int S,
                                                                float* __restrict__ db, int pair_blocks) {
    __shared__ float us[16][17];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= pair_blocks) {                // bias gradient: the blocks behind the pair blocks
        const long m = (long)(blockIdx.x - pair_blocks) * 256 + tid;
        if (db && m < Co) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += slab[5 * batch_slab + (long)k * stride + (long)Co * Ci + m];
            db[m] = s;
        }
        return;
    }
    const int i = tid & 15, p = tid >> 4;
    const long idx = (long)blockIdx.x * 16 + i;
    const bool ok = idx < (long)Co * Ci;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        const float* src = slab + p * batch_slab + idx;
        int k = 0;
        for (; k + 1 < S; k += 2) { s0 += src[(long)k * stride]; s1 += src[(long)(k + 1) * stride]; }
        if (k < S) s0 += src[(long)k * stride];
    }
    us[p][i] = s0 + s1;
    __syncthreads();
    if (tid >= 16 || !ok) return;
    float r[3][4];       // G^T u
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float u0 = us[c][i], u1 = us[4 + c][i], u2 = us[8 + c][i], u3 = us[12 + c][i];
        const float hs = 0.5f * (u1 + u2), hd = 0.5f * (u1 - u2);
        r[0][c] = u0 + hs;
        r[1][c] = hd;
        r[2][c] = hs + u3;
    }
    float* d = dw + idx * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float hs = 0.5f * (r[a][1] + r[a][2]), hd = 0.5f * (r[a][1] - r[a][2]);
        d[3 * a + 0] = r[a][0] + hs;
        d[3 * a + 1] = hd;
        d[3 * a + 2] = hs + r[a][3];
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3 forward / input gradient in the Winograd F(2x2, 3x3) domain for the deep layers (256/512 channels at
// 120^2 and below), where the channel counts make the 16 per-position GEMMs (tiles x Cin) . (Cout x Cin)^T efficient
// and the 4x larger transformed tensors small:   V = B^T d B  ->  M_p = V_p . U_p^T  ->  Y = A^T M A (+ epilogue).
// U = G g G^T per (co, ci) is re-derived from the weights once per step (pack kernel below); the input gradient is
// the same pipeline over dy with the filter rotated by 180 degrees and its channel roles swapped.
// ---------------------------------------------------------------------------------------------
// mode 0: U[p][co][ci] (forward);  mode 1: Ud[p][ci][co] from the rotated filter (dgrad).  thread = one (row, col)
// of the output matrix, col fastest (coalesced stores)
__global__ void wino_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int mode) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Co * Ci) return;
    int co, ci;
    if (mode == 0) { co = idx / Ci; ci = idx - (long)co * Ci; }
    else { ci = idx / Co; co = idx - (long)ci * Co; }
    const float* gsrc = w + ((long)co * Ci + ci) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = mode == 0 ? gsrc[3 * a + b] : gsrc[8 - (3 * a + b)];
    float r[4][3];       // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float hs = 0.5f * (g[0][b] + g[2][b]), hm = 0.5f * g[1][b];
        r[0][b] = g[0][b];
        r[1][b] = hs + hm;
        r[2][b] = hs - hm;
        r[3][b] = g[2][b];
    }
    const long ps = (long)Co * Ci;
    float* out = U + idx;
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (.) G^T
        const float hs = 0.5f * (r[a][0] + r[a][2]), hm = 0.5f * r[a][1];
        out[(4 * a + 0) * ps] = r[a][0];
        out[(4 * a + 1) * ps] = hs + hm;
        out[(4 * a + 2) * ps] = hs - hm;
        out[(4 * a + 3) * ps] = r[a][2];
    }
}

// thread = (tile, 4 channels): Y = A^T M A for the tile's 2x2 outputs, then the conv epilogue (bias / mask /
// accumulate / second ReLU'd output) on the pixels inside the image
__global__ __launch_bounds__(256) void wino_output_transform_kernel(const float* __restrict__ Mt, const float* __restrict__ bias,
                                                                    const float* __restrict__ mask, float* __restrict__ y,
                                                                    float* __restrict__ y_relu, float* __restrict__ y_pool,
                                                                    int pool_relu, const WinoGeom g, int accum) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int Q = g.C >> 2;
    if (idx >= g.T * Q) return;
    const int t = fast_div((int)idx, g.dQ);
    const int cq = (int)idx - t * Q;
    const int bi = fast_div(t, g.dTw);
    const int j = t - bi * g.Tw;
    const int b = fast_div(bi, g.dTh);
    const int i = bi - b * g.Th;
    const float* src = Mt + (long)t * g.C + 4 * cq;
    const long ps = g.ps;
#define F4(op, a, b) make_float4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w)
    float4 s[2][4];      // rows: A^T m
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 m0 = ld4(src + (0 + c) * ps), m1 = ld4(src + (4 + c) * ps), m2 = ld4(src + (8 + c) * ps),
                     m3 = ld4(src + (12 + c) * ps);
        const float4 t12 = F4(+, m1, m2), d12 = F4(-, m1, m2);
        s[0][c] = F4(+, m0, t12);
        s[1][c] = F4(-, d12, m3);
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = ld4(bias + 4 * cq);
    // a 2x2 output tile is exactly one window of the 2x2 / stride-2 max-pool that may follow the layer (floor mode: a
    // tile that hangs over an odd border has no pooled pixel)
    const float ninf = -__builtin_inff();
    float4 pm = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int h = 2 * i + r;
        if (h >= g.H) break;
        const float4 t12 = F4(+, s[r][1], s[r][2]), d12 = F4(-, s[r][1], s[r][2]);
        float4 o[2];
        o[0] = F4(+, s[r][0], t12);
        o[1] = F4(-, d12, s[r][3]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int w = 2 * j + c;
            if (w >= g.W) break;
            const long off = (((long)b * g.H + h) * g.W + w) * g.C + 4 * cq;
            float4 v = F4(+, o[c], bv);
            if (mask) {
                const float4 mk = ld4(mask + off);
                v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
            if (accum) {
                const float4 old = ld4(y + off);
                v = F4(+, v, old);
            }
            st4(y + off, v);
            if (y_relu) st4(y_relu + off, relu4(v));
            pm.x = fmaxf(pm.x, v.x); pm.y = fmaxf(pm.y, v.y); pm.z = fmaxf(pm.z, v.z); pm.w = fmaxf(pm.w, v.w);
        }
    }
    if (y_pool && 2 * i + 1 < g.H && 2 * j + 1 < g.W) {
        const int Hp = g.H >> 1, Wp = g.W >> 1;
        st4(y_pool + (((long)b * Hp + i) * Wp + j) * g.C + 4 * cq, pool_relu ? relu4(pm) : pm);
    }
#undef F4
}

extern "C" size_t wesup_winograd_weight_floats(int Cin, int Cout) { return (size_t)16 * Cin * Cout; }

// w (Cout,Cin,3,3) -> u_fwd [16][Cout][Cin] and/or u_dgrad [16][Cin][Cout] (either may be NULL)
extern "C" int wesup_winograd_pack_weight(const float* w, float* u_fwd, float* u_dgrad, int Cout, int Cin, void* stream) {
    if (!w || Cout <= 0 || Cin <= 0 || (!u_fwd && !u_dgrad)) return WESUP_ERR_INVALID;
    const long tot = (long)Cout * Cin;
    hipStream_t st = (hipStream_t)stream;
    if (u_fwd) {
        hipLaunchKernelGGL(wino_weight_transform_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, u_fwd, Cout,
                           Cin, 0);
        WESUP_CHECK_LAUNCH();
    }
    if (u_dgrad) {
        hipLaunchKernelGGL(wino_weight_transform_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, u_dgrad, Cout,
                           Cin, 1);
        WESUP_CHECK_LAUNCH();
    }
    return WESUP_OK;
}

static WinoGeom wino_geom(int B, int H, int W, int C, long plane_elems = 0) {
    WinoGeom g;
    g.H = H; g.W = W; g.Th = (H + 1) / 2; g.Tw = (W + 1) / 2; g.T = wino_tiles(B, H, W);
    g.ps = plane_elems > 0 ? plane_elems : g.T * C;
    g.dTw = make_fastdiv(g.Tw); g.dTh = make_fastdiv(g.Th);
    g.C = C; g.dQ = make_fastdiv(C / 4);
    return g;
}

// x (B,H,W,C) -> V [16][tiles][C]
extern "C" int wesup_winograd_input_transform(const float* x, float* V, long plane_elems, int B, int H, int W, int C,
                                              int relu_in, void* stream) {
    if (!x || !V || !wino_shape_ok(B, H, W, C, C) || (((uintptr_t)x | (uintptr_t)V) & 15) || (plane_elems % 4) ||
        (plane_elems > 0 && plane_elems < wino_tiles(B, H, W) * C))
        return WESUP_ERR_INVALID;
    const WinoGeom g = wino_geom(B, H, W, C, plane_elems);
    hipLaunchKernelGGL(wino_input_transform_kernel, dim3((unsigned)ceil_div(g.T * (C / 4), 256l)), dim3(256), 0,
                       (hipStream_t)stream, x, V, g, relu_in);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// Mt [16][tiles][C] -> y (B,H,W,C) = A^T M A + bias, masked by mask_src > 0, added to the old y if accumulate;
// y_relu: optional second output max(y, 0);  y_pool: optional third output (B,H/2,W/2,C) = the 2x2 / stride-2 max-pool
// of y (a tile is one pooling window), ReLU'd if pool_relu
extern "C" int wesup_winograd_output_transform(const float* Mt, long plane_elems, const float* bias, const float* mask_src,
                                               float* y, float* y_relu, float* y_pool, int pool_relu, int B, int H, int W,
                                               int C, int accumulate, void* stream) {
    if (!Mt || !y || !wino_shape_ok(B, H, W, C, C) || (plane_elems % 4) ||
        (plane_elems > 0 && plane_elems < wino_tiles(B, H, W) * C) ||
        (((uintptr_t)Mt | (uintptr_t)y | (uintptr_t)y_relu | (uintptr_t)y_pool | (uintptr_t)mask_src | (uintptr_t)bias) & 15))
        return WESUP_ERR_INVALID;
    const WinoGeom g = wino_geom(B, H, W, C, plane_elems);
    hipLaunchKernelGGL(wino_output_transform_kernel, dim3((unsigned)ceil_div(g.T * (C / 4), 256l)), dim3(256), 0,
                       (hipStream_t)stream, Mt, bias, mask_src, y, y_relu, y_pool, pool_relu, g, accumulate);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// dy (B,H,W,C) -> dM [16][tiles][C] = A dY A^T per 2x2 tile (the weight gradient's second operand)
extern "C" int wesup_winograd_outgrad_transform(const float* dy, float* dM, int B, int H, int W, int C, void* stream) {
    if (!dy || !dM || !wino_shape_ok(B, H, W, C, C) || (((uintptr_t)dy | (uintptr_t)dM) & 15)) return WESUP_ERR_INVALID;
    const WinoGeom g = wino_geom(B, H, W, C);
    hipLaunchKernelGGL(wino_outgrad_transform_kernel, dim3((unsigned)ceil_div(g.T * (C / 4), 256l)), dim3(256), 0,
                       (hipStream_t)stream, dy, dM, g);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// slabs [16][S][Cout*Cin + Cout] (split-K partial products of dU_p, each followed by Cout column sums of dM_p) ->
// dw (Cout,Cin,3,3) = G^T (sum over S) G, db (Cout) = the column sums of position 5.  slab_stride = elements between two
// splits, batch_stride = between two positions.
extern "C" int wesup_winograd_filter_grad(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs,
                                          float* db, int Cout, int Cin, void* stream) {
    if (!slabs || !dw_kcrs || S <= 0 || Cout <= 0 || Cin <= 0 || slab_stride < (long)Cout * Cin + Cout ||
        batch_stride < (long)S * slab_stride)
        return WESUP_ERR_INVALID;
    const long tot = (long)Cout * Cin;
    const int pair_blocks = (int)((tot + 15) / 16);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)(pair_blocks + (Cout + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, slabs, slab_stride, batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
