// Winograd F(m x m, 3x3) transforms for gfx950, m = 2 and m = 4: everything of the Winograd-domain convolutions that is
// NOT a GEMM -- the memory-bound passes between the activations / filters and the [(m+2)^2 positions][tiles][C] operands of
// the batched GEMMs in gemm.hip (wesup_gemm_nt_batched, the TN launch inside wesup_conv3x3_wgrad_winograd).  DESIGN.md 3.1.1.
//
//   forward / input gradient:   V = B^T d B  ->  M_p = V_p . U_p^T  (gemm.hip)  ->  Y = A^T M A (+ epilogue)
//   weight gradient:            dM = A dY A^T, V as above  ->  dU_p = dM_p^T . V_p  (gemm.hip)  ->  dg = G^T dU G
//   filters:                    U = G g G^T once per step (forward), and of the rotated filter (input gradient)
//
// B^T, G, A^T: F(2x2,3x3) are the published minimal-filtering matrices (points 0, +-1, inf; entries 0, +-1, +-1/2; 16
// positions, 4x the activation bytes, 4/9 of the direct form's multiply-adds).  F(4x4,3x3) (36 positions, 2.25x the bytes,
// 1/4 of the multiply-adds) is the same Toom-Cook construction over the points 0, +-3/4, +-3/2, inf instead of the textbook
// 0, +-1, +-2, inf: ~4x less fp32 error (about 1e-6 of the tensor's maximum, the level of the direct kernels; the textbook
// set gave 4-7e-6), every entry of B^T and A^T a dyadic rational (exact in fp32), the +- pairs share their sub-expressions
// (oracle/winograd_oracle.py toom_cook()).  A transformed tensor is position-major so that each
// GEMM operand is a plain row-major matrix.  Every kernel: one thread per (tile, 4 channels), channels fastest across
// lanes, 16-byte loads and stores of contiguous channel rows; tiles that hang over a ragged border read zeros and skip
// the stores.
#include "winograd.hpp"
#include <cstdlib>

struct WinoGeom {
    int H, W, C, Th, Tw;
    int nt;              // 1: the transformed tensor is written with non-temporal (streaming) stores, see wino_nt_stores()
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

// dw[co][ci][3][3] = G^T (sum_s slab[p][s][co][ci]) G ;  db[co] (F(2x2) only) = sum_s (column sums of dM at position (1,1))
// NP = m + 2.  (A thread per pair walked NP^2 x S dependent loads -- 512 at conv2_2 -- with only Co*Ci/256 blocks on the
// chip: 113 us per launch on average, 1.1 ms per step; hence a thread per (pair, position group).)  Fixed order.
template <int NP>
__device__ __forceinline__ void wino_gt(const float (&u)[NP], float (&r)[3]) {      // one row of G^T (.)
    if constexpr (NP == 4) {
        const float hs = 0.5f * (u[1] + u[2]), hd = 0.5f * (u[1] - u[2]);
        r[0] = u[0] + hs;
        r[1] = hd;
        r[2] = hs + u[3];
    } else {
        const float s12 = u[1] + u[2], s34 = u[3] + u[4];        // G^T of the F(4x4,3x3) point set (columns of wino4_g)
        r[0] = (64.f / 81.f) * u[0] - (128.f / 243.f) * s12 + (32.f / 243.f) * s34;
        r[1] = (32.f / 81.f) * (u[2] - u[1]) + (16.f / 81.f) * (u[3] - u[4]);
        r[2] = (8.f / 27.f) * (s34 - s12) + u[5];
    }
}
// block = 64 (co, ci) pairs x all NP^2 positions: thread t adds the S split-K slabs of pair t & 63 at the positions
// t >> 6, (t >> 6) + 4, ... -- every load instruction of a wave reads 256 consecutive bytes of one slab (with 7 pairs per
// block, as many as fit one thread per (pair, position) at 36 positions, the rows were 28 bytes: 65 us per launch, 0.65 ms
// per step); the sums of a pair meet in LDS and 64 threads apply G^T (.) G.
// Bias gradient, in the blocks behind the pair blocks: F(2x2) from the column sums stored behind each slab of position
// (1,1); F(4x4) from the per-block rows wino4_outgrad_transform_kernel left (bias_part [bias_rows][Co]): a block takes 16
// channels, 64 row groups of 4 lanes each walk the rows, LDS folds the groups in a fixed order.
// PB pairs per block: 64, or 16 for the narrow layers (64 x 64 ... 128 x 128 filters: with 64 pairs a block the grid is 64 ..
// 256 blocks and every thread walks 9 positions x S <= 28 slabs one load after the other; 16 pairs x 16 position groups
// quarter that walk and fill the chip.  The sum over the slabs of one (position, pair) keeps its order: same bits.)
__device__ __forceinline__ float4 wr_add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// VEC: four consecutive pairs per thread (16-byte loads; strides and the pair count divisible by 4), every load of a thread's
// positions in flight before the first sum -- beside the conv chain's kernels a load takes several times as long as alone, and
// with two scalar loads in flight per thread this kernel ran 70 us there against 15 us alone.  Same sums in the same order.
template <int NP, int PB, bool VEC = false>
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, long stride, long batch_slab,
                                                                float* __restrict__ dw, int Co, int Ci, int S,
                                                                float* __restrict__ db, int pair_blocks,
                                                                const float* __restrict__ bias_part, int bias_rows) {
    constexpr int P = NP * NP, PG = 256 / PB;            // PG position groups
    constexpr int QB = VEC ? 4 * PB : PB;                // pairs per block
    __shared__ __attribute__((aligned(16))) float us[(P * (QB + 1) > 1024 ? P * (QB + 1) : 1024)];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= pair_blocks) {                // bias gradient
        const int bb = blockIdx.x - pair_blocks;
        if (!db) return;
        if (bias_part) {
            float4* sh = reinterpret_cast<float4*>(us);              // 256 float4 = 4 KiB <= sizeof(us)
            const int cq = tid & 3, rg = tid >> 2, c = 16 * bb + 4 * cq;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < Co) {
                // (57 600 rows for a 480^2 x 64 gradient, 4 blocks: eight rows in flight per thread, eight running sums
                // folded in a fixed order)
                float4 a8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a8[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                int r = rg;
                for (; r + 7 * 64 < bias_rows; r += 8 * 64) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld4(bias_part + (long)(r + 64 * u) * Co + c);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a8[u].x += v[u].x; a8[u].y += v[u].y; a8[u].z += v[u].z; a8[u].w += v[u].w; }
                }
                for (; r < bias_rows; r += 64) {
                    const float4 v = ld4(bias_part + (long)r * Co + c);
                    a8[0].x += v.x; a8[0].y += v.y; a8[0].z += v.z; a8[0].w += v.w;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += a8[u].x; acc.y += a8[u].y; acc.z += a8[u].z; acc.w += a8[u].w; }
            }
            sh[tid] = acc;
            __syncthreads();
            if (tid < 4 && c < Co) {
                for (int k = 1; k < 64; ++k) {
                    const float4 v = sh[4 * k + tid];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                st4(db + c, acc);
            }
            return;
        }
        const long m = (long)bb * 256 + tid;
        if (m < Co) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += slab[(NP + 1) * batch_slab + (long)k * stride + (long)Co * Ci + m];
            db[m] = s;
        }
        return;
    }
    int i = tid & (PB - 1);
    long idx = (long)blockIdx.x * PB + i;
    bool ok = idx < (long)Co * Ci;
    if constexpr (VEC) {
        const long idx4 = ((long)blockIdx.x * PB + i) * 4;
        const bool ok4 = idx4 < (long)Co * Ci;          // (the pair count is a multiple of 4)
        constexpr int NPOS = (P + PG - 1) / PG;
        const int pg = tid / PB;
#pragma unroll
        for (int n = 0; n < NPOS; ++n) {
            const int p = pg + n * PG;
            if (p >= P) break;
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
            if (ok4) {
                const float* src = slab + p * batch_slab + idx4;
                int k = 0;
                for (; k + 3 < S; k += 4) {
                    const float4 v0 = ld4(src + (long)k * stride), v1 = ld4(src + (long)(k + 1) * stride);
                    const float4 v2 = ld4(src + (long)(k + 2) * stride), v3 = ld4(src + (long)(k + 3) * stride);
                    s0 = wr_add4(s0, v0); s1 = wr_add4(s1, v1); s0 = wr_add4(s0, v2); s1 = wr_add4(s1, v3);
                }
                for (; k + 1 < S; k += 2) {
                    const float4 v0 = ld4(src + (long)k * stride), v1 = ld4(src + (long)(k + 1) * stride);
                    s0 = wr_add4(s0, v0); s1 = wr_add4(s1, v1);
                }
                if (k < S) s0 = wr_add4(s0, ld4(src + (long)k * stride));
            }
            float* u = us + p * (QB + 1) + 4 * i;
            u[0] = s0.x + s1.x; u[1] = s0.y + s1.y; u[2] = s0.z + s1.z; u[3] = s0.w + s1.w;
        }
        __syncthreads();
        if (tid >= QB) return;
        i = tid;
        idx = (long)blockIdx.x * QB + i;
        ok = idx < (long)Co * Ci;
        if (!ok) return;
    } else {
    for (int p = tid / PB; p < P; p += PG) {
        float s0 = 0.f, s1 = 0.f;
        if (ok) {
            const float* src = slab + p * batch_slab + idx;
            int k = 0;
            for (; k + 1 < S; k += 2) { s0 += src[(long)k * stride]; s1 += src[(long)(k + 1) * stride]; }
            if (k < S) s0 += src[(long)k * stride];
        }
        us[p * (PB + 1) + i] = s0 + s1;
    }
    __syncthreads();
    if (tid >= PB || !ok) return;
    }
    float r[3][NP];      // G^T u
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        float col[NP], o[3];
#pragma unroll
        for (int a = 0; a < NP; ++a) col[a] = us[(NP * a + c) * (QB + 1) + i];
        wino_gt<NP>(col, o);
        r[0][c] = o[0]; r[1][c] = o[1]; r[2][c] = o[2];
    }
    float* d = dw + idx * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float o[3];
        wino_gt<NP>(r[a], o);
        d[3 * a + 0] = o[0];
        d[3 * a + 1] = o[1];
        d[3 * a + 2] = o[2];
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

// ---------------------------------------------------------------------------------------------
// F(4x4, 3x3): 6x6 input patches at stride 4, 36 positions.  Same thread mapping as above; the block index goes through
// the XCD remap so that one XCD walks a contiguous range of tiles (neighbouring tile rows share two of their six patch
// rows: with round-robin blocks those re-reads would meet in eight different L2s).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// s * a + b
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
// 1-D transforms over the points 0, +-3/4, +-3/2, inf (rows as in oracle/winograd_oracle.py _BT[4], _AT[4]).
// B^T (.) for a 6-vector: rows [81/64,0,-45/16,0,1,0], [0,-27/16,-9/4,3/4,1,0], [0,27/16,-9/4,-3/4,1,0],
// [0,-27/32,-9/16,3/2,1,0], [0,27/32,-9/16,-3/2,1,0], [0,81/64,0,-45/16,0,1]
__device__ __forceinline__ void wino4_bt(const float4 (&d)[6], float4 (&t)[6]) {
    t[0] = f4fma(81.f / 64.f, d[0], f4fma(-45.f / 16.f, d[2], d[4]));
    const float4 a = f4fma(-9.f / 4.f, d[2], d[4]), b = f4fma(-9.f / 4.f, d[1], d[3]);        // b: (3/4) b below
    t[1] = f4fma(0.75f, b, a);
    t[2] = f4fma(-0.75f, b, a);
    const float4 c = f4fma(-9.f / 16.f, d[2], d[4]), e = f4fma(-9.f / 16.f, d[1], d[3]);
    t[3] = f4fma(1.5f, e, c);
    t[4] = f4fma(-1.5f, e, c);
    t[5] = f4fma(81.f / 64.f, d[1], f4fma(-45.f / 16.f, d[3], d[5]));
}
// A^T (.) for a 6-vector: rows [1,1,1,1,1,0], [0,3/4,-3/4,3/2,-3/2,0], [0,9/16,9/16,9/4,9/4,0], [0,27/64,-27/64,27/8,-27/8,1]
__device__ __forceinline__ void wino4_at(const float4 (&m)[6], float4 (&o)[4]) {
    const float4 s12 = f4add(m[1], m[2]), d12 = f4sub(m[1], m[2]), s34 = f4add(m[3], m[4]), d34 = f4sub(m[3], m[4]);
    o[0] = f4add(f4add(m[0], s12), s34);
    o[1] = f4fma(1.5f, d34, f4scale(0.75f, d12));
    o[2] = f4fma(9.f / 4.f, s34, f4scale(9.f / 16.f, s12));
    o[3] = f4add(f4fma(27.f / 8.f, d34, f4scale(27.f / 64.f, d12)), m[5]);
}
// A (.) for a 4-vector (A = (A^T)^T): rows [1,0,0,0], [1,+-3/4,9/16,+-27/64], [1,+-3/2,9/4,+-27/8], [0,0,0,1]
__device__ __forceinline__ void wino4_a(const float4 (&y)[4], float4 (&t)[6]) {
    const float4 e1 = f4fma(9.f / 16.f, y[2], y[0]), o1 = f4fma(9.f / 16.f, y[3], y[1]);
    const float4 e2 = f4fma(9.f / 4.f, y[2], y[0]), o2 = f4fma(9.f / 4.f, y[3], y[1]);
    t[0] = y[0];
    t[1] = f4fma(0.75f, o1, e1);
    t[2] = f4fma(-0.75f, o1, e1);
    t[3] = f4fma(1.5f, o2, e2);
    t[4] = f4fma(-1.5f, o2, e2);
    t[5] = y[3];
}

struct WinoTile { int t, cq, b, i, j; };
__device__ __forceinline__ bool wino4_decode(const WinoGeom& g, WinoTile& o) {
    const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int Q = g.C >> 2;
    if (idx >= g.T * Q) return false;
    o.t = fast_div((int)idx, g.dQ);
    o.cq = (int)idx - o.t * Q;
    const int bi = fast_div(o.t, g.dTw);
    o.j = o.t - bi * g.Tw;
    o.b = fast_div(bi, g.dTh);
    o.i = bi - o.b * g.Th;
    return true;
}

// thread = (tile, 4 channels): 36 float4 loads (patch rows 4i-1 .. 4i+4), B^T d B, 36 float4 stores
// bits (optional, [B][H][W][C/4] bytes): bit j of a byte = x[..., 4 cq + j] > 0 for the 4x4 pixels the tile owns (its
// patch without the halo) -- the ReLU decisions of the layer below, kept for its backward (16x smaller than the mask tensor).
template <bool KEEP_CORE = false>
__device__ __forceinline__ void wino4_input_body(const float* __restrict__ x, float* __restrict__ V, const WinoGeom& g,
                                                 const WinoTile& q, int relu, unsigned char* __restrict__ bits,
                                                 float4 (*core)[4] = nullptr) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // All 36 loads first, nothing between them: with the sign-bit stores in the same loop every load was followed by its own
    // s_waitcnt (36 serial round trips per thread -- unnoticed alone, where other waves cover them, and the reason this kernel
    // doubled its time beside the other streams' kernels).
    float4 d[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int w = 4 * q.j - 1 + c;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int h = 4 * q.i - 1 + r;
            const bool in = (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
            d[r][c] = in ? ld4(x + (((long)q.b * g.H + h) * g.W + w) * g.C + 4 * q.cq) : z;
        }
    }
    if constexpr (KEEP_CORE) {           // the tile's own 4x4 values for the caller (the dual transform's second half)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) core[r][c] = d[r + 1][c + 1];
    }
    if (bits) {
        unsigned bt[4][4];       // (all sixteen values first: a store whose source register is reused waits for the one before)
#pragma unroll
        for (int c = 1; c <= 4; ++c)
#pragma unroll
            for (int r = 1; r <= 4; ++r) {
                const float4 v = d[r][c];
                bt[r - 1][c - 1] = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            }
        unsigned char* b0 = bits + (((long)q.b * g.H + 4 * q.i) * g.W + 4 * q.j) * (g.C >> 2) + q.cq;
        const long qs = g.C >> 2, rsb = (long)g.W * qs;
        if (4 * q.i + 4 <= g.H && 4 * q.j + 4 <= g.W) {      // the tile lies inside the image (all but the last row / column of
#pragma unroll                                              // tiles): sixteen stores in one block, none waiting for another
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) b0[r * rsb + c * qs] = (unsigned char)bt[r][c];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * q.i + r < g.H && 4 * q.j + c < g.W) b0[r * rsb + c * qs] = (unsigned char)bt[r][c];
        }
    }
    float4 m[6][6];      // B^T d
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float4 col[6], t[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = relu ? relu4(d[r][c]) : d[r][c];
        wino4_bt(col, t);
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r][c] = t[r];
    }
    float* out = V + (long)q.t * g.C + 4 * q.cq;
    const long ps = g.ps;
#pragma unroll
    for (int r = 0; r < 6; ++r) {   // (.) B
        float4 o[6];
        wino4_bt(m[r], o);
#pragma unroll
        for (int c = 0; c < 6; ++c) st4s(out + (6 * r + c) * ps, o[c], g.nt);
    }
}
__global__ __launch_bounds__(256) void wino4_input_transform_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                    const WinoGeom g, int relu, unsigned char* __restrict__ bits) {
    WinoTile q;
    if (!wino4_decode(g, q)) return;
    wino4_input_body(x, V, g, q, relu, bits);
}

// thread = (tile, 4 channels): the tile's 4x4 gradients -> A dY A^T (6x6).
// colsum_part (optional, [blocks][C]): the bias gradient rides along.  This point set has no point 1, so no single position
// of dM is the plain sum of a tile's gradients (for F(2x2) position (1,1) is, and the TN GEMM sums its column); instead
// every thread adds up the 16 gradients it has loaded anyway, the tiles of a block meet in LDS in a fixed order (needs
// C/4 | 256), and wesup_colsum folds the per-block rows: dy is not read a second time, no float atomics.
template <bool HAVE_CORE = false>
__device__ __forceinline__ float4 wino4_outgrad_body(const float* __restrict__ dy, float* __restrict__ dM, const WinoGeom& g,
                                                     const WinoTile& q, const float4 (*core)[4] = nullptr) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tot = z;
    {
        float4 m[6][4];      // A dY
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int w = 4 * q.j + c;
            float4 y[4], t[6];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = 4 * q.i + r;
                if constexpr (HAVE_CORE) y[r] = core[r][c];      // (out-of-image positions were loaded as zeros)
                else y[r] = (h < g.H && w < g.W) ? ld4(dy + (((long)q.b * g.H + h) * g.W + w) * g.C + 4 * q.cq) : z;
            }
            tot = f4add(tot, f4add(f4add(y[0], y[1]), f4add(y[2], y[3])));
            wino4_a(y, t);
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r][c] = t[r];
        }
        float* out = dM + (long)q.t * g.C + 4 * q.cq;
        const long ps = g.ps;
#pragma unroll
        for (int r = 0; r < 6; ++r) {   // (.) A^T
            float4 o[6];
            wino4_a(m[r], o);
#pragma unroll
            for (int c = 0; c < 6; ++c) st4s(out + (6 * r + c) * ps, o[c], g.nt);
        }
    }
    return tot;
}
// the per-block column sums of the gradients the block's threads loaded (tot), one row of colsum_part per block
__device__ __forceinline__ void wino4_block_colsum(float4 tot, const WinoGeom& g, float* __restrict__ colsum_part, float4* sh) {
    const int Q = g.C >> 2, tid = threadIdx.x;
    sh[tid] = tot;
    __syncthreads();
    if (tid < Q) {
        float4 s = sh[tid];
        for (int k = tid + Q; k < 256; k += Q) s = f4add(s, sh[k]);
        st4(colsum_part + (long)xcd_remap(blockIdx.x, gridDim.x) * g.C + 4 * tid, s);
    }
}
__global__ __launch_bounds__(256) void wino4_outgrad_transform_kernel(const float* __restrict__ dy, float* __restrict__ dM,
                                                                      const WinoGeom g, float* __restrict__ colsum_part) {
    __shared__ float4 sh[256];
    WinoTile q;
    const bool active = wino4_decode(g, q);
    const float4 tot = active ? wino4_outgrad_body(dy, dM, g, q) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (!colsum_part) return;            // uniform
    wino4_block_colsum(tot, g, colsum_part, sh);
}
// Both transforms of one gradient tensor in one pass over it: V = B^T dY B of the 6x6 patches (the input of the layer's input
// gradient) and dM = A dY A^T of their 4x4 cores (the second operand of its weight gradient).  A thread does the first, then the
// second on the 16 core values it re-reads (cache hits: it loaded them a moment ago); dy comes from HBM once instead of once
// per stream, and one launch replaces two.
__global__ __launch_bounds__(256) void wino4_dual_transform_kernel(const float* __restrict__ dy, float* __restrict__ V,
                                                                   float* __restrict__ dM, const WinoGeom g,
                                                                   float* __restrict__ colsum_part) {
    __shared__ float4 sh[256];
    WinoTile q;
    const bool active = wino4_decode(g, q);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        float4 core[4][4];           // dy is read once: the 16 values both transforms share stay in registers
        wino4_input_body<true>(dy, V, g, q, 0, nullptr, core);
        tot = wino4_outgrad_body<true>(dy, dM, g, q, core);
    }
    if (!colsum_part) return;            // uniform
    wino4_block_colsum(tot, g, colsum_part, sh);
}

// thread = (tile, 4 channels): Y = A^T M A for the tile's 4x4 outputs, then the conv epilogue on the pixels inside the
// image; a 4x4 output tile holds four windows of the 2x2 / stride-2 max-pool that may follow the layer
__global__ __launch_bounds__(256) void wino4_output_transform_kernel(const float* __restrict__ Mt, const float* __restrict__ bias,
                                                                     const float* __restrict__ mask, float* __restrict__ y,
                                                                     float* __restrict__ y_relu, float* __restrict__ y_pool,
                                                                     int pool_relu, const WinoGeom g, int accum,
                                                                     const WinoUnpool up) {
    WinoTile q;
    if (!wino4_decode(g, q)) return;
    const float* src = Mt + (long)q.t * g.C + 4 * q.cq;
    const long ps = g.ps;
    float4 s[4][6];      // A^T m
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float4 m[6], o[4];
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r] = ld4(src + (6 * r + c) * ps);
        wino4_at(m, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r][c] = o[r];
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = ld4(bias + 4 * q.cq);
    const float ninf = -__builtin_inff();
    const int Hp = g.H >> 1, Wp = g.W >> 1;
    float4 pm[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = 4 * q.i + r;
        if ((r & 1) == 0) pm[0] = pm[1] = make_float4(ninf, ninf, ninf, ninf);
        if (h < g.H && !up.src) {
            // the four pixels of a tile row together: what the epilogue reads per pixel (ReLU mask, the old gradient) is loaded for
            // all of them before the first use; columns outside the image are clamped and predicated
            float4 o[4], mk[4], old[4];
            long offc[4];
            wino4_at(s[r], o);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int w = min(4 * q.j + c, g.W - 1);
                offc[c] = (((long)q.b * g.H + h) * g.W + w) * g.C + 4 * q.cq;
                if (mask) mk[c] = ld4(mask + offc[c]);
                if (accum) old[c] = ld4(y + offc[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (4 * q.j + c >= g.W) break;
                float4 v = f4add(o[c], bv);
                if (mask) {
                    v.x = mk[c].x > 0.f ? v.x : 0.f; v.y = mk[c].y > 0.f ? v.y : 0.f;
                    v.z = mk[c].z > 0.f ? v.z : 0.f; v.w = mk[c].w > 0.f ? v.w : 0.f;
                }
                if (accum) v = f4add(v, old[c]);
                st4(y + offc[c], v);
                if (y_relu) st4(y_relu + offc[c], relu4(v));
                float4& p = pm[c >> 1];
                p.x = fmaxf(p.x, v.x); p.y = fmaxf(p.y, v.y); p.z = fmaxf(p.z, v.z); p.w = fmaxf(p.w, v.w);
            }
        } else if (h < g.H) {
            float4 o[4];
            wino4_at(s[r], o);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int w = 4 * q.j + c;
                if (w >= g.W) break;
                const long off = (((long)q.b * g.H + h) * g.W + w) * g.C + 4 * q.cq;
                float4 v = f4add(o[c], bv);
                if (mask) {
                    const float4 mk = ld4(mask + off);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                    v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                if (up.src) {        // input gradient at pooled resolution: straight through the max-pool backward
                    wino_unpool_add(up, nullptr, q.b, h, w, g.C, 4 * q.cq, v);
                    continue;
                }
                if (accum) v = f4add(v, ld4(y + off));
                st4(y + off, v);
                if (y_relu) st4(y_relu + off, relu4(v));
                float4& p = pm[c >> 1];
                p.x = fmaxf(p.x, v.x); p.y = fmaxf(p.y, v.y); p.z = fmaxf(p.z, v.z); p.w = fmaxf(p.w, v.w);
            }
        }
        if (y_pool && (r & 1)) {         // rows 4i + r - 1 and 4i + r are done: pooled row 2i + r/2 (floor mode)
            const int ph = 2 * q.i + (r >> 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pw = 2 * q.j + k;
                if (ph < Hp && pw < Wp)
                    st4(y_pool + (((long)q.b * Hp + ph) * Wp + pw) * g.C + 4 * q.cq, pool_relu ? relu4(pm[k]) : pm[k]);
            }
        }
    }
}

// F(4x4,3x3) filters: U = G g G^T (6x6) per (co, ci); mode as in wino_weight_transform_kernel
// G (.) for a 3-vector: rows [64/81,0,0], [-128/243,-+32/81,-8/27], [32/243,+-16/81,8/27], [0,0,1]
__device__ __forceinline__ void wino4_g(float g0, float g1, float g2, float (&u)[6]) {
    const float a = fmaf(-128.f / 243.f, g0, (-8.f / 27.f) * g2), b = (32.f / 81.f) * g1;
    const float c = fmaf(32.f / 243.f, g0, (8.f / 27.f) * g2), d = (16.f / 81.f) * g1;
    u[0] = (64.f / 81.f) * g0;
    u[1] = a - b;
    u[2] = a + b;
    u[3] = c + d;
    u[4] = c - d;
    u[5] = g2;
}
__global__ void wino4_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int mode) {
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
    float r[6][3];       // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float u[6];
        wino4_g(g[0][b], g[1][b], g[2][b], u);
#pragma unroll
        for (int a = 0; a < 6; ++a) r[a][b] = u[a];
    }
    const long ps = (long)Co * Ci;
    float* out = U + idx;
#pragma unroll
    for (int a = 0; a < 6; ++a) {   // (.) G^T
        float u[6];
        wino4_g(r[a][0], r[a][1], r[a][2], u);
#pragma unroll
        for (int b = 0; b < 6; ++b) out[(6 * a + b) * ps] = u[b];
    }
}

extern "C" size_t wesup_winograd_weight_floats(int Cin, int Cout, int m) {
    return wino_m_ok(m) ? (size_t)wino_positions(m) * Cin * Cout : 0;
}
extern "C" long wesup_winograd_tiles(int B, int H, int W, int m) { return wino_m_ok(m) ? wino_tiles(B, H, W, m) : 0; }

// w (Cout,Cin,3,3) -> u_fwd [P][Cout][Cin] and/or u_dgrad [P][Cin][Cout] (either may be NULL), P = (m+2)^2
extern "C" int wesup_winograd_pack_weight(const float* w, float* u_fwd, float* u_dgrad, int Cout, int Cin, int m, void* stream) {
    if (!w || Cout <= 0 || Cin <= 0 || (!u_fwd && !u_dgrad) || !wino_m_ok(m)) return WESUP_ERR_INVALID;
    const long tot = (long)Cout * Cin;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((tot + 255) / 256));
    for (int mode = 0; mode < 2; ++mode) {
        float* u = mode == 0 ? u_fwd : u_dgrad;
        if (!u) continue;
        if (m == 2) WESUP_LAUNCH(wino_weight_transform_kernel, grid, dim3(256), 0, st, w, u, Cout, Cin, mode);
        else WESUP_LAUNCH(wino4_weight_transform_kernel, grid, dim3(256), 0, st, w, u, Cout, Cin, mode);
        WESUP_CHECK_LAUNCH();
    }
    return WESUP_OK;
}

// The filters of SEVERAL layers in one launch (F(4x4,3x3)): the step re-derives 12 forward and 12 rotated filter sets from the
// updated weights, 24 launches of 4-16 us that sat at the head of every iteration.  A block finds its job by a scan of the
// (at most 32) block ranges, then works as wino4_weight_transform_kernel does.
#define WINO_WB_MAX 32
struct WinoWBatch {
    const float* w[WINO_WB_MAX];
    float* u[WINO_WB_MAX];
    int co[WINO_WB_MAX], ci[WINO_WB_MAX], mode[WINO_WB_MAX];
    int first[WINO_WB_MAX + 1];      // first block of job j; first[n] = grid size
    int n;
};
__device__ __forceinline__ void wino4_weight_transform_one(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci,
                                                           int mode, long idx) {
    int co, ci;
    if (mode == 0) { co = idx / Ci; ci = idx - (long)co * Ci; }
    else { ci = idx / Co; co = idx - (long)ci * Co; }
    const float* gsrc = w + ((long)co * Ci + ci) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = mode == 0 ? gsrc[3 * a + b] : gsrc[8 - (3 * a + b)];
    float r[6][3];       // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float u[6];
        wino4_g(g[0][b], g[1][b], g[2][b], u);
#pragma unroll
        for (int a = 0; a < 6; ++a) r[a][b] = u[a];
    }
    const long ps = (long)Co * Ci;
    float* out = U + idx;
#pragma unroll
    for (int a = 0; a < 6; ++a) {   // (.) G^T
        float u[6];
        wino4_g(r[a][0], r[a][1], r[a][2], u);
#pragma unroll
        for (int b = 0; b < 6; ++b) out[(6 * a + b) * ps] = u[b];
    }
}
__global__ void wino4_weight_transform_batched_kernel(const WinoWBatch p) {
    int j = 0;
    while (j + 1 < p.n && (int)blockIdx.x >= p.first[j + 1]) ++j;
    const long idx = (long)(blockIdx.x - p.first[j]) * blockDim.x + threadIdx.x;
    if (idx >= (long)p.co[j] * p.ci[j]) return;
    wino4_weight_transform_one(p.w[j], p.u[j], p.co[j], p.ci[j], p.mode[j], idx);
}
// layers[i] = {w (Cout,Cin,3,3), u_fwd or NULL, u_dgrad or NULL, Cout, Cin}: the same results as n calls of
// wesup_winograd_pack_weight(..., m = 4), in one launch; at most 32 filter sets (a NULL panel does not count)
extern "C" int wesup_winograd_pack_weights(const WesupWinoFilter* layers /* host */, int n, void* stream) {
    if (!layers || n <= 0) return WESUP_ERR_INVALID;
    WinoWBatch p = {};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const WesupWinoFilter& L = layers[i];
        if (!L.w || L.Cout <= 0 || L.Cin <= 0 || (!L.u_fwd && !L.u_dgrad)) return WESUP_ERR_INVALID;
        for (int mode = 0; mode < 2; ++mode) {
            float* u = mode == 0 ? L.u_fwd : L.u_dgrad;
            if (!u) continue;
            if (p.n >= WINO_WB_MAX) return WESUP_ERR_INVALID;
            p.w[p.n] = L.w; p.u[p.n] = u; p.co[p.n] = L.Cout; p.ci[p.n] = L.Cin; p.mode[p.n] = mode;
            p.first[p.n] = blocks;
            blocks += (int)(((long)L.Cout * L.Cin + 255) / 256);
            ++p.n;
        }
    }
    p.first[p.n] = blocks;
    WESUP_LAUNCH(wino4_weight_transform_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

static WinoGeom wino_geom(int B, int H, int W, int C, int m, long plane_elems = 0) {
    WinoGeom g = {};
    g.H = H; g.W = W; g.Th = (H + m - 1) / m; g.Tw = (W + m - 1) / m; g.T = wino_tiles(B, H, W, m);
    g.ps = plane_elems > 0 ? plane_elems : g.T * C;
    g.dTw = make_fastdiv(g.Tw); g.dTh = make_fastdiv(g.Th);
    g.C = C; g.dQ = make_fastdiv(C / 4);
    return g;
}

// x (B,H,W,C) -> V [P][tiles][C]
static int wino_input_launch(const float* x, float* V, long plane_elems, unsigned char* bits, int B, int H, int W, int C,
                             int relu_in, int m, void* stream) {
    if (!x || !V || (bits && m != 4) || !wino_shape_ok(B, H, W, C, C, m) || (((uintptr_t)x | (uintptr_t)V) & 15) || (plane_elems % 4) ||
        (plane_elems > 0 && plane_elems < wino_tiles(B, H, W, m) * C))
        return WESUP_ERR_INVALID;
    const WinoGeom g = wino_geom(B, H, W, C, m, plane_elems);
    const dim3 grid((unsigned)ceil_div(g.T * (C / 4), 256l));
    if (m == 2) WESUP_LAUNCH(wino_input_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, V, g, relu_in);
    else {
        WinoGeom g2 = g;
        g2.nt = wino_nt_stores(4.0 * 36 * g.T * C);
        WESUP_LAUNCH(wino4_input_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, V, g2, relu_in, bits);
    }
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_winograd_input_transform(const float* x, float* V, long plane_elems, int B, int H, int W, int C,
                                              int relu_in, int m, void* stream) {
    return wino_input_launch(x, V, plane_elems, nullptr, B, H, W, C, relu_in, m, stream);
}
// ... F(4x4), also leaving the sign bits of x (one byte per pixel and channel quad: bit j = x[..., 4 q + j] > 0) in
// relu_bits [B][H][W][C/4]: what the input gradient of the layer that produced x needs of it (its ReLU mask)
extern "C" int wesup_winograd_input_transform_bits(const float* x, float* V, long plane_elems, unsigned char* relu_bits, int B,
                                                   int H, int W, int C, int relu_in, void* stream) {
    if (!relu_bits) return WESUP_ERR_INVALID;
    return wino_input_launch(x, V, plane_elems, relu_bits, B, H, W, C, relu_in, 4, stream);
}
// Mt [P][tiles][C] -> y (B,H,W,C) = A^T M A + bias, masked by mask_src > 0, added to the old y if accumulate;
// y_relu: optional second output max(y, 0);  y_pool: optional third output (B,H/2,W/2,C) = the 2x2 / stride-2 max-pool
// of y (an output tile is one pooling window for m = 2, four for m = 4), ReLU'd if pool_relu
static int wino_output_launch(const float* Mt, long plane_elems, const float* bias, const float* mask_src, float* y,
                              float* y_relu, float* y_pool, int pool_relu, int B, int H, int W, int C, int accumulate, int m,
                              const WinoUnpool& up, void* stream) {
    if (!Mt || (!y && !up.src) || !wino_shape_ok(B, H, W, C, C, m) || (plane_elems % 4) ||
        (plane_elems > 0 && plane_elems < wino_tiles(B, H, W, m) * C) ||
        (((uintptr_t)Mt | (uintptr_t)y | (uintptr_t)y_relu | (uintptr_t)y_pool | (uintptr_t)mask_src | (uintptr_t)bias |
          (uintptr_t)up.src | (uintptr_t)up.dst) & 15))
        return WESUP_ERR_INVALID;
    // the unpooling epilogue: F(4x4) only, in place of the plain store (no second / pooled output, no accumulate into y);
    // the pooled map is floor(Hu / 2) x floor(Wu / 2)
    if (up.src && (m != 4 || !up.dst || y_relu || y_pool || accumulate || up.Hu / 2 != H || up.Wu / 2 != W)) return WESUP_ERR_INVALID;
    const WinoGeom g = wino_geom(B, H, W, C, m, plane_elems);
    const dim3 grid((unsigned)ceil_div(g.T * (C / 4), 256l));
    if (m == 2)
        WESUP_LAUNCH(wino_output_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, Mt, bias, mask_src, y, y_relu,
                           y_pool, pool_relu, g, accumulate);
    else
        WESUP_LAUNCH(wino4_output_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, Mt, bias, mask_src, y, y_relu,
                           y_pool, pool_relu, g, accumulate, up);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_winograd_output_transform(const float* Mt, long plane_elems, const float* bias, const float* mask_src,
                                               float* y, float* y_relu, float* y_pool, int pool_relu, int B, int H, int W,
                                               int C, int accumulate, int m, void* stream) {
    return wino_output_launch(Mt, plane_elems, bias, mask_src, y, y_relu, y_pool, pool_relu, B, H, W, C, accumulate, m,
                              WinoUnpool{nullptr, nullptr, 0, 0}, stream);
}
// The same transform with the max-pool backward as its epilogue (m = 4): Mt -> the input gradient at pooled resolution
// (B,H,W,C), each value added to unpool_dst (B,Hu,Wu,C) at the first positive maximum of its 2x2 window of unpool_src;
// H == Hu / 2, W == Wu / 2.  Replaces wesup_winograd_output_transform + wesup_maxpool2_bwd(accumulate) for the layers
// behind a pooling step; mask_src (optional) applies to the pooled-resolution gradient first.
extern "C" int wesup_winograd_output_transform_unpool(const float* Mt, long plane_elems, const float* bias,
                                                      const float* mask_src, const float* unpool_src, float* unpool_dst,
                                                      int B, int H, int W, int Hu, int Wu, int C, int m, void* stream) {
    if (!unpool_src || !unpool_dst) return WESUP_ERR_INVALID;
    return wino_output_launch(Mt, plane_elems, bias, mask_src, nullptr, nullptr, nullptr, 0, B, H, W, C, 0, m,
                              WinoUnpool{unpool_src, unpool_dst, Hu, Wu}, stream);
}
// dy (B,H,W,C) -> dM [P][tiles][C] = A dY A^T per m x m tile (the weight gradient's second operand).
// db (optional, m = 4 only, [C]): the bias gradient sum_pixels dy, from the values the transform loads anyway (m = 2 takes it
// from the TN GEMM: the column sums of position (1,1)).  Workspace: per-block partial sums + wesup_colsum's own.
static bool wino4_block_colsum_ok(int C) { const int Q = C / 4; return Q <= 256 && 256 % Q == 0; }
static long wino_transform_blocks(int B, int H, int W, int C, int m) { return ceil_div(wino_tiles(B, H, W, m) * (C / 4), 256l); }
// internal (winograd.hpp): rows of per-block column sums the F(4x4) outgrad transform leaves for a bias gradient
long wino4_bias_rows(int B, int H, int W, int C) { return wino4_block_colsum_ok(C) ? wino_transform_blocks(B, H, W, C, 4) : 0; }

extern "C" size_t wesup_winograd_outgrad_workspace_bytes(int B, int H, int W, int C, int m) {
    if (m != 4 || !wino_shape_ok(B, H, W, C, C, m)) return 0;
    if (!wino4_block_colsum_ok(C)) return wesup_colsum_workspace_bytes(B * H * W, C);
    const long blocks = wino_transform_blocks(B, H, W, C, m);
    return align_up((size_t)blocks * C * sizeof(float), 256) + wesup_colsum_workspace_bytes((int)blocks, C);
}
// internal (winograd.hpp): the transform with the per-block sums written to bias_part ([wino4_bias_rows][C], m = 4) or not
int wino_outgrad_launch(const float* dy, float* dM, float* bias_part, int B, int H, int W, int C, int m, void* stream) {
    WinoGeom g = wino_geom(B, H, W, C, m);
    g.nt = m == 4 ? wino_nt_stores(4.0 * 36 * g.T * C) : 0;
    const dim3 grid((unsigned)wino_transform_blocks(B, H, W, C, m));
    if (m == 2) WESUP_LAUNCH(wino_outgrad_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dM, g);
    else WESUP_LAUNCH(wino4_outgrad_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dM, g, bias_part);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// rows of per-block column sums the F(4x4) gradient transforms leave for the bias gradient (0: C / 4 does not divide 256)
extern "C" long wesup_winograd_bias_rows(int B, int H, int W, int C) {
    return wino_shape_ok(B, H, W, C, C, 4) ? wino4_bias_rows(B, H, W, C) : 0;
}
// dy (B,H,W,C) -> V [36][tiles][C] (input transform, no ReLU) AND dM [36][tiles][C] (outgrad transform) in one launch (m = 4);
// bias_part (optional, [wesup_winograd_bias_rows][C]): the per-block column sums of dy for wesup_conv3x3_wgrad_winograd_pre.
extern "C" int wesup_winograd_dual_transform(const float* dy, float* V, float* dM, float* bias_part, int B, int H, int W, int C,
                                             void* stream) {
    if (!dy || !V || !dM || !wino_shape_ok(B, H, W, C, C, 4) || (((uintptr_t)dy | (uintptr_t)V | (uintptr_t)dM | (uintptr_t)bias_part) & 15) ||
        (bias_part && !wino4_block_colsum_ok(C)))
        return WESUP_ERR_INVALID;
    WinoGeom g = wino_geom(B, H, W, C, 4);
    g.nt = wino_nt_stores(2 * 4.0 * 36 * g.T * C);      // (two transformed tensors per launch)
    const dim3 grid((unsigned)wino_transform_blocks(B, H, W, C, 4));
    WESUP_LAUNCH(wino4_dual_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, V, dM, g, bias_part);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_winograd_outgrad_transform(const float* dy, float* dM, float* db, int B, int H, int W, int C, int m,
                                                void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !dM || !wino_shape_ok(B, H, W, C, C, m) || (((uintptr_t)dy | (uintptr_t)dM | (uintptr_t)ws) & 15))
        return WESUP_ERR_INVALID;
    if (db && m != 4) return WESUP_ERR_INVALID;
    if (db && (!ws || ws_bytes < wesup_winograd_outgrad_workspace_bytes(B, H, W, C, m))) return WESUP_ERR_WORKSPACE;
    const bool in_kernel = db && wino4_block_colsum_ok(C);
    int rc = wino_outgrad_launch(dy, dM, in_kernel ? (float*)ws : nullptr, B, H, W, C, m, stream);
    if (rc) return rc;
    if (in_kernel) {
        const long blocks = wino_transform_blocks(B, H, W, C, m);
        const size_t pb = align_up((size_t)blocks * C * sizeof(float), 256);
        return wesup_colsum((const float*)ws, C, db, (int)blocks, C, (char*)ws + pb, ws_bytes - pb, stream);
    }
    if (db) return wesup_colsum(dy, C, db, B * H * W, C, ws, ws_bytes, stream);
    return WESUP_OK;
}
// internal (winograd.hpp): wesup_winograd_filter_grad with the F(4x4) bias rows of wino_outgrad_launch
int wino_filter_grad_launch(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs, float* db, int Cout,
                            int Cin, int m, const float* bias_part, int bias_rows, void* stream) {
    const long tot = (long)Cout * Cin;
    const bool narrow = m == 4 && tot / 64 < 512;        // few pairs: 16 per block
    const bool vec = m == 4 && tot % 4 == 0 && slab_stride % 4 == 0 && batch_stride % 4 == 0 && !((uintptr_t)slabs & 15);
    const int per_block = (narrow ? 16 : 64) * (vec ? 4 : 1);
    const int pair_blocks = (int)((tot + per_block - 1) / per_block);
    const int bias_blocks = !db ? 0 : bias_part ? (Cout + 15) / 16 : (Cout + 255) / 256;
    const dim3 grid((unsigned)(pair_blocks + bias_blocks));
    if (vec && narrow)
        WESUP_LAUNCH((wino_wgrad_reduce_kernel<6, 16, true>), grid, dim3(256), 0, (hipStream_t)stream, slabs, slab_stride,
                           batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks, bias_part, bias_rows);
    else if (vec)
        WESUP_LAUNCH((wino_wgrad_reduce_kernel<6, 64, true>), grid, dim3(256), 0, (hipStream_t)stream, slabs, slab_stride,
                           batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks, bias_part, bias_rows);
    else if (m == 2)
        WESUP_LAUNCH((wino_wgrad_reduce_kernel<4, 64>), grid, dim3(256), 0, (hipStream_t)stream, slabs, slab_stride,
                           batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks, (const float*)nullptr, 0);
    else if (narrow)
        WESUP_LAUNCH((wino_wgrad_reduce_kernel<6, 16>), grid, dim3(256), 0, (hipStream_t)stream, slabs, slab_stride,
                           batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks, bias_part, bias_rows);
    else
        WESUP_LAUNCH((wino_wgrad_reduce_kernel<6, 64>), grid, dim3(256), 0, (hipStream_t)stream, slabs, slab_stride,
                           batch_stride, dw_kcrs, Cout, Cin, S, db, pair_blocks, bias_part, bias_rows);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// slabs [P][S][Cout*Cin + Cout] (split-K partial products of dU_p, each followed by Cout column sums of dM_p) ->
// dw (Cout,Cin,3,3) = G^T (sum over S) G; db (Cout, m = 2 only, else NULL) = the column sums of position (1,1) (index 5).
// slab_stride = elements between two splits, batch_stride = between two positions.
extern "C" int wesup_winograd_filter_grad(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs,
                                          float* db, int Cout, int Cin, int m, void* stream) {
    if (!slabs || !dw_kcrs || S <= 0 || Cout <= 0 || Cin <= 0 || slab_stride < (long)Cout * Cin + Cout ||
        batch_stride < (long)S * slab_stride || !wino_m_ok(m) || (db && m != 2))
        return WESUP_ERR_INVALID;
    return wino_filter_grad_launch(slabs, slab_stride, batch_stride, S, dw_kcrs, db, Cout, Cin, m, nullptr, 0, stream);
}
