// GPU SLIC superpixels: replaces the CPU skimage.segmentation.slic call of
// WESUPTrainer.preprocess (models/wesup.py:471-476; parameters sp_area 200 -> n_segments = H*W/200,
// compactness 40, skimage defaults max_num_iter 10, sigma 0, convert2lab, enforce_connectivity with
// min_size_factor 0.5).  skimage is a third-party dependency that is absent from /root/reference and unpinned
// (requirements.txt:9): PARITY IS UNPINNED.  This file restates the published algorithm (Achanta et al., SLIC,
// TPAMI 2012) with skimage's parameterisation:
//   1. sRGB -> CIE Lab (D65), colours divided by the compactness m;
//   2. cluster centres on a regular grid: step = round(sqrt(HW/n)), first centre at floor(sqrt(HW/n)/2);
//   3. max_iter rounds of { every pixel takes the nearest centre among those whose 2S x 2S window holds it,
//      D^2 = |lab/m - c_lab/m|^2 + (dy^2 + dx^2)/S^2, ties to the lower centre index; centres move to the mean of
//      their members };  sums are 64-bit fixed point, so the result does not depend on the order of the atomics;
//   4. connectivity: 4-connected components of the label image by lock-free union-find (root = first pixel in
//      raster order); components smaller than min_size_factor * HW/n are absorbed by the component of the pixel
//      left of (else above) their first pixel; surviving components are renumbered 0..K-1 in raster order of their
//      first pixel -- exactly the contiguous 0-based ids _preprocess_superpixels needs (models/wesup.py:41).
// Everything is integer-deterministic except the float distance comparison itself.
#include "common.hpp"

#define SLIC_FIX 1048576.0   // 2^20 fixed-point scale of the centre sums

__device__ __forceinline__ float srgb_to_linear(float c) {
    return c > 0.04045f ? powf((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}
__device__ __forceinline__ float lab_f(float t) {
    return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f;
}
// img [B][3][HW] in [0,1] -> lab [B][HW] float4 {L, a, b, 0} * (1/compactness)
__global__ void slic_lab_kernel(const float* __restrict__ img, float4* __restrict__ lab, int B, long HW, float inv_m) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW, p = idx - b * HW;
    const float* s = img + b * 3 * HW + p;
    const float r = srgb_to_linear(s[0]), g = srgb_to_linear(s[HW]), bl = srgb_to_linear(s[2 * HW]);
    float x = 0.412453f * r + 0.357580f * g + 0.180423f * bl;
    float y = 0.212671f * r + 0.715160f * g + 0.072169f * bl;
    float z = 0.019334f * r + 0.119193f * g + 0.950227f * bl;
    x /= 0.95047f;
    z /= 1.08883f;
    const float fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
    lab[idx] = make_float4((116.f * fy - 16.f) * inv_m, 500.f * (fx - fy) * inv_m, 200.f * (fy - fz) * inv_m, 0.f);
}

struct SlicGrid {
    int gy, gx, step, start, H, W;
};

// centres [B][Kc][5] = {y, x, L, a, b}
__global__ void slic_init_kernel(const float4* __restrict__ lab, float* __restrict__ cen, SlicGrid g, int B) {
    const int Kc = g.gy * g.gx;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Kc) return;
    const int b = idx / Kc, k = idx - b * Kc;
    const int y = g.start + (k / g.gx) * g.step, x = g.start + (k % g.gx) * g.step;
    const float4 c = lab[(long)b * g.H * g.W + (long)y * g.W + x];
    float* o = cen + (long)idx * 5;
    o[0] = (float)y; o[1] = (float)x; o[2] = c.x; o[3] = c.y; o[4] = c.z;
}

// nearest centre among the 7x7 grid neighbourhood whose 2S window contains the pixel; accumulates fixed-point sums
__global__ void slic_assign_kernel(const float4* __restrict__ lab, const float* __restrict__ cen,
                                   int32_t* __restrict__ label, long long* __restrict__ sums, SlicGrid g, int B,
                                   int accumulate) {
    const long HW = (long)g.H * g.W;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW;
    const int p = idx - (long)b * HW;
    const int y = p / g.W, x = p - y * g.W;
    const int Kc = g.gy * g.gx;
    const float4 c = lab[idx];
    const int cy = min(max((y - g.start + g.step / 2) / g.step, 0), g.gy - 1);
    const int cx = min(max((x - g.start + g.step / 2) / g.step, 0), g.gx - 1);
    const float S = (float)g.step, inv_s2 = 1.f / (S * S), win = 2.f * S;
    float best = 3.0e38f;
    int bk = -1;
    for (int dy = -3; dy <= 3; ++dy) {          // 7x7 grid cells: a centre within 2S of the pixel is at most 3 cells away
        const int ky = cy + dy;
        if (ky < 0 || ky >= g.gy) continue;
        for (int dx = -3; dx <= 3; ++dx) {
            const int kx = cx + dx;
            if (kx < 0 || kx >= g.gx) continue;
            const int k = ky * g.gx + kx;
            const float* q = cen + ((long)b * Kc + k) * 5;
            const float ddy = (float)y - q[0], ddx = (float)x - q[1];
            if (fabsf(ddy) > win || fabsf(ddx) > win) continue;
            const float dl = c.x - q[2], da = c.y - q[3], db = c.z - q[4];
            const float d = (dl * dl + da * da + db * db) + (ddy * ddy + ddx * ddx) * inv_s2;
            if (d < best) { best = d; bk = k; }          // ascending k: ties keep the lower centre index
        }
    }
    if (bk < 0) bk = cy * g.gx + cx;
    label[idx] = bk;
    if (accumulate) {
        long long* s = sums + ((long)b * Kc + bk) * 6;
        atomicAdd((unsigned long long*)&s[0], (unsigned long long)(long long)y);
        atomicAdd((unsigned long long*)&s[1], (unsigned long long)(long long)x);
        atomicAdd((unsigned long long*)&s[2], (unsigned long long)(long long)llrint((double)c.x * SLIC_FIX));
        atomicAdd((unsigned long long*)&s[3], (unsigned long long)(long long)llrint((double)c.y * SLIC_FIX));
        atomicAdd((unsigned long long*)&s[4], (unsigned long long)(long long)llrint((double)c.z * SLIC_FIX));
        atomicAdd((unsigned long long*)&s[5], 1ull);
    }
}
// The same assignment, one block per 16 x 16 pixel tile (round 4).  The per-pixel form above sends six 64-bit atomics per pixel
// to HBM-side memory (55 M of them per segmentation of a 4 x 480 x 480 batch: 4.3 ms).  A tile's pixels can only choose among
// the centres of the grid cells within three cells of the tile (<= 13 x 13 for a grid step >= 3): those centres are staged in
// LDS, the pixels accumulate their fixed-point contributions there with LDS atomics, and the block adds one value per
// (centre, field) it touched to the global sums -- integer sums, so the result is bit-identical to the per-pixel form whatever
// the grouping.  Candidates are visited in the same ascending-centre order (ties keep the lower index).
#define SLIC_TILE 16
#define SLIC_MAXC 176
__global__ __launch_bounds__(256) void slic_assign_tile_kernel(const float4* __restrict__ lab, const float* __restrict__ cen,
                                                                int32_t* __restrict__ label, long long* __restrict__ sums,
                                                                const SlicGrid g, int accumulate) {
    __shared__ float scen[SLIC_MAXC][5];
    __shared__ unsigned long long ssum[SLIC_MAXC][6];
    const int b = blockIdx.z, tid = threadIdx.x;
    const int ty0 = blockIdx.y * SLIC_TILE, tx0 = blockIdx.x * SLIC_TILE;
    const int ty1 = min(ty0 + SLIC_TILE, g.H) - 1, tx1 = min(tx0 + SLIC_TILE, g.W) - 1;
    const int Kc = g.gy * g.gx;
    auto cell = [&](int v, int n) { return min(max((v - g.start + g.step / 2) / g.step, 0), n - 1); };
    const int cy_lo = max(cell(ty0, g.gy) - 3, 0), cy_hi = min(cell(ty1, g.gy) + 3, g.gy - 1);
    const int cx_lo = max(cell(tx0, g.gx) - 3, 0), cx_hi = min(cell(tx1, g.gx) + 3, g.gx - 1);
    const int ncx = cx_hi - cx_lo + 1, nc = (cy_hi - cy_lo + 1) * ncx;      // <= SLIC_MAXC (the host checked the step)
    for (int i = tid; i < nc * 5; i += 256) {
        const int c = i / 5, f = i - c * 5;
        const int k = (cy_lo + c / ncx) * g.gx + cx_lo + c % ncx;
        scen[c][f] = cen[((long)b * Kc + k) * 5 + f];
    }
    for (int i = tid; i < nc * 6; i += 256) ssum[i / 6][i % 6] = 0ull;
    __syncthreads();
    const int y = ty0 + (tid >> 4), x = tx0 + (tid & 15);
    if (y < g.H && x < g.W) {
        const long idx = ((long)b * g.H + y) * g.W + x;
        const float4 c = lab[idx];
        const int cy = cell(y, g.gy), cx = cell(x, g.gx);
        const float S = (float)g.step, inv_s2 = 1.f / (S * S), win = 2.f * S;
        float best = 3.0e38f;
        int bl = -1;
        for (int dy = -3; dy <= 3; ++dy) {
            const int ky = cy + dy;
            if (ky < 0 || ky >= g.gy) continue;
            for (int dx = -3; dx <= 3; ++dx) {
                const int kx = cx + dx;
                if (kx < 0 || kx >= g.gx) continue;
                const int l = (ky - cy_lo) * ncx + (kx - cx_lo);
                const float* q = scen[l];
                const float ddy = (float)y - q[0], ddx = (float)x - q[1];
                if (fabsf(ddy) > win || fabsf(ddx) > win) continue;
                const float dl = c.x - q[2], da = c.y - q[3], db = c.z - q[4];
                const float d = (dl * dl + da * da + db * db) + (ddy * ddy + ddx * ddx) * inv_s2;
                if (d < best) { best = d; bl = l; }
            }
        }
        if (bl < 0) bl = (cy - cy_lo) * ncx + (cx - cx_lo);
        label[idx] = (cy_lo + bl / ncx) * g.gx + cx_lo + bl % ncx;
        if (accumulate) {
            atomicAdd(&ssum[bl][0], (unsigned long long)(long long)y);
            atomicAdd(&ssum[bl][1], (unsigned long long)(long long)x);
            atomicAdd(&ssum[bl][2], (unsigned long long)(long long)llrint((double)c.x * SLIC_FIX));
            atomicAdd(&ssum[bl][3], (unsigned long long)(long long)llrint((double)c.y * SLIC_FIX));
            atomicAdd(&ssum[bl][4], (unsigned long long)(long long)llrint((double)c.z * SLIC_FIX));
            atomicAdd(&ssum[bl][5], 1ull);
        }
    }
    if (!accumulate) return;
    __syncthreads();
    for (int i = tid; i < nc * 6; i += 256) {
        const int c = i / 6, f = i - c * 6;
        if (ssum[c][5] == 0ull) continue;                    // no pixel of this tile chose that centre
        const int k = (cy_lo + c / ncx) * g.gx + cx_lo + c % ncx;
        atomicAdd((unsigned long long*)&sums[((long)b * Kc + k) * 6 + f], ssum[c][f]);
    }
}
__global__ void slic_update_kernel(float* __restrict__ cen, long long* __restrict__ sums, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long long* s = sums + (long)idx * 6;
    const long long n = s[5];
    if (n > 0) {
        float* o = cen + (long)idx * 5;
        const double inv = 1.0 / (double)n;
        o[0] = (float)((double)s[0] * inv);
        o[1] = (float)((double)s[1] * inv);
        o[2] = (float)((double)s[2] * inv / SLIC_FIX);
        o[3] = (float)((double)s[3] * inv / SLIC_FIX);
        o[4] = (float)((double)s[4] * inv / SLIC_FIX);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = 0;
}

// ---- connected components (4-connectivity) by union-find; parent[] holds pixel indices of one image
__device__ __forceinline__ int uf_find(int* parent, int a) {
    int r = a;
    while (true) {
        const int pr = parent[r];
        if (pr == r) break;
        r = pr;
    }
    return r;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    while (a != b) {
        if (a < b) { const int t = a; a = b; b = t; }      // a > b: hang the larger root under the smaller
        const int old = atomicMin(&parent[a], b);
        if (old == a) break;                                // a was a root and now points to b
        a = uf_find(parent, old);                            // somebody else re-parented a: continue from there
        b = uf_find(parent, b);
    }
}
__global__ void ccl_init_kernel(int* __restrict__ parent, long total, long HW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) parent[idx] = (int)(idx % HW);
}
__global__ void ccl_merge_kernel(const int32_t* __restrict__ label, int* __restrict__ parent, int H, int W, int B) {
    const long HW = (long)H * W;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW;
    const int p = idx - (long)b * HW;
    const int y = p / W, x = p - y * W;
    const int32_t* lab = label + (long)b * HW;
    int* par = parent + (long)b * HW;
    const int l = lab[p];
    if (x + 1 < W && lab[p + 1] == l) uf_union(par, p, p + 1);
    if (y + 1 < H && lab[p + W] == l) uf_union(par, p, p + W);
}
// flatten + component sizes
__global__ void ccl_flatten_kernel(int* __restrict__ parent, int32_t* __restrict__ size, long HW, int B) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW;
    int* par = parent + b * HW;
    const int r = uf_find(par, (int)(idx - b * HW));
    par[idx - b * HW] = r;
    atomicAdd(&size[b * HW + r], 1);
}
// roots of small components point to the (flattened) root of the pixel left of / above them
__global__ void ccl_absorb_kernel(const int* __restrict__ parent, const int32_t* __restrict__ size,
                                  int32_t* __restrict__ target, int W, long HW, int B, int min_size) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW;
    const int p = (int)(idx - b * HW);
    const int* par = parent + b * HW;
    int t = p;
    if (par[p] == p && size[idx] < min_size) {
        const int x = p % W;
        if (x > 0) t = par[p - 1];
        else if (p >= W) t = par[p - W];
    }
    target[idx] = t;            // for non-roots and large roots: itself
}
// final root of a pixel: follow parent then the absorb chain (targets are strictly earlier pixels: terminates);
// flag surviving roots for the compaction scan
__global__ void ccl_resolve_kernel(const int* __restrict__ parent, const int32_t* __restrict__ target,
                                   int32_t* __restrict__ final_root, int32_t* __restrict__ is_root, long HW, int B) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW;
    const int p = (int)(idx - b * HW);
    int r = parent[idx];
    while (true) {
        const int t = target[b * HW + r];
        if (t == r) break;
        r = t;
    }
    final_root[idx] = r;
    is_root[idx] = (r == p) ? 1 : 0;
}
// exclusive scan of is_root over each image (three small kernels: per-block sums, scan of block sums, apply)
#define SCAN_BLOCK 1024
__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_sums(const int32_t* __restrict__ flags, int32_t* __restrict__ bsum,
                                                              long HW, int nblk) {
    __shared__ int sh[SCAN_BLOCK];
    const int b = blockIdx.y, blk = blockIdx.x;
    const long p = (long)blk * SCAN_BLOCK + threadIdx.x;
    sh[threadIdx.x] = (p < HW) ? flags[(long)b * HW + p] : 0;
    __syncthreads();
    for (int off = SCAN_BLOCK / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[(long)b * nblk + blk] = sh[0];
}
__global__ void scan_of_block_sums(int32_t* __restrict__ bsum, int32_t* __restrict__ n_labels, int nblk) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < nblk; ++i) {
            const int v = bsum[(long)b * nblk + i];
            bsum[(long)b * nblk + i] = run;
            run += v;
        }
        n_labels[b] = run;
    }
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const int32_t* __restrict__ flags, const int32_t* __restrict__ bsum,
                                                         int32_t* __restrict__ newid, long HW, int nblk) {
    __shared__ int sh[SCAN_BLOCK];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const long p = (long)blk * SCAN_BLOCK + tid;
    const int v = (p < HW) ? flags[(long)b * HW + p] : 0;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        const int t = (tid >= off) ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    if (p < HW) newid[(long)b * HW + p] = bsum[(long)b * nblk + blk] + sh[tid] - v;     // exclusive
}
__global__ void relabel_kernel(const int32_t* __restrict__ final_root, const int32_t* __restrict__ newid,
                               int32_t* __restrict__ labels, long HW, int B) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const long b = idx / HW;
    labels[idx] = newid[b * HW + final_root[idx]];
}

static SlicGrid make_grid(int H, int W, int n_segments) {
    SlicGrid g;
    g.H = H; g.W = W;
    const double s = sqrt((double)H * W / (double)(n_segments > 0 ? n_segments : 1));
    g.step = (int)floor(s + 0.5);
    if (g.step < 1) g.step = 1;
    g.start = (int)floor(s / 2.0);
    if (g.start >= H) g.start = H - 1;
    if (g.start >= W) g.start = W - 1;
    g.gy = (H - 1 - g.start) / g.step + 1;
    g.gx = (W - 1 - g.start) / g.step + 1;
    return g;
}

extern "C" int wesup_slic_num_centers(int H, int W, int n_segments) {
    if (H <= 0 || W <= 0 || n_segments <= 0) return 0;
    const SlicGrid g = make_grid(H, W, n_segments);
    return g.gy * g.gx;
}
extern "C" size_t wesup_slic_workspace_bytes(int B, int H, int W, int n_segments) {
    if (B <= 0 || H <= 0 || W <= 0 || n_segments <= 0) return 0;
    const size_t HW = (size_t)H * W;
    const size_t Kc = (size_t)wesup_slic_num_centers(H, W, n_segments);
    const size_t nblk = (HW + SCAN_BLOCK - 1) / SCAN_BLOCK;
    size_t b = 0;
    b += align_up(B * HW * 16, 256);            // lab
    b += align_up(B * Kc * 5 * 4, 256);         // centres
    b += align_up(B * Kc * 6 * 8, 256);         // fixed-point sums
    b += 6 * align_up(B * HW * 4, 256);         // centre label, parent, size, target, final_root / is_root, newid
    b += align_up(B * nblk * 4, 256);           // block sums
    return b;
}

extern "C" int wesup_slic(const float* img_nchw, int32_t* labels, int32_t* n_labels, int B, int H, int W, int n_segments,
                          float compactness, int max_iter, int enforce_connectivity, float min_size_factor, void* ws,
                          size_t ws_bytes, void* stream) {
    if (!img_nchw || !labels || !n_labels || !ws || B <= 0 || H <= 0 || W <= 0 || n_segments <= 0 || compactness <= 0.f ||
        max_iter < 1)
        return WESUP_ERR_INVALID;
    if ((long)H * W >= (1l << 30)) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_slic_workspace_bytes(B, H, W, n_segments)) return WESUP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const SlicGrid g = make_grid(H, W, n_segments);
    const long HW = (long)H * W;
    const int Kc = g.gy * g.gx;
    const int nblk = (int)((HW + SCAN_BLOCK - 1) / SCAN_BLOCK);
    char* w = (char*)ws;
    float4* lab = (float4*)w;            w += align_up((size_t)B * HW * 16, 256);
    float* cen = (float*)w;              w += align_up((size_t)B * Kc * 5 * 4, 256);
    long long* sums = (long long*)w;     w += align_up((size_t)B * Kc * 6 * 8, 256);
    const size_t plane = align_up((size_t)B * HW * 4, 256);
    int32_t* clabel = (int32_t*)w;       w += plane;
    int* parent = (int*)w;               w += plane;
    int32_t* size = (int32_t*)w;         w += plane;
    int32_t* target = (int32_t*)w;       w += plane;
    int32_t* final_root = (int32_t*)w;   w += plane;
    int32_t* newid = (int32_t*)w;        w += plane;
    int32_t* bsum = (int32_t*)w;
    int32_t* is_root = size;             // size[] is dead once targets are known

    const long tot = (long)B * HW;
    const unsigned pb = (unsigned)((tot + 255) / 256);
    WESUP_LAUNCH(slic_lab_kernel, dim3(pb), dim3(256), 0, st, img_nchw, lab, B, HW, 1.f / compactness);
    WESUP_LAUNCH(slic_init_kernel, dim3(ceil_div(B * Kc, 256)), dim3(256), 0, st, lab, cen, g, B);
    if (wesup_fill_words_(sums, 0u, ((size_t)B * Kc * 6 * 8) / 4, st) != WESUP_OK) return WESUP_ERR_LAUNCH;
    // tile form when the candidate centres of a 16 x 16 tile fit its LDS table: (15 / step + 2 + 6)^2 <= SLIC_MAXC
    const int span = (SLIC_TILE - 1) / g.step + 2 + 6;
    const bool tiled = span * span <= SLIC_MAXC && B <= 65535;
    const dim3 tgrid(ceil_div(W, SLIC_TILE), ceil_div(H, SLIC_TILE), B);
    for (int it = 0; it < max_iter; ++it) {
        if (tiled) WESUP_LAUNCH(slic_assign_tile_kernel, tgrid, dim3(256), 0, st, lab, cen, clabel, sums, g, 1);
        else WESUP_LAUNCH(slic_assign_kernel, dim3(pb), dim3(256), 0, st, lab, cen, clabel, sums, g, B, 1);
        WESUP_LAUNCH(slic_update_kernel, dim3(ceil_div(B * Kc, 256)), dim3(256), 0, st, cen, sums, B * Kc);
    }
    if (!enforce_connectivity) {
        // raw k-means labels: ids are centre indices (not necessarily all used)
        if (hipMemcpyAsync(labels, clabel, (size_t)tot * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return WESUP_ERR_LAUNCH;
        WESUP_CHECK_LAUNCH();
        return WESUP_OK;
    }
    const int min_size = (int)(min_size_factor * (float)HW / (float)n_segments);
    WESUP_LAUNCH(ccl_init_kernel, dim3(pb), dim3(256), 0, st, parent, tot, HW);
    if (wesup_fill_words_(size, 0u, ((size_t)tot * 4) / 4, st) != WESUP_OK) return WESUP_ERR_LAUNCH;
    WESUP_LAUNCH(ccl_merge_kernel, dim3(pb), dim3(256), 0, st, clabel, parent, H, W, B);
    WESUP_LAUNCH(ccl_flatten_kernel, dim3(pb), dim3(256), 0, st, parent, size, HW, B);
    WESUP_LAUNCH(ccl_absorb_kernel, dim3(pb), dim3(256), 0, st, parent, size, target, W, HW, B, min_size);
    WESUP_LAUNCH(ccl_resolve_kernel, dim3(pb), dim3(256), 0, st, parent, target, final_root, is_root, HW, B);
    WESUP_LAUNCH(scan_block_sums, dim3(nblk, B), dim3(SCAN_BLOCK), 0, st, is_root, bsum, HW, nblk);
    WESUP_LAUNCH(scan_of_block_sums, dim3(B), dim3(64), 0, st, bsum, n_labels, nblk);
    WESUP_LAUNCH(scan_apply, dim3(nblk, B), dim3(SCAN_BLOCK), 0, st, is_root, bsum, newid, HW, nblk);
    WESUP_LAUNCH(relabel_kernel, dim3(pb), dim3(256), 0, st, final_root, newid, labels, HW, B);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
