// Superpixel kernels: preprocessing from the label map (histograms, reference ordering, a stable counting
// sort of the pixels by superpixel), the scatter-mean pooling forward (row gather, HBM-bound) and backward,
// paint-back, and the dense-sp_maps compatibility argmax.
//
// Feature maps are PIXEL-major ([B][HW][ldf], 2112 fp32 channels contiguous per pixel), so the scatter-mean
// of models/wesup.py:283-285 becomes a gather of whole 8448-byte rows through the sorted pixel list: every
// load instruction of a wave is 1 KiB contiguous, there are no atomics, and the summation order is fixed
// (ascending pixel index), so results are bitwise reproducible.
#include "common.hpp"

#define SP_CHUNK 2048            // pixels per counting-sort chunk
#define SP_MAX_K 16384           // LDS histogram capacity (ids per image)
#define SP_SEG 512               // pixels per pooling segment (see the scatter-mean kernels)

// ------------------------------------------------------------------ preprocessing: four launches, every phase parallel
// (Round 4 ran this as two launches whose first ended in ONE block per image summing nchunk x Kmax x (1 + C) strided words and
// re-walking them for the sort bases: 0.23 ms in the configs[1] step, 4.4 ms at the 1024 x 1024 shard, on the stream the first
// pooling waits for -- and the class counts in LDS capped Kmax at 13 312.  Now:)
//   0. sp_zero: the per-image sums (area and class counts by id) and the per-chunk {largest id, status} words are zeroed;
//   1. sp_hist_kernel, one block per 2048-pixel chunk: histogram and class counts of the chunk in LDS (class counts as 16-bit
//      halves: a chunk has 2048 pixels), the histogram row stored for the counting sort, every non-zero entry ADDED to the
//      image's sums with integer atomics (exact in any order: the result does not depend on the arrival order);
//   2. sp_scan_order_kernel, two kinds of block in one launch, independent of each other: (a) blocks of 64 ids x 16 chunk ranges
//      turn the histogram columns into exclusive prefixes over the chunks (where chunk g starts inside the pixel list of id i),
//      all loads of a thread in flight at once; (b) one block per image does what needs all ids: the reference ordering, labels,
//      row starts, segment table -- O(Kmax / 1024) per thread, from the sums;
//   3. sp_place_kernel, one wave per chunk: places the pixels (stable: ascending pixel index inside a row).
// No hand-off between blocks inside a launch, no library-side state: any B, any number of streams.
#define SP_SCAN_COLS 64
#define SP_SCAN_SEGS 16

// block-wide exclusive scan of one int per thread (1024 threads); returns the exclusive prefix, total in *total
__device__ int block_excl_scan(int v, int* total, int* sh /*[1024]*/) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = (tid >= off) ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const int incl = sh[tid];
    *total = sh[1023];
    __syncthreads();
    return incl - v;
}

struct SpPre {
    const int32_t* labels;
    const uint8_t* mask;
    int HW, C, Kmax, nchunk, Umax;
    int nslice;              // blocks of SP_SCAN_COLS ids per image in launch 2
    int32_t* chunk_hist;     // [B][nchunk][Kmax]: histogram, then the chunk's start inside the pixel list of each id
    int32_t* chunk_info;     // [B][nchunk][2] {largest id + 1, status bits} of the chunk | zeroed by launch 0, summed by launch 1
    int32_t* cnt;            // [B][Kmax * C] class counts by id         |
    int32_t* area_old;       // [B][Kmax] pixels by id                   |
    int32_t *n_sp, *n_l, *perm, *inv_perm, *area_new, *row_start, *status, *seg_start, *unit_row;
    float* sp_labels;
};

__global__ __launch_bounds__(1024) void sp_hist_kernel(const SpPre p) {
    extern __shared__ int32_t lds[];      // [Kmax] histogram, [(Kmax * C + 1) / 2] class counts (two 16-bit counts per word)
    int32_t* hist = lds;
    int32_t* lcnt = lds + p.Kmax;
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int Kmax = p.Kmax, C = p.C, HW = p.HW;
    const bool has_mask = p.mask != nullptr;
    const int words = Kmax + (has_mask ? (Kmax * C + 1) / 2 : 0);
    for (int i = tid; i < words; i += 1024) lds[i] = 0;
    __syncthreads();
    const int p0 = g * SP_CHUNK, p1 = min(HW, p0 + SP_CHUNK);
    int lmax = 0, bad = 0;
    for (int q = p0 + tid; q < p1; q += 1024) {
        const int l = p.labels[(long)b * HW + q];
        if (l < 0 || l >= Kmax) {
            bad = 1;
            continue;
        }
        atomicAdd(&hist[l], 1);
        lmax = max(lmax, l + 1);
        if (has_mask)
            for (int c = 0; c < C; ++c)
                if (p.mask[((long)b * C + c) * HW + q]) {
                    const int e = l * C + c;
                    atomicAdd(&lcnt[e >> 1], 1 << (16 * (e & 1)));          // (a chunk has 2048 pixels: no carry into the other half)
                }
    }
    // the chunk's largest id and status: wave-level reduction, then one atomic per wave on the CHUNK's own words (sixteen waves of
    // every chunk block adding to one word per image serialised 65 K atomics on a cache line at 8 x 1024^2: 0.65 ms)
    for (int off = 32; off > 0; off >>= 1) {
        lmax = max(lmax, __shfl_xor(lmax, off));
        bad |= __shfl_xor(bad, off);
    }
    if ((tid & 63) == 0) {
        if (lmax) atomicMax(&p.chunk_info[((long)b * p.nchunk + g) * 2], lmax);
        if (bad) atomicOr(&p.chunk_info[((long)b * p.nchunk + g) * 2 + 1], 1);
    }
    __syncthreads();
    int32_t* oh = p.chunk_hist + ((long)b * p.nchunk + g) * Kmax;
    int32_t* area = p.area_old + (long)b * Kmax;
    int32_t* cnt = p.cnt + (long)b * Kmax * C;
    for (int i = tid; i < Kmax; i += 1024) {
        const int h = hist[i];
        oh[i] = h;
        if (h) {
            atomicAdd(&area[i], h);
            if (has_mask)
                for (int c = 0; c < C; ++c) {
                    const int e = i * C + c;
                    const int v = (lcnt[e >> 1] >> (16 * (e & 1))) & 0xffff;
                    if (v) atomicAdd(&cnt[e], v);
                }
        }
    }
}

// rows: labelled ids ascending, then unlabelled ids ascending (models/wesup.py:45-47); label = multi-hot of the classes whose
// pixel count equals the row maximum (models/wesup.py:50-52, integer form).  One block of 1024 threads per image.
__device__ void sp_order_image(const SpPre& p, int b, int* sh /*[1024]*/) {
    const int tid = threadIdx.x, Kmax = p.Kmax, C = p.C;
    const bool has_mask = p.mask != nullptr;
    const int32_t* cnt = p.cnt + (long)b * Kmax * C;
    const int32_t* area_old = p.area_old + (long)b * Kmax;
    // largest id and status over the chunks
    int lm = 0, stat = 0;
    for (int g = tid; g < p.nchunk; g += 1024) {
        lm = max(lm, p.chunk_info[((long)b * p.nchunk + g) * 2]);
        stat |= p.chunk_info[((long)b * p.nchunk + g) * 2 + 1];
    }
    sh[tid] = lm;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) sh[tid] = max(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    const int n = min(sh[0], Kmax);
    __syncthreads();
    sh[tid] = stat;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) sh[tid] |= sh[tid + off];
        __syncthreads();
    }
    stat = sh[0];
    __syncthreads();
    const int per = (Kmax + 1023) / 1024;
    const int i0 = min(Kmax, tid * per), i1 = min(Kmax, i0 + per);
    int nlab = 0;
    bool empty = false;
    for (int i = i0; i < min(i1, n); ++i) {
        int s = 0;
        if (has_mask)
            for (int c = 0; c < C; ++c) s += cnt[i * C + c];
        if (area_old[i] == 0) empty = true;
        if (s > 0) ++nlab;
    }
    sh[tid] = empty ? 1 : 0;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) sh[tid] |= sh[tid + off];
        __syncthreads();
    }
    if (tid == 0) p.status[b] = stat | (sh[0] ? 2 : 0);
    __syncthreads();
    int total_l;
    const int pre_l = block_excl_scan(nlab, &total_l, sh);
    // second pass: assign rows
    int rl = pre_l, ru = total_l + (min(i0, n) - pre_l);
    int asum = 0;
    for (int i = i0; i < i1; ++i) {
        int row;
        if (i < n) {
            int s = 0, mx = 0;
            if (has_mask)
                for (int c = 0; c < C; ++c) {
                    const int v = cnt[i * C + c];
                    s += v;
                    mx = max(mx, v);
                }
            if (s > 0) {
                row = rl++;
                for (int c = 0; c < C; ++c) p.sp_labels[((long)b * Kmax + row) * C + c] = (cnt[i * C + c] == mx) ? 1.f : 0.f;
            } else {
                row = ru++;
                for (int c = 0; c < C; ++c) p.sp_labels[((long)b * Kmax + row) * C + c] = 0.f;
            }
        } else {
            row = i;      // padding rows keep their place, area 0
            for (int c = 0; c < C; ++c) p.sp_labels[((long)b * Kmax + row) * C + c] = 0.f;
        }
        p.perm[(long)b * Kmax + row] = i;
        p.inv_perm[(long)b * Kmax + i] = row;
        p.area_new[(long)b * Kmax + row] = area_old[i];
    }
    if (tid == 0) {
        p.n_sp[b] = n;
        p.n_l[b] = total_l;
    }
    __syncthreads();      // (area_new of this image: written and read by this block)
    // row_start = exclusive scan of area_new over rows
    int32_t* rs = p.row_start + (long)b * (Kmax + 1);
    for (int r = i0; r < i1; ++r) asum += p.area_new[(long)b * Kmax + r];
    int tot;
    int pre = block_excl_scan(asum, &tot, sh);
    for (int r = i0; r < i1; ++r) {
        rs[r] = pre;
        pre += p.area_new[(long)b * Kmax + r];
    }
    if (tid == 1023) rs[Kmax] = tot;
    __syncthreads();
    // segment table (rows cut into <= SP_SEG-pixel segments) for the load-balanced pooling kernels
    if (p.seg_start) {
        int c = 0;
        for (int r = i0; r < i1; ++r) c += max(1, (rs[r + 1] - rs[r] + SP_SEG - 1) / SP_SEG);
        int total;
        int ps = block_excl_scan(c, &total, sh);
        for (int r = i0; r < i1; ++r) {
            const int k = max(1, (rs[r + 1] - rs[r] + SP_SEG - 1) / SP_SEG);
            p.seg_start[(long)b * (Kmax + 1) + r] = ps;
            for (int j = 0; j < k && ps + j < p.Umax; ++j) p.unit_row[(long)b * p.Umax + ps + j] = r;
            ps += k;
        }
        if (tid == 1023) p.seg_start[(long)b * (Kmax + 1) + Kmax] = min(total, p.Umax);
    }
}

__global__ __launch_bounds__(1024) void sp_scan_order_kernel(const SpPre p) {
    __shared__ int32_t sh[1024];
    const int b = blockIdx.y, tid = threadIdx.x;
    if ((int)blockIdx.x == p.nslice) {
        sp_order_image(p, b, sh);
        return;
    }
    // chunk_hist[b][g][i] -> sum over g' < g of chunk_hist[b][g'][i]: thread (i, seg) owns the chunks [seg L, (seg + 1) L) of id i
    const int Kmax = p.Kmax, nchunk = p.nchunk;
    const int col = tid & (SP_SCAN_COLS - 1), seg = tid / SP_SCAN_COLS;
    const int i = blockIdx.x * SP_SCAN_COLS + col;
    const int L = (nchunk + SP_SCAN_SEGS - 1) / SP_SCAN_SEGS;
    const int g0 = min(nchunk, seg * L), g1 = min(nchunk, g0 + L);
    int32_t* h = p.chunk_hist + (long)b * nchunk * Kmax + i;
    int sum = 0;
    if (i < Kmax) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = h[(long)(g + k) * Kmax];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; g < g1; ++g) sum += h[(long)g * Kmax];
    }
    sh[tid] = sum;
    __syncthreads();
    int run = 0;
    for (int s2 = 0; s2 < seg; ++s2) run += sh[s2 * SP_SCAN_COLS + col];
    if (i < Kmax) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = h[(long)(g + k) * Kmax];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h[(long)(g + k) * Kmax] = run;
                run += v[k];
            }
        }
        for (; g < g1; ++g) {
            const int v = h[(long)g * Kmax];
            h[(long)g * Kmax] = run;
            run += v;
        }
    }
}

// ------------------------------------------------------------------ stable placement: one wave per chunk
// The wave keeps a cursor per id in LDS, started at row_start[row of the id] + the chunk's prefix (three coalesced / gathered loads
// per id, all independent), so that the pixel loop holds no dependent global load: per 64 pixels the labels (and the rows they map
// to) are loaded one iteration ahead, the lanes of equal label rank themselves with ballots.
__global__ __launch_bounds__(64) void sp_place_kernel(const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ chunk_base,
                                                      const int32_t* __restrict__ inv_perm,
                                                      const int32_t* __restrict__ row_start, int HW, int Kmax, int nchunk,
                                                      int32_t* __restrict__ pix_sorted, int32_t* __restrict__ new_row) {
    extern __shared__ int32_t cur[];
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int32_t* base = chunk_base + ((long)b * nchunk + g) * Kmax;
    const int32_t* ip = inv_perm + (long)b * Kmax;
    const int32_t* rs = row_start + (long)b * (Kmax + 1);
    for (int i = lane; i < Kmax; i += 64) cur[i] = rs[ip[i]] + base[i];
    __syncthreads();
    const int p0 = g * SP_CHUNK, p1 = min(HW, p0 + SP_CHUNK);
    int l_nxt = (p0 + lane < p1) ? labels[(long)b * HW + p0 + lane] : -1;
    if (l_nxt >= Kmax) l_nxt = -1;
    int r_nxt = l_nxt >= 0 ? ip[l_nxt] : 0;
    for (int s = p0; s < p1; s += 64) {
        const int p = s + lane;
        const int l = l_nxt, row = r_nxt;
        if (s + 64 < p1) {
            const int pn = p + 64;
            l_nxt = (pn < p1) ? labels[(long)b * HW + pn] : -1;
            if (l_nxt >= Kmax) l_nxt = -1;
            r_nxt = l_nxt >= 0 ? ip[l_nxt] : 0;
        }
        unsigned long long rem = __ballot(l >= 0);
        int pos = -1;
        while (rem) {
            const int leader = __ffsll((long long)rem) - 1;
            const int ll = __shfl(l, leader);
            const unsigned long long m = __ballot(l == ll);
            const int c0 = cur[ll];                     // (every lane reads the same word: a broadcast)
            if (l == ll) pos = c0 + __popcll(m & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == leader) cur[ll] = c0 + __popcll(m);
            __syncthreads();
            rem &= ~m;
        }
        if (l >= 0) {
            pix_sorted[(long)b * HW + pos] = p;
            new_row[(long)b * HW + p] = row;
        } else if (p < p1) {
            new_row[(long)b * HW + p] = 0;
        }
    }
}

__global__ void sp_zero_kernel(int32_t* __restrict__ w, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0;
}

extern "C" size_t wesup_sp_preprocess_workspace_bytes(int B, int HW, int C, int Kmax) {
    if (B <= 0 || HW <= 0 || Kmax <= 0) return 0;
    const size_t nchunk = (HW + SP_CHUNK - 1) / SP_CHUNK;
    const size_t Cc = C > 0 ? C : 1;
    size_t bytes = align_up((size_t)B * nchunk * Kmax * 4, 256);     // chunk_hist / chunk_base
    bytes += align_up((size_t)B * (nchunk * 2 + Kmax * (1 + Cc)) * 4, 256);      // {largest id + 1, status} per chunk, class counts, areas by id
    return bytes;
}

extern "C" int wesup_sp_max_units(int HW, int Kmax) { return (HW > 0 && Kmax > 0) ? Kmax + HW / SP_SEG : 0; }

extern "C" int wesup_sp_preprocess(const int32_t* labels, const uint8_t* mask, int B, int HW, int C, int Kmax,
                                   int32_t* n_sp, int32_t* n_l, int32_t* perm, int32_t* inv_perm, int32_t* area_new,
                                   float* sp_labels, int32_t* new_row, int32_t* row_start, int32_t* pix_sorted,
                                   int32_t* status, int32_t* seg_start, int32_t* unit_row, int Umax, void* ws, size_t ws_bytes,
                                   void* stream) {
    if (!labels || !n_sp || !n_l || !perm || !inv_perm || !area_new || !sp_labels || !new_row || !row_start ||
        !pix_sorted || !status || !ws)
        return WESUP_ERR_INVALID;
    if (B <= 0 || B > 65535 || HW <= 0 || C <= 0 || Kmax <= 0 || Kmax > SP_MAX_K) return WESUP_ERR_INVALID;
    if ((seg_start || unit_row) && (!seg_start || !unit_row || Umax < Kmax)) return WESUP_ERR_INVALID;
    const size_t lds = ((size_t)Kmax + (mask ? ((size_t)Kmax * C + 1) / 2 : 0)) * 4;
    if (lds > 160 * 1024) return WESUP_ERR_INVALID;       // (a mask of up to 3 classes at Kmax = 16 384; Kmax * (2 + C) <= 81 920)
    if (ws_bytes < wesup_sp_preprocess_workspace_bytes(B, HW, C, Kmax)) return WESUP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = (HW + SP_CHUNK - 1) / SP_CHUNK;
    SpPre p = {};
    char* w = (char*)ws;
    p.chunk_hist = (int32_t*)w; w += align_up((size_t)B * nchunk * Kmax * 4, 256);
    p.chunk_info = (int32_t*)w;
    p.cnt = p.chunk_info + (size_t)B * nchunk * 2;
    p.area_old = p.cnt + (size_t)B * Kmax * C;
    const long sums = (long)B * ((long)nchunk * 2 + (long)Kmax * (1 + C));
    p.labels = labels; p.mask = mask; p.HW = HW; p.C = C; p.Kmax = Kmax; p.nchunk = nchunk; p.Umax = Umax;
    p.nslice = (Kmax + SP_SCAN_COLS - 1) / SP_SCAN_COLS;
    p.n_sp = n_sp; p.n_l = n_l; p.perm = perm; p.inv_perm = inv_perm; p.area_new = area_new; p.row_start = row_start;
    p.status = status; p.seg_start = seg_start; p.unit_row = unit_row; p.sp_labels = sp_labels;
    {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(sp_hist_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return WESUP_ERR_LAUNCH;
        static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(sp_place_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, SP_MAX_K * 4);
        if (attr2 != hipSuccess) return WESUP_ERR_LAUNCH;
    }
    WESUP_LAUNCH(sp_zero_kernel, dim3((unsigned)((sums + 255) / 256)), dim3(256), 0, st, p.chunk_info, sums);
    WESUP_LAUNCH(sp_hist_kernel, dim3(nchunk, B), dim3(1024), lds, st, p);
    WESUP_LAUNCH(sp_scan_order_kernel, dim3(p.nslice + 1, B), dim3(1024), 0, st, p);
    WESUP_LAUNCH(sp_place_kernel, dim3(nchunk, B), dim3(64), (size_t)Kmax * 4, st, labels, p.chunk_hist, inv_perm, row_start, HW,
                 Kmax, nchunk, pix_sorted, new_row);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ dense compat: labels from (N,H,W) maps
__global__ void spmaps_argmax_kernel(const float* __restrict__ maps, int32_t* __restrict__ labels, int N, long HW) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float best = maps[p];
    int bi = 0;
    for (int n = 1; n < N; ++n) {
        const float v = maps[(long)n * HW + p];
        if (v > best) { best = v; bi = n; }
    }
    labels[p] = bi;
}
extern "C" int wesup_spmaps_to_labels(const float* sp_maps, int32_t* labels, int N, int HW, void* stream) {
    if (!sp_maps || !labels || N <= 0 || HW <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(spmaps_argmax_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, sp_maps, labels, N,
                       (long)HW);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ scatter-mean forward
// One wave per (row, 256-channel slab): lane owns 4 channels, walks the row's pixel list 8 pixels per
// iteration (8 x 1 KiB loads in flight per wave), acc = fma(v, 1/area, acc) as the reference's pre-normalised
// GEMM does.  Algorithmic bytes per image: C*HW*4 (features) + HW*4 (pixel list) + N*C*4 (output).
//
// Skewed maps: a superpixel is cut into segments of at most SP_SEG pixels (wesup_sp_segments builds, per image,
// seg_start[row] and the inverse unit_row[segment]); the work unit of a wave is one SEGMENT.  Rows of one segment
// (the normal case) are written directly; longer rows leave one partial sum per segment in the workspace and
// sp_pool_combine_kernel adds them in segment order -- still a fixed summation order, and a superpixel 50x the
// median no longer serialises on one wave.
#define POOL_UNROLL 8
struct SegInfo {
    int r, j0, j1, nseg, u;
    float inv;
};
// wave-uniform: everything comes from scalar loads
__device__ __forceinline__ bool seg_lookup(const int32_t* __restrict__ row_start, const int32_t* __restrict__ seg_start,
                                           const int32_t* __restrict__ unit_row, int b, int u, int Kmax, int Umax,
                                           SegInfo& s) {
    const int32_t* ss = seg_start + (long)b * (Kmax + 1);
    if (u >= ss[Kmax]) return false;
    const int r = unit_row[(long)b * Umax + u];
    const int s0 = ss[r];
    const int a0 = row_start[(long)b * (Kmax + 1) + r], a1 = row_start[(long)b * (Kmax + 1) + r + 1];
    s.r = r;
    s.u = u;
    s.nseg = ss[r + 1] - s0;
    s.j0 = a0 + (u - s0) * SP_SEG;
    s.j1 = min(a1, s.j0 + SP_SEG);
    s.inv = (a1 > a0) ? 1.f / (float)(a1 - a0) : 0.f;
    return true;
}
__global__ __launch_bounds__(1024) void sp_segments_kernel(const int32_t* __restrict__ row_start, int Kmax, int Umax,
                                                           int32_t* __restrict__ seg_start, int32_t* __restrict__ unit_row) {
    __shared__ int sh[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (Kmax + 1023) / 1024;
    const int i0 = min(Kmax, tid * per), i1 = min(Kmax, i0 + per);
    const int32_t* rs = row_start + (long)b * (Kmax + 1);
    int cnt = 0;
    for (int r = i0; r < i1; ++r) cnt += max(1, (rs[r + 1] - rs[r] + SP_SEG - 1) / SP_SEG);
    int total;
    int pre = block_excl_scan(cnt, &total, sh);
    for (int r = i0; r < i1; ++r) {
        const int n = max(1, (rs[r + 1] - rs[r] + SP_SEG - 1) / SP_SEG);
        seg_start[(long)b * (Kmax + 1) + r] = pre;
        for (int k = 0; k < n && pre + k < Umax; ++k) unit_row[(long)b * Umax + pre + k] = r;
        pre += n;
    }
    if (tid == 1023) seg_start[(long)b * (Kmax + 1) + Kmax] = min(total, Umax);
}
// (the same table from a row_start of the caller's own; wesup_sp_preprocess writes it itself when seg_start is given)
extern "C" int wesup_sp_segments(const int32_t* row_start, int B, int Kmax, int Umax, int32_t* seg_start,
                                 int32_t* unit_row, void* stream) {
    if (!row_start || !seg_start || !unit_row || B <= 0 || Kmax <= 0 || Umax < Kmax) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(sp_segments_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, row_start, Kmax, Umax, seg_start,
                 unit_row);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// out[b][r][c] = sum over the row's segments (in order) of part[b][seg][c]; rows with one segment were written directly
__global__ void sp_pool_combine_kernel(const float* __restrict__ part, const int32_t* __restrict__ seg_start,
                                       float* __restrict__ out, int Kmax, int Umax, int C4, int ldo, int coff) {
    const int b = blockIdx.y;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Kmax * C4) return;
    const int r = idx / C4, q = idx - (long)r * C4;
    const int s0 = seg_start[(long)b * (Kmax + 1) + r], s1 = seg_start[(long)b * (Kmax + 1) + r + 1];
    if (s1 - s0 <= 1) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = s0; u < s1; ++u) {
        const float4 v = ld4(part + (((long)b * Umax + u) * C4 + q) * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    st4(out + ((long)b * Kmax + r) * ldo + coff + 4 * q, acc);
}
extern "C" size_t wesup_sp_pool_workspace_bytes(int B, int Umax, int C) {
    return (B > 0 && Umax > 0 && C > 0) ? (size_t)B * Umax * C * sizeof(float) : 0;
}

__global__ __launch_bounds__(256) void sp_pool_fwd_kernel(const float* __restrict__ fm,
                                                          const int32_t* __restrict__ pix_sorted,
                                                          const int32_t* __restrict__ row_start,
                                                          const int32_t* __restrict__ seg_start,
                                                          const int32_t* __restrict__ unit_row, float* __restrict__ part,
                                                          float* __restrict__ sp_feat, int HW, int ldf, int C, int Kmax,
                                                          int Umax, int nslab, int units_per_img) {
    const int b = blockIdx.y;
    // wave-uniform unit id -> row bounds and pixel indices live in SGPRs (scalar loads)
    const int unit = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (unit >= units_per_img) return;
    const int lane = threadIdx.x & 63;
    const int u = unit / nslab, slab = unit - u * nslab;
    const int c = slab * 256 + 4 * lane;
    SegInfo sg;
    if (!seg_lookup(row_start, seg_start, unit_row, b, u, Kmax, Umax, sg)) return;
    if (c >= C) return;
    const int r = sg.r, j0 = sg.j0, j1 = sg.j1;
    const float inv = sg.inv;
    const int32_t* list = pix_sorted + (long)b * HW;
    const float* base = fm + (long)b * HW * ldf + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = j0;
    for (; j + POOL_UNROLL <= j1; j += POOL_UNROLL) {
        int pix[POOL_UNROLL];
#pragma unroll
        for (int u = 0; u < POOL_UNROLL; ++u) pix[u] = list[j + u];
        float4 v[POOL_UNROLL];
#pragma unroll
        for (int u = 0; u < POOL_UNROLL; ++u) v[u] = ld4(base + (long)pix[u] * ldf);
#pragma unroll
        for (int u = 0; u < POOL_UNROLL; ++u) {
            acc.x = fmaf(v[u].x, inv, acc.x);
            acc.y = fmaf(v[u].y, inv, acc.y);
            acc.z = fmaf(v[u].z, inv, acc.z);
            acc.w = fmaf(v[u].w, inv, acc.w);
        }
    }
    for (; j < j1; ++j) {
        const float4 v = ld4(base + (long)list[j] * ldf);
        acc.x = fmaf(v.x, inv, acc.x);
        acc.y = fmaf(v.y, inv, acc.y);
        acc.z = fmaf(v.z, inv, acc.z);
        acc.w = fmaf(v.w, inv, acc.w);
    }
    if (sg.nseg == 1) st4(sp_feat + ((long)b * Kmax + r) * C + c, acc);
    else st4(part + ((long)b * Umax + sg.u) * C + c, acc);
}
extern "C" int wesup_sp_pool_fwd(const float* fm, const int32_t* pix_sorted, const int32_t* row_start,
                                 const int32_t* seg_start, const int32_t* unit_row, float* sp_feat, int B, int HW, int ldf,
                                 int C, int Kmax, int Umax, void* ws, size_t ws_bytes, void* stream) {
    if (!fm || !pix_sorted || !row_start || !seg_start || !unit_row || !sp_feat || !ws || B <= 0 || HW <= 0 || Kmax <= 0 ||
        Umax < Kmax || (C % 4) || (ldf % 4) || C > ldf)
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_sp_pool_workspace_bytes(B, Umax, C)) return WESUP_ERR_WORKSPACE;
    const int nslab = ceil_div(C, 256);
    const int units = Umax * nslab;
    hipStream_t st = (hipStream_t)stream;
    WESUP_LAUNCH(sp_pool_fwd_kernel, dim3(ceil_div(units, 4), B), dim3(256), 0, st, fm, pix_sorted, row_start,
                       seg_start, unit_row, (float*)ws, sp_feat, HW, ldf, C, Kmax, Umax, nslab, units);
    const long tot = (long)Kmax * (C / 4);
    WESUP_LAUNCH(sp_pool_combine_kernel, dim3((unsigned)((tot + 255) / 256), B), dim3(256), 0, st, (const float*)ws,
                       seg_start, sp_feat, Kmax, Umax, C / 4, C, 0);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ scatter-mean of the UPSAMPLED side output, fused
// sp_feat[b][r][coff+c] = (1/area_r) * sum_{p in row r} bilinear(s[b], p)[c]  -- the same value wesup_upsample_fwd +
// wesup_sp_pool_fwd produce, without ever writing the (HW x 2112) feature map: s (the side conv output at its
// native h x w resolution, a few MB, L2/MALL resident) is sampled on the fly.  One wave per row; LPP = C/4 lanes
// cover the channels of one pixel, so a wave walks 64/LPP pixels at a time; lane groups are combined by a fixed
// xor-shuffle tree (deterministic).  Pixel order inside a group is ascending list order.
struct Lerp2 {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp2 lerp2_of(int dst, float scale, int in) {
    Lerp2 r;
    const float src = scale * (float)dst;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
    r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
    r.l0 = 1.f - r.l1;
    return r;
}
#define SP_UP_UNROLL 4
#ifndef SP_ID_UNROLL
#define SP_ID_UNROLL 4          // native-resolution branch: row loads in flight per lane group
#endif
#define SP_CELL_CAP 1024          // cells of a segment's box of the coarse grid kept in LDS (8 KiB per wave)
template <int LPP>
__global__ __launch_bounds__(256) void sp_pool_up_fwd_kernel(const float* __restrict__ s, const int32_t* __restrict__ pix_sorted,
                                                             const int32_t* __restrict__ row_start,
                                                             const int32_t* __restrict__ seg_start,
                                                             const int32_t* __restrict__ unit_row, float* __restrict__ part,
                                                             float* __restrict__ sp_feat, int h, int w, int H, int W,
                                                             FastDiv dW, int lds, int ldo, int coff, int Kmax, int Umax,
                                                             float sh, float sw) {
    constexpr int PPW = 64 / LPP;
    constexpr int C = LPP * 4;
    const int b = blockIdx.y;
    const int u = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SegInfo sg;
    if (u >= Umax || !seg_lookup(row_start, seg_start, unit_row, b, u, Kmax, Umax, sg)) return;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPP, cl = lane % LPP;
    const int HW = H * W;
    const int r = sg.r, j0 = sg.j0, j1 = sg.j1;
    const float inv = sg.inv;
    const int32_t* list = pix_sorted + (long)b * HW;
    const float* base = s + (long)b * h * w * lds + 4 * cl;       // lds: floats between two cells of s (>= C)
    const bool ident = (h == H && w == W);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Coarse maps: "sample 4 cells per pixel, then average" is regrouped by CELL.  The segment's pixels touch only the
    // cells of a small box of the coarse grid (a 400-pixel superpixel under a 120x120 map: ~8x8 cells), so the wave
    // first sums the bilinear weights per cell of that box in LDS (2^-40 fixed point, 64-bit integer atomics: the sums
    // do not depend on arrival order) and then fetches every touched cell ONCE: ~25x fewer cache reads than 4 loads per
    // pixel.  A segment whose box does not fit (long thin diagonal shapes) takes the per-pixel loop below.
    __shared__ unsigned long long cellbuf[4][SP_CELL_CAP];
    bool by_cell = false;
    if (!ident) {
        unsigned long long* cell = cellbuf[threadIdx.x >> 6];
        int y0 = 1 << 30, y1 = -1, x0 = 1 << 30, x1 = -1;
        // the lane's pixels of the segment (<= SP_SEG / 64), loaded once, all loads in flight together, for both passes
        int pl[SP_SEG / 64];
#pragma unroll
        for (int k = 0; k < SP_SEG / 64; ++k) pl[k] = (j0 + lane + 64 * k < j1) ? list[j0 + lane + 64 * k] : -1;
#pragma unroll
        for (int k = 0; k < SP_SEG / 64; ++k) {
            const int p = pl[k];
            if (p < 0) continue;
            const int Y = fast_div(p, dW), X = p - Y * W;
            const Lerp2 ly = lerp2_of(Y, sh, h), lx = lerp2_of(X, sw, w);
            y0 = min(y0, ly.i0); y1 = max(y1, ly.i1);
            x0 = min(x0, lx.i0); x1 = max(x1, lx.i1);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
            x0 = min(x0, __shfl_xor(x0, off)); x1 = max(x1, __shfl_xor(x1, off));
        }
        const int bw = (y1 >= 0) ? x1 - x0 + 1 : 0, ncell = (y1 >= 0) ? bw * (y1 - y0 + 1) : 0;
        by_cell = y1 >= 0 && ncell <= SP_CELL_CAP;          // wave-uniform
        if (by_cell) {
            for (int q = lane; q < ncell; q += 64) cell[q] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const float FX = 1099511627776.f;      // 2^40
#pragma unroll
            for (int k = 0; k < SP_SEG / 64; ++k) {
                const int p = pl[k];
                if (p < 0) continue;
                const int Y = fast_div(p, dW), X = p - Y * W;
                const Lerp2 ly = lerp2_of(Y, sh, h), lx = lerp2_of(X, sw, w);
                const int a0 = (ly.i0 - y0) * bw - x0, a1 = (ly.i1 - y0) * bw - x0;
                atomicAdd(&cell[a0 + lx.i0], (unsigned long long)(ly.l0 * lx.l0 * FX + 0.5f));
                atomicAdd(&cell[a0 + lx.i1], (unsigned long long)(ly.l0 * lx.l1 * FX + 0.5f));
                atomicAdd(&cell[a1 + lx.i0], (unsigned long long)(ly.l1 * lx.l0 * FX + 0.5f));
                atomicAdd(&cell[a1 + lx.i1], (unsigned long long)(ly.l1 * lx.l1 * FX + 0.5f));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const float scale = inv * (1.f / FX);
            // SP_UP_UNROLL cells in flight per lane group (same order of additions as one at a time: a cell nobody touched
            // has weight 0 and its load is skipped)
            for (int q0 = grp; q0 < ncell; q0 += PPW * SP_UP_UNROLL) {
                float wgt[SP_UP_UNROLL];
                float4 v[SP_UP_UNROLL];
#pragma unroll
                for (int u = 0; u < SP_UP_UNROLL; ++u) {
                    const int q = q0 + u * PPW;
                    const unsigned long long wq = (q < ncell) ? cell[q] : 0ull;
                    wgt[u] = (float)wq * scale;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (wq != 0ull) {
                        const int cy = q / bw, cx = q - cy * bw;
                        v[u] = ld4(base + ((long)(y0 + cy) * w + x0 + cx) * lds);
                    }
                }
#pragma unroll
                for (int u = 0; u < SP_UP_UNROLL; ++u) {       // (untouched cell: weight 0 times the 0 put in v)
                    acc.x = fmaf(v[u].x, wgt[u], acc.x);
                    acc.y = fmaf(v[u].y, wgt[u], acc.y);
                    acc.z = fmaf(v[u].z, wgt[u], acc.z);
                    acc.w = fmaf(v[u].w, wgt[u], acc.w);
                }
            }
        }
    }
    if (ident) {
        // native resolution: the pixel's own row of s, added in list order.  The segment's pixel list (<= SP_SEG entries) is loaded
        // ONCE, all loads in flight, one entry per lane and pass; a lane group then takes its pixel indices from those registers by
        // shuffle -- the row loads no longer wait for a list load each (round 6: list -> pixel -> row was a chain of two dependent
        // loads per step).  Same additions in the same order as before.
        int pl[SP_SEG / 64];
#pragma unroll
        for (int k = 0; k < SP_SEG / 64; ++k) pl[k] = (j0 + lane + 64 * k < j1) ? list[j0 + lane + 64 * k] : -1;
        const int npass = (j1 - j0 + 63) >> 6;                       // wave-uniform
#pragma unroll
        for (int k = 0; k < SP_SEG / 64; ++k) {
            if (k >= npass) break;
            // the 64 entries of pass k: group grp takes entries grp, grp + PPW, ...; SP_ID_UNROLL row loads in flight per group
#pragma unroll 1
            for (int e0 = 0; e0 < 64; e0 += PPW * SP_ID_UNROLL) {
                int pix[SP_ID_UNROLL];
                float4 v[SP_ID_UNROLL];
#pragma unroll
                for (int u = 0; u < SP_ID_UNROLL; ++u) {
                    const int e = e0 + u * PPW + grp;                // < 64 whenever PPW * SP_ID_UNROLL divides 64
                    pix[u] = (PPW * SP_ID_UNROLL <= 64 || e < 64) ? __shfl(pl[k], e & 63) : -1;
                }
#pragma unroll
                for (int u = 0; u < SP_ID_UNROLL; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pix[u] >= 0) v[u] = ld4(base + (long)pix[u] * lds);
                }
#pragma unroll
                for (int u = 0; u < SP_ID_UNROLL; ++u) {             // (an entry beyond the segment adds 0 * inv)
                    acc.x = fmaf(v[u].x, inv, acc.x);
                    acc.y = fmaf(v[u].y, inv, acc.y);
                    acc.z = fmaf(v[u].z, inv, acc.z);
                    acc.w = fmaf(v[u].w, inv, acc.w);
                }
            }
        }
    } else if (!by_cell)
    for (int j = j0 + grp; j < j1; j += PPW) {
        const int p = list[j];
        float4 v;
        {
            const int Y = fast_div(p, dW), X = p - Y * W;
            const Lerp2 ly = lerp2_of(Y, sh, h), lx = lerp2_of(X, sw, w);
            const float4 v00 = ld4(base + ((long)ly.i0 * w + lx.i0) * lds);
            const float4 v01 = ld4(base + ((long)ly.i0 * w + lx.i1) * lds);
            const float4 v10 = ld4(base + ((long)ly.i1 * w + lx.i0) * lds);
            const float4 v11 = ld4(base + ((long)ly.i1 * w + lx.i1) * lds);
            v.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
            v.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
            v.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
            v.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
        }
        acc.x = fmaf(v.x, inv, acc.x);
        acc.y = fmaf(v.y, inv, acc.y);
        acc.z = fmaf(v.z, inv, acc.z);
        acc.w = fmaf(v.w, inv, acc.w);
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    if (grp == 0) {
        if (sg.nseg == 1) st4(sp_feat + ((long)b * Kmax + r) * ldo + coff + 4 * cl, acc);
        else st4(part + ((long)b * Umax + sg.u) * C + 4 * cl, acc);
    }
}
extern "C" int wesup_sp_pool_upsample_fwd(const float* s, const int32_t* pix_sorted, const int32_t* row_start,
                                          const int32_t* seg_start, const int32_t* unit_row, float* sp_feat, int B, int h,
                                          int w, int H, int W, int C, int ldo, int coff, int Kmax, int Umax, void* ws,
                                          size_t ws_bytes, void* stream) {
    if (!s || !pix_sorted || !row_start || !seg_start || !unit_row || !sp_feat || !ws || B <= 0 || h <= 0 || w <= 0 ||
        H <= 0 || W <= 0 || Kmax <= 0 || Umax < Kmax || (ldo % 4) || (coff % 4) || coff + C > ldo)
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_sp_pool_workspace_bytes(B, Umax, C)) return WESUP_ERR_WORKSPACE;
    const dim3 grid(ceil_div(Umax, 4), B);
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const FastDiv dW = make_fastdiv(W);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
#define WESUP_LAUNCH_PU(L, c0)                                                                                             \
    WESUP_LAUNCH(sp_pool_up_fwd_kernel<L>, grid, dim3(256), 0, st, s + (c0), pix_sorted, row_start, seg_start, unit_row, \
                       part, sp_feat, h, w, H, W, dW, C, ldo, coff + (c0), Kmax, Umax, sh, sw)
    // a wave's 64 lanes cover 256 channels of a pixel: wider maps go in slabs of 256 channels (row stride C)
    if (C > 256 && (C % 256)) return WESUP_ERR_INVALID;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = C - c0 < 256 ? C - c0 : 256;
        switch (cw) {
            case 32: WESUP_LAUNCH_PU(8, c0); break;
            case 64: WESUP_LAUNCH_PU(16, c0); break;
            case 128: WESUP_LAUNCH_PU(32, c0); break;
            case 256: WESUP_LAUNCH_PU(64, c0); break;
            default: return WESUP_ERR_INVALID;
        }
        const long tot = (long)Kmax * (cw / 4);
        WESUP_LAUNCH(sp_pool_combine_kernel, dim3((unsigned)((tot + 255) / 256), B), dim3(256), 0, st, (const float*)ws,
                           seg_start, sp_feat, Kmax, Umax, cw / 4, ldo, coff + c0);
    }
#undef WESUP_LAUNCH_PU
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ interpolation-pooling matrix of a coarse grid
// For a side output at a coarse h x w grid, "bilinear upsample to H x W, then average over superpixel r" is one
// linear map of the h*w coarse cells:   Wm[b][r][q] = (1/area_r) * sum_{p in row r} bw(p, q),  bw = the (<= 4) bilinear
// weights of full-resolution pixel p (torch align_corners=True formula, as in wesup_upsample_fwd).  With it the
// fused forward is  sp_feat[b][:, slice] = Wm[b] . s[b]  and the fused backward  ds[b] = Wm[b]^T . g[b][:, slice]:
// two small MFMA GEMMs per image (wesup_gemm_tn on Wm^T / Wm) instead of H*W*C gathers per layer.  It pays for the
// deep layers (60x60 / 30x30 at 480x480: 1536 of the 2112 channels), where Wm is a few MB per image.
// One block per row r; the weights are accumulated in LDS as 2^-40 fixed point with 64-bit integer atomics, so the
// result does not depend on the order in which the pixels arrive (deterministic).
__global__ __launch_bounds__(256) void sp_interp_matrix_kernel(const int32_t* __restrict__ pix_sorted,
                                                               const int32_t* __restrict__ row_start,
                                                               float* __restrict__ Wm, int H, int W, int h, int w,
                                                               int Kmax, FastDiv dW, float sh, float sw) {
    extern __shared__ unsigned long long cell[];
    const int r = blockIdx.x, b = blockIdx.y;
    const int hw = h * w;
    const int j0 = row_start[b * (Kmax + 1) + r], j1 = row_start[b * (Kmax + 1) + r + 1];
    for (int q = threadIdx.x; q < hw; q += 256) cell[q] = 0ull;
    __syncthreads();
    const int32_t* list = pix_sorted + (long)b * H * W;
    const float FX = 1099511627776.f;      // 2^40
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
        const int p = list[j];
        const int Y = fast_div(p, dW), X = p - Y * W;
        const Lerp2 ly = lerp2_of(Y, sh, h), lx = lerp2_of(X, sw, w);
        // (a tap with zero weight adds nothing; i0 == i1 at the last row / column just adds twice into one cell)
        atomicAdd(&cell[ly.i0 * w + lx.i0], (unsigned long long)(ly.l0 * lx.l0 * FX + 0.5f));
        atomicAdd(&cell[ly.i0 * w + lx.i1], (unsigned long long)(ly.l0 * lx.l1 * FX + 0.5f));
        atomicAdd(&cell[ly.i1 * w + lx.i0], (unsigned long long)(ly.l1 * lx.l0 * FX + 0.5f));
        atomicAdd(&cell[ly.i1 * w + lx.i1], (unsigned long long)(ly.l1 * lx.l1 * FX + 0.5f));
    }
    __syncthreads();
    const float scale = (j1 > j0) ? 1.f / ((float)(j1 - j0) * FX) : 0.f;
    float* out = Wm + ((long)b * Kmax + r) * hw;
    for (int q = threadIdx.x; q < hw; q += 256) out[q] = (float)cell[q] * scale;
}
extern "C" int wesup_sp_interp_matrix(const int32_t* pix_sorted, const int32_t* row_start, float* Wm, int B, int H,
                                      int W, int h, int w, int Kmax, void* stream) {
    if (!pix_sorted || !row_start || !Wm || B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || h > H || w > W ||
        Kmax <= 0 || (long)h * w > 8192)
        return WESUP_ERR_INVALID;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    WESUP_LAUNCH(sp_interp_matrix_kernel, dim3(Kmax, B), dim3(256), (size_t)h * w * sizeof(unsigned long long),
                       (hipStream_t)stream, pix_sorted, row_start, Wm, H, W, h, w, Kmax, make_fastdiv(W), sh, sw);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ scatter-mean backward (row broadcast)
__global__ void sp_pool_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ new_row,
                                   const int32_t* __restrict__ area, float* __restrict__ dfm, long HW, int ldf, int C4,
                                   int Kmax, long total) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = idx % C4;
        const long bp = idx / C4;              // b*HW + p
        const long b = bp / HW;
        const int r = new_row[bp];
        const float inv = 1.f / (float)area[b * Kmax + r];
        const float4 v = ld4(g + (b * Kmax + r) * (long)(C4 * 4) + 4 * c);
        st4(dfm + bp * ldf + 4 * c, make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv));
    }
}
extern "C" int wesup_sp_pool_bwd(const float* g, const int32_t* new_row, const int32_t* area_new, float* dfm, int B,
                                 int HW, int ldf, int C, int Kmax, void* stream) {
    if (!g || !new_row || !area_new || !dfm || B <= 0 || HW <= 0 || Kmax <= 0 || (C % 4) || (ldf % 4) || C > ldf)
        return WESUP_ERR_INVALID;
    const long total = (long)B * HW * (C / 4);
    const long blocks = (total + 255) / 256;
    WESUP_LAUNCH(sp_pool_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                       (hipStream_t)stream, g, new_row, area_new, dfm, (long)HW, ldf, C / 4, Kmax, total);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ paint-back
__global__ void paint_kernel(const float* __restrict__ sp_pred, const int32_t* __restrict__ new_row,
                             float* __restrict__ pred, long HW, int Kmax, int C, int cls, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long b = idx / HW;
    pred[idx] = sp_pred[(b * Kmax + new_row[idx]) * C + cls];
}
extern "C" int wesup_paint_fwd(const float* sp_pred, const int32_t* new_row, float* pred, int B, int HW, int Kmax, int C,
                               int cls, void* stream) {
    if (!sp_pred || !new_row || !pred || B <= 0 || HW <= 0 || Kmax <= 0 || cls < 0 || cls >= C) return WESUP_ERR_INVALID;
    const long total = (long)B * HW;
    WESUP_LAUNCH(paint_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sp_pred,
                       new_row, pred, (long)HW, Kmax, C, cls, total);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
