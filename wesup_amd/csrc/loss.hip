// Head and loss kernels: classifier + softmax, label propagation over the superpixel affinity
// exp(-|fi-fj|^2), semi-supervised cross entropy, fused SGD, segmentation metric sums.
// All reductions are fixed-order (no float atomics): results are bitwise reproducible.
#include "common.hpp"

// block-wide sum of `v` over 256 threads, fixed tree order; result valid in all threads
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ classifier Linear(D,2) + Softmax(dim=1)
__global__ void classifier_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wc,
                                      const float* __restrict__ bc, float* __restrict__ pred, int R, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float z0 = bc[0], z1 = bc[1];
    const float* f = feat + (long)r * D;
    for (int k = 0; k < D; ++k) {
        const float v = f[k];
        z0 = fmaf(v, Wc[k], z0);
        z1 = fmaf(v, Wc[D + k], z1);
    }
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    const float inv = 1.f / (e0 + e1);
    pred[2 * r] = e0 * inv;
    pred[2 * r + 1] = e1 * inv;
}
extern "C" int wesup_classifier_fwd(const float* feat, const float* Wc, const float* bc, float* pred, int R, int D,
                                    void* stream) {
    if (!feat || !Wc || !bc || !pred || R <= 0 || D <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(classifier_fwd_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, feat, Wc, bc, pred,
                       R, D);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// backward: dz = p * (dp - sum_c dp_c p_c); dfeat = (dz . Wc + extra) masked by feat > 0; partial dWc/dbc per 64 rows
#define CLS_ROWS 64
// The two sums of two products below are written with their one fused multiply-add spelled out: left to the compiler
// (-ffp-contract=fast) either product may become the addend, and classifier_bwd_kernel and head_bwd_kernel -- which must agree
// bit for bit -- are contracted separately.
__device__ __forceinline__ void cls_dz(float p0, float p1, float d0, float d1, float& a, float& b) {
    const float s = fmaf(d1, p1, d0 * p0);
    a = p0 * (d0 - s);
    b = p1 * (d1 - s);
}
__device__ __forceinline__ float cls_dfeat(float dz0, float dz1, float w0, float w1) { return fmaf(dz1, w1, dz0 * w0); }
__global__ void classifier_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wc,
                                      const float* __restrict__ pred, const float* __restrict__ dpred,
                                      const float* __restrict__ extra, float* __restrict__ dfeat,
                                      float* __restrict__ part, int R, int D) {
    __shared__ float dz[CLS_ROWS][2];
    const int r0 = blockIdx.x * CLS_ROWS;
    const int tid = threadIdx.x;                      // 256 threads
    if (tid < CLS_ROWS) {
        const int r = r0 + tid;
        float a = 0.f, b = 0.f;
        if (r < R) {
            const float p0 = pred[2 * r], p1 = pred[2 * r + 1];
            const float d0 = dpred[2 * r], d1 = dpred[2 * r + 1];
            cls_dz(p0, p1, d0, d1, a, b);
        }
        dz[tid][0] = a;
        dz[tid][1] = b;
    }
    __syncthreads();
    // dfeat
    for (int e = tid; e < CLS_ROWS * D; e += 256) {
        const int rr = e / D, k = e - rr * D;
        const int r = r0 + rr;
        if (r < R) {
            float g = cls_dfeat(dz[rr][0], dz[rr][1], Wc[k], Wc[D + k]);
            if (extra) g += extra[(long)r * D + k];
            dfeat[(long)r * D + k] = feat[(long)r * D + k] > 0.f ? g : 0.f;
        }
    }
    // partial dWc[c][k] (2*D entries) and dbc[c] (2 entries) for this block of rows
    float* pout = part + (long)blockIdx.x * (2 * D + 2);
    for (int e = tid; e < 2 * D + 2; e += 256) {
        float s = 0.f;
        if (e < 2 * D) {
            const int c = e / D, k = e - c * D;
            for (int rr = 0; rr < CLS_ROWS; ++rr) {
                const int r = r0 + rr;
                if (r < R) s = fmaf(dz[rr][c], feat[(long)r * D + k], s);
            }
        } else {
            const int c = e - 2 * D;
            for (int rr = 0; rr < CLS_ROWS; ++rr) s += dz[rr][c];
        }
        pout[e] = s;
    }
}
__global__ void classifier_bwd_reduce(const float* __restrict__ part, float* __restrict__ dWc, float* __restrict__ dbc,
                                      int nblk, int D) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * D + 2) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * (2 * D + 2) + e];
    if (e < 2 * D) dWc[e] = s;
    else dbc[e - 2 * D] = s;
}
extern "C" size_t wesup_classifier_bwd_workspace_bytes(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (size_t)ceil_div(R, CLS_ROWS) * (2 * D + 2) * sizeof(float);
}
extern "C" int wesup_classifier_bwd(const float* feat, const float* Wc, const float* pred, const float* dpred,
                                    const float* dfeat_extra, float* dfeat, float* dWc, float* dbc, int R, int D,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !Wc || !pred || !dpred || !dfeat || !dWc || !dbc || !ws || R <= 0 || D <= 0) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_classifier_bwd_workspace_bytes(R, D)) return WESUP_ERR_WORKSPACE;
    const int nblk = ceil_div(R, CLS_ROWS);
    hipStream_t st = (hipStream_t)stream;
    WESUP_LAUNCH(classifier_bwd_kernel, dim3(nblk), dim3(256), 0, st, feat, Wc, pred, dpred, dfeat_extra, dfeat,
                       (float*)ws, R, D);
    WESUP_LAUNCH(classifier_bwd_reduce, dim3(ceil_div(2 * D + 2, 64)), dim3(64), 0, st, (const float*)ws, dWc, dbc,
                       nblk, D);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ label propagation (models/wesup.py:99-139)
// One launch: a block owns PROP_ROWS consecutive rows of an image.  It first writes what every row has whatever happens next --
// y_all = the row's own labels (labelled rows) or zeros, src_idx = -1, max_sim = 0 (a launch of its own until round 5) -- and
// then, for its unlabelled rows, one wave per row: lanes stride over the labelled rows j (staged in LDS 256 at a time, row stride
// D+1 floats so that lanes hit distinct banks).  d_ij = sum_k (f_j,k - f_i,k)^2 as a direct difference in ascending k (not
// |a|^2+|b|^2-2ab: near-ties must not flip); W = expf(-d); the row maximum takes the FIRST (lowest) labelled index on ties, as
// torch.max(dim=1) does; propagate iff W > threshold (strict).
#define PROP_TILE 256
#define PROP_ROWS 16
__global__ __launch_bounds__(256) void prop_kernel(const float* __restrict__ feat, const float* __restrict__ sp_labels,
                                                   const int32_t* __restrict__ n_sp, const int32_t* __restrict__ n_l,
                                                   float thr, int enable, float* __restrict__ y_all, int32_t* __restrict__ src_idx,
                                                   float* __restrict__ max_sim, int Kmax, int D, int C,
                                                   const float* __restrict__ Wc, const float* __restrict__ bc,
                                                   float* __restrict__ pred) {
    extern __shared__ float sh[];
    float* fl = sh;                               // [PROP_TILE][D+1]
    float* fi = sh + PROP_TILE * (D + 1);         // [PROP_ROWS][D]
    const int b = blockIdx.y;
    const int nl = n_l[b], ns = n_sp[b];
    const int i_blk = blockIdx.x * PROP_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < PROP_ROWS * C; e += 256) {
        const int r = i_blk + e / C;
        if (r < Kmax) y_all[((long)b * Kmax + i_blk) * C + e] = (r < nl) ? sp_labels[((long)b * Kmax + i_blk) * C + e] : 0.f;
    }
    if (tid < PROP_ROWS && i_blk + tid < Kmax) {
        src_idx[(long)b * Kmax + i_blk + tid] = -1;
        max_sim[(long)b * Kmax + i_blk + tid] = 0.f;
    }
    // wesup_head_fwd: the classifier + softmax of this block's rows rides along (classifier_fwd_kernel's arithmetic, row by row:
    // a launch of its own on the chain between the fc layers and the loss otherwise)
    if (pred && tid < PROP_ROWS && i_blk + tid < Kmax) {
        const long r = (long)b * Kmax + i_blk + tid;
        float z0 = bc[0], z1 = bc[1];
        const float* f = feat + r * D;
        for (int k = 0; k < D; ++k) {
            const float v = f[k];
            z0 = fmaf(v, Wc[k], z0);
            z1 = fmaf(v, Wc[D + k], z1);
        }
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        const float inv = 1.f / (e0 + e1);
        pred[2 * r] = e0 * inv;
        pred[2 * r + 1] = e1 * inv;
    }
    // rows of this block that take part in the propagation: unlabelled and present (uniform per block)
    if (!enable || nl <= 0 || i_blk + PROP_ROWS <= nl || i_blk >= ns) return;
    __syncthreads();                              // (the defaults above are written before a propagated row overwrites its own)
    const float* fb = feat + (long)b * Kmax * D;
    for (int e = tid; e < PROP_ROWS * D; e += 256) {
        const int rr = e / D, k = e - rr * D;
        fi[e] = (i_blk + rr < ns) ? fb[(long)(i_blk + rr) * D + k] : 0.f;
    }
    float best_w[PROP_ROWS / 4];
    int best_j[PROP_ROWS / 4];
#pragma unroll
    for (int q = 0; q < PROP_ROWS / 4; ++q) {
        best_w[q] = -1.f;
        best_j[q] = 0x7fffffff;
    }
    for (int j0 = 0; j0 < nl; j0 += PROP_TILE) {
        const int nj = min(PROP_TILE, nl - j0);
        __syncthreads();
        for (int e = tid; e < nj * D; e += 256) {
            const int jj = e / D, k = e - jj * D;
            fl[jj * (D + 1) + k] = fb[(long)(j0 + jj) * D + k];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PROP_ROWS / 4; ++q) {
            const int i = i_blk + wave * (PROP_ROWS / 4) + q;
            if (i < nl || i >= ns) continue;      // (wave-uniform: a labelled or absent row)
            const float* f_i = fi + (wave * (PROP_ROWS / 4) + q) * D;
            for (int jj = lane; jj < nj; jj += 64) {
                const float* f_j = fl + jj * (D + 1);
                float d = 0.f;
                for (int k = 0; k < D; ++k) {
                    const float t = f_j[k] - f_i[k];
                    d = fmaf(t, t, d);
                }
                const float w = expf(-d);
                if (w > best_w[q]) {              // strict: earlier j wins ties inside a lane
                    best_w[q] = w;
                    best_j[q] = j0 + jj;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PROP_ROWS / 4; ++q) {
        float w = best_w[q];
        int j = best_j[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ow = __shfl_xor(w, off);
            const int oj = __shfl_xor(j, off);
            if (ow > w || (ow == w && oj < j)) {
                w = ow;
                j = oj;
            }
        }
        const int i = i_blk + wave * (PROP_ROWS / 4) + q;
        if (lane == 0 && i >= nl && i < ns) {
            max_sim[(long)b * Kmax + i] = w;
            src_idx[(long)b * Kmax + i] = j;
            if (w > thr)
                for (int c = 0; c < C; ++c)
                    y_all[((long)b * Kmax + i) * C + c] = sp_labels[((long)b * Kmax + j) * C + c];
        }
    }
}
extern "C" int wesup_propagate(const float* feat, const float* sp_labels, const int32_t* n_sp, const int32_t* n_l,
                               float threshold, int enable, float* y_all, int32_t* src_idx, float* max_sim, int B,
                               int Kmax, int D, int C, void* stream) {
    if (!feat || !sp_labels || !n_sp || !n_l || !y_all || !src_idx || !max_sim || B <= 0 || Kmax <= 0 || D <= 0 ||
        D > 256 || C <= 0)
        return WESUP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)PROP_TILE * (D + 1) + (size_t)PROP_ROWS * D) * sizeof(float);
    WESUP_LAUNCH(prop_kernel, dim3(ceil_div(Kmax, PROP_ROWS), B), dim3(256), lds, st, feat, sp_labels, n_sp, n_l, threshold,
                 enable, y_all, src_idx, max_sim, Kmax, D, C, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// classifier + softmax (wesup_classifier_fwd) and label propagation (wesup_propagate) of the same features in ONE launch: both
// read feat, neither reads the other's result.  pred (B*Kmax, 2).  Bit-identical to the two entries.
extern "C" int wesup_head_fwd(const float* feat, const float* Wc, const float* bc, float* pred, const float* sp_labels,
                              const int32_t* n_sp, const int32_t* n_l, float threshold, int enable, float* y_all,
                              int32_t* src_idx, float* max_sim, int B, int Kmax, int D, int C, void* stream) {
    if (!feat || !Wc || !bc || !pred || !sp_labels || !n_sp || !n_l || !y_all || !src_idx || !max_sim || B <= 0 || Kmax <= 0 ||
        D <= 0 || D > 256 || C <= 0)
        return WESUP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)PROP_TILE * (D + 1) + (size_t)PROP_ROWS * D) * sizeof(float);
    WESUP_LAUNCH(prop_kernel, dim3(ceil_div(Kmax, PROP_ROWS), B), dim3(256), lds, st, feat, sp_labels, n_sp, n_l, threshold,
                 enable, y_all, src_idx, max_sim, Kmax, D, C, Wc, bc, pred);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ loss (models/wesup.py:66-96, 492-531)
__device__ __forceinline__ float ce_row(const float* p, const float* y, int C, float eps, float* ysum,
                                        const float* cw = nullptr) {
    float s = 0.f, ys = 0.f;
    for (int c = 0; c < C; ++c) {
        const float yc = y[c];
        ys += yc;
        // torch.clamp keeps a NaN (models/wesup.py:83), fminf/fmaxf would swallow it and a NaN prediction would end
        // in a finite loss instead of the ValueError of models/base.py:202-203
        const float pc = (p[c] != p[c]) ? p[c] : fminf(fmaxf(p[c], eps), 1.f - eps);
        const float ce = -yc * logf(pc);
        s += cw ? ce * cw[c] : ce;               // models/wesup.py:93-94
    }
    *ysum = ys;
    return s;
}
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ y_all,
                                                       const int32_t* __restrict__ n_sp, const int32_t* __restrict__ n_l,
                                                       float eps, float prop_weight, float* __restrict__ terms, int Kmax,
                                                       int C) {
    __shared__ float sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ns = n_sp[b], nl = n_l[b];
    float sup = 0.f, supc = 0.f, pro = 0.f, proc = 0.f, plab = 0.f;
    for (int r = tid; r < ns; r += 256) {
        float ys;
        const float ce = ce_row(pred + ((long)b * Kmax + r) * C, y_all + ((long)b * Kmax + r) * C, C, eps, &ys);
        if (r < nl) {
            sup += ce;
            supc += (ys > 0.f) ? 1.f : 0.f;
        } else {
            pro += ce;
            proc += (ys > 0.f) ? 1.f : 0.f;
            plab += ys;
        }
    }
    sup = block_sum256(sup, sh);
    supc = block_sum256(supc, sh);
    pro = block_sum256(pro, sh);
    proc = block_sum256(proc, sh);
    plab = block_sum256(plab, sh);
    if (tid == 0) {
        float l = (supc > 0.f) ? sup / supc : 0.f;
        if (nl < ns && proc > 0.f) l += prop_weight * (pro / proc);
        float* t = terms + (long)b * 8;
        t[0] = sup; t[1] = supc; t[2] = pro; t[3] = proc; t[4] = plab; t[5] = l; t[6] = 0.f; t[7] = 0.f;
    }
}
__global__ void loss_mean_kernel(const float* __restrict__ terms, float* __restrict__ loss, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += terms[(long)b * 8 + 5];
        loss[0] = s / (float)B;
    }
}
extern "C" int wesup_loss_fwd(const float* pred, const float* y_all, const int32_t* n_sp, const int32_t* n_l, float eps,
                              float prop_weight, float* terms, float* loss, int B, int Kmax, int C, void* stream) {
    if (!pred || !y_all || !n_sp || !n_l || !terms || B <= 0 || Kmax <= 0 || C <= 0) return WESUP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    WESUP_LAUNCH(loss_fwd_kernel, dim3(B), dim3(256), 0, st, pred, y_all, n_sp, n_l, eps, prop_weight, terms, Kmax, C);
    // (loss == NULL: the caller forms the mean of terms[b][5] itself -- the step runner does, on the host, from the block it
    // reads back anyway: one launch less on the chain between forward and backward)
    if (loss) WESUP_LAUNCH(loss_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)terms, loss, B);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
// d loss / d pred: -y/p where eps <= p <= 1-eps (torch.clamp passes the gradient on the closed interval)
__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ y_all,
                                const int32_t* __restrict__ n_sp, const int32_t* __restrict__ n_l,
                                const float* __restrict__ terms, const float* __restrict__ dloss, float eps,
                                float prop_weight, float* __restrict__ dpred, int B, int Kmax, int C, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long br = idx / C;
    const int b = br / Kmax, r = br - (long)b * Kmax;
    const int ns = n_sp[b], nl = n_l[b];
    float g = 0.f;
    if (r < ns) {
        const float* t = terms + (long)b * 8;
        float coef;
        if (r < nl) coef = (t[1] > 0.f) ? 1.f / t[1] : 0.f;
        else coef = (nl < ns && t[3] > 0.f) ? prop_weight / t[3] : 0.f;
        const float p = pred[idx];
        if (p >= eps && p <= 1.f - eps) g = dloss[0] * (1.f / (float)B) * coef * (-y_all[idx] / p);
    }
    dpred[idx] = g;
}
extern "C" int wesup_loss_bwd(const float* pred, const float* y_all, const int32_t* n_sp, const int32_t* n_l,
                              const float* terms, const float* dloss, float eps, float prop_weight, float* dpred, int B,
                              int Kmax, int C, void* stream) {
    if (!pred || !y_all || !n_sp || !n_l || !terms || !dloss || !dpred || B <= 0 || Kmax <= 0 || C <= 0)
        return WESUP_ERR_INVALID;
    const long total = (long)B * Kmax * C;
    WESUP_LAUNCH(loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                       y_all, n_sp, n_l, terms, dloss, eps, prop_weight, dpred, B, Kmax, C, total);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// Loss, its gradient and the classifier's backward in ONE launch (wesup_loss_fwd + wesup_loss_bwd + wesup_classifier_bwd's first
// kernel: three launches on the chain between the fc layers' forward and backward).  A block owns CLS_ROWS = 64 rows of one image
// (Kmax % 64 == 0).  The per-image sums the gradient needs are a reduction over the whole image: every block of an image forms
// them ITSELF, with loss_fwd_kernel's loop and tree -- the same numbers in every block, no hand-off between blocks; the image's
// first block writes them to terms.  Then dpred of the block's rows (loss_bwd_kernel's expression) and classifier_bwd_kernel's
// body on them.  Two classes (the classifier is Linear(D, 2)).  Bit-identical to the three entries.
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wc,
                                                       const float* __restrict__ pred, const float* __restrict__ y_all,
                                                       const int32_t* __restrict__ n_sp, const int32_t* __restrict__ n_l,
                                                       const float* __restrict__ dloss, float eps, float prop_weight,
                                                       float* __restrict__ terms, float* __restrict__ dpred,
                                                       float* __restrict__ dfeat, float* __restrict__ part, int B, int Kmax,
                                                       int D) {
    constexpr int C = 2;
    __shared__ float sh[256];
    __shared__ float dz[CLS_ROWS][2];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * CLS_ROWS;                 // first row (of B * Kmax) of this block
    const int b = r0 / Kmax, i0 = r0 - b * Kmax;
    const int ns = n_sp[b], nl = n_l[b];
    float sup = 0.f, supc = 0.f, pro = 0.f, proc = 0.f, plab = 0.f;
    for (int r = tid; r < ns; r += 256) {
        float ys;
        const float ce = ce_row(pred + ((long)b * Kmax + r) * C, y_all + ((long)b * Kmax + r) * C, C, eps, &ys);
        if (r < nl) {
            sup += ce;
            supc += (ys > 0.f) ? 1.f : 0.f;
        } else {
            pro += ce;
            proc += (ys > 0.f) ? 1.f : 0.f;
            plab += ys;
        }
    }
    sup = block_sum256(sup, sh);
    supc = block_sum256(supc, sh);
    pro = block_sum256(pro, sh);
    proc = block_sum256(proc, sh);
    plab = block_sum256(plab, sh);
    if (tid == 0 && i0 == 0) {
        float l = (supc > 0.f) ? sup / supc : 0.f;
        if (nl < ns && proc > 0.f) l += prop_weight * (pro / proc);
        float* t = terms + (long)b * 8;
        t[0] = sup; t[1] = supc; t[2] = pro; t[3] = proc; t[4] = plab; t[5] = l; t[6] = 0.f; t[7] = 0.f;
    }
    // dpred of the block's rows, then dz = p * (dp - sum_c dp_c p_c)
    if (tid < CLS_ROWS) {
        const int r = i0 + tid;                           // row inside the image (< Kmax)
        const long gr = (long)r0 + tid;
        float g0 = 0.f, g1 = 0.f;
        const float p0 = pred[2 * gr], p1 = pred[2 * gr + 1];
        if (r < ns) {
            float coef;
            if (r < nl) coef = (supc > 0.f) ? 1.f / supc : 0.f;
            else coef = (nl < ns && proc > 0.f) ? prop_weight / proc : 0.f;
            if (p0 >= eps && p0 <= 1.f - eps) g0 = dloss[0] * (1.f / (float)B) * coef * (-y_all[2 * gr] / p0);
            if (p1 >= eps && p1 <= 1.f - eps) g1 = dloss[0] * (1.f / (float)B) * coef * (-y_all[2 * gr + 1] / p1);
        }
        dpred[2 * gr] = g0;
        dpred[2 * gr + 1] = g1;
        float a, bq;
        cls_dz(p0, p1, g0, g1, a, bq);
        dz[tid][0] = a;
        dz[tid][1] = bq;
    }
    __syncthreads();
    for (int e = tid; e < CLS_ROWS * D; e += 256) {
        const int rr = e / D, k = e - rr * D;
        const long r = (long)r0 + rr;
        const float g = cls_dfeat(dz[rr][0], dz[rr][1], Wc[k], Wc[D + k]);
        dfeat[r * D + k] = feat[r * D + k] > 0.f ? g : 0.f;
    }
    float* pout = part + (long)blockIdx.x * (2 * D + 2);
    for (int e = tid; e < 2 * D + 2; e += 256) {
        float s = 0.f;
        if (e < 2 * D) {
            const int c = e / D, k = e - c * D;
            for (int rr = 0; rr < CLS_ROWS; ++rr) s = fmaf(dz[rr][c], feat[((long)r0 + rr) * D + k], s);
        } else {
            const int c = e - 2 * D;
            for (int rr = 0; rr < CLS_ROWS; ++rr) s += dz[rr][c];
        }
        pout[e] = s;
    }
}
// ws: wesup_classifier_bwd_workspace_bytes(B * Kmax, D); it holds the per-block partial sums of the classifier's weight gradient
// until wesup_classifier_bwd_finish adds them up (any stream behind this launch: nothing on the way to the fc layers' gradients
// reads dWc / dbc).
extern "C" int wesup_head_bwd(const float* feat, const float* Wc, const float* pred, const float* y_all, const int32_t* n_sp,
                              const int32_t* n_l, const float* dloss, float eps, float prop_weight, float* terms, float* dpred,
                              float* dfeat, int B, int Kmax, int D, int C, void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !Wc || !pred || !y_all || !n_sp || !n_l || !dloss || !terms || !dpred || !dfeat || !ws || B <= 0 || Kmax <= 0 ||
        (Kmax % CLS_ROWS) || D <= 0 || C != 2)
        return WESUP_ERR_INVALID;
    const int R = B * Kmax;
    if (ws_bytes < wesup_classifier_bwd_workspace_bytes(R, D)) return WESUP_ERR_WORKSPACE;
    WESUP_LAUNCH(head_bwd_kernel, dim3(R / CLS_ROWS), dim3(256), 0, (hipStream_t)stream, feat, Wc, pred, y_all, n_sp, n_l, dloss,
                 eps, prop_weight, terms, dpred, dfeat, (float*)ws, B, Kmax, D);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_classifier_bwd_finish(const void* ws, size_t ws_bytes, float* dWc, float* dbc, int R, int D, void* stream) {
    if (!ws || !dWc || !dbc || R <= 0 || D <= 0) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_classifier_bwd_workspace_bytes(R, D)) return WESUP_ERR_WORKSPACE;
    WESUP_LAUNCH(classifier_bwd_reduce, dim3(ceil_div(2 * D + 2, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)ws, dWc,
                 dbc, ceil_div(R, CLS_ROWS), D);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// generic _cross_entropy on (n, C)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ y_hat, const float* __restrict__ y_true,
                                                     const float* __restrict__ cw, float eps, float* __restrict__ out2,
                                                     int n, int C) {
    __shared__ float sh[256];
    float s = 0.f, cnt = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) {
        float ys;
        s += ce_row(y_hat + (long)r * C, y_true + (long)r * C, C, eps, &ys, cw);
        cnt += (ys > 0.f) ? 1.f : 0.f;
    }
    s = block_sum256(s, sh);
    cnt = block_sum256(cnt, sh);
    if (threadIdx.x == 0) {
        out2[0] = s;
        out2[1] = cnt;
        out2[2] = (cnt > 0.f) ? s / cnt : 0.f;      // models/wesup.py:88-96
        out2[3] = 0.f;
    }
}
__global__ void ce_bwd_kernel(const float* __restrict__ y_hat, const float* __restrict__ y_true,
                              const float* __restrict__ cw, const float* __restrict__ out2,
                              const float* __restrict__ dloss, float eps, float* __restrict__ dy, long total, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float cnt = out2[1];
    const float p = y_hat[idx];
    float g = 0.f;
    if (cnt > 0.f && p >= eps && p <= 1.f - eps) g = dloss[0] * (-y_true[idx] / p) / cnt;
    if (cw) g *= cw[idx % C];
    dy[idx] = g;
}
extern "C" int wesup_cross_entropy_fwd(const float* y_hat, const float* y_true, const float* class_weights, float eps,
                                       float* out2, int n, int C, void* stream) {
    if (!y_hat || !y_true || !out2 || n < 0 || C <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(ce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, y_hat, y_true, class_weights, eps, out2,
                       n, C);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_cross_entropy_bwd(const float* y_hat, const float* y_true, const float* class_weights,
                                       const float* out2, const float* dloss, float eps, float* dy_hat, int n, int C,
                                       void* stream) {
    if (!y_hat || !y_true || !out2 || !dloss || !dy_hat || n <= 0 || C <= 0) return WESUP_ERR_INVALID;
    const long total = (long)n * C;
    WESUP_LAUNCH(ce_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y_hat,
                       y_true, class_weights, out2, dloss, eps, dy_hat, total, C);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ SGD with momentum + weight decay
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, size_t n, float lr,
                           float mu, float wd, float gs, int first) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = ld4(p + 4 * i);
        const float4 gg = ld4(g + 4 * i);
        float4 vv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(v + 4 * i);
        const float gx = gg.x * gs + wd * pp.x, gy = gg.y * gs + wd * pp.y;
        const float gz = gg.z * gs + wd * pp.z, gw = gg.w * gs + wd * pp.w;
        vv.x = first ? gx : mu * vv.x + gx;
        vv.y = first ? gy : mu * vv.y + gy;
        vv.z = first ? gz : mu * vv.z + gz;
        vv.w = first ? gw : mu * vv.w + gw;
        pp.x -= lr * vv.x;
        pp.y -= lr * vv.y;
        pp.z -= lr * vv.z;
        pp.w -= lr * vv.w;
        st4(v + 4 * i, vv);
        st4(p + 4 * i, pp);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float gi = g[i] * gs + wd * p[i];
        const float vi = first ? gi : mu * v[i] + gi;
        v[i] = vi;
        p[i] -= lr * vi;
    }
}
extern "C" int wesup_sgd_step(float* p, const float* g, float* v, size_t n, float lr, float momentum, float weight_decay,
                              float grad_scale, int first_step, void* stream) {
    if (!p || !g || !v || n == 0) return WESUP_ERR_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)v) & 15) return WESUP_ERR_INVALID;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    WESUP_LAUNCH(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, v, n, lr, momentum,
                       weight_decay, grad_scale, first_step);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ accuracy / dice sums
// SEG_BLOCKS blocks per image write partial sums; a second tiny kernel adds them in a fixed order.
#define SEG_BLOCKS 64
__global__ __launch_bounds__(256) void seg_metrics_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ part, int HW, int C) {
    __shared__ float sh[256];
    const int b = blockIdx.y;
    float eq = 0.f, pg = 0.f, sp = 0.f, sg = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += SEG_BLOCKS * 256) {
        const float P = rintf(pred[(long)b * HW + p]);        // round half to even, as torch.round
        int gi = 0;
        uint8_t best = mask[((long)b * C) * HW + p];
        for (int c = 1; c < C; ++c) {
            const uint8_t v = mask[((long)b * C + c) * HW + p];
            if (v > best) { best = v; gi = c; }
        }
        const float G = (float)gi;
        eq += (P == G) ? 1.f : 0.f;
        pg += P * G;
        sp += P;
        sg += G;
    }
    eq = block_sum256(eq, sh);
    pg = block_sum256(pg, sh);
    sp = block_sum256(sp, sh);
    sg = block_sum256(sg, sh);
    if (threadIdx.x == 0) {
        float* o = part + ((long)b * SEG_BLOCKS + blockIdx.x) * 4;
        o[0] = eq; o[1] = pg; o[2] = sp; o[3] = sg;
    }
}
__global__ void seg_metrics_reduce(const float* __restrict__ part, float* __restrict__ out4, int B) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 4) return;
    const int b = idx >> 2, k = idx & 3;
    float s = 0.f;
    for (int i = 0; i < SEG_BLOCKS; ++i) s += part[((long)b * SEG_BLOCKS + i) * 4 + k];
    out4[idx] = s;
}
extern "C" size_t wesup_seg_metrics_workspace_bytes(int B) { return B > 0 ? (size_t)B * SEG_BLOCKS * 4 * sizeof(float) : 0; }
extern "C" int wesup_seg_metrics(const float* pred, const uint8_t* mask, float* out4, int B, int HW, int C, void* ws,
                                 size_t ws_bytes, void* stream) {
    if (!pred || !mask || !out4 || !ws || B <= 0 || HW <= 0 || C <= 0) return WESUP_ERR_INVALID;
    if (ws_bytes < (size_t)B * SEG_BLOCKS * 4 * sizeof(float)) return WESUP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    WESUP_LAUNCH(seg_metrics_kernel, dim3(SEG_BLOCKS, B), dim3(256), 0, st, pred, mask, (float*)ws, HW, C);
    WESUP_LAUNCH(seg_metrics_reduce, dim3(ceil_div(B * 4, 64)), dim3(64), 0, st, (const float*)ws, out4, B);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ misc
extern "C" int wesup_abi_version(void) { return 6; }
extern "C" const char* wesup_strerror(int code) {
    switch (code) {
        case WESUP_OK: return "ok";
        case WESUP_ERR_INVALID: return "invalid argument (shape, alignment or null pointer)";
        case WESUP_ERR_LAUNCH: return "HIP kernel launch failed";
        case WESUP_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
