// The entry points SURVEY.md 8(b) lists by name ("minimum set" of the C ABI) that the step itself reaches through more
// general entries: the 1x1 side convolution and the fc layers are wesup_gemm_nt / wesup_gemm_tn, the dense bilinear
// up-sampling is wesup_upsample_fwd / bwd, the per-superpixel statistics are the first stage of wesup_sp_preprocess, the
// fused softmax + cross entropy is wesup_classifier_fwd followed by wesup_cross_entropy_fwd.  A reference-side binding
// that wants exactly one call per ATen op it replaces (models/wesup.py:208-209,253 Conv2d 1x1; :213-220 Linear;
// :254-255 F.interpolate; :34-42 the per-id mask sums; :66-96 with :231 Softmax) binds these.
#include "common.hpp"

// ------------------------------------------------------------------ K9: per-superpixel area and class counts
__global__ void sp_stats_kernel(const int32_t* __restrict__ labels, const uint8_t* __restrict__ mask,
                                int32_t* __restrict__ area, int32_t* __restrict__ counts, int32_t* __restrict__ status,
                                int HW, int C, int Kmax) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int id = labels[(long)b * HW + p];
    if (id < 0 || id >= Kmax) { atomicOr(status + b, 1); return; }
    atomicAdd(area + (long)b * Kmax + id, 1);                      // integer atomics: order-independent
    if (mask)
        for (int c = 0; c < C; ++c)
            if (mask[((long)b * C + c) * HW + p]) atomicAdd(counts + ((long)b * Kmax + id) * C + c, 1);
}
extern "C" int wesup_sp_stats(const int32_t* labels, const uint8_t* mask, int B, int HW, int C, int Kmax, int32_t* area,
                              int32_t* counts, int32_t* status, void* stream) {
    if (!labels || !area || !status || B <= 0 || HW <= 0 || Kmax <= 0 || C <= 0 || B > 65535) return WESUP_ERR_INVALID;
    if (mask && !counts) return WESUP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (wesup_fill_words_(area, 0u, (sizeof(int32_t) * (size_t)B * Kmax) / 4, st) != WESUP_OK) return WESUP_ERR_LAUNCH;
    if (wesup_fill_words_(status, 0u, (sizeof(int32_t) * (size_t)B) / 4, st) != WESUP_OK) return WESUP_ERR_LAUNCH;
    if (counts && wesup_fill_words_(counts, 0u, (sizeof(int32_t) * (size_t)B * Kmax * C) / 4, st) != WESUP_OK) return WESUP_ERR_LAUNCH;
    WESUP_LAUNCH(sp_stats_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, st, labels, mask, area, counts,
                       status, HW, C, Kmax);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ------------------------------------------------------------------ K3: 1x1 convolution on NHWC = GEMM over pixels
extern "C" size_t wesup_conv1x1_workspace_bytes(int P, int Cin, int Cout) {
    if (P <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const size_t a = wesup_gemm_tn_workspace_bytes(Cout, Cin, P), b = wesup_gemm_nt_workspace_bytes(P, Cout, Cin);
    const size_t c = wesup_gemm_nt_workspace_bytes(P, Cin, Cout);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
extern "C" int wesup_conv1x1_fwd(const float* x, const float* w, const float* bias, float* y, int P, int Cin, int Cout,
                                 void* ws, size_t ws_bytes, void* stream) {
    return wesup_gemm_nt(x, Cin, w, Cin, bias, y, Cout, nullptr, 0, P, Cout, Cin, 0, ws,
                         ws_bytes >= wesup_gemm_nt_workspace_bytes(P, Cout, Cin) ? wesup_gemm_nt_workspace_bytes(P, Cout, Cin) : 0,
                         stream);
}
// dx[P][Cin] (+)= dy[P][Cout] . w   with w_t = w transposed, [Cin][Cout] (wesup_transpose makes it)
extern "C" int wesup_conv1x1_dgrad(const float* dy, const float* w_t, float* dx, int P, int Cin, int Cout, int accumulate,
                                   void* ws, size_t ws_bytes, void* stream) {
    return wesup_gemm_nt(dy, Cout, w_t, Cout, nullptr, dx, Cin, nullptr, 0, P, Cin, Cout, accumulate ? WESUP_ACCUM : 0, ws,
                         ws_bytes >= wesup_gemm_nt_workspace_bytes(P, Cin, Cout) ? wesup_gemm_nt_workspace_bytes(P, Cin, Cout) : 0,
                         stream);
}
// dw[Cout][Cin] = dy^T . x ; db[Cout] = column sums of dy (NULL: skipped); ws >= wesup_conv1x1_workspace_bytes
extern "C" int wesup_conv1x1_wgrad(const float* dy, const float* x, float* dw, float* db, int P, int Cin, int Cout,
                                   void* ws, size_t ws_bytes, void* stream) {
    return wesup_gemm_tn(dy, Cout, x, Cin, dw, Cin, db, Cout, Cin, P, 0, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------ K7: Linear (+ ReLU)
extern "C" size_t wesup_linear_workspace_bytes(int R, int In, int Out) { return wesup_conv1x1_workspace_bytes(R, In, Out); }
extern "C" int wesup_linear_fwd(const float* x, const float* w, const float* bias, float* y, int R, int In, int Out,
                                int relu, void* ws, size_t ws_bytes, void* stream) {
    return wesup_gemm_nt(x, In, w, In, bias, y, Out, nullptr, 0, R, Out, In, relu ? WESUP_RELU_OUT : 0, ws,
                         ws_bytes >= wesup_gemm_nt_workspace_bytes(R, Out, In) ? wesup_gemm_nt_workspace_bytes(R, Out, In) : 0,
                         stream);
}
// dx = dy . w (masked by relu_src > 0 when the layer's INPUT came out of a ReLU: relu_src = that input), dw = dy^T . x,
// db = column sums of dy.  w_t = w transposed, [In][Out].  dx may be NULL (first layer).
extern "C" int wesup_linear_bwd(const float* dy, const float* x, const float* w_t, const float* relu_src, float* dx,
                                float* dw, float* db, int R, int In, int Out, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !dw) return WESUP_ERR_INVALID;
    int rc = wesup_gemm_tn(dy, Out, x, In, dw, In, db, Out, In, R, 0, ws, ws_bytes, stream);
    if (rc || !dx) return rc;
    if (!w_t) return WESUP_ERR_INVALID;
    return wesup_gemm_nt(dy, Out, w_t, Out, nullptr, dx, In, relu_src, In, R, In, Out, relu_src ? WESUP_MASK : 0, nullptr, 0,
                         stream);
}

// ------------------------------------------------------------------ K4: F.interpolate(bilinear, align_corners=True)
extern "C" int wesup_upsample_bilinear_ac_fwd(const float* s, float* out, int B, int h, int w, int H, int W, int C,
                                              int ld_out, int coff, void* stream) {
    return wesup_upsample_fwd(s, out, B, h, w, H, W, C, ld_out, coff, stream);
}
extern "C" int wesup_upsample_bilinear_ac_bwd(const float* dout, float* ds, int B, int h, int w, int H, int W, int C,
                                              int ld_out, int coff, void* stream) {
    return wesup_upsample_bwd(dout, nullptr, nullptr, ds, B, h, w, H, W, C, ld_out, coff, 0, stream);
}

// ------------------------------------------------------------------ K11: Softmax(dim=1) + _cross_entropy, fused
// probs = softmax(logits); out2 = {sum(-y log clamp(probs) [* cw]), #rows with sum(y) > 0, loss, 0} (models/wesup.py:66-96)
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                             const float* __restrict__ cw, float eps,
                                                             float* __restrict__ probs, float* __restrict__ out2, int n, int C) {
    __shared__ float sh[2][256];
    float s = 0.f, cnt = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) {
        const float* z = logits + (long)r * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[c] - m);
        float ys = 0.f;
        for (int c = 0; c < C; ++c) {
            const float pc = expf(z[c] - m) / den;
            probs[(long)r * C + c] = pc;
            const float yc = y[(long)r * C + c];
            ys += yc;
            const float q = (pc != pc) ? pc : fminf(fmaxf(pc, eps), 1.f - eps);
            const float ce = -yc * logf(q);
            s += cw ? ce * cw[c] : ce;
        }
        cnt += ys > 0.f ? 1.f : 0.f;
    }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {                      // fixed tree: deterministic
        if (threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = sh[0][0]; out2[1] = sh[1][0];
        out2[2] = sh[1][0] > 0.f ? sh[0][0] / sh[1][0] : 0.f;
        out2[3] = 0.f;
    }
}
// dlogits = dloss * d loss / d logits: through the clamp (zero outside [eps, 1 - eps]) and the softmax Jacobian
__global__ void softmax_ce_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ y,
                                      const float* __restrict__ cw, const float* __restrict__ out2,
                                      const float* __restrict__ dloss, float eps, float* __restrict__ dlogits, int n, int C) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float cnt = out2[1];
    const float* p = probs + (long)r * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
        float g = 0.f;
        if (cnt > 0.f && p[c] >= eps && p[c] <= 1.f - eps) g = dloss[0] * (-y[(long)r * C + c] / p[c]) / cnt * (cw ? cw[c] : 1.f);
        dot += g * p[c];
    }
    for (int c = 0; c < C; ++c) {
        float g = 0.f;
        if (cnt > 0.f && p[c] >= eps && p[c] <= 1.f - eps) g = dloss[0] * (-y[(long)r * C + c] / p[c]) / cnt * (cw ? cw[c] : 1.f);
        dlogits[(long)r * C + c] = p[c] * (g - dot);
    }
}
extern "C" int wesup_softmax_ce_fwd(const float* logits, const float* y_true, const float* class_weights, float eps,
                                    float* probs, float* out2, int n, int C, void* stream) {
    if (!logits || !y_true || !probs || !out2 || n < 0 || C <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(softmax_ce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, y_true, class_weights, eps,
                       probs, out2, n, C);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_softmax_ce_bwd(const float* probs, const float* y_true, const float* class_weights, const float* out2,
                                    const float* dloss, float eps, float* dlogits, int n, int C, void* stream) {
    if (!probs || !y_true || !out2 || !dloss || !dlogits || n <= 0 || C <= 0) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(softmax_ce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs,
                       y_true, class_weights, out2, dloss, eps, dlogits, n, C);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
