// Shared helpers for the gfx950 kernels of libwesup_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/wesup_hip.h"
#include "launch.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WESUP_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return WESUP_ERR_LAUNCH; \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Exact n / d for n*d < 2^40 via one 64-bit multiply (n < 2^26 pixels, d < 2^12 here).
struct FastDiv {
    unsigned long long m;
    int d;
    int pad_;            // (explicit: by-value kernel parameters carry no indeterminate bytes, launch.hpp)
};
static inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.pad_ = 0;
    f.d = d;
    f.m = ((1ull << 40) + (unsigned long long)d - 1) / (unsigned long long)d;
    return f;
}
__device__ __forceinline__ int fast_div(int n, const FastDiv f) {
    return (int)(((unsigned long long)(unsigned)n * f.m) >> 40);
}

// Bijective XCD-aware block remap (cdna_hip_programming.md 5, "XCD swizzle must be bijective"):
// blocks b and b+8 share an XCD; give each XCD a contiguous range of logical tiles so that
// neighbouring tiles (which share activation rows / halos) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// ReLU-on-load in the GEMM loops as ONE vector instruction: a signed-integer max of the float's bits with 0 (a float
// with the sign bit clear is a non-negative integer and keeps its bits, one with the sign bit set -- negative values
// and -0 -- is a negative integer and becomes +0).  fmaxf() and even __builtin_amdgcn_fmed3f(x, 0, inf) compile to
// TWO v_max_f32 (llvm.maxnum canonicalises a possibly-signalling input first): 64 instead of 32 vector instructions
// per K-step in both GEMM loops, and vector instructions do not overlap with MFMAs on a SIMD.  (An asm v_max would be
// outside the compiler's s_waitcnt bookkeeping and read a fragment before its ds_read has landed.)
// NaN: the integer view makes the result depend on the NaN's SIGN bit -- a NaN with the sign clear (what 0/0-free fp32
// arithmetic produces on this hardware: 0x7fc00000) is a large positive integer and survives, as in torch.relu; a NaN
// with the sign set (0xffc00000: a negated NaN, or one loaded from a host-made checkpoint / input) is a negative integer
// and becomes +0.  The training step does not depend on it: ReLU-on-load is off its default path (relu_on_store: the
// producing kernel's epilogue applies relu1() below, which keeps both), and a NaN conv output reaches the loss through
// the pre-ReLU side tap either way.  tests/test_kernels_gpu.py::test_relu_on_load_nan_semantics pins this behaviour.
__device__ __forceinline__ float vmax1(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}
// ReLU of an epilogue.  NaN-preserving like torch.relu (fmaxf would turn a NaN into 0): a NaN in any conv output
// reaches the superpixel features through the pre-ReLU side tap, and from there it must survive the fc_layers' ReLUs
// so that the loss is NaN and the trainer raises before the weights are touched (models/base.py:202-203).
__device__ __forceinline__ float relu1(float x) { return x < 0.f ? 0.f : x; }
__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)); }

#include <cstdlib>
// Streaming stores.  A transformed tensor of the big layers is 2.25 x its activation (conv1_2 at B = 4, 480 x 480: 531 MB) --
// larger than the 256 MB memory-side cache -- and is read once, by a later kernel: written with ordinary stores it only evicts
// what the next kernels would have found there, and the write stream itself runs at half the copy rate (input transform of
// conv1_2 alone 231 us = 3.3 TB/s; with non-temporal stores 144 us = 5.3 TB/s; 240^2 x 128: 112 -> 79 us).  A tensor that
// is small (<= WESUP_NT_MB per launch, default 16 MB) keeps ordinary stores: its consumer finds it in L2.  Alone on the GPU a
// 133 MB tensor is better off with ordinary stores (42 vs 47 us: it fits the memory-side cache); inside the step, where
// three streams stream through that cache, the low bar wins (threshold 100000 / 160 / 64 / 16 MB: 9.03 / 8.97 / 8.97 / 8.93 ms).
__device__ __forceinline__ void st4s(float* p, float4 v, int nt) {
    if (nt) {
        f32x4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
    } else st4(p, v);
}
static inline int wino_nt_stores(double bytes_written) {
    static const double limit = [] { const char* e = getenv("WESUP_NT_MB"); return (e ? atof(e) : 16.0) * 1048576.0; }();
    return bytes_written > limit ? 1 : 0;
}
