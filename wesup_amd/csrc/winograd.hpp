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

// internal entry points of winograd.hip used by the weight-gradient pass in gemm.hip (not part of the C ABI): the F(4x4)
// bias gradient travels as per-block rows from the outgrad transform to the filter-gradient reduce, with no launch of its own
long wino4_bias_rows(int B, int H, int W, int C);          // 0: C / 4 does not divide 256 (use wesup_colsum on dy instead)
int wino_outgrad_launch(const float* dy, float* dM, float* bias_part, int B, int H, int W, int C, int m, void* stream);
int wino_filter_grad_launch(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs, float* db, int Cout,
                            int Cin, int m, const float* bias_part, int bias_rows, void* stream);
