// Shared by gemm.hip and winograd.hip: tile counting and the shape limits of the Winograd-domain entries.
#pragma once
#include "common.hpp"

static inline long wino_tiles(int B, int H, int W) { return (long)B * ((H + 1) / 2) * ((W + 1) / 2); }
static inline bool wino_shape_ok(int B, int H, int W, int Ci, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Ci < 32 || Cout < 32 || (Ci % 4) || (Cout % 4)) return false;
    const long T = wino_tiles(B, H, W);
    const long cmax = Ci > Cout ? Ci : Cout;
    // thread index and the FastDiv range (n * d < 2^40, quotient < 2^24)
    return T < (1l << 24) && T * (cmax / 4) < (1l << 31) && T * (cmax / 4) * (cmax / 4) < (1l << 40);
}
