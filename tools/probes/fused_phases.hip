// Where the one-kernel Winograd product route (csrc/wino_fused.hip) spends a launch: per block the start, the end of the 36 K/64
// product stages and the end of the epilogue (shader clock), with the CU / wave slot it ran on.  Includes the product source with
// WESUP_FUSED_TRACE defined; the library itself is never built that way.
//   gpurun -- 'bash tools/fused_phases.sh'      (compiles against csrc/plan.o of the built library; shapes: K N H B, default 64 64 480 4
//                                               = conv1_2 at configs[1])
#define WESUP_FUSED_TRACE 1
#include "../../wesup_amd/csrc/wino_fused.hip"
#include <vector>
#include <algorithm>
#include <map>

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 64, H = argc > 3 ? atoi(argv[3]) : 480, B = argc > 4 ? atoi(argv[4]) : 4;
    const int W = H;
    const long T = wino_tiles(B, H, W, 4);
    float *V, *U, *y, *bias;
    hipMalloc(&V, 36 * T * K * 4); hipMalloc(&U, 36l * N * K * 4); hipMalloc(&y, (long)B * H * W * N * 4); hipMalloc(&bias, N * 4);
    hipMemset(V, 0, 36 * T * K * 4); hipMemset(U, 0, 36l * N * K * 4); hipMemset(bias, 0, N * 4);
    const long blocks = ((T + 31) / 32) * (N / 64);
    unsigned long long* tr;
    hipMalloc(&tr, blocks * 9 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_fused_trace), &tr, sizeof(tr));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        int rc = wesup_winograd_gemm_output_transform(V, 0, U, bias, nullptr, y, nullptr, 0, nullptr, nullptr, 0, 0, B, H, W, K, N, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    std::vector<unsigned long long> h(blocks * 9);
    hipMemcpy(h.data(), tr, blocks * 9 * 8, hipMemcpyDeviceToHost);
    unsigned long long t_min = ~0ull, t_max = 0;
    double main_sum = 0, epi_sum = 0;
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;      // (xcc, se, cu) -> block intervals
    for (long b = 0; b < blocks; ++b) {
        const unsigned long long* t = &h[b * 9];
        t_min = std::min(t_min, t[0]); t_max = std::max(t_max, t[2]);
        main_sum += (double)(t[1] - t[0]); epi_sum += (double)(t[2] - t[1]);
        const unsigned hw = (unsigned)t[3], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = (unsigned)t[4] & 15;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({t[0], t[1]});
        per_cu[(1u << 20) | (xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({t[1], t[2]});
    }
    const double span = (double)(t_max - t_min);
    printf("K=%d N=%d %dx%d B=%d: %ld blocks, %.1f us by events; first start to last end %.1f us\n", K, N, H, W, B, blocks, ms * 1e3, span * 0.01);
    printf("per block: products %.1f us, epilogue %.1f us (%.0f %% of a block)\n", main_sum / blocks * 0.01, epi_sum / blocks * 0.01,
           100.0 * epi_sum / (main_sum + epi_sum));
    {   // the epilogue's own phases (thread 0's clock): image of rows 0-1 written + barrier | its stores issued | barrier + image of rows 2-3 + barrier | stores | end
        double d[5] = {0, 0, 0, 0, 0};
        for (long b = 0; b < blocks; ++b) {
            const unsigned long long* x = &h[b * 9];
            d[0] += (double)(x[5] - x[1]); d[1] += (double)(x[6] - x[5]); d[2] += (double)(x[7] - x[6]); d[3] += (double)(x[8] - x[7]); d[4] += (double)(x[2] - x[8]);
        }
        printf("epilogue phases, us: image 0 %.2f | stores 0 %.2f | image 1 %.2f | stores 1 %.2f | tail %.2f\n", d[0] / blocks * 0.01, d[1] / blocks * 0.01,
               d[2] / blocks * 0.01, d[3] / blocks * 0.01, d[4] / blocks * 0.01);
    }
    {   // chip-wide: how many blocks are in their epilogue at a time (20 samples over the span)
        printf("blocks in products / in epilogue at 5 %% steps of the span:");
        for (int k = 1; k < 20; ++k) {
            const unsigned long long t = t_min + (unsigned long long)(span * k / 20.0);
            int np = 0, ne = 0;
            for (long b = 0; b < blocks; ++b) {
                const unsigned long long* x = &h[b * 9];
                if (x[0] <= t && t < x[1]) ++np; else if (x[1] <= t && t < x[2]) ++ne;
            }
            printf(" %d/%d", np, ne);
        }
        printf("\n");
    }
    // per CU: ticks during which 0 / 1 / 2 blocks are in their product stages
    double in0 = 0, in1 = 0, in2 = 0, ncu = 0;
    for (auto& kv : per_cu) {
        if (kv.first >> 20) continue;
        std::vector<std::pair<unsigned long long, int>> ev;
        for (auto& iv : kv.second) { ev.push_back({iv.first, +1}); ev.push_back({iv.second, -1}); }
        std::sort(ev.begin(), ev.end());
        unsigned long long last = t_min; int depth = 0;
        for (auto& e : ev) {
            const double d = (double)(e.first - last);
            (depth == 0 ? in0 : depth == 1 ? in1 : in2) += d;
            depth += e.second; last = e.first;
        }
        in0 += (double)(t_max - last);
        ncu += 1;
    }
    printf("CUs seen %.0f; of the launch's span a CU had 0 / 1 / 2+ blocks in their product stages for %.0f / %.0f / %.0f %%\n", ncu,
           100 * in0 / (ncu * span), 100 * in1 / (ncu * span), 100 * in2 / (ncu * span));
    printf("MFMA floor of the products: %ld blocks x 36 x %d stages x 1024 cycles x 4 waves / 1024 SIMDs = %.0f cycles = %.1f us at 2.1 GHz\n", blocks,
           K / 64, (double)blocks * 36 * (K / 64) * 1024 * 4 / 1024, (double)blocks * 36 * (K / 64) * 1024 * 4 / 1024 / 2100.0);
    return 0;
}
