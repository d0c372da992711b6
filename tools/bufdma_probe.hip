// Probe of the buffer form of the LDS-DMA on gfx950: what lands in LDS for a lane whose offset is out of range, and
// whether the scalar offset takes part in the range check.   hipcc -O3 --offload-arch=gfx950 tools/bufdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const float* src, float* out, unsigned num_records, unsigned soff, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = -7.f;
    __syncthreads();
    const unsigned long a = (unsigned long)src;
    i32x4 srd;
    srd[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    srd[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    srd[2] = __builtin_amdgcn_readfirstlane((int)num_records);
    srd[3] = 0x00020000;
    unsigned voff = lane * 16;
    if (mode == 1 && (lane & 1)) voff = 0x80000000u;          // odd lanes out of range by their own offset
    const unsigned ldsa = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lptr_t)lds);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)"
                 ::"v"(voff), "s"(srd), "s"(so), "s"(ldsa) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out, h[4096], o[256];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    (void)hipMalloc(&src, sizeof(h)); (void)hipMalloc(&out, sizeof(o));
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    struct { unsigned nr, so; int mode; const char* what; } cases[] = {
        {16384, 0, 0, "all in range"},
        {16384, 0, 1, "odd lanes voffset 0x80000000"},
        {512, 0, 0, "num_records 512 B: lanes >= 32 out of range by voffset"},
        {1024, 512, 0, "num_records 1024, soffset 512: lanes 32..63 in range by voffset, out by voffset+soffset"},
        {16384, 4096, 0, "soffset 4096 in range"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, c.nr, c.so, c.mode);
        (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        printf("%s\n  lane0: %g %g | lane1: %g %g | lane31: %g | lane32: %g | lane33: %g | lane63: %g %g\n", c.what, o[0], o[1], o[4], o[5],
               o[124], o[128], o[132], o[252], o[255]);
    }
    return 0;
}
