// LDS-DMA staging helpers of the MFMA GEMM kernels (gemm.hip, wino_fused.hip): a wave-instruction copies 64 x 16 B from
// per-lane addresses straight into 1 KiB of LDS, no staging registers and no ds_write.
#pragma once
#include "common.hpp"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA through inline asm: hipcc cannot prove that the ds_reads of the current tile do not alias the DMA
// destination (same __shared__ array, runtime buffer index) and, for the builtin form, parks an s_waitcnt vmcnt(0)
// right behind the DMA issue -- the whole DMA latency in front of the MFMA phase, every K-step.  An asm statement
// is outside its wait bookkeeping (cdna_hip_programming.md 5.7): the only wait is glds_wait() placed by hand in front
// of the barrier that ends the K-step.  M0 carries the wave-uniform LDS byte address and is restored.
__device__ __forceinline__ void glds16(const float* src, unsigned lds_byte_addr_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte_addr_uniform)
                 : "memory");
}
// Buffer form of the same DMA (buffer_load_dwordx4 ... lds): the base lives in a scalar resource descriptor, the lane
// supplies one 32-bit byte offset and the K-step adds a scalar offset, so a staging instruction needs no vector
// address arithmetic at all (the 64-bit global form needed an add-with-carry and two selects per instruction), and
// a lane whose offset is >= num_records gets ZEROS written to LDS -- implicit zero padding without a zero page
// (round-2 probe bufdma_probe.hip, tools/README.md: out-of-range lanes store 0, the scalar offset takes part in the range check).
// round-2 probe gemm_lab.hip, tools/README.md: +2.4 % on the 128x128 two-blocks-per-CU loop.
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define WESUP_OOB 0x80000000u            // per-lane offset of a masked lane; every descriptor has num_records <= 2 GiB
__device__ __forceinline__ i32x4 make_srd(const float* base, unsigned num_records = WESUP_OOB) {
    const unsigned long a = (unsigned long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)num_records);
    r[3] = 0x00020000;
    return r;
}
// M0 contract.  The DMA takes its LDS base from M0; the asm writes M0 and LEAVES it (saving and restoring it around
// every DMA cost 0.7 %), and says so in the clobber list.  M0 is a register hipcc reserves, hence its warning
// "inline asm clobber list contains reserved registers: m0" (472 of them, one per inlined copy) -- silenced here on
// purpose: the clobber is the truth, and what makes it safe is that nothing else in these kernels READS M0 (gfx9 LDS
// instructions do not; the GEMM loops have no indirect register indexing, no s_movrel / v_movrel, no GWS, no
// sendmsg).  tests/test_isa_cpu.py disassembles the built library and fails if any GEMM kernel gains an M0 reader
// other than the LDS-DMA itself (or a scratch access, or loses an MFMA of its unrolled K-step).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void bglds16(unsigned voff, i32x4 srd, unsigned soff_uniform, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(srd), "s"(soff_uniform), "s"(lds_byte_addr_uniform)
                 : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(unsigned long)(lptr_t)p;      // LDS byte offset of a pointer into __shared__ memory
}

