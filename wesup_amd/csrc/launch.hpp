// Every kernel launch of the library goes through WESUP_LAUNCH: it launches, and -- while the calling thread records a step
// plan (plan.hip, wesup_plan_begin) -- also appends the launch (kernel, grid, block, LDS bytes, stream, a copy of the
// arguments) to that plan.  wesup_plan_replay later re-issues the recorded launches straight from C: the host logic of the
// entries (shape checks, tile selection, workspace carving) and the Python walk above them run once per shape, not once per
// step.  Replaces the hipGraph capture a tracing runtime would use (measured: this runtime replays a three-stream graph at
// half the rate of eager launches).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <tuple>
#include <type_traits>
#include <utility>

struct WesupPlan;
// plan.hip: the plan this thread is recording into (NULL: none), and the appenders
WesupPlan* wesup_plan_recording_();
void wesup_plan_add_kernel_(WesupPlan* plan, const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, const void* blob,
                            size_t blob_bytes, const unsigned* offs, int nargs);

template <typename Tuple, size_t... I>
static inline void wesup_arg_offsets_(const Tuple& t, unsigned* offs, std::index_sequence<I...>) {
    ((offs[I] = (unsigned)((const char*)&std::get<I>(t) - (const char*)&t)), ...);
}

template <typename... KA, typename... A>
static inline void wesup_launch(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... a) {
    static_assert(sizeof...(KA) == sizeof...(A), "argument count differs from the kernel's parameter list");
    static_assert((std::is_trivially_copyable<std::decay_t<KA>>::value && ...), "kernel parameters are copied byte-wise into a plan");
    WesupPlan* rec = wesup_plan_recording_();
    if (rec) {
        const std::tuple<std::decay_t<KA>...> args(static_cast<std::decay_t<KA>>(a)...);
        unsigned offs[sizeof...(KA) > 0 ? sizeof...(KA) : 1];
        wesup_arg_offsets_(args, offs, std::index_sequence_for<KA...>{});
        wesup_plan_add_kernel_(rec, reinterpret_cast<const void*>(kern), grid, block, lds, st, &args, sizeof(args), offs,
                               (int)sizeof...(KA));
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<std::decay_t<KA>>(a)...);
}
#define WESUP_LAUNCH(kern, grid, block, lds, st, ...) wesup_launch(kern, grid, block, lds, st, __VA_ARGS__)

// hipMemsetAsync of whole 32-bit words as a kernel (a memset is a runtime node of its own kind; as a kernel it records and
// replays like every other launch).  common.hpp's callers: ptr 4-byte aligned, bytes a multiple of 4.
int wesup_fill_words_(void* ptr, unsigned value, size_t words, hipStream_t st);
