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
void wesup_plan_add_kernel_(WesupPlan* plan, const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t st,
                            const void* const* args, const unsigned* sizes, int nargs);

template <typename Tuple, size_t... I>
static inline void wesup_arg_pointers_(const Tuple& t, const void** ptrs, std::index_sequence<I...>) {
    ((ptrs[I] = &std::get<I>(t)), ...);
}

// A plan is compared with a second recording of the same step byte by byte (wesup_plan_diff), so an argument copy must not
// carry indeterminate bytes: every by-value kernel parameter is a scalar, a pointer, or a struct WITHOUT padding (explicit
// `pad_` members where the layout has holes).  Structs with float members cannot be checked by the compiler
// (has_unique_object_representations is false for floating point): they opt in with WESUP_NO_PADDING(T, bytes-of-members).
template <typename T>
struct wesup_no_padding : std::integral_constant<bool, std::is_scalar<T>::value || std::has_unique_object_representations<T>::value> {};
#define WESUP_NO_PADDING(T, member_bytes)                                                         \
    static_assert(sizeof(T) == (member_bytes), #T " has padding: add explicit pad_ members");    \
    template <>                                                                                   \
    struct wesup_no_padding<T> : std::true_type {}

template <typename... KA, typename... A>
static inline void wesup_launch(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... a) {
    static_assert(sizeof...(KA) == sizeof...(A), "argument count differs from the kernel's parameter list");
    static_assert((std::is_trivially_copyable<std::decay_t<KA>>::value && ...), "kernel parameters are copied byte-wise into a plan");
    static_assert((wesup_no_padding<std::decay_t<KA>>::value && ...), "a by-value kernel parameter has padding bytes (launch.hpp)");
    WesupPlan* rec = wesup_plan_recording_();
    if (rec) {
        // the arguments as the kernel receives them (converted to its parameter types), copied one by one: the plan lays them
        // out itself, so no byte between two arguments is left to chance either
        const std::tuple<std::decay_t<KA>...> args(static_cast<std::decay_t<KA>>(a)...);
        const void* ptrs[sizeof...(KA) > 0 ? sizeof...(KA) : 1];
        const unsigned sizes[sizeof...(KA) > 0 ? sizeof...(KA) : 1] = {(unsigned)sizeof(std::decay_t<KA>)...};
        wesup_arg_pointers_(args, ptrs, std::index_sequence_for<KA...>{});
        wesup_plan_add_kernel_(rec, reinterpret_cast<const void*>(kern), grid, block, lds, st, ptrs, sizes, (int)sizeof...(KA));
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<std::decay_t<KA>>(a)...);
}
#define WESUP_LAUNCH(kern, grid, block, lds, st, ...) wesup_launch(kern, grid, block, lds, st, __VA_ARGS__)

// hipMemsetAsync of whole 32-bit words as a kernel (a memset is a runtime node of its own kind; as a kernel it records and
// replays like every other launch).  common.hpp's callers: ptr 4-byte aligned, bytes a multiple of 4.
int wesup_fill_words_(void* ptr, unsigned value, size_t words, hipStream_t st);
