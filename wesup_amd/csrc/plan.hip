// Step plans: record the launches of one training iteration once, replay them from C (include/wesup_hip.h, "step plans").
//
// The reference walks its iteration in Python every step (models/base.py:184-211: preprocess, forward, loss, backward,
// optimiser) and so does the engine above this library -- ~330 launches on three streams, each crossing Python and ctypes.
// At batch 4 the GPU hides that; at the reference's own batch size of 1 (models/wesup.py:178) the host needs longer to queue
// a step than the GPU needs to run it.  A plan keeps what the walk produced -- for every launch the kernel, its grid, its
// stream and a byte copy of its arguments; for every ordering edge the event slot and the stream -- and wesup_plan_replay
// re-issues exactly that.  Nothing is re-derived at replay: pointers, sizes and tile choices are those of the recorded
// step, so the caller replays a plan only while every buffer it saw is still alive at the same address (the engine keeps
// its buffers per shape; inputs are copied into buffers the plan knows).
#include "common.hpp"
#include "launch.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

enum { NODE_KERNEL = 0, NODE_RECORD = 1, NODE_WAIT = 2, NODE_COPY = 3 };
struct PlanNode {
    int kind;
    const void* fn;
    dim3 grid, block;
    unsigned lds;
    hipStream_t st;
    size_t blob;         // offset of the argument copy
    size_t offs;         // index of the first per-argument offset
    size_t argv;         // index of the first argument pointer (after wesup_plan_end)
    int nargs;
    int slot;            // NODE_RECORD / NODE_WAIT
    void* dst;           // NODE_COPY
    const void* src;
    size_t bytes;
    int copy_kind;
};
struct WesupPlan {
    std::vector<PlanNode> nodes;
    std::vector<char> blob;
    std::vector<unsigned> offs;
    std::vector<void*> argv;
    mutable std::vector<long long> host_ns;      // WESUP_PLAN_TIMING: host time of each node's issue in the latest replay
    mutable std::vector<long long> host_at;      // ... and when it was issued (ns since the replay began)
    bool sealed = false;
};
static thread_local WesupPlan* t_rec = nullptr;
static inline long long now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

WesupPlan* wesup_plan_recording_() { return t_rec; }

void wesup_plan_add_kernel_(WesupPlan* plan, const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t st,
                            const void* const* args, const unsigned* sizes, int nargs) {
    PlanNode n = {};
    n.kind = NODE_KERNEL; n.fn = fn; n.grid = grid; n.block = block; n.lds = (unsigned)lds; n.st = st; n.nargs = nargs;
    n.blob = align_up(plan->blob.size(), 16);
    n.offs = plan->offs.size();
    size_t end = n.blob;
    for (int k = 0; k < nargs; ++k) {              // every argument on a 16-byte boundary of a zero-filled blob
        const size_t at = align_up(end, 16);
        plan->offs.push_back((unsigned)(at - n.blob));
        end = at + sizes[k];
    }
    plan->blob.resize(align_up(end, 16), 0);
    for (int k = 0; k < nargs; ++k) memcpy(plan->blob.data() + n.blob + plan->offs[n.offs + k], args[k], sizes[k]);
    n.bytes = plan->blob.size() - n.blob;
    plan->nodes.push_back(n);
}

// ------------------------------------------------------------------ ordering edges between streams
// A fixed pool of events addressed by slot number: the engine names its edges (weights ready, G_l ready, ...) by constant
// slots, so a recorded edge replays on the same event without handles travelling through the plan.
#define SYNC_SLOTS 256
static hipEvent_t g_events[SYNC_SLOTS];
static bool g_event_made[SYNC_SLOTS];
static hipEvent_t* sync_event(int slot) {
    if (slot < 0 || slot >= SYNC_SLOTS) return nullptr;
    if (!g_event_made[slot]) {
        // (hipEventReleaseToDevice instead of the default system-scope fence of a record: measured, no difference -- DESIGN.md 6)
        if (hipEventCreateWithFlags(&g_events[slot], hipEventDisableTiming) != hipSuccess) return nullptr;
        g_event_made[slot] = true;
    }
    return &g_events[slot];
}
extern "C" int wesup_sync_slots(void) { return SYNC_SLOTS; }
extern "C" int wesup_sync_record(int slot, void* stream) {
    hipEvent_t* ev = sync_event(slot);
    if (!ev) return WESUP_ERR_INVALID;
    if (hipEventRecord(*ev, (hipStream_t)stream) != hipSuccess) return WESUP_ERR_LAUNCH;
    if (t_rec) {
        PlanNode n = {};
        n.kind = NODE_RECORD; n.slot = slot; n.st = (hipStream_t)stream;
        t_rec->nodes.push_back(n);
    }
    return WESUP_OK;
}
extern "C" int wesup_sync_wait(int slot, void* stream) {
    hipEvent_t* ev = sync_event(slot);
    if (!ev) return WESUP_ERR_INVALID;
    if (hipStreamWaitEvent((hipStream_t)stream, *ev, 0) != hipSuccess) return WESUP_ERR_LAUNCH;
    if (t_rec) {
        PlanNode n = {};
        n.kind = NODE_WAIT; n.slot = slot; n.st = (hipStream_t)stream;
        t_rec->nodes.push_back(n);
    }
    return WESUP_OK;
}
// host-synchronous by definition: the caller's one wait per iteration (the loss read-back)
extern "C" int wesup_sync_synchronize(int slot) {
    hipEvent_t* ev = sync_event(slot);
    if (!ev) return WESUP_ERR_INVALID;
    return hipEventSynchronize(*ev) == hipSuccess ? WESUP_OK : WESUP_ERR_LAUNCH;
}
extern "C" int wesup_sync_query(int slot) {      // 1 = reached, 0 = not yet
    hipEvent_t* ev = sync_event(slot);
    if (!ev) return WESUP_ERR_INVALID;
    return hipEventQuery(*ev) == hipSuccess ? 1 : 0;
}

// ------------------------------------------------------------------ copies and fills (recordable)
static int plan_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, void* stream) {
    if (!dst || !src) return WESUP_ERR_INVALID;
    if (bytes == 0) return WESUP_OK;
    if (hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream) != hipSuccess) return WESUP_ERR_LAUNCH;
    if (t_rec) {
        PlanNode n = {};
        n.kind = NODE_COPY; n.dst = dst; n.src = src; n.bytes = bytes; n.copy_kind = (int)kind; n.st = (hipStream_t)stream;
        t_rec->nodes.push_back(n);
    }
    return WESUP_OK;
}
extern "C" int wesup_copy_to_host(void* dst_host_pinned, const void* src, size_t bytes, void* stream) {
    return plan_copy(dst_host_pinned, src, bytes, hipMemcpyDeviceToHost, stream);
}
extern "C" int wesup_copy(void* dst, const void* src, size_t bytes, void* stream) {
    return plan_copy(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
}

__global__ void fill_words_kernel(unsigned* __restrict__ p, unsigned v, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 + 4 <= n && (((uintptr_t)(p + i0)) & 15) == 0) {
        *reinterpret_cast<uint4*>(p + i0) = make_uint4(v, v, v, v);
    } else {
        for (size_t i = i0; i < n && i < i0 + 4; ++i) p[i] = v;
    }
}
int wesup_fill_words_(void* ptr, unsigned value, size_t words, hipStream_t st) {
    if (!ptr || (((uintptr_t)ptr) & 3)) return WESUP_ERR_INVALID;
    if (words == 0) return WESUP_OK;
    const size_t threads = (words + 3) / 4;
    if (threads > (size_t)0x7fffffff * 256) return WESUP_ERR_INVALID;
    WESUP_LAUNCH(fill_words_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (unsigned*)ptr, value, words);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}
extern "C" int wesup_fill_words(void* ptr, uint32_t value, size_t words, void* stream) {
    return wesup_fill_words_(ptr, value, words, (hipStream_t)stream);
}

// ------------------------------------------------------------------ the plan object
extern "C" int wesup_plan_create(WesupPlan** out) {
    if (!out) return WESUP_ERR_INVALID;
    *out = new (std::nothrow) WesupPlan();
    return *out ? WESUP_OK : WESUP_ERR_INVALID;
}
extern "C" int wesup_plan_destroy(WesupPlan* plan) {
    if (!plan) return WESUP_ERR_INVALID;
    if (t_rec == plan) t_rec = nullptr;
    delete plan;
    return WESUP_OK;
}
// From here until wesup_plan_end, every launch, edge and copy this THREAD issues through the library is executed as usual
// and appended to the plan.  One plan records at a time per thread; a plan that was recorded before starts over.
extern "C" int wesup_plan_begin(WesupPlan* plan) {
    if (!plan || t_rec) return WESUP_ERR_INVALID;
    plan->nodes.clear(); plan->blob.clear(); plan->offs.clear(); plan->argv.clear();
    plan->sealed = false;
    t_rec = plan;
    return WESUP_OK;
}
extern "C" int wesup_plan_end(WesupPlan* plan) {
    if (!plan || t_rec != plan) return WESUP_ERR_INVALID;
    t_rec = nullptr;
    plan->argv.clear();
    for (PlanNode& n : plan->nodes) {
        if (n.kind != NODE_KERNEL) continue;
        n.argv = plan->argv.size();
        for (int k = 0; k < n.nargs; ++k) plan->argv.push_back(plan->blob.data() + n.blob + plan->offs[n.offs + k]);
    }
    plan->sealed = true;
    return WESUP_OK;
}
// nodes recorded so far: taken between two calls while recording, it marks a position the caller can later split a replay at
// (host work between two parts of a step: the NaN check in front of the optimiser, a gradient bucket handed to RCCL)
extern "C" int wesup_plan_size(const WesupPlan* plan) { return plan ? (int)plan->nodes.size() : WESUP_ERR_INVALID; }
extern "C" int wesup_plan_kernels(const WesupPlan* plan) {
    if (!plan) return WESUP_ERR_INVALID;
    int k = 0;
    for (const PlanNode& n : plan->nodes) k += n.kind == NODE_KERNEL;
    return k;
}
// re-issue nodes [first, last) in recorded order, each on its recorded stream
extern "C" int wesup_plan_replay(const WesupPlan* plan, int first, int last) {
    if (!plan || !plan->sealed || first < 0 || last < first || (size_t)last > plan->nodes.size()) return WESUP_ERR_INVALID;
    void* const* argv = plan->argv.data();
    static const bool timing = getenv("WESUP_PLAN_TIMING") != nullptr;
    static thread_local long long t_origin = 0;
    if (timing) {
        plan->host_ns.resize(plan->nodes.size(), 0);
        plan->host_at.resize(plan->nodes.size(), 0);
        if (first == 0) t_origin = now_ns();
    }
    for (int i = first; i < last; ++i) {
        const PlanNode& n = plan->nodes[i];
        hipError_t e = hipSuccess;
        const long long t0 = timing ? now_ns() : 0;
        switch (n.kind) {
        case NODE_KERNEL:
            e = hipLaunchKernel(n.fn, n.grid, n.block, const_cast<void**>(argv + n.argv), n.lds, n.st);
            break;
        case NODE_RECORD:
            e = hipEventRecord(g_events[n.slot], n.st);
            break;
        case NODE_WAIT:
            e = hipStreamWaitEvent(n.st, g_events[n.slot], 0);
            break;
        case NODE_COPY:
            e = hipMemcpyAsync(n.dst, n.src, n.bytes, (hipMemcpyKind)n.copy_kind, n.st);
            break;
        }
        if (timing) { plan->host_at[i] = t0 - t_origin; plan->host_ns[i] = now_ns() - t0; }
        if (e != hipSuccess) return WESUP_ERR_LAUNCH;
    }
    return WESUP_OK;
}
// WESUP_PLAN_TIMING=1: host nanoseconds the latest replay spent issuing node i (out[0]) and when, since the replay began (out[1])
extern "C" int wesup_plan_node_host_ns(const WesupPlan* plan, int i, long long* out /* host [2] */) {
    if (!plan || !out || i < 0 || (size_t)i >= plan->host_ns.size()) return WESUP_ERR_INVALID;
    out[0] = plan->host_ns[i]; out[1] = plan->host_at[i];
    return WESUP_OK;
}
// Compares two plans node by node (kernel, geometry, stream, argument bytes; edges; copies): 0 = identical, k > 0 = the first
// difference is at node k - 1, -1 = bad arguments.  A caller that records the same step twice learns whether anything the
// walk produces (a pointer, a size, a tile choice) moved between the two -- i.e. whether replaying the first is safe.
extern "C" int wesup_plan_diff(const WesupPlan* a, const WesupPlan* b) {
    if (!a || !b || !a->sealed || !b->sealed) return -1;
    const size_t n = a->nodes.size() < b->nodes.size() ? a->nodes.size() : b->nodes.size();
    for (size_t i = 0; i < n; ++i) {
        const PlanNode &x = a->nodes[i], &y = b->nodes[i];
        bool same = x.kind == y.kind && x.st == y.st;
        if (same && x.kind == NODE_KERNEL) {
            same = x.fn == y.fn && x.grid.x == y.grid.x && x.grid.y == y.grid.y && x.grid.z == y.grid.z && x.block.x == y.block.x &&
                   x.lds == y.lds && x.nargs == y.nargs;
            same = same && x.bytes == y.bytes && memcmp(a->blob.data() + x.blob, b->blob.data() + y.blob, x.bytes) == 0;
        } else if (same && (x.kind == NODE_RECORD || x.kind == NODE_WAIT)) {
            same = x.slot == y.slot;
        } else if (same && x.kind == NODE_COPY) {
            same = x.dst == y.dst && x.src == y.src && x.bytes == y.bytes && x.copy_kind == y.copy_kind;
        }
        if (!same) return (int)i + 1;
    }
    return a->nodes.size() == b->nodes.size() ? 0 : (int)n + 1;
}
// what node i is, for diagnostics (a plan_diff mismatch, a dump of the recorded schedule): the kernel's name, or
// "<record slot>" / "<wait slot>" / "<copy bytes>"; *stream_out (optional) receives the node's stream handle
extern "C" const char* wesup_plan_node_name(const WesupPlan* plan, int i) {
    static thread_local char buf[64];
    if (!plan || i < 0 || (size_t)i >= plan->nodes.size()) return "";
    const PlanNode& n = plan->nodes[i];
    if (n.kind == NODE_RECORD) { snprintf(buf, sizeof buf, "<record %d>", n.slot); return buf; }
    if (n.kind == NODE_WAIT) { snprintf(buf, sizeof buf, "<wait %d>", n.slot); return buf; }
    if (n.kind == NODE_COPY) { snprintf(buf, sizeof buf, "<copy %zu>", n.bytes); return buf; }
    const char* s = hipKernelNameRefByPtr(n.fn, n.st);
    return s ? s : "<kernel>";
}
extern "C" void* wesup_plan_node_stream(const WesupPlan* plan, int i) {
    return (!plan || i < 0 || (size_t)i >= plan->nodes.size()) ? nullptr : (void*)plan->nodes[i].st;
}
