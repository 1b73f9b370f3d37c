// common.hpp -- context, device workspace arena, error plumbing shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/obia_hip.h"

namespace obia {

void set_error(const char *fmt, ...);

#define OBIA_HIP_TRY(expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            obia::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return OBIA_E_HIP;                                                                   \
        }                                                                                        \
    } while (0)

#define OBIA_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != OBIA_OK) return rc__; \
    } while (0)

// Bump allocator over a few large hipMalloc blocks.  reset() keeps the memory; when a call needed
// more than one block the blocks are merged into one at the next reset, so a steady-state call
// performs no hipMalloc at all.
class Arena {
  public:
    ~Arena() { release(); }
    void *alloc(size_t bytes);
    template <typename T> T *get(size_t n) { return static_cast<T *>(alloc(n * sizeof(T))); }
    void reset();
    void release();
    // mark()/rewind(): scoped reuse inside one call (per tile batch)
    struct Mark { size_t block, used, total; };
    Mark mark() const;
    void rewind(const Mark &m);
    size_t capacity() const;
    bool failed() const { return failed_; }

  private:
    struct Block { char *p; size_t size; size_t used; };
    std::vector<Block> blocks_;
    size_t high_water_ = 0, cur_total_ = 0, cur_ = 0;
    bool failed_ = false;
};

struct Timing {
    double assign_ms = 0, prepass_ms = 0, feat_ms = 0, cc_ms = 0, zonal_ms = 0, total_ms = 0;
    double assign_px = 0, prepass_px = 0;   // pixels processed by the timed colour / pre-pass sweeps (sum over launches)
    double assign_store_px = 0;             // ... of the colour sweeps that also stored their labels (the last sweep of a batch)
    int sweeps = 0;
    int batch_repeats = 0;                  // batches whose sweeps ran again with every sweep storing its labels (a valid pixel no window reached)
    // time during which at least one colour (pre-pass) sweep was running: equals assign_ms (prepass_ms) when the sweeps run one
    // after the other, less when groups of problems run side by side (slic_run_sweeps)
    double assign_busy_ms = 0, prepass_busy_ms = 0;
};

}  // namespace obia

struct obia_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    obia::Arena arena;
    void *pinned = nullptr;          // small pinned host staging buffer for scalar read-backs
    size_t pinned_bytes = 0;
    unsigned long long *defer_buf = nullptr;   // pinned landing area of a read-back that is looked at after a LATER synchronisation
    bool defer_pending = false;                // (slic_run_sweeps with defer: the orphan flag and pixel counters of the sweeps)
    // visited map of the connectivity stage's component walks (cc.hip): every walk clears its own marks, so the map is all zero
    // between calls and is cleared as a whole only when it grows or after a call that did not complete (cc_visited_dirty)
    int32_t *cc_visited = nullptr;
    size_t cc_visited_px = 0;
    bool cc_visited_dirty = false;
    char *up_buf = nullptr;          // pinned ring for small host -> device tables (upload_async): no stream sync per upload
    size_t up_bytes = 0, up_used = 0;
    int profiling = 0;               // 0 off, 1 every span, 2 only the colour sweeps (obia_set_profiling)
    obia::Timing timing;
    // event pairs recorded around kernels of interest; resolved (one sync) at the end of the call
    struct Span { int kind; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    // side streams of the sweep loop (slic_run_sweeps: groups of problems run their prep / sweep chains side by side, so
    // that one group's sweep fills the ramp, the tail and the launch gaps of the others'); created on first use
    static constexpr int MAX_SIDE = 3;
    hipStream_t side[MAX_SIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[MAX_SIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;   // the tiler's white feature pass beside the black sweeps (tiling.hip)
};

namespace obia {

enum TimeKind { T_ASSIGN = 0, T_FEAT = 2, T_CC = 3, T_ZONAL = 4, T_TOTAL = 5, T_PREPASS = 6 };

// Records a HIP event pair around a region of the context's stream when profiling is on; no host
// synchronisation happens until resolve_timing().
struct ScopedSpan {
    obia_ctx *ctx; bool on; size_t idx;
    ScopedSpan(obia_ctx *c, int kind);
    ~ScopedSpan();
};
// The same bookkeeping for ONE kernel: the event pair is handed to hipExtLaunchKernelGGL, which binds it to the dispatch
// itself, so the elapsed time is the kernel's own start-to-end (what a rocprofv3 kernel trace reports) and not the distance
// between two stream markers around it (that also times the dispatch, a few microseconds).  Both null when profiling is off.
struct KernelSpan {
    hipEvent_t a = nullptr, b = nullptr;
    KernelSpan(obia_ctx *c, int kind);
};
void begin_timing(obia_ctx *ctx);
void resolve_timing(obia_ctx *ctx);

int read_back(obia_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);  // async copy + stream sync
// Host table -> device without a stream synchronisation: the bytes are copied into a pinned ring first, so the caller's
// (pageable, short-lived) buffer is free at once; the ring is recycled at the next read_back (everything queued before it
// has then executed).
int upload_async(obia_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int side_streams(obia_ctx *ctx, int n);   // makes sure side[0..n) and their events exist
// Developer aid (OBIA_DEBUG_SYNC=1): synchronise the stream and report the stage on stderr, so that an asynchronous GPU fault
// is pinned on the stage that caused it.  A no-op otherwise.
void debug_sync(obia_ctx *ctx, const char *stage);

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Loads of data that ONE workgroup reads ONCE per pass (feature planes in a sweep, the raster in the feature and statistics
// passes): non-temporal, so the lines do not displace what IS re-read from the caches -- centroid records, bin heads, accumulator
// lines.  Measured on the colour sweep: 0.2144 -> 0.2065 ms, 0.2143 -> 0.2087 ms per launch (two pairs on one box).
#if defined(__HIPCC__)
typedef float obia_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream_f4(const void *p) {
    const obia_v4f t = __builtin_nontemporal_load(reinterpret_cast<const obia_v4f *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
#endif

}  // namespace obia
