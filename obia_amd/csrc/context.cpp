// context.cpp -- context lifetime, workspace arena, error string, timing spans.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <utility>

#include <cstdarg>

namespace obia {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

void *Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (; cur_ < blocks_.size(); ++cur_) {
        Block &b = blocks_[cur_];
        if (b.size - b.used >= bytes) {
            void *p = b.p + b.used;
            b.used += bytes;
            cur_total_ += bytes;
            if (cur_total_ > high_water_) high_water_ = cur_total_;
            return p;
        }
    }
    size_t want = bytes;
    size_t grow = blocks_.empty() ? (size_t(64) << 20) : 2 * blocks_.back().size;
    if (grow > (size_t(4) << 30)) grow = size_t(4) << 30;
    if (want < grow) want = grow;
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        if (want == bytes || hipMalloc(&p, bytes) != hipSuccess) {
            failed_ = true;
            set_error("workspace hipMalloc of %zu bytes failed", bytes);
            return nullptr;
        }
        want = bytes;
    }
    blocks_.push_back(Block{static_cast<char *>(p), want, bytes});
    cur_ = blocks_.size() - 1;
    cur_total_ += bytes;
    if (cur_total_ > high_water_) high_water_ = cur_total_;
    return p;
}

Arena::Mark Arena::mark() const {
    Mark m;
    m.block = cur_ < blocks_.size() ? cur_ : blocks_.size();
    m.used = m.block < blocks_.size() ? blocks_[m.block].used : 0;
    m.total = cur_total_;
    return m;
}

void Arena::rewind(const Mark &m) {
    for (size_t i = m.block + 1; i < blocks_.size(); ++i) blocks_[i].used = 0;
    if (m.block < blocks_.size()) blocks_[m.block].used = m.used;
    cur_ = m.block;
    cur_total_ = m.total;
}

void Arena::reset() {
    failed_ = false;
    cur_total_ = 0;
    cur_ = 0;
    if (blocks_.size() > 1) {
        // merge: one block large enough for the high-water mark of the previous calls
        size_t total = 0;
        for (auto &b : blocks_) total += b.size;
        for (auto &b : blocks_) (void)hipFree(b.p);
        blocks_.clear();
        void *p = nullptr;
        if (hipMalloc(&p, high_water_ + (high_water_ >> 3)) == hipSuccess) blocks_.push_back(Block{static_cast<char *>(p), high_water_ + (high_water_ >> 3), 0});
        (void)total;
    } else if (!blocks_.empty()) {
        blocks_[0].used = 0;
    }
}

void Arena::release() {
    for (auto &b : blocks_) (void)hipFree(b.p);
    blocks_.clear();
}

size_t Arena::capacity() const {
    size_t t = 0;
    for (auto &b : blocks_) t += b.size;
    return t;
}

static hipEvent_t get_event(obia_ctx *ctx) {
    if (ctx->events_used == ctx->event_pool.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        ctx->event_pool.push_back(e);
    }
    return ctx->event_pool[ctx->events_used++];
}

ScopedSpan::ScopedSpan(obia_ctx *c, int kind) : ctx(c), on(c->profiling == 1 || (c->profiling == 2 && kind == T_ASSIGN)), idx(0) {
    if (!on) return;
    obia_ctx::Span s{kind, get_event(ctx), get_event(ctx)};
    (void)hipEventRecord(s.a, ctx->stream);
    idx = ctx->spans.size();
    ctx->spans.push_back(s);
}
ScopedSpan::~ScopedSpan() {
    if (on) (void)hipEventRecord(ctx->spans[idx].b, ctx->stream);
}

int side_streams(obia_ctx *ctx, int n) {
    if (n > obia_ctx::MAX_SIDE) n = obia_ctx::MAX_SIDE;
    if (!ctx->fork_ev) OBIA_HIP_TRY(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    if (!ctx->aux_fork) OBIA_HIP_TRY(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    if (!ctx->aux_join) OBIA_HIP_TRY(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
    for (int i = 0; i < n; ++i) {
        if (!ctx->side[i]) OBIA_HIP_TRY(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
        if (!ctx->join_ev[i]) OBIA_HIP_TRY(hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming));
    }
    return OBIA_OK;
}

KernelSpan::KernelSpan(obia_ctx *c, int kind) {
    if (!(c->profiling == 1 || (c->profiling == 2 && kind == T_ASSIGN))) return;
    a = get_event(c);
    b = get_event(c);
    c->spans.push_back(obia_ctx::Span{kind, a, b});
}

void begin_timing(obia_ctx *ctx) {
    ctx->spans.clear();
    ctx->events_used = 0;
    ctx->timing = Timing();
}

void resolve_timing(obia_ctx *ctx) {
    if (!ctx->profiling) return;
    (void)hipStreamSynchronize(ctx->stream);
    // sweeps: [start, end) of every kernel on the clock of the first one (end = distance from the first kernel's start to this
    // kernel's end, start = end - the kernel's own duration), for the length of their union
    std::vector<std::pair<double, double>> iv[2];
    hipEvent_t ref = nullptr;
    for (auto &s : ctx->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) { (void)hipGetLastError(); continue; }   // (a launch that was skipped)
        if (s.kind == T_ASSIGN || s.kind == T_PREPASS) {
            if (!ref) ref = s.a;
            float end = 0;
            if (hipEventElapsedTime(&end, ref, s.b) == hipSuccess) iv[s.kind == T_ASSIGN ? 0 : 1].emplace_back((double)end - ms, (double)end);
            else (void)hipGetLastError();
        }
        switch (s.kind) {
            case T_ASSIGN: ctx->timing.assign_ms += ms; ctx->timing.sweeps += 1; break;
            case T_FEAT: ctx->timing.feat_ms += ms; break;
            case T_CC: ctx->timing.cc_ms += ms; break;
            case T_ZONAL: ctx->timing.zonal_ms += ms; break;
            case T_TOTAL: ctx->timing.total_ms += ms; break;
            case T_PREPASS: ctx->timing.prepass_ms += ms; break;
            default: break;
        }
    }
    for (int c = 0; c < 2; ++c) {
        std::sort(iv[c].begin(), iv[c].end());
        double busy = 0, lo = 0, hi = -1;
        for (auto &x : iv[c]) {
            if (hi < lo || x.first > hi) { if (hi > lo) busy += hi - lo; lo = x.first; hi = x.second; }
            else if (x.second > hi) hi = x.second;
        }
        if (hi > lo) busy += hi - lo;
        (c == 0 ? ctx->timing.assign_busy_ms : ctx->timing.prepass_busy_ms) += busy;
    }
    ctx->spans.clear();
}

int read_back(obia_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes > ctx->pinned_bytes) {
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        size_t want = bytes < 4096 ? 4096 : bytes;
        OBIA_HIP_TRY(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    OBIA_HIP_TRY(hipMemcpyAsync(ctx->pinned, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->up_used = 0;   // every upload queued so far has executed: the ring is free again
    memcpy(host_dst, ctx->pinned, bytes);
    return OBIA_OK;
}

void debug_sync(obia_ctx *ctx, const char *stage) {
    static const bool on = std::getenv("OBIA_DEBUG_SYNC") != nullptr;
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    fprintf(stderr, "[obia debug] %s: %s\n", stage, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
}

int upload_async(obia_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
    if (bytes == 0) return OBIA_OK;
    constexpr size_t RING = 8u << 20;
    if (!ctx->up_buf) {
        void *p = nullptr;
        OBIA_HIP_TRY(hipHostMalloc(&p, RING, hipHostMallocDefault));
        ctx->up_buf = static_cast<char *>(p);
        ctx->up_bytes = RING;
        ctx->up_used = 0;
    }
    if (bytes > ctx->up_bytes) {   // too big for the ring: plain synchronous copy
        OBIA_HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->up_used = 0;
        return OBIA_OK;
    }
    if (ctx->up_used + bytes > ctx->up_bytes) {   // ring full: wait for what is queued, start over
        OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->up_used = 0;
    }
    char *slot = ctx->up_buf + ctx->up_used;
    memcpy(slot, host_src, bytes);
    OBIA_HIP_TRY(hipMemcpyAsync(dev_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->up_used += (bytes + 63) & ~(size_t)63;
    return OBIA_OK;
}

}  // namespace obia

using namespace obia;

extern "C" {

int obia_abi_version(void) { return OBIA_ABI_VERSION; }

const char *obia_last_error(void) { return g_last_error.c_str(); }

static obia_ctx *create_impl(int device_id, void *stream, bool own) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available");
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        set_error("device_id %d out of range (%d devices)", device_id, n);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", device_id);
        return nullptr;
    }
    obia_ctx *ctx = new obia_ctx();
    ctx->device = device_id;
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed");
            delete ctx;
            return nullptr;
        }
        ctx->own_stream = true;
    } else {
        ctx->stream = static_cast<hipStream_t>(stream);
    }
    return ctx;
}

obia_ctx *obia_create(int device_id) { return create_impl(device_id, nullptr, true); }
obia_ctx *obia_create_on_stream(int device_id, void *hip_stream) { return create_impl(device_id, hip_stream, false); }

void obia_destroy(obia_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->arena.release();
    if (ctx->up_buf) (void)hipHostFree(ctx->up_buf);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->defer_buf) (void)hipHostFree(ctx->defer_buf);
    if (ctx->cc_visited) (void)hipFree(ctx->cc_visited);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < obia_ctx::MAX_SIDE; ++i) {
        if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]);
        if (ctx->join_ev[i]) (void)hipEventDestroy(ctx->join_ev[i]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int obia_synchronize(obia_ctx *ctx) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OBIA_OK;
}

int64_t obia_workspace_bytes(obia_ctx *ctx) { return ctx ? (int64_t)ctx->arena.capacity() : 0; }

void obia_slic_default_params(obia_slic_params *p) {
    if (!p) return;
    p->n_segments = 100;
    p->compactness = 10.0;
    p->max_num_iter = 10;
    p->convert2lab = -1;
    p->enforce_connectivity = 1;
    p->min_size_factor = 0.5;
    p->max_size_factor = 3.0;
    p->slic_zero = 0;
    p->start_label = 1;
    p->normalize_bands = 0;
    p->exit_on_fixed_point = 0;
    p->reserved = 0;
    p->sigma_zyx[0] = p->sigma_zyx[1] = p->sigma_zyx[2] = 0.0;
    p->spacing_zyx[0] = p->spacing_zyx[1] = p->spacing_zyx[2] = 1.0;
}

int obia_set_profiling(obia_ctx *ctx, int enabled) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    ctx->profiling = enabled < 0 ? 0 : (enabled > 2 ? 1 : enabled);
    return OBIA_OK;
}

double obia_last_timing(obia_ctx *ctx, int what) {
    if (!ctx) return -1.0;
    switch (what) {
        case 0: return ctx->timing.assign_ms;
        case 1: return (double)ctx->timing.sweeps;
        case 2: return ctx->timing.feat_ms;
        case 3: return ctx->timing.cc_ms;
        case 4: return ctx->timing.zonal_ms;
        case 5: return ctx->timing.total_ms;
        case 6: return ctx->timing.prepass_ms;
        case 7: return ctx->timing.assign_px;
        case 8: return ctx->timing.prepass_px;
        case 9: return ctx->timing.assign_store_px;
        case 10: return ctx->timing.assign_busy_ms;
        case 11: return ctx->timing.prepass_busy_ms;
        case 12: return (double)ctx->timing.batch_repeats;
        default: return -1.0;
    }
}

}  // extern "C"
