// common.hpp -- context, workspace arena, error plumbing shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/gauspcc.h"

#include "errors.hpp"

namespace gpcc {

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return gpcc::fail(GPCC_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

// GAUSPCC_DEBUG_SYNC=1 (developer): every launch site prints itself and waits for the device -- the last line on stderr
// before a "Memory access fault" abort names the kernel that faulted
inline bool debug_sync_on() { static const bool on = [] { const char *e = getenv("GAUSPCC_DEBUG_SYNC"); return e && atoi(e) != 0; }(); return on; }
inline hipError_t launch_check(const char *file, int line)
{
    hipError_t e = hipGetLastError();
    ++g_launches;
    if (e == hipSuccess && debug_sync_on()) { fprintf(stderr, "[launch] %s:%d\n", file, line); fflush(stderr); e = hipDeviceSynchronize(); }
    return e;
}
#define LAUNCH_CHECK() HIP_TRY(gpcc::launch_check(__FILE__, __LINE__))

constexpr int CB = 1 << 20;        // coordinate bias at level 0
constexpr int CLIM = CB - 8;       // |coordinate| limit
constexpr int NSTAGE = 4;
constexpr int CH = 32;             // channels the MFMA kernels are specialised for

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over one hipMalloc'd slab; reset at the start of every API call.  Two ends: take() grows from the bottom
// (data that outlives a step of the caller's loop), take_top() from the top (per-step work buffers, rewound by the caller).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0, top = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return GPCC_OK;
        if (base) HIP_TRY(hipFree(base));
        base = nullptr; cap = 0;
        size_t want = bytes + (bytes >> 3) + (1 << 20);
        HIP_TRY(hipMalloc((void **)&base, want));
        cap = want; top = want & ~size_t(255);
        return GPCC_OK;
    }
    // flipped: take / mark / rewind serve the TOP end.  Work enqueued on a second stream takes its temporaries there, so
    // that a rewind on one stream never hands memory to the other while kernels still use it (the encoder's rank pass).
    // The host rewinds `top` as soon as the second stream's launches are ENQUEUED, while its kernels may still be running:
    // top_low remembers the lowest top a flipped section reached, and bottom takes stay below it until the caller has made
    // the first stream wait for the second one (release_top_low) -- otherwise a nearly full arena could hand the rank
    // pass's live temporaries to the first stream's next buffers.
    bool flip = false;
    size_t top_low = ~size_t(0);
    void release_top_low() { top_low = ~size_t(0); }
    void reset() { off = 0; top = cap & ~size_t(255); flip = false; top_low = ~size_t(0); }
    template <typename T> T *take(size_t count)
    {
        if (flip) return take_top<T>(count);
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (off + bytes > (top < top_low ? top : top_low)) return nullptr;
        T *p = reinterpret_cast<T *>(base + off);
        off += bytes;
        return p;
    }
    template <typename T> T *take_top(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (off + bytes > top) return nullptr;
        top -= bytes;
        if (flip && top < top_low) top_low = top;
        return reinterpret_cast<T *>(base + top);
    }
    size_t top_mark() const { return top; }
    void top_rewind(size_t m) { top = m; }
    size_t mark() const { return flip ? top : off; }
    void rewind(size_t m) { if (flip) top = m; else off = m; }
};

// Environment knobs and per-device one-time set-up, safe when several host threads (one context each) enter for the first
// time together: function-local statics initialised from a lambda are initialised once (C++11), and what has to happen once
// PER DEVICE (hipFuncSetAttribute is a per-device setting) runs under a mutex with a bit per device.
inline int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
inline long long env_ll(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
// Knobs that change WHICH kernel runs (block classes, parked loops, fused / unfused paths ...) are developer switches: honoured only
// with GAUSPCC_DEV=1 in the environment, so that a stray variable in a user's shell cannot silently change the kernel mix.  One
// line on stderr says so when such a variable is set without it.  (The tests of the variants set GAUSPCC_DEV=1.)
inline bool dev_mode() { static const bool on = env_int("GAUSPCC_DEV", 0) != 0; return on; }
inline long long dev_env_ll(const char *name, long long dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    if (dev_mode()) return atoll(e);
    static std::mutex m;
    static bool told = false;
    std::lock_guard<std::mutex> g(m);
    if (!told) { told = true; fprintf(stderr, "[gauspcc] %s is a developer knob and is ignored without GAUSPCC_DEV=1 (as are the other kernel-selection knobs)\n", name); }
    return dflt;
}
inline int dev_env_int(const char *name, int dflt) { return (int)dev_env_ll(name, dflt); }
struct PerDeviceOnce {
    std::mutex m;
    uint64_t done = 0;
    template <typename F> int run(int device, F f)
    {
        std::lock_guard<std::mutex> g(m);
        const uint64_t bit = 1ull << (device & 63);
        if (done & bit) return GPCC_OK;
        const int rc = f();
        if (rc == GPCC_OK) done |= bit;
        return rc;
    }
};

template <typename T> struct HostBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return GPCC_OK;
        if (p) HIP_TRY(hipHostFree(p));
        p = nullptr; cap = 0;
        size_t want = n + (n >> 2) + 1024;
        HIP_TRY(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
        return GPCC_OK;
    }
};

}  // namespace gpcc

namespace gpcc {
// HIP-event timing of the dominant kernel (k_sparse_conv), on the stream it is launched on.  Convolutions that are
// enqueued back to back with nothing between them (a trunk's five, a stage's two) share ONE pair of events: an event
// costs ~3 us of stream time, and one pair per launch slowed a decode by 1.5 ms.
struct ConvRec { int e0, e1, level, njobs, R, H; long long n, nblk; int launches; int fused = 0; };   // fused: a persistent small-level launch (fused.hip), `launches` = the convolutions inside
struct Prof {
    bool on = false;
    bool chain_open = false;   // between conv_chain_begin and conv_chain_end: sparse_conv records no events of its own
    ConvRec chain = {};
    std::vector<hipEvent_t> pool;
    int used = 0;
    std::vector<ConvRec> recs;
    double conv_ms = 0.0;
    int64_t conv_launches = 0, conv_pair_jobs = 0;
    double fused_ms = 0.0;
    int64_t fused_launches = 0, fused_pair_jobs = 0;
    // gpcc_profile_enable(ctx, 2): the HBM-bound stages of the path are bracketed by events too (a handful per level: the
    // brackets cost stream time, so bench.py measures them in a pass of their own behind the timed region)
    bool stages = false;
    struct StageRec { int id, e0, e1; };
    std::vector<StageRec> srecs;
    double stage_ms[8] = {0}, stage_bytes[8] = {0};
    double stage_crit_ms[8] = {0};   // of stage_ms: the part during which no convolution bracket (either family) was open -- stream time the step really waits for
    int64_t stage_n[8] = {0};
};
enum { ST_OCTREE = 0, ST_TILES, ST_ELEM, ST_HEADS, ST_CODER, ST_COUNT };
}  // namespace gpcc

namespace gpcc {
// GAUSPCC_HOST_TRACE=1: host-side phase times of encode / decode on stderr (where the CPU thread is when the GPU idles)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    HostTrace() : on(getenv("GAUSPCC_HOST_TRACE") && atoi(getenv("GAUSPCC_HOST_TRACE")) > 0) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what, long long a = -1, long long b = -1)
    {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[host] %-22s %8.3f ms  (+%.3f)", what, std::chrono::duration<double, std::milli>(t - t0).count(), std::chrono::duration<double, std::milli>(t - last).count());
        if (a >= 0) fprintf(stderr, "  %lld", a);
        if (b >= 0) fprintf(stderr, " %lld", b);
        fputc('\n', stderr);
        last = t;
    }
};
}  // namespace gpcc

struct gpcc_ctx {
    int device = 0;
    gpcc::Prof prof;
    gpcc::Arena arena;              // device workspace
    gpcc::HostBuf<uint8_t> hbytes;  // pinned output / staging bytes
    gpcc::HostBuf<uint8_t> hstage;  // pinned small staging (counts, flags, descriptors)
    gpcc::HostBuf<uint8_t> hcoder;  // pinned: what the host coder of the reference layout reads / writes (coder words, CDF rows, symbols; hostcoder.hpp)
    gpcc::HostBuf<uint8_t> hbatch;  // pinned staging of the batched calls (forest.hpp: per-scene tables); reserved once per call, before anything reads it asynchronously
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream of the codec: octree / tile-list work of the next step runs beside the convolutions of this one
    hipStream_t side = nullptr, xfer = nullptr;   // xfer: the container's host -> device copy of a decode, beside both
    hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_bytes = nullptr, ev_tables = nullptr;
    // partial products of the two-launch convolution of the small levels (network.hip: k_conv_products / k_conv_sum): one
    // buffer per context, grown on demand, used by one convolution at a time (the launches of a context's convolutions are
    // ordered on one stream)
    float *conv_products = nullptr;
    size_t conv_products_cap = 0;   // floats
    // persistent small-level launches of the decoder (fused.hip): two grid-barrier blocks that alternate between launches and a
    // sticky timeout word; fused_off: a launch timed out on this context (its workgroups were not all resident) -- the
    // launch-per-layer path from then on
    // single-pass scans (primitives.hip: k_scan_lookback): per stream of the context one tile-status array, a ticket word and a launch
    // epoch (status words of earlier launches are invalid by their epoch: no reset between launches)
    struct ScanState { hipStream_t st; unsigned long long *status; uint32_t *ticket; uint32_t epoch; };
    std::vector<ScanState> scan_states;
    uint32_t *dev_err_dev = nullptr;   // the device's address of dev_err (hipHostGetDevicePointer, once)
    uint32_t *dev_err = nullptr;   // sticky device-side error word in pinned, device-visible host memory (primitives.hip: the look-back scan's bounded wait);
                                   // read after a call's final sync by device_error_check()
    int container_version = 4;     // what gpcc_encode / gpcc_rc_encode write for chunk_log2 != 0 (gpcc_ctx_set_container_version: 3 or 4)
    void *fused_state = nullptr;
    int fused_flip = 0;
    bool fused_off = false;
    // a timeout may have been transient (another process's long kernel held the CUs): the persistent path is tried again after
    // fused_rearm_after decodes on the launch-per-layer path, and the wait doubles with every further timeout (32, 64, ... 4096)
    int fused_off_decodes = 0, fused_rearm_after = 32;
    void fused_note_decode()
    {
        if (!fused_off) return;
        if (++fused_off_decodes >= fused_rearm_after) { fused_off = false; fused_off_decodes = 0; fused_rearm_after = fused_rearm_after < 4096 ? 2 * fused_rearm_after : 4096; }
    }
    hipEvent_t fused_ev = nullptr;   // recorded behind every persistent launch of this context while other contexts use the device (fused.hip: FusedGate)
    // developer trace (gpcc_debug_trace_*): checksums of intermediate buffers of a decode, one (tag, sum) per mark, computed
    // on the stream that produced the buffer -- to find the first stage whose output differs between two runs
    bool dbg_on = false;
    unsigned long long *dbg_dev = nullptr;
    std::vector<int> dbg_tags;
    struct DbgCap { int tag; void *dev; size_t cap, bytes; };
    std::vector<DbgCap> dbg_caps;   // gpcc_debug_capture: copies of selected buffers (tag -> device copy)
    int dbg_capture_tag_mod = -1;   // capture buffers whose tag % 100 equals this (-1: none)
    int side_init()
    {
        if (side) return GPCC_OK;
        HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&xfer, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_bytes, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_tables, hipEventDisableTiming));
        return GPCC_OK;
    }
};

namespace gpcc {
int dbg_mark(gpcc_ctx *ctx, hipStream_t st, int tag, const void *p, size_t bytes);   // api.hip (developer trace)
int prof_event(gpcc_ctx *ctx, hipStream_t st, int *idx);   // network.hip
// RAII bracket of one HBM-bound stage on the stream its kernels are enqueued on; bytes = the stage's ALGORITHMIC traffic
// (what an ideal layer-by-layer implementation reads and writes, HISTORY.md section 4), accumulated beside the time
struct StageTimer {
    gpcc_ctx *c; hipStream_t st; int id, e0 = -1;
    StageTimer(gpcc_ctx *ctx, hipStream_t stream, int stage, double bytes) : c(ctx), st(stream), id(stage)
    {
        if (!c || !c->prof.on || !c->prof.stages) { c = nullptr; return; }
        if (prof_event(c, st, &e0) != GPCC_OK) { c = nullptr; return; }
        c->prof.stage_bytes[id] += bytes;
    }
    void add_bytes(double b) { if (c) c->prof.stage_bytes[id] += b; }
    ~StageTimer()
    {
        int e1;
        if (c && prof_event(c, st, &e1) == GPCC_OK) c->prof.srecs.push_back(Prof::StageRec{id, e0, e1});
    }
};
}  // namespace gpcc

#define TAKE_TOP(var, T, count)                                                                    \
    T *var = ctx->arena.take_top<T>((size_t)(count));                                              \
    if (!var) return gpcc::fail(GPCC_ERR_NOMEM, "%s:%d workspace arena exhausted (%s x %lld)", __FILE__, __LINE__, #T, (long long)(count))

#define TAKE(var, T, count)                                                                        \
    T *var = ctx->arena.take<T>((size_t)(count));                                                  \
    if (!var) return gpcc::fail(GPCC_ERR_NOMEM, "%s:%d workspace arena exhausted (%s x %lld)", __FILE__, __LINE__, #T, (long long)(count))
