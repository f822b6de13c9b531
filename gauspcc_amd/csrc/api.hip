// api.hip -- context / model management, calculate_morton_order, and the stage-level C-ABI entry
// points used by the parity tests (include/gauspcc.h).
#include <algorithm>

#include "fused.hpp"
#include "network.hpp"
#include "octree.hpp"
#include "primitives.hpp"
#include "rangecoder_dev.hpp"

using namespace gpcc;

namespace gpcc {
thread_local char g_err[512] = "";
thread_local long long g_launches = 0;
}

extern "C" const char *gpcc_last_error(void) { return gpcc::g_err; }
extern "C" int gpcc_version(void) { return 100; }

extern "C" int gpcc_ctx_create(int device, gpcc_ctx **out)
{
    if (!out) return fail(GPCC_ERR_ARG, "null argument");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(GPCC_ERR_ARG, "device %d not present (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(GPCC_ERR_ARG, "libgauspcc is built for gfx950 (MI355X); device %d is %s", device, prop.gcnArchName);
    gpcc_ctx *c = new gpcc_ctx();
    c->device = device;
    *out = c;
    return GPCC_OK;
}

extern "C" void gpcc_ctx_destroy(gpcc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->prof.pool) (void)hipEventDestroy(e);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); (void)hipStreamSynchronize(c->xfer); (void)hipStreamDestroy(c->xfer); (void)hipEventDestroy(c->ev_main); (void)hipEventDestroy(c->ev_side); (void)hipEventDestroy(c->ev_bytes); (void)hipEventDestroy(c->ev_tables); }
    if (c->arena.base) (void)hipFree(c->arena.base);
    if (c->conv_products) (void)hipFree(c->conv_products);
    fused_ctx_release(c);
    if (c->fused_state) (void)hipFree(c->fused_state);
    for (auto &ss : c->scan_states) if (ss.status) (void)hipFree(ss.status);
    if (c->dev_err) (void)hipHostFree(c->dev_err);
    if (c->hbytes.p) (void)hipHostFree(c->hbytes.p);
    if (c->dbg_dev) (void)hipFree(c->dbg_dev);
    for (auto &e : c->dbg_caps) if (e.dev) (void)hipFree(e.dev);
    if (c->hstage.p) (void)hipHostFree(c->hstage.p);
    if (c->hcoder.p) (void)hipHostFree(c->hcoder.p);
    if (c->hbatch.p) (void)hipHostFree(c->hbatch.p);
    delete c;
}

// Version gpcc_encode (and the stage-level gpcc_rc_encode / gpcc_rc_decode) use for chunked containers: 4 (default: the
// carry-propagating coder in the lanes, HISTORY.md section 5) or 3 (torchac's coder in the lanes: what round 3 wrote).  Readers take 0-4.
// kernel launches enqueued by the calling thread since the last reset (every launch site of the library passes LAUNCH_CHECK);
// bench.py's `kernels per decode` and the batch's launch-count bar read it
// the sticky device-side error word of the context (primitives.hpp: device_error_check): for callers of the stage-level entry points that
// do not synchronise themselves -- call after synchronising the stream
extern "C" int gpcc_device_error_check(gpcc_ctx *ctx)
{
    if (!ctx) return fail(GPCC_ERR_ARG, "null argument");
    return device_error_check(ctx);
}

extern "C" long long gpcc_debug_launches(int reset)
{
    const long long v = gpcc::g_launches;
    if (reset) gpcc::g_launches = 0;
    return v;
}

extern "C" int gpcc_ctx_set_container_version(gpcc_ctx *c, int version)
{
    if (!c) return fail(GPCC_ERR_ARG, "null argument");
    if (version != 3 && version != 4) return fail(GPCC_ERR_ARG, "container version must be 3 or 4");
    c->container_version = version;
    return GPCC_OK;
}

// what the context holds right now: device bytes (workspace arena + the small levels' product buffer + developer buffers) and
// pinned host bytes.  torch.cuda.max_memory_allocated() cannot see any of it (the library allocates with hipMalloc itself).
extern "C" int gpcc_ctx_bytes(const gpcc_ctx *c, int64_t *device_bytes, int64_t *pinned_bytes)
{
    if (!c) return fail(GPCC_ERR_ARG, "null argument");
    size_t d = c->arena.cap + c->conv_products_cap * sizeof(float) + (c->dbg_dev ? 8 * (size_t)4096 : 0);
    d += c->scan_states.size() * (8 * (size_t)65536 + 256);      // single-pass scan states (primitives.hip: scan_state; at most 8)
    if (c->fused_state) d += 2 * 3456 + 128;                     // grid-barrier blocks of the persistent launches (fused.hip)
    for (const auto &e : c->dbg_caps) d += e.cap;
    if (device_bytes) *device_bytes = (int64_t)d;
    if (pinned_bytes) *pinned_bytes = (int64_t)(c->hbytes.cap + c->hstage.cap + c->hbatch.cap + c->hcoder.cap);
    return GPCC_OK;
}

namespace gpcc {
constexpr int DBG_MAX = 4096;
__global__ __launch_bounds__(256) void k_dbg_sum(const uint32_t *__restrict__ p, size_t words, unsigned long long *__restrict__ out)
{
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) a += (unsigned long long)p[i] * (2ull * i + 1ull);
    atomicAdd(out, a);
}
int dbg_mark(gpcc_ctx *ctx, hipStream_t st, int tag, const void *p, size_t bytes)
{
    if (!ctx || !ctx->dbg_on || (int)ctx->dbg_tags.size() >= DBG_MAX) return GPCC_OK;
    const size_t words = bytes / 4;
    unsigned long long *slot = ctx->dbg_dev + ctx->dbg_tags.size();
    HIP_TRY(hipMemsetAsync(slot, 0, 8, st));
    if (words) k_dbg_sum<<<(unsigned)std::min<size_t>(1024, (words + 255) / 256), 256, 0, st>>>(static_cast<const uint32_t *>(p), words, slot);
    LAUNCH_CHECK();
    ctx->dbg_tags.push_back(tag);
    if (ctx->dbg_capture_tag_mod >= 0 && tag % 100 == ctx->dbg_capture_tag_mod && bytes) {
        gpcc_ctx::DbgCap *c = nullptr;
        for (auto &e : ctx->dbg_caps) if (e.tag == tag) c = &e;
        if (!c) { ctx->dbg_caps.push_back(gpcc_ctx::DbgCap{tag, nullptr, 0, 0}); c = &ctx->dbg_caps.back(); }
        if (c->cap < bytes) {
            if (c->dev) HIP_TRY(hipFree(c->dev));
            c->dev = nullptr; c->cap = 0;
            HIP_TRY(hipMalloc(&c->dev, bytes));
            c->cap = bytes;
        }
        c->bytes = bytes;
        HIP_TRY(hipMemcpyAsync(c->dev, p, bytes, hipMemcpyDeviceToDevice, st));
    }
    return GPCC_OK;
}
}  // namespace gpcc

// developer trace: keep a device copy of every marked buffer whose tag % 100 == tag_mod (-1: none); _get copies one out
extern "C" int gpcc_debug_capture(gpcc_ctx *ctx, int tag_mod)
{
    if (!ctx) return fail(GPCC_ERR_ARG, "null context");
    ctx->dbg_capture_tag_mod = tag_mod;
    return GPCC_OK;
}
extern "C" int gpcc_debug_exclusive_scan(gpcc_ctx *ctx, const uint32_t *in_dev, uint32_t *out_dev, const uint32_t *in2_dev, uint32_t *out2_dev, int64_t n,
                                         uint32_t *total_dev, void *stream)
{
    if (!ctx || !in_dev || !out_dev || n < 0 || (in2_dev == nullptr) != (out2_dev == nullptr)) return fail(GPCC_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    GP_TRY(ctx->arena.reserve((size_t)n / 64 + ((size_t)1 << 20)));   // (the three-launch scan of unaligned arrays keeps its block sums there)
    ctx->arena.reset();
    if (in2_dev) {
        GP_TRY(exclusive_scan_pair_u32(ctx, st, in_dev, out_dev, in2_dev, out2_dev, n));
        if (total_dev) return fail(GPCC_ERR_ARG, "no total with a pair of scans");
        return GPCC_OK;
    }
    return exclusive_scan_u32(ctx, st, in_dev, out_dev, n, total_dev);
}

extern "C" long long gpcc_debug_capture_get(gpcc_ctx *ctx, int tag, void *host, long long cap)
{
    if (!ctx) return -1;
    for (auto &e : ctx->dbg_caps)
        if (e.tag == tag) {
            const size_t n = std::min<size_t>(e.bytes, (size_t)std::max<long long>(cap, 0));
            if (n && hipMemcpy(host, e.dev, n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            return (long long)e.bytes;
        }
    return -1;
}

extern "C" int gpcc_debug_trace_enable(gpcc_ctx *ctx, int on)
{
    if (!ctx) return fail(GPCC_ERR_ARG, "null context");
    if (on && !ctx->dbg_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->dbg_dev), 8 * (size_t)DBG_MAX));
    ctx->dbg_on = on != 0;
    ctx->dbg_tags.clear();
    return GPCC_OK;
}
// marks recorded since the last enable / get (the decode that recorded them has returned: its streams are idle); clears the list
extern "C" int gpcc_debug_trace_get(gpcc_ctx *ctx, int *tags, unsigned long long *sums, int cap)
{
    if (!ctx || !ctx->dbg_dev) return 0;
    const int n = std::min<int>(cap, (int)ctx->dbg_tags.size());
    if (n > 0) {
        if (hipMemcpy(sums, ctx->dbg_dev, 8 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
        for (int i = 0; i < n; ++i) tags[i] = ctx->dbg_tags[(size_t)i];
    }
    ctx->dbg_tags.clear();
    return n;
}

extern "C" int gpcc_profile_enable(gpcc_ctx *ctx, int on)
{
    if (!ctx) return fail(GPCC_ERR_ARG, "null argument");
    if (on == 3) { ctx->prof.on = false; ctx->prof.stages = false; return GPCC_OK; }   // pause: what was collected stays readable
    ctx->prof.on = on != 0;
    ctx->prof.stages = on >= 2;
    ctx->prof.conv_ms = 0.0; ctx->prof.conv_launches = 0; ctx->prof.conv_pair_jobs = 0;
    ctx->prof.fused_ms = 0.0; ctx->prof.fused_launches = 0; ctx->prof.fused_pair_jobs = 0;
    ctx->prof.recs.clear(); ctx->prof.srecs.clear(); ctx->prof.used = 0; ctx->prof.chain_open = false;
    for (int i = 0; i < 8; ++i) { ctx->prof.stage_ms[i] = 0.0; ctx->prof.stage_bytes[i] = 0.0; ctx->prof.stage_n[i] = 0; ctx->prof.stage_crit_ms[i] = 0.0; }
    return GPCC_OK;
}

extern "C" int gpcc_profile_stages(gpcc_ctx *ctx, gpcc_stage *out, int cap, int *n_out)
{
    if (!ctx || !out || !n_out) return fail(GPCC_ERR_ARG, "null argument");
    static const char *const names[ST_COUNT] = {"octree+ranks (radix sort, level build)", "tile lists (cell maps -> tiles)", "elementwise (embeddings, stage inputs)",
                                                "heads (Linear-ReLU-Linear-softmax-CDF)", "range coder"};
    int n = 0;
    for (int i = 0; i < ST_COUNT && n < cap; ++i, ++n) {
        memset(&out[n], 0, sizeof out[n]);
        strncpy(out[n].name, names[i], sizeof out[n].name - 1);
        out[n].ms = ctx->prof.stage_ms[i]; out[n].bytes = ctx->prof.stage_bytes[i]; out[n].brackets = ctx->prof.stage_n[i]; out[n].critical_ms = ctx->prof.stage_crit_ms[i];
    }
    *n_out = n;
    return GPCC_OK;
}

extern "C" int gpcc_profile_get(gpcc_ctx *ctx, gpcc_profile *out)
{
    if (!ctx || !out) return fail(GPCC_ERR_ARG, "null argument");
    out->conv_ms = ctx->prof.conv_ms;
    out->conv_launches = ctx->prof.conv_launches;
    out->conv_pair_jobs = ctx->prof.conv_pair_jobs;
    out->fused_ms = ctx->prof.fused_ms;
    out->fused_launches = ctx->prof.fused_launches;
    out->fused_pair_jobs = ctx->prof.fused_pair_jobs;
    return GPCC_OK;
}

// the runtime's blit path moved 12 MB in 220 us (54 GB/s) here; a plain grid of 16-byte copies does it in ~10
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int gpcc_memcpy_d2d(gpcc_ctx *ctx, void *dst, const void *src, int64_t nbytes, void *stream)
{
    if (!ctx || !dst || !src || nbytes < 0) return fail(GPCC_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t n16 = nbytes / 16;
    if (n16 > 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0) {
        k_copy16<<<(unsigned)std::min<int64_t>(cdiv(n16, 256), 8192), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
        LAUNCH_CHECK();
        if (nbytes % 16) HIP_TRY(hipMemcpyAsync(static_cast<char *>(dst) + 16 * n16, static_cast<const char *>(src) + 16 * n16, (size_t)(nbytes % 16), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    } else
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GPCC_OK;
}

// ------------------------------------------------------------------ model
extern "C" int gpcc_model_create(gpcc_ctx *ctx, int channels, int kernel_size, const float *const *t, gpcc_model **out)
{
    if (!ctx || !t || !out) return fail(GPCC_ERR_ARG, "null argument");
    if (channels != 16 && channels != CH && channels != 64) return fail(GPCC_ERR_ARG, "channels must be 16, 32 or 64 (got %d)", channels);
    if (kernel_size != 3 && kernel_size != 5 && kernel_size != 7) return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
    for (int i = 0; i < GPCC_T_COUNT; ++i) if (!t[i]) return fail(GPCC_ERR_ARG, "tensor %d is null", i);
    HIP_TRY(hipSetDevice(ctx->device));
    const int C = channels, K = kernel_size * kernel_size * kernel_size;
    const bool fast = C == CH;   // 32: MFMA fragments and the physical channel order; 16 / 64: upstream layouts (network_any.hip)
    std::vector<float> h;
    std::vector<size_t> off;
    auto push_rows_phys = [&](const float *src, int rows) {  // (rows, C) logical -> physical channel order (32 channels only)
        off.push_back(h.size());
        size_t b = h.size();
        h.resize(b + (size_t)rows * C);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < C; ++c) h[b + (size_t)r * C + (fast ? phys_of(c) : c)] = src[(size_t)r * C + c];
        while (h.size() % 64) h.push_back(0.0f);
    };
    auto push_raw = [&](const float *src, size_t count) {
        off.push_back(h.size());
        h.insert(h.end(), src, src + count);
        while (h.size() % 64) h.push_back(0.0f);
    };
    push_rows_phys(t[GPCC_T_PRIOR_EMB], 256);
    for (int ci = 0; ci < 18; ++ci) {  // (K, C, C) -> MFMA B fragments per offset
        if (!fast) { push_raw(t[GPCC_T_CONV0 + ci], (size_t)K * C * C); continue; }
        off.push_back(h.size());
        size_t b = h.size();
        h.resize(b + 2 * (size_t)K * C * C);
        conv_weight_fragments(t[GPCC_T_CONV0 + ci], K, h.data() + b);
        conv_weight_fragments_t(t[GPCC_T_CONV0 + ci], K, h.data() + b + (size_t)K * C * C);
    }
    push_rows_phys(t[GPCC_T_TEMB], 8);
    for (int s = 0; s < 4; ++s) push_raw(t[GPCC_T_HW1 + s], (size_t)C * C);
    for (int s = 0; s < 4; ++s) push_raw(t[GPCC_T_HB1 + s], (size_t)C);
    for (int s = 0; s < 4; ++s) push_raw(t[GPCC_T_HW2 + s], (size_t)STAGE_M[s] * C);
    for (int s = 0; s < 4; ++s) push_raw(t[GPCC_T_HB2 + s], (size_t)STAGE_M[s]);
    for (int s = 0; s < 4 && fast; ++s) {   // MFMA fragments of the heads: B[k][c] = W[c][k] through the conv fragment layout
        off.push_back(h.size());
        const size_t b = h.size();
        h.resize(b + HEAD_FRAG_FLOATS, 0.0f);
        std::vector<float> wt((size_t)C * C), fr((size_t)C * C);
        for (int c = 0; c < C; ++c) for (int k = 0; k < C; ++k) wt[(size_t)k * C + c] = t[GPCC_T_HW1 + s][(size_t)c * C + k];
        conv_weight_fragments(wt.data(), 1, fr.data());
        std::copy(fr.begin(), fr.end(), h.begin() + b);
        std::fill(wt.begin(), wt.end(), 0.0f);
        for (int j = 0; j < STAGE_M[s]; ++j) for (int k = 0; k < C; ++k) wt[(size_t)k * C + j] = t[GPCC_T_HW2 + s][(size_t)j * C + k];
        conv_weight_fragments(wt.data(), 1, fr.data());
        std::copy(fr.begin(), fr.begin() + 512, h.begin() + b + 1024);          // output half 0 = columns 0..15
        std::copy(t[GPCC_T_HB1 + s], t[GPCC_T_HB1 + s] + C, h.begin() + b + 1536);
        std::copy(t[GPCC_T_HB2 + s], t[GPCC_T_HB2 + s] + STAGE_M[s], h.begin() + b + 1568);
    }
    static const int semb_rows[3] = {2, 4, 16};
    for (int s = 0; s < 3; ++s) push_rows_phys(t[GPCC_T_SEMB + s], semb_rows[s]);
    gpcc_model *m = new gpcc_model();
    m->C = C; m->k = kernel_size; m->K = K;
    if (hipMalloc((void **)&m->slab, h.size() * sizeof(float)) != hipSuccess) { delete m; return fail(GPCC_ERR_NOMEM, "hipMalloc of the model slab failed"); }
    if (hipMemcpy(m->slab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(m->slab); delete m; return fail(GPCC_ERR_HIP, "model upload failed"); }
    size_t i = 0;
    m->prior_emb = m->slab + off[i++];
    for (int ci = 0; ci < 18; ++ci) m->conv[ci] = m->slab + off[i++];
    m->temb = m->slab + off[i++];
    for (int s = 0; s < 4; ++s) m->hw1[s] = m->slab + off[i++];
    for (int s = 0; s < 4; ++s) m->hb1[s] = m->slab + off[i++];
    for (int s = 0; s < 4; ++s) m->hw2[s] = m->slab + off[i++];
    for (int s = 0; s < 4; ++s) m->hb2[s] = m->slab + off[i++];
    for (int s = 0; s < 4; ++s) m->hfrag[s] = fast ? m->slab + off[i++] : nullptr;
    for (int s = 0; s < 3; ++s) m->semb[s] = m->slab + off[i++];
    *out = m;
    return GPCC_OK;
}

extern "C" void gpcc_model_destroy(gpcc_model *m)
{
    if (!m) return;
    if (m->slab) (void)hipFree(m->slab);
    delete m;
}

// ------------------------------------------------------------------ calculate_morton_order
namespace {

constexpr int TB = 256;

template <typename T>
__global__ __launch_bounds__(TB) void k_minmax(const T *__restrict__ x, int64_t n, double *__restrict__ part /* [grid][6] */)
{
    __shared__ double sm[4][6];
    T mn[3], mx[3];
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T v = x[3 * i + a];
            if (!any) { mn[a] = v; mx[a] = v; }
            else { mn[a] = v < mn[a] ? v : mn[a]; mx[a] = v > mx[a] ? v : mx[a]; }
        }
        any = true;
    }
    double dmn[3], dmx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { dmn[a] = any ? (double)mn[a] : 1e300; dmx[a] = any ? (double)mx[a] : -1e300; }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            dmn[a] = fmin(dmn[a], __shfl_xor(dmn[a], d, 64));
            dmx[a] = fmax(dmx[a], __shfl_xor(dmx[a], d, 64));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { sm[wave][a] = dmn[a]; sm[wave][3 + a] = dmx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fmin(v, sm[w][threadIdx.x]) : fmax(v, sm[w][threadIdx.x]);
        part[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// key = x' + y'*M + z'*M^2 with x' = int64(x - min) evaluated in the input dtype (pcc_utils.py:18-20)
template <typename T>
__global__ __launch_bounds__(TB) void k_order_keys(const T *__restrict__ x, int64_t n, T mnx, T mny, T mnz, uint64_t M, uint64_t flip,
                                                   uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t sx = (uint64_t)(int64_t)(T)(x[3 * i] - mnx), sy = (uint64_t)(int64_t)(T)(x[3 * i + 1] - mny), sz = (uint64_t)(int64_t)(T)(x[3 * i + 2] - mnz);
    key[i] = (sx + sy * M + sz * M * M) ^ flip;
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(TB) void k_widen_perm(const uint32_t *__restrict__ idx, int64_t n, int64_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) out[i] = (int64_t)idx[i];
}

template <typename T>
int raster_order_t(gpcc_ctx *ctx, hipStream_t st, const T *x, int64_t n, int64_t *perm)
{
    ctx->arena.reset();
    const int grid = (int)std::min<int64_t>(cdiv(n, TB), 512);
    TAKE(part, double, (size_t)grid * 6);
    GP_TRY(ctx->hstage.reserve((size_t)grid * 48 + 64));
    k_minmax<T><<<grid, TB, 0, st>>>(x, n, part);
    LAUNCH_CHECK();
    double *hp = reinterpret_cast<double *>(ctx->hstage.p);
    HIP_TRY(hipMemcpyAsync(hp, part, (size_t)grid * 48, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < grid; ++b)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], hp[b * 6 + a]); mx[a] = std::max(mx[a], hp[b * 6 + 3 + a]); }
    int64_t mmax = 0;
    for (int a = 0; a < 3; ++a) mmax = std::max<int64_t>(mmax, (int64_t)(T)((T)mx[a] - (T)mn[a]));
    const uint64_t M = (uint64_t)mmax + 1;
    int bits; uint64_t flip = 0;
    if (M <= (1ull << 21)) { const uint64_t top = M * M * M - 1; bits = 1; while (bits < 64 && (top >> bits)) ++bits; }
    else { bits = 64; flip = 0x8000000000000000ull; }  // the reference's int64 arithmetic may wrap here; sort as signed
    TAKE(ka, uint64_t, n); TAKE(kb, uint64_t, n); TAKE(va, uint32_t, n); TAKE(vb, uint32_t, n);
    k_order_keys<T><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(x, n, (T)mn[0], (T)mn[1], (T)mn[2], M, flip, ka, va);
    LAUNCH_CHECK();
    uint64_t *k0 = ka, *k1 = kb; uint32_t *v0 = va, *v1 = vb;
    GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, n, bits));
    k_widen_perm<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(v0, n, perm);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// a1: out = int32(rint(((x / d1) + add) / d2)) evaluated in the array's own type T, one IEEE operation after the other
// (division correctly rounded; -ffp-contract=off keeps the add out of an fma), as numpy / torch on the CPU evaluate
// `xyz / 0.001 + 131072` and `torch.round(xyz / posQ).int()`; values that do not fit an int32 saturate
template <typename T>
__global__ __launch_bounds__(TB) void k_voxelise(const T *__restrict__ x, int64_t n3, T d1, T add, T d2, int flags, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n3) return;
    T t = x[i];
    if (flags & 1) { t = t / d1; t = t + add; }
    if (flags & 2) t = t / d2;
    const double r = rint((double)t);    // exact for float and double; round-half-even like torch.round
    out[i] = r >= 2147483647.0 ? INT32_MAX : (r <= -2147483648.0 ? INT32_MIN : (int32_t)r);
}

}  // namespace

extern "C" int gpcc_voxelise(gpcc_ctx *ctx, const void *xyz, int dtype, int64_t n, double d1, double add, double d2, int flags, int32_t *out, void *stream)
{
    if (!ctx || !xyz || !out) return fail(GPCC_ERR_ARG, "null argument");
    if (n < 0) return fail(GPCC_ERR_ARG, "negative point count");
    if (n == 0) return GPCC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n3 = 3 * n;
    const unsigned grid = (unsigned)cdiv(n3, TB);
    switch (dtype) {
    case GPCC_F32: k_voxelise<float><<<grid, TB, 0, st>>>((const float *)xyz, n3, (float)d1, (float)add, (float)d2, flags, out); break;
    case GPCC_F64: k_voxelise<double><<<grid, TB, 0, st>>>((const double *)xyz, n3, d1, add, d2, flags, out); break;
    default: return fail(GPCC_ERR_ARG, "gpcc_voxelise: dtype must be GPCC_F32 or GPCC_F64");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

extern "C" int gpcc_raster_order(gpcc_ctx *ctx, const void *xyz, int dtype, int64_t n, int64_t *perm, void *stream)
{
    if (!ctx || !xyz || !perm) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0) return fail(GPCC_ERR_ARG, "empty point cloud");
    if (n >= ((int64_t)1 << 32)) return fail(GPCC_ERR_ARG, "too many points");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t want = (size_t)n * 64 + ((size_t)8 << 20);
    GP_TRY(ctx->arena.reserve(want));
    switch (dtype) {
    case GPCC_F32: return raster_order_t<float>(ctx, st, (const float *)xyz, n, perm);
    case GPCC_F64: return raster_order_t<double>(ctx, st, (const double *)xyz, n, perm);
    case GPCC_I32: return raster_order_t<int32_t>(ctx, st, (const int32_t *)xyz, n, perm);
    case GPCC_I64: return raster_order_t<int64_t>(ctx, st, (const int64_t *)xyz, n, perm);
    default: return fail(GPCC_ERR_ARG, "unsupported dtype %d", dtype);
    }
}

// ------------------------------------------------------------------ stage-level entry points
namespace {

__global__ __launch_bounds__(TB) void k_zyx_keys(const int32_t *__restrict__ xyz, int64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    key[i] = rkey3((uint32_t)(xyz[3 * i] + CB), (uint32_t)(xyz[3 * i + 1] + CB), (uint32_t)(xyz[3 * i + 2] + CB));
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(TB) void k_double_coords(const int32_t *__restrict__ in, int64_t n3, int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n3) out[i] = in[i] * 2;
}

// rows given in raster order, logical channels  <->  Morton order, physical channels
__global__ __launch_bounds__(TB) void k_rows_in(const float *__restrict__ in, const uint32_t *__restrict__ m2r, int64_t n, float *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * 32) return;
    const int64_t i = t >> 5; const int c = (int)(t & 31);
    out[i * 32 + phys_of(c)] = in[(int64_t)m2r[i] * 32 + c];
}
__global__ __launch_bounds__(TB) void k_rows_out(const float *__restrict__ in, const uint32_t *__restrict__ m2r, int64_t n, float *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * 32) return;
    const int64_t i = t >> 5; const int c = (int)(t & 31);
    out[(int64_t)m2r[i] * 32 + c] = in[i * 32 + phys_of(c)];
}
// the same for any channel count, logical order on both sides (network_any.hip); dir 0: raster -> Morton, 1: Morton -> raster
__global__ __launch_bounds__(TB) void k_rows_any(const float *__restrict__ in, const uint32_t *__restrict__ m2r, int64_t n, int C, int dir, float *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * C) return;
    const int64_t i = t / C; const int c = (int)(t % C);
    if (dir == 0) out[i * C + c] = in[(int64_t)m2r[i] * C + c];
    else out[(int64_t)m2r[i] * C + c] = in[i * C + c];
}

}  // namespace

extern "C" int gpcc_sort_zyx(gpcc_ctx *ctx, const int32_t *xyz, int64_t n, uint32_t *perm, void *stream)
{
    if (!ctx || !xyz || !perm || n <= 0) return fail(GPCC_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    GP_TRY(ctx->arena.reserve((size_t)n * 64 + ((size_t)8 << 20)));
    ctx->arena.reset();
    TAKE(ka, uint64_t, n); TAKE(kb, uint64_t, n); TAKE(va, uint32_t, n); TAKE(vb, uint32_t, n);
    k_zyx_keys<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(xyz, n, ka, va);
    LAUNCH_CHECK();
    uint64_t *k0 = ka, *k1 = kb; uint32_t *v0 = va, *v1 = vb;
    GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, n, 63));
    HIP_TRY(hipMemcpyAsync(perm, v0, 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return device_error_check(ctx);
}

extern "C" int gpcc_build_octree(gpcc_ctx *ctx, const int32_t *xyz, int64_t n, int32_t *levels_out, int64_t *level_nodes_out,
                                 int32_t **coords_out_host, uint8_t **occ_out_host, int64_t cap_nodes, void *stream)
{
    if (!ctx || !xyz || !levels_out || !level_nodes_out) return fail(GPCC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    size_t want = (size_t)n * 400 + ((size_t)32 << 20);
    int rc = GPCC_OK;
    Tree T;
    for (int attempt = 0; attempt < 4; ++attempt) {
        GP_TRY(ctx->arena.reserve(want));
        ctx->arena.reset();
        T = Tree();
        rc = tree_build(ctx, st, xyz, n, &T);
        if (rc == GPCC_OK) rc = tree_ranks(ctx, st, &T);
        if (rc != GPCC_ERR_NOMEM) break;
        want *= 2;
    }
    GP_TRY(rc);
    *levels_out = T.L;
    for (int d = 0; d < T.L; ++d) {
        level_nodes_out[d] = T.lv[d].n;
        if (coords_out_host && occ_out_host) {
            if (T.lv[d].n > cap_nodes) return fail(GPCC_ERR_ARG, "level %d has %lld nodes, capacity %lld", d, (long long)T.lv[d].n, (long long)cap_nodes);
            size_t mk = ctx->arena.mark();
            TAKE(dx, int32_t, 3 * T.lv[d].n); TAKE(dox, uint8_t, T.lv[d].n);
            GP_TRY(level_to_raster(ctx, st, &T.lv[d], T.bias, dx, dox));
            HIP_TRY(hipMemcpyAsync(coords_out_host[d], dx, 12 * (size_t)T.lv[d].n, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(occ_out_host[d], dox, (size_t)T.lv[d].n, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            ctx->arena.rewind(mk);
        }
    }
    return GPCC_OK;
}

extern "C" int gpcc_conv3d(gpcc_ctx *ctx, const int32_t *xyz_sorted, int64_t n, int channels, int kernel_size, const float *in_dev,
                           const float *w_host, const float *res_dev, int relu, float *out_dev, int64_t *pairs_out, void *stream)
{
    if (!ctx || !xyz_sorted || !in_dev || !w_host || !out_dev) return fail(GPCC_ERR_ARG, "null argument");
    if (channels != 16 && channels != CH && channels != 64) return fail(GPCC_ERR_ARG, "channels must be 16, 32 or 64");
    if (kernel_size != 3 && kernel_size != 5 && kernel_size != 7) return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
    if (channels != CH && (relu & 2)) return fail(GPCC_ERR_ARG, "the pair-plan convolution is the 32-channel path");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int K = kernel_size * kernel_size * kernel_size;
    const int C = channels;
    size_t want = (size_t)n * (size_t)(4 * 125 * 2 + K * 81 / 16 + 1300 + 12 * C) + (size_t)K * C * C * 4 + ((size_t)32 << 20);
    for (int attempt = 0;; ++attempt) {
        GP_TRY(ctx->arena.reserve(want));
        ctx->arena.reset();
        int rc = [&]() -> int {
            // The points become the finest STORED level by building the tree of 2*xyz (every point alone in
            // its voxel pair), so the neighbour map comes from the production top-down path.
            TAKE(x2, int32_t, 3 * n);
            k_double_coords<<<(unsigned)cdiv(3 * n, TB), TB, 0, st>>>(xyz_sorted, 3 * n, x2);
            LAUNCH_CHECK();
            Tree T;
            GP_TRY(tree_build(ctx, st, x2, n, &T));
            GP_TRY(tree_ranks(ctx, st, &T));
            const Level *fin = &T.lv[T.L - 1];
            if (fin->n != n) return fail(GPCC_ERR_ARG, "internal: finest level has %lld nodes for %lld points", (long long)fin->n, (long long)n);
            // tile lists of the whole tree from the production top-down path (tiles.hip); the convolution runs on the finest
            TAKE(pairs, unsigned long long, MAXLV);
            HIP_TRY(hipMemsetAsync(pairs, 0, 8 * MAXLV, st));
            const int NPc = cell_map_entries(kernel_size);
            TileLevel tl[MAXLV];
            const int32_t *cell_prev = nullptr;
            for (int d = 0; d < T.L; ++d) {
                int32_t *own = nullptr;
                if (d + 1 < T.L) { TAKE(cm, int32_t, (int64_t)NPc * T.lv[d].n); own = cm; }
                tl[d] = TileLevel{&T.lv[d], d ? &T.lv[d - 1] : nullptr, cell_prev, own};
                cell_prev = own;
            }
            const int R = conv_pick_rows(n, kernel_size);
            TilePool pool;
            GP_TRY(tiles_build(ctx, st, tl, T.L, kernel_size, R, conv_pick_height(n, R), &pool, pairs));
            ConvTiles tiles;
            const int64_t zero_base[1] = {0};
            GP_TRY(tiles_view(ctx, st, pool, T.L - 1, T.L, zero_base, &tiles));
            // weights -> B-fragment order (32 channels); other widths keep the upstream (K, C, C) layout (network_any.hip)
            std::vector<float> wf(C == CH ? (size_t)K * 2048 : (size_t)K * C * C);
            if (C == CH) {
                conv_weight_fragments(w_host, K, wf.data());
                conv_weight_fragments_t(w_host, K, wf.data() + (size_t)K * 1024);
            } else std::copy(w_host, w_host + (size_t)K * C * C, wf.begin());
            TAKE(dw, float, wf.size()); TAKE(xin, float, n * C); TAKE(xres, float, n * C); TAKE(xout, float, n * C);
            HIP_TRY(hipMemcpyAsync(dw, wf.data(), wf.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (C == CH) {
                k_rows_in<<<(unsigned)cdiv(n * 32, TB), TB, 0, st>>>(in_dev, fin->m2r, n, xin);
                if (res_dev) k_rows_in<<<(unsigned)cdiv(n * 32, TB), TB, 0, st>>>(res_dev, fin->m2r, n, xres);
            } else {
                k_rows_any<<<(unsigned)cdiv(n * C, TB), TB, 0, st>>>(in_dev, fin->m2r, n, C, 0, xin);
                if (res_dev) k_rows_any<<<(unsigned)cdiv(n * C, TB), TB, 0, st>>>(res_dev, fin->m2r, n, C, 0, xres);
            }
            LAUNCH_CHECK();
            ConvBatch cb = {};
            cb.C = C;
            cb.job[0] = ConvJob{xin, dw, res_dev ? xres : nullptr, xout};
            if (relu & 2) {
                // test knob: the pair-plan form of the decoder's small levels (fused.hpp) -- products + ordered sums on a level-wide plan
                if (T.L < 2 || !fused_level_ok(n, kernel_size)) return fail(GPCC_ERR_ARG, "the pair-plan convolution takes levels of at most %lld nodes below a parent level", (long long)FUSE_MAX_NODES);
                PairPlan plan;
                GP_TRY(pairplan_build(ctx, st, &T.lv[T.L - 2], tl[T.L - 1].cell_par, fin, nullptr, kernel_size, &plan, nullptr));
                TAKE(P, float, plan.pcap * 32);
                GP_TRY(plan_conv(st, plan, cb.job[0], P, relu & 1));
            } else
            GP_TRY(sparse_conv(nullptr, -1, st, cb, 1, tiles, n, relu & 1));
            if (C == CH) k_rows_out<<<(unsigned)cdiv(n * 32, TB), TB, 0, st>>>(xout, fin->m2r, n, out_dev);
            else k_rows_any<<<(unsigned)cdiv(n * C, TB), TB, 0, st>>>(xout, fin->m2r, n, C, 1, out_dev);
            LAUNCH_CHECK();
            unsigned long long hpairs = 0;
            HIP_TRY(hipMemcpyAsync(&hpairs, pairs + (T.L - 1), 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (pairs_out) *pairs_out = (int64_t)hpairs;
            return GPCC_OK;
        }();
        if (rc != GPCC_ERR_NOMEM || attempt >= 3) return rc;
        want *= 2;
    }
}

extern "C" int gpcc_head_cdf(gpcc_ctx *ctx, const float *x_dev, int64_t n, int channels, int m, const float *w1, const float *b1,
                             const float *w2, const float *b2, float *prob_dev, uint16_t *cdf_dev, void *stream)
{
    if (!ctx || !x_dev || !w1 || !b1 || !w2 || !b2) return fail(GPCC_ERR_ARG, "null argument");
    if (channels != 16 && channels != CH && channels != 64) return fail(GPCC_ERR_ARG, "channels must be 16, 32 or 64");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    GP_TRY(ctx->arena.reserve((size_t)1 << 20));
    ctx->arena.reset();
    const int C = channels;
    const size_t o_b1 = (size_t)C * C, o_w2 = o_b1 + C, o_b2 = o_w2 + 16 * (size_t)C;
    TAKE(dw, float, o_b2 + 16 + 64);
    HIP_TRY(hipMemcpyAsync(dw, w1, 4 * (size_t)C * C, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dw + o_b1, b1, 4 * (size_t)C, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dw + o_w2, w2, 4 * (size_t)m * C, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dw + o_b2, b2, 4 * (size_t)m, hipMemcpyHostToDevice, st));
    HeadArgs ha = {};
    ha.C = C;
    ha.x = x_dev; ha.n = n; ha.stage_m = m; ha.w1 = dw; ha.b1 = dw + o_b1; ha.w2 = dw + o_w2; ha.b2 = dw + o_b2;
    ha.prob = prob_dev; ha.cdf = cdf_dev; ha.mode = 2;
    GP_TRY(head_cdf(st, ha));
    HIP_TRY(hipStreamSynchronize(st));
    return device_error_check(ctx);
}

// One stream of the container as gpcc_encode writes it for a level of n nodes (rangecoder.hpp: rc_plan, version 3):
// chunk_log2 = 0 -> one lane, the bare coder bytes; else the chunk table (rangecoder.hpp: rc_table_*), then the chunks (forward lane + reversed
// backward lane each).
extern "C" int gpcc_rc_encode(gpcc_ctx *ctx, const uint16_t *cdf_dev, int lp, const uint8_t *sym_dev, int64_t n, int chunk_log2,
                              const uint8_t **bytes_out, int64_t *nbytes_out, void *stream)
{
    if (!ctx || !cdf_dev || !sym_dev || !bytes_out || !nbytes_out || n <= 0) return fail(GPCC_ERR_ARG, "bad argument");
    if (chunk_log2 != 0 && (chunk_log2 < 6 || chunk_log2 > 14)) return fail(GPCC_ERR_ARG, "chunk_log2 must be 0 or 6..14");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int version = ctx->container_version;
    const RcPlan pl = rc_plan(n, chunk_log2, version);
    const int64_t S = chunk_log2 ? (int64_t)1 << pl.llog : n;
    const int nch = (int)pl.nlanes;
    const uint32_t stride = rc_scratch_stride((uint32_t)std::min<int64_t>(S, n));
    GP_TRY(ctx->arena.reserve((size_t)n * 8 + (size_t)nch * S * 4 + 2 * (size_t)nch * stride + ((size_t)4 << 20)));
    ctx->arena.reset();
    std::vector<RcChunk> chunks((size_t)nch);
    for (int c = 0; c < nch; ++c) chunks[(size_t)c] = RcChunk{(uint32_t)c, (uint32_t)nch, (uint32_t)pl.lane_syms(n, (uint32_t)c), 0, 0, 0};
    TAKE(lohi, uint32_t, (int64_t)nch * S); TAKE(dch, RcChunk, nch); TAKE(dcnt, uint32_t, nch + 1); TAKE(doff, uint32_t, nch + 1);
    TAKE(scratch, uint8_t, (size_t)nch * stride); TAKE(payload, uint8_t, (size_t)nch * stride);
    HIP_TRY(hipMemcpyAsync(dch, chunks.data(), sizeof(RcChunk) * (size_t)nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    GP_TRY(rc_pack_lohi(st, cdf_dev, lp, sym_dev, n, pl.llog, (uint32_t)nch, lohi));
    GP_TRY(rc_encode_launch(st, lohi, dch, nch, scratch, stride, dcnt, chunk_log2 ? rc_coder_of_version(version) : RC_CODER_CARRYLESS));
    GP_TRY(exclusive_scan_u32(ctx, st, dcnt, doff, nch, doff + nch));
    GP_TRY(rc_compact_launch(st, scratch, stride, dcnt, doff, nullptr, nch, payload, pl.dual ? dch : nullptr));
    std::vector<uint32_t> hcnt((size_t)nch + 1);
    HIP_TRY(hipMemcpyAsync(hcnt.data(), dcnt, 4 * (size_t)nch, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hcnt.data() + nch, doff + nch, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t total = hcnt[(size_t)nch];
    size_t hdr = 0;
    auto chunk_bytes = [&](uint32_t c) { const size_t l = 2 * (size_t)c; return hcnt[l] + (l + 1 < (size_t)nch ? hcnt[l + 1] : 0u); };
    if (chunk_log2) hdr = rc_table_size(chunk_bytes, (uint32_t)((nch + 1) / 2));
    GP_TRY(ctx->hbytes.reserve(total + hdr + 16));
    uint8_t *out = ctx->hbytes.p;
    if (chunk_log2 && rc_table_put(out, chunk_bytes, (uint32_t)((nch + 1) / 2)) != hdr) return fail(GPCC_ERR_HIP, "internal: chunk table size mismatch");
    if (total) HIP_TRY(hipMemcpyAsync(out + hdr, payload, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *bytes_out = out; *nbytes_out = (int64_t)(total + hdr);
    return GPCC_OK;
}

extern "C" int gpcc_rc_decode(gpcc_ctx *ctx, const uint16_t *cdf_dev, int lp, const uint8_t *bytes, int64_t nbytes, int64_t n,
                              int chunk_log2, uint8_t *sym_dev, void *stream)
{
    if (!ctx || !cdf_dev || !bytes || !sym_dev || n <= 0 || nbytes < 0) return fail(GPCC_ERR_ARG, "bad argument");
    if (chunk_log2 != 0 && (chunk_log2 < 6 || chunk_log2 > 14)) return fail(GPCC_ERR_ARG, "chunk_log2 must be 0 or 6..14");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int version = ctx->container_version;
    const RcPlan pl = rc_plan(n, chunk_log2, version);
    const int64_t S = chunk_log2 ? (int64_t)1 << pl.llog : n;
    const int nch = (int)pl.nlanes;
    GP_TRY(ctx->arena.reserve((size_t)nbytes + sizeof(RcChunk) * (size_t)nch + (size_t)rc_rows_capacity(nch, S) * 32 + (size_t)n + ((size_t)4 << 20)));
    ctx->arena.reset();
    std::vector<RcChunk> chunks((size_t)nch);
    uint32_t win = 0;
    constexpr int64_t FRONT = 16;   // the uploaded copy starts 16 bytes into its buffer: a backwards lane's staging loads reach up to 3 bytes in front of its chunk
    if (const char *err = rc_parse_table(bytes, FRONT, nbytes, pl, n, chunk_log2 ? version : 0, chunks.data(), &win)) return fail(GPCC_ERR_FORMAT, "%s", err);
    TAKE(db, uint8_t, nbytes + FRONT + 16); TAKE(dch, RcChunk, nch);
    TAKE(rows, uint16_t, rc_rows_capacity(nch, S) * rc_row_stride(lp) + 64);
    TAKE(symbuf, uint8_t, n + 4);   // the decoder stores groups of four
    GP_TRY(rc_pack_rows(st, cdf_dev, lp, n, pl.llog, (uint32_t)nch, rows));
    HIP_TRY(hipMemsetAsync(db, 0, FRONT, st));
    HIP_TRY(hipMemcpyAsync(db + FRONT, bytes, (size_t)nbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dch, chunks.data(), sizeof(RcChunk) * (size_t)nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    GP_TRY(rc_decode_launch(st, rows, lp, db, dch, nch, win, pl.dual, symbuf, chunk_log2 ? rc_coder_of_version(version) : RC_CODER_CARRYLESS));
    HIP_TRY(hipMemcpyAsync(sym_dev, symbuf, (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return device_error_check(ctx);
}
