// primitives.hip -- exclusive scan + radix sort for gfx950 (wave64).
//
// Radix sort: 8-bit digits, three kernels per pass
//   k_radix_hist     per 4096-key tile (256 threads): LDS-atomic digit histogram -> hist[digit][tile]
//   exclusive scan   over the digit-major table (global digit offsets per tile)
//   k_radix_scatter  one 256-thread workgroup per tile; each wave ranks its 1 024 keys stably by wave ballots (8 ballots give the
//                    lanes that hold the same digit, popcount of the lower lanes is the rank; running per-digit counters of the wave in
//                    LDS), the waves' counts are chained, the tile is ordered by digit in LDS and leaves in per-digit runs.
// HBM-bound integer work: per pass it reads keys twice and writes them once.
#include <algorithm>
#include "primitives.hpp"
#include <atomic>

namespace gpcc {

constexpr int SCAN_T = 256;             // threads per scan block
constexpr int SCAN_E = 4;               // elements per thread
constexpr int SCAN_TILE = SCAN_T * SCAN_E;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix,
// *total = block sum (valid in every thread).
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *total, uint32_t *lds /*>=8*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    uint32_t base = wave == 0 ? 0 : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2;
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------ one-wave tiles: scans WITHOUT an LDS allocation
// Round 4: the scans of the octree stage run on the context's second stream beside a convolution whose four waves hold all 160 KiB of
// every CU's LDS -- a workgroup that wants even the 40 bytes of block_excl_scan_256 waits for a conv workgroup to retire (the same
// launch: 5.6 us on a free device, 40-160 us beside a convolution; tools/scan_hist.py).  A 64-thread workgroup needs none: a tile is
// 4 096 elements = 16 coalesced 16-byte loads per lane (row i = elements 256 i .. 256 i + 255, lane l its elements 4 l .. 4 l + 3), the
// 16 row sums are scanned across the wave with cross-lane shuffles (no LDS memory involved), rows chained through a register.
constexpr int WT_ROWS = 16, WT_TILE = 64 * 4 * WT_ROWS;
__device__ __forceinline__ void wave_tile_load(const uint32_t *__restrict__ in, int64_t base, int64_t n, int lane, uint4 v[WT_ROWS])
{
    if (base + WT_TILE <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base) + lane;
#pragma unroll
        for (int i = 0; i < WT_ROWS; ++i) v[i] = p[i * 64];
    } else {
#pragma unroll
        for (int i = 0; i < WT_ROWS; ++i) {
            const int64_t e = base + (int64_t)(i * 64 + lane) * 4;
            v[i].x = e < n ? in[e] : 0u; v[i].y = e + 1 < n ? in[e + 1] : 0u; v[i].z = e + 2 < n ? in[e + 2] : 0u; v[i].w = e + 3 < n ? in[e + 3] : 0u;
        }
    }
}
// ex[i] = sum of the tile's elements in front of this lane's first element of row i; returns the tile's sum (in every lane)
__device__ __forceinline__ uint32_t wave_tile_scan(const uint4 v[WT_ROWS], int lane, uint32_t ex[WT_ROWS])
{
    uint32_t s[WT_ROWS], inc[WT_ROWS];
#pragma unroll
    for (int i = 0; i < WT_ROWS; ++i) { s[i] = v[i].x + v[i].y + v[i].z + v[i].w; inc[i] = s[i]; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {      // the 16 rows' scans side by side: 16 independent shuffles per step
        uint32_t t[WT_ROWS];
#pragma unroll
        for (int i = 0; i < WT_ROWS; ++i) t[i] = (uint32_t)__shfl_up((int)inc[i], d, 64);
#pragma unroll
        for (int i = 0; i < WT_ROWS; ++i) inc[i] += lane >= d ? t[i] : 0u;
    }
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < WT_ROWS; ++i) {
        ex[i] = carry + inc[i] - s[i];
        carry += (uint32_t)__shfl((int)inc[i], 63, 64);
    }
    return carry;
}
__device__ __forceinline__ void wave_tile_store(uint32_t *__restrict__ out, int64_t base, int64_t n, int lane, const uint4 v[WT_ROWS], const uint32_t ex[WT_ROWS], uint32_t carry)
{
    const bool full = base + WT_TILE <= n;
    uint4 *p = reinterpret_cast<uint4 *>(out + base) + lane;
#pragma unroll
    for (int i = 0; i < WT_ROWS; ++i) {
        uint4 o;
        o.x = carry + ex[i]; o.y = o.x + v[i].x; o.z = o.y + v[i].y; o.w = o.z + v[i].z;
        if (full) p[i * 64] = o;
        else {
            const int64_t e = base + (int64_t)(i * 64 + lane) * 4;
            if (e < n) out[e] = o.x;
            if (e + 1 < n) out[e + 1] = o.y;
            if (e + 2 < n) out[e + 2] = o.z;
            if (e + 3 < n) out[e + 3] = o.w;
        }
    }
}

// one wave walks the whole array (n <= SCAN_SINGLE_MAX: at most four tiles); blockIdx.x picks one of two independent scans
__global__ __launch_bounds__(64) void k_scan_wave(const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, int64_t n, uint32_t *total_out)
{
    const uint32_t *in = blockIdx.x ? in1 : in0;
    uint32_t *out = blockIdx.x ? out1 : out0;
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < n; b0 += WT_TILE) {
        uint4 v[WT_ROWS];
        uint32_t ex[WT_ROWS];
        wave_tile_load(in, b0, n, lane, v);
        const uint32_t total = wave_tile_scan(v, lane, ex);
        wave_tile_store(out, b0, n, lane, v, ex, carry);      // in == out allowed: the tile is in registers
        carry += total;
    }
    if (lane == 0 && total_out && blockIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t *__restrict__ bsum, int64_t n)
{
    __shared__ uint32_t lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_E;
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e)
        if (base + e < n) s += in[base + e];
    uint32_t total;
    block_excl_scan_256(s, &total, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: exclusive scan of bsum[0..nb) in place, total -> *total_out (may be null)
__global__ __launch_bounds__(SCAN_T) void k_scan_bsums(uint32_t *bsum, int64_t nb, uint32_t *total_out)
{
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += SCAN_T) {
        int64_t i = b0 + threadIdx.x;
        uint32_t v = i < nb ? bsum[i] : 0;
        uint32_t total;
        uint32_t ex = block_excl_scan_256(v, &total, lds);
        if (i < nb) bsum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_apply(const uint32_t *in, uint32_t *out, const uint32_t *__restrict__ bsum, int64_t n)
{
    __shared__ uint32_t lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_E;
    uint32_t v[SCAN_E];
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) {
        v[e] = base + e < n ? in[base + e] : 0;
        s += v[e];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan_256(s, &total, lds) + bsum[blockIdx.x];
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) {
        if (base + e < n) out[base + e] = ex;
        ex += v[e];
    }
}

// one block walks the whole array (carry in a register): for the many short scans of the small octree levels one launch
// instead of three.  in == out allowed.
__global__ __launch_bounds__(SCAN_T) void k_scan_single(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total_out)
{
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < n; b0 += SCAN_TILE) {
        const int64_t base = b0 + (int64_t)threadIdx.x * SCAN_E;
        uint32_t v[SCAN_E];
        uint32_t s = 0;
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) {
            v[e] = base + e < n ? in[base + e] : 0;
            s += v[e];
        }
        uint32_t total;
        uint32_t ex = carry + block_excl_scan_256(s, &total, lds);
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) {
            if (base + e < n) out[base + e] = ex;
            ex += v[e];
        }
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}
constexpr int64_t SCAN_SINGLE_MAX = 16 * SCAN_TILE;
// two independent short scans of the same length in ONE launch (workgroup 0 / 1): the two flag scans of the rank derivation
__global__ __launch_bounds__(SCAN_T) void k_scan_single2(const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, int64_t n)
{
    __shared__ uint32_t lds[8];
    const uint32_t *in = blockIdx.x ? in1 : in0;
    uint32_t *out = blockIdx.x ? out1 : out0;
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < n; b0 += SCAN_TILE) {
        const int64_t base = b0 + (int64_t)threadIdx.x * SCAN_E;
        uint32_t v[SCAN_E];
        uint32_t s = 0;
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) {
            v[e] = base + e < n ? in[base + e] : 0;
            s += v[e];
        }
        uint32_t total;
        uint32_t ex = carry + block_excl_scan_256(s, &total, lds);
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) {
            if (base + e < n) out[base + e] = ex;
            ex += v[e];
        }
        carry += total;
    }
}

// ------------------------------------------------------------------ single-pass scan (decoupled look-back)
// Round 3 scanned every array above 16 k elements with three launches (reduce, scan of the block sums, apply): 1080 scan launches
// and 5.6 ms of stream time per encode + decode step, at ~300 GB/s.  Here ONE launch: a workgroup draws a tile (4096 elements:
// four 16-byte loads per thread) from a ticket counter (round 6: one of eight, see the kernel) -- tiles are started in order, so a tile only ever waits for tiles
// that are running or done --, scans it, publishes (epoch | AGGREGATE | sum) as ONE 8-byte agent-scope word (the value is the
// flag: no fence), looks back over its predecessors' words 64 at a time until it meets an INCLUSIVE prefix, publishes its own
// inclusive prefix and writes the tile.  The status words carry the launch's epoch: no reset between launches.
constexpr int LB_T = 64, LB_TILE = WT_TILE;    // one wave per tile: no LDS allocation (see "one-wave tiles" above)
constexpr int64_t LB_MAX_TILES = 65536;
constexpr int LB_SHARDS = 8, LB_SHARD_STRIDE = 32;      // ticket counters per scan state, 128 bytes apart
constexpr size_t LB_TAIL = (size_t)LB_SHARDS * LB_SHARD_STRIDE * 4;
constexpr int64_t LB_SHARD_MIN_TILES = 1024;            // launches of at least this many tiles (4 M elements) draw from all the counters
constexpr unsigned long long LB_AGG = 1ull << 32, LB_INC = 2ull << 32;
__device__ __forceinline__ unsigned long long lb_pack(uint32_t epoch, unsigned long long flag, uint32_t v) { return ((unsigned long long)epoch << 34) | flag | v; }

__global__ __launch_bounds__(LB_T) void k_scan_lookback(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n, unsigned long long *status, uint32_t *ticket,
                                                        uint32_t epoch, uint32_t *__restrict__ total_out, uint32_t *err, uint32_t spin_limit, uint32_t skip_tile, uint32_t shards)
{
    const int lane = threadIdx.x;
    uint32_t t = 0;
    if (lane == 0) {
        // LB_SHARDS ticket counters, workgroup b draws from counter b mod LB_SHARDS and becomes tile LB_SHARDS x ticket + (b mod LB_SHARDS): device-scope
        // atomics on ONE word serialise at ~10 ns each (measured: 24 of the 52 us of a 2 442-tile launch were the tiles queueing for their ticket).
        // Within a shard tiles start in id order; across shards a started tile can meet a predecessor whose workgroup has not started only while the
        // dispatcher is about to start it (workgroups leave the dispatcher in blockIdx order, so the shards' counts differ by at most one) -- and should
        // that ever not hold, the bounded wait below ends the launch with the context's error word instead of a hang.
        // (shards = 1 below LB_SHARD_MIN_TILES: the scans of the codec -- a few hundred tiles, beside a convolution on the other stream -- keep the strict
        // start order of ONE counter, where a tile never waits for a workgroup that has not started; their tickets cost 2-6 us)
        const uint32_t c = blockIdx.x % shards;
        const uint32_t mine = (gridDim.x - c + shards - 1u) / shards;      // tickets of this shard in this launch
        uint32_t *tk = ticket + c * LB_SHARD_STRIDE;
        const uint32_t k = atomicAdd(tk, 1u);
        if (k == mine - 1u) atomicExch(tk, 0u);              // every ticket of the shard is taken: ready for the next launch on this stream
        t = k * shards + c;
    }
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    const int64_t base = (int64_t)tile * LB_TILE;
    uint4 v[WT_ROWS];
    uint32_t ex[WT_ROWS];
    wave_tile_load(in, base, n, lane, v);
    const uint32_t total = wave_tile_scan(v, lane, ex);
    uint32_t excl = 0;
    // (skip_tile: developer fault injection -- that tile never publishes; 0xFFFFFFFF: none)
    if (lane == 0 && tile != skip_tile) __hip_atomic_store(status + tile, lb_pack(epoch, tile == 0 ? LB_INC : LB_AGG, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile > 0) {
        int64_t j = (int64_t)tile - 1;       // nearest predecessor not yet accounted for
        uint32_t idle = 0;
        for (;;) {
            const int64_t idx = j - lane;
            // before tile 0: a virtual inclusive prefix of 0
            const unsigned long long w = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lb_pack(epoch, LB_INC, 0u);
            const bool valid = (uint32_t)(w >> 34) == epoch && (w & (LB_AGG | LB_INC)) != 0ull;
            const bool inc = valid && (w & LB_INC) != 0ull;
            const unsigned long long bad = __ballot(!valid), pre = __ballot(inc);
            const int first_bad = bad ? __builtin_ctzll(bad) : 64, first_inc = pre ? __builtin_ctzll(pre) : 64;
            const int take = first_inc < first_bad ? first_inc + 1 : first_bad;   // lanes 0 .. take - 1 are usable in order
            uint32_t part = lane < take ? (uint32_t)w : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
            excl += part;
            if (first_inc < first_bad) break;
            j -= take;
            if (take == 0) {
                __builtin_amdgcn_s_sleep(1);
                // predecessors are running workgroups (tickets are drawn in start order): no progress for ~10 s means the launch's
                // state was damaged.  Raise the context's sticky error word (host memory: the caller reads it after its final sync and
                // fails the call, device_error_check) and let the launch run out -- this tile publishes an inclusive prefix, so its
                // successors stop waiting too; the output is garbage, the process and its other contexts live on (a trap would end them all)
                if (++idle > spin_limit) {
                    if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            } else idle = 0;
        }
        if (lane == 0) __hip_atomic_store(status + tile, lb_pack(epoch, LB_INC, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0 && total_out && tile == gridDim.x - 1u) *total_out = excl + total;
    wave_tile_store(out, base, n, lane, v, ex, excl);
}

int device_error_check(gpcc_ctx *ctx)
{
    if (!ctx || !ctx->dev_err || *static_cast<volatile uint32_t *>(ctx->dev_err) == 0u) return GPCC_OK;
    *static_cast<volatile uint32_t *>(ctx->dev_err) = 0u;
    for (auto &ss : ctx->scan_states) {   // tickets and status words of an abandoned launch are undefined: start every stream's state over
        (void)hipStreamSynchronize(ss.st);
        (void)hipMemsetAsync(ss.status, 0, 8 * (size_t)LB_MAX_TILES + LB_TAIL, ss.st);
        ss.epoch = 0u;
    }
    (void)hipGetLastError();
    return fail(GPCC_ERR_HIP, "a device-wide scan made no progress (look-back state damaged): the results of this call are invalid; the scan state was reset");
}

// the scan state of (context, stream): created on first use (one hipMalloc + memset per stream of a context)
static int scan_state(gpcc_ctx *ctx, hipStream_t st, gpcc_ctx::ScanState **out)
{
    for (size_t i = 0; i < ctx->scan_states.size(); ++i)
        if (ctx->scan_states[i].st == st) {
            // most recently used last: the context's own streams and the caller's usual stream stay, a stream seen once is what goes
            if (i + 1 != ctx->scan_states.size()) std::rotate(ctx->scan_states.begin() + (ptrdiff_t)i, ctx->scan_states.begin() + (ptrdiff_t)i + 1, ctx->scan_states.end());
            *out = &ctx->scan_states.back();
            return GPCC_OK;
        }
    // A caller that cycles through raw stream handles must not grow this for ever (a context's own calls use its three streams and
    // the caller's): beyond SCAN_STATES_MAX the LEAST RECENTLY USED state of a foreign stream is retired (never one of the context's own
    // streams: round 5 evicted first-in-first-out, i.e. exactly those, and paid a drain + free + malloc on every later call).  Its handle may
    // already be destroyed, so it is not synchronised -- the device is: nothing of any stream is in flight when the buffer changes hands.
    constexpr size_t SCAN_STATES_MAX = 8;
    unsigned long long *recycled = nullptr;
    if (ctx->scan_states.size() >= SCAN_STATES_MAX) {
        size_t victim = ctx->scan_states.size();
        for (size_t i = 0; i < ctx->scan_states.size(); ++i) {
            const hipStream_t s = ctx->scan_states[i].st;
            if (s != ctx->side && s != ctx->xfer) { victim = i; break; }
        }
        if (victim == ctx->scan_states.size()) victim = 0;
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        recycled = ctx->scan_states[victim].status;
        ctx->scan_states.erase(ctx->scan_states.begin() + (ptrdiff_t)victim);
    }
    if (!ctx->dev_err) {
        void *h = nullptr;
        HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
        *static_cast<volatile uint32_t *>(h) = 0u;
        void *d = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
        ctx->dev_err = static_cast<uint32_t *>(h);
        ctx->dev_err_dev = static_cast<uint32_t *>(d);
    }
    gpcc_ctx::ScanState ns = {st, nullptr, nullptr, 0u};
    void *p = recycled;
    if (!p) HIP_TRY(hipMalloc(&p, 8 * (size_t)LB_MAX_TILES + LB_TAIL));
    // ON THE STREAM that will use it: a hipMemset on the null stream is not ordered with a non-blocking stream, and may return before it
    // has run -- it then zeroed the ticket word under a running scan (found with tools/inflight_side.py: tiles drawn twice, others
    // never, their successors spinning for ever)
    HIP_TRY(hipMemsetAsync(p, 0, 8 * (size_t)LB_MAX_TILES + LB_TAIL, st));
    ns.status = static_cast<unsigned long long *>(p);
    ns.ticket = reinterpret_cast<uint32_t *>(ns.status + LB_MAX_TILES);
    ctx->scan_states.push_back(ns);
    *out = &ctx->scan_states.back();
    return GPCC_OK;
}

// the one-wave kernels move 16-byte words (GAUSPCC_SCAN_WAVE=0: the 256-thread kernels with their LDS words, kept as the cross-check)
static bool scan_wave_ok(const uint32_t *in, const uint32_t *out)
{
    static const bool on = dev_env_int("GAUSPCC_SCAN_WAVE", 1) != 0;
    return on && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
}

int exclusive_scan_u32(gpcc_ctx *ctx, hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total_dev);
int exclusive_scan_pair_u32(gpcc_ctx *ctx, hipStream_t st, const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, int64_t n)
{
    if (n > 0 && n <= SCAN_SINGLE_MAX) {
        if (scan_wave_ok(in0, out0) && scan_wave_ok(in1, out1)) k_scan_wave<<<2, 64, 0, st>>>(in0, out0, in1, out1, n, nullptr);
        else k_scan_single2<<<2, SCAN_T, 0, st>>>(in0, out0, in1, out1, n);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    GP_TRY(exclusive_scan_u32(ctx, st, in0, out0, n, nullptr));
    return exclusive_scan_u32(ctx, st, in1, out1, n, nullptr);
}

int exclusive_scan_u32(gpcc_ctx *ctx, hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total_dev)
{
    if (n <= 0) {
        if (total_dev) HIP_TRY(hipMemsetAsync(total_dev, 0, 4, st));
        return GPCC_OK;
    }
    if (n <= SCAN_SINGLE_MAX) {
        if (scan_wave_ok(in, out)) k_scan_wave<<<1, 64, 0, st>>>(in, out, in, out, n, total_dev);
        else k_scan_single<<<1, SCAN_T, 0, st>>>(in, out, n, total_dev);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    static const bool lookback = dev_env_int("GAUSPCC_SCAN_LOOKBACK", 1) != 0;   // (0: the three-launch scan of rounds 1-3, kept as the cross-check)
    const int64_t tiles = cdiv(n, LB_TILE);
    if (lookback && ctx && tiles <= LB_MAX_TILES && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        gpcc_ctx::ScanState *ss = nullptr;
        GP_TRY(scan_state(ctx, st, &ss));
        if (++ss->epoch >= (1u << 30)) {   // the epoch field wraps: start over with clean status words
            HIP_TRY(hipMemsetAsync(ss->status, 0, 8 * (size_t)LB_MAX_TILES, st));
            ss->epoch = 1u;
        }
        // developer fault injection (tests/test_gpu_robustness.py): in the N-th look-back launch of the process tile 0 never publishes -- tile 1 waits
        // for it until the (shortened) limit and the error word goes up
        static const int fault_at = dev_env_int("GAUSPCC_SCAN_FAULT", 0);
        static std::atomic<int> launches{0};
        uint32_t limit = 1u << 24, skip = 0xFFFFFFFFu;
        if (fault_at > 0 && ++launches == fault_at) { limit = 1u << 12; skip = 0u; }
        k_scan_lookback<<<(unsigned)tiles, LB_T, 0, st>>>(in, out, n, ss->status, ss->ticket, ss->epoch, total_dev, ctx->dev_err_dev, limit, skip,
                                                                 tiles >= LB_SHARD_MIN_TILES ? (uint32_t)LB_SHARDS : 1u);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    const int64_t nb = cdiv(n, SCAN_TILE);
    size_t mk = ctx->arena.mark();
    TAKE(bsum, uint32_t, nb);
    k_scan_reduce<<<dim3((unsigned)nb), SCAN_T, 0, st>>>(in, bsum, n);
    k_scan_bsums<<<1, SCAN_T, 0, st>>>(bsum, nb, total_dev);
    k_scan_apply<<<dim3((unsigned)nb), SCAN_T, 0, st>>>(in, out, bsum, n);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);  // stream-ordered reuse: later kernels on `st` run after these
    return GPCC_OK;
}

// ------------------------------------------------------------------ radix sort
// One 8-bit pass = digit histogram per tile -> one scan of [digit][tile] -> scatter.  Round 6: a tile is 4 096 keys on a 256-thread workgroup
// (until then 1 024 keys on one wave, every key stored straight to its global slot: 64 unrelated 8-byte stores per instruction, 31 Gkeys/s
// per pass -- the two sorts were 1.1 ms of the 4.5 ms RD frame).  The tile is first ordered by digit IN LDS -- each wave ranks its 1 024
// keys with the stable ballot ranking below, the four waves' per-digit counts are chained -- and then written out slot by slot, so that the
// keys of a digit leave as one contiguous run (16 keys = 128 bytes on average) instead of one by one.
constexpr int RS_SMALL = 1024;                 // up to this many keys: the whole sort in one launch of one wave (k_radix_small)
constexpr int RS_TILE = 4096;                  // keys per tile
constexpr int RS_T = 256;                      // threads per tile
constexpr int RS_ROUNDS = RS_TILE / RS_T;      // keys per thread; a wave owns RS_TILE / 4 consecutive keys

__global__ __launch_bounds__(RS_T) void k_radix_hist(const uint64_t *__restrict__ keys, int64_t n, int shift, uint32_t *__restrict__ hist, int64_t ntiles)
{
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    uint64_t k[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) { const int64_t i = base + r * RS_T + threadIdx.x; k[r] = i < n ? keys[i] : 0; }
    // (a wave whose 64 keys share the digit -- the upper bytes of depths, tile ids, Morton codes of one region -- adds its count once: 64 atomics on one
    //  LDS word serialise, which made the last pass of a 32-bit sort twice as slow as the first)
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool ok = base + r * RS_T + threadIdx.x < n;
        const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);     // lane 0's key: in range whenever any key of the wave is
        const uint64_t act = __ballot(ok), same = __ballot(ok && d == d0);
        if (same == act) {
            if ((threadIdx.x & 63) == 0 && act) atomicAdd(&cnt[d0], (uint32_t)__popcll(act));
        } else if (ok) atomicAdd(&cnt[d], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

template <bool HAS_VAL>
__global__ __launch_bounds__(RS_T) void k_radix_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                        uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
                                                        int shift, const uint32_t *__restrict__ offs, int64_t ntiles)
{
    __shared__ uint64_t skey[RS_TILE];
    __shared__ uint32_t sval[HAS_VAL ? RS_TILE : 1];
    __shared__ uint32_t cnt[4][256];     // per wave: digit counts, then the first tile slot of the wave's keys of that digit
    __shared__ uint32_t dstart[256];     // per digit: its first global slot for this tile minus its first tile slot
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][tid] = 0;
    // digit tid's first global slot for this tile: requested now, used by the copy-out at the end through LDS (fetched inside the copy-out loop it was a
    // dependent round trip in front of every one of a thread's 16 stores)
    const uint32_t my_off = offs[(int64_t)tid * ntiles + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (RS_TILE / 4);   // the wave's first key
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // the wave's 1 024 keys, 16 per lane, all requested before anything is ranked (key r of lane l: index base + 64 r + l)
    uint64_t key[RS_ROUNDS];
    uint32_t val[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = base + r * 64 + lane;
        key[r] = i < n ? keys[i] : 0;
        if (HAS_VAL) val[r] = i < n ? vals[i] : 0u;
    }
    __syncthreads();
    // stable rank inside the wave: 8 ballots give the lanes that hold my digit; the wave's running count per digit lives in its own LDS
    // row (the LDS operations of a wave execute in order: every lane has read cnt[d] before the first peer writes it)
    uint32_t slot[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool ok = base + r * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        const uint32_t prior = cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (ok && rank == 0) cnt[wave][d] = prior + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        slot[r] = prior + rank;
    }
    __syncthreads();
    // digit d = tid: chain the four waves' counts, then an exclusive scan over the 256 digits gives every (wave, digit) its first tile slot
    {
        const uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        const uint32_t tot = c0 + c1 + c2 + c3;
        const uint32_t inc = wave_incl_scan(tot, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t wb = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) wb += w < wave ? wsum[w] : 0u;
        const uint32_t ex = wb + inc - tot;
        dstart[tid] = my_off - ex;       // global slot of tile slot t of digit tid: dstart[tid] + t (mod 2^32)
        cnt[0][tid] = ex; cnt[1][tid] = ex + c0; cnt[2][tid] = ex + c0 + c1; cnt[3][tid] = ex + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (base + r * 64 + lane < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t at = cnt[wave][d] + slot[r];
            skey[at] = key[r];
            if (HAS_VAL) sval[at] = val[r];
        }
    }
    __syncthreads();
    // out: tile slot t -> global slot offs[digit][tile] + (t - first tile slot of the digit): a digit's keys leave as one run
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    const int count = (int)min((int64_t)RS_TILE, n - tile0);
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int t = r * RS_T + tid;
        if (t < count) {
            const uint64_t k = skey[t];
            const uint32_t d = (uint32_t)(k >> shift) & 255u;
            const int64_t pos = (int64_t)(uint32_t)(dstart[d] + (uint32_t)t);
            keys_out[pos] = k;
            if (HAS_VAL) vals_out[pos] = sval[t];
        }
    }
}

// n <= RS_SMALL: the whole sort (every 8-bit pass) in ONE launch of one wave, keys and payloads ping-ponging in LDS.
// Same stable ballot ranking as k_radix_scatter.  The small octree levels sort a few hundred keys at a time, where a
// pass of the tiled sort costs five launches.
template <bool HAS_VAL>
__global__ __launch_bounds__(64) void k_radix_small(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, int n, int bits)
{
    __shared__ uint64_t k[2][RS_SMALL];
    __shared__ uint32_t v[2][HAS_VAL ? RS_SMALL : 1];
    __shared__ uint32_t cnt[256];
    const int lane = threadIdx.x;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int i = lane; i < n; i += 64) {
        k[0][i] = keys[i];
        if (HAS_VAL) v[0][i] = vals[i];
    }
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cnt[lane + 64 * i] = 0;
        __syncthreads();
        for (int i = lane; i < n; i += 64) atomicAdd(&cnt[(uint32_t)(k[cur][i] >> shift) & 255u], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 digit counts: 4 per lane + wave scan
            const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
            const uint32_t s = c0 + c1 + c2 + c3;
            const uint32_t ex = wave_incl_scan(s, lane) - s;
            __syncthreads();
            cnt[4 * lane] = ex; cnt[4 * lane + 1] = ex + c0; cnt[4 * lane + 2] = ex + c0 + c1; cnt[4 * lane + 3] = ex + c0 + c1 + c2;
        }
        __syncthreads();
        for (int r = 0; r < (n + 63) / 64; ++r) {
            const int i = r * 64 + lane;
            const bool ok = i < n;
            const uint64_t key = ok ? k[cur][i] : 0;
            const uint32_t d = (uint32_t)(key >> shift) & 255u;
            uint64_t peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
            const uint32_t prior = cnt[d];
            __syncthreads();
            if (ok && rank == 0) cnt[d] = prior + (uint32_t)__popcll(peers);
            __syncthreads();
            if (ok) {
                k[cur ^ 1][prior + rank] = key;
                if (HAS_VAL) v[cur ^ 1][prior + rank] = v[cur][i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = lane; i < n; i += 64) {
        keys[i] = k[cur][i];
        if (HAS_VAL) vals[i] = v[cur][i];
    }
}

int radix_sort_u64(gpcc_ctx *ctx, hipStream_t st, uint64_t **keys_io, uint64_t **keys_tmp_io, uint32_t **vals_io,
                   uint32_t **vals_tmp_io, int64_t n, int bits)
{
    if (n <= 1 || bits <= 0) return GPCC_OK;
    if (bits > 64) bits = 64;
    if (n <= RS_SMALL) {   // in place: the *_io pointers keep pointing at the result
        if (vals_io && *vals_io) k_radix_small<true><<<1, 64, 0, st>>>(*keys_io, *vals_io, (int)n, bits);
        else k_radix_small<false><<<1, 64, 0, st>>>(*keys_io, nullptr, (int)n, bits);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    const int64_t ntiles = cdiv(n, RS_TILE);
    size_t mk = ctx->arena.mark();
    TAKE(hist, uint32_t, 256 * ntiles);
    const bool has_val = vals_io && *vals_io;
    for (int shift = 0; shift < bits; shift += 8) {
        k_radix_hist<<<dim3((unsigned)ntiles), RS_T, 0, st>>>(*keys_io, n, shift, hist, ntiles);
        LAUNCH_CHECK();
        GP_TRY(exclusive_scan_u32(ctx, st, hist, hist, 256 * ntiles, nullptr));
        if (has_val)
            k_radix_scatter<true><<<dim3((unsigned)ntiles), RS_T, 0, st>>>(*keys_io, *vals_io, *keys_tmp_io, *vals_tmp_io, n, shift, hist, ntiles);
        else
            k_radix_scatter<false><<<dim3((unsigned)ntiles), RS_T, 0, st>>>(*keys_io, nullptr, *keys_tmp_io, nullptr, n, shift, hist, ntiles);
        LAUNCH_CHECK();
        uint64_t *tk = *keys_io; *keys_io = *keys_tmp_io; *keys_tmp_io = tk;
        if (has_val) { uint32_t *tv = *vals_io; *vals_io = *vals_tmp_io; *vals_tmp_io = tv; }
    }
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

}  // namespace gpcc
