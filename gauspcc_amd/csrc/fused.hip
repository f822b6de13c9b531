// fused.hip -- pair plans and the persistent small-level kernels of the decoder (fused.hpp has the design).
#include "fused.hpp"

#include <atomic>

#include <mutex>

#include "network_dev.hpp"
#include "primitives.hpp"
#include "rangecoder_dev.hpp"

namespace gpcc {

// ------------------------------------------------------------------ policy
int fused_mode()
{
    static const int m = [] { const int v = dev_env_int("GAUSPCC_FUSED", 1); return v < 0 || v > 2 ? 1 : v; }();
    return m;
}
bool fused_enabled() { return fused_mode() != 0; }
int fused_device_cap();   // below: workgroups of a persistent launch the current device holds at once
bool fused_level_ok(int64_t n, int k)
{
    if (fused_device_cap() < 16) return false;   // a partitioned / masked device too small for a persistent grid: the launch-per-layer path
    static const int64_t nmax = std::min<int64_t>(dev_env_ll("GAUSPCC_FUSED_MAX", FUSE_MAX_NODES), FUSE_HARD_MAX);
    const int64_t K = (int64_t)k * k * k;
    return fused_enabled() && n >= 1 && n <= nmax && (n * K + 1) * 128 <= ((int64_t)768 << 20);
}

// ------------------------------------------------------------------ pair plan
namespace {

// exclusive scan of one value per thread over a 1024-thread workgroup; returns the exclusive prefix, *total = the sum
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *total, uint32_t *sh /* [17] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += u;
    }
    __syncthreads();                     // (sh may still be read from the previous call)
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? sh[lane] : 0u, wi = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)wi, d, 64);
            if (lane >= d) wi += u;
        }
        if (lane < 16) sh[lane] = wi - w;
        if (lane == 15) sh[16] = wi;
    }
    __syncthreads();
    *total = sh[16];
    return sh[wave] + inc - v;
}

// workgroup o < K: rows that have a neighbour at offset o; workgroup K: rowstart = exclusive scan of the per-row counts (in place)
__global__ __launch_bounds__(1024) void k_pp_count(const int32_t *__restrict__ nbr, int n, int K, uint32_t *__restrict__ cnt_o, uint32_t *__restrict__ rowstart,
                                                   unsigned long long *__restrict__ pairs_out)
{
    __shared__ uint32_t sh[17];
    const int o = blockIdx.x;
    if (o < K) {
        uint32_t c = 0;
        for (int r = threadIdx.x; r < n; r += 1024) c += nbr[(size_t)o * n + r] >= 0 ? 1u : 0u;
        uint32_t tot;
        (void)block_scan_1024(c, &tot, sh);
        if (threadIdx.x == 0) cnt_o[o] = tot;
        return;
    }
    uint32_t carry = 0;
    for (int r0 = 0; r0 < n; r0 += 1024) {
        const int r = r0 + (int)threadIdx.x;
        const uint32_t v = r < n ? rowstart[r] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_1024(v, &tot, sh);
        if (r < n) rowstart[r] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { rowstart[n] = carry; if (pairs_out) *pairs_out += carry; }
}

// workgroup o: the rows with a neighbour at offset o, in row order, packed 16 to a tile behind the tiles of the offsets below
__global__ __launch_bounds__(1024) void k_pp_fill(const int32_t *__restrict__ nbr, const uint16_t *__restrict__ rk, const uint32_t *__restrict__ cnt_o, const uint32_t *__restrict__ rowstart,
                                                  int n, int K, uint32_t tcap, uint32_t pdummy, int32_t *__restrict__ ot_j, uint32_t *__restrict__ ot_q, uint32_t *__restrict__ ot_o,
                                                  uint32_t *__restrict__ ntiles)
{
    __shared__ uint32_t sh[17];
    const int o = blockIdx.x;
    uint32_t part = 0;
    for (int q = threadIdx.x; q < o; q += 1024) part += (cnt_o[q] + 15u) >> 4;
    uint32_t base;
    (void)block_scan_1024(part, &base, sh);
    const uint32_t cnt = cnt_o[o];
    const uint32_t mine = (cnt + 15u) >> 4;
    if (o == K - 1 && threadIdx.x == 0) *ntiles = min(base + mine, tcap);
    if (base + mine > tcap) return;          // (cannot happen: tcap is the bound K ceil(n / 16))
    uint32_t carry = 0;
    for (int r0 = 0; r0 < n; r0 += 1024) {
        const int r = r0 + (int)threadIdx.x;
        const int32_t j = r < n ? nbr[(size_t)o * n + r] : -1;
        uint32_t tot;
        const uint32_t pos = carry + block_scan_1024(j >= 0 ? 1u : 0u, &tot, sh);
        if (j >= 0) {
            const size_t slot = (size_t)base * 16 + pos;
            ot_j[slot] = j;
            ot_q[slot] = min(rowstart[r] + (uint32_t)rk[(size_t)o * n + r], pdummy);
        }
        carry += tot;
    }
    for (uint32_t p = cnt + threadIdx.x; p < mine * 16u; p += 1024) { ot_j[(size_t)base * 16 + p] = 0; ot_q[(size_t)base * 16 + p] = pdummy; }
    for (uint32_t i = threadIdx.x; i < mine; i += 1024) ot_o[base + i] = (uint32_t)o | (min(16u, cnt - 16u * i) << 16);
}

}  // namespace

int pairplan_build(gpcc_ctx *ctx, hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int32_t *cell_own, int k, PairPlan *plan,
                   unsigned long long *pairs_dev)
{
    const int64_t n = chi->n;
    const int K = k * k * k;
    if (!fused_level_ok(n, k)) return fail(GPCC_ERR_ARG, "internal: pair plan of a level of %lld nodes", (long long)n);
    plan->n = n; plan->K = K;
    plan->tcap = (int64_t)K * cdiv(n, 16);
    plan->pcap = n * K + 1;
    TAKE(rowstart, uint32_t, n + 1);
    TAKE(ot_j, int32_t, plan->tcap * 16);
    TAKE(ot_q, uint32_t, plan->tcap * 16);
    TAKE(ot_o, uint32_t, plan->tcap);
    TAKE(ntiles, uint32_t, 1);
    // build-time scratch (dense map, ranks, per-offset counts): from the top, released when the build is ENQUEUED -- the
    // caller's next top allocations are used by work ordered behind this stream's event
    const size_t mk = ctx->arena.top_mark();
    TAKE_TOP(nbr, int32_t, (int64_t)K * n);
    TAKE_TOP(rk, uint16_t, (int64_t)K * n);
    TAKE_TOP(cnt_o, uint32_t, K);
    GP_TRY(tiles_dense_map(st, par, cell_par, chi, cell_own, k, nbr, rk, rowstart));
    k_pp_count<<<(unsigned)K + 1u, 1024, 0, st>>>(nbr, (int)n, K, cnt_o, rowstart, pairs_dev);
    LAUNCH_CHECK();
    k_pp_fill<<<(unsigned)K, 1024, 0, st>>>(nbr, rk, cnt_o, rowstart, (int)n, K, (uint32_t)plan->tcap, (uint32_t)(plan->pcap - 1), ot_j, ot_q, ot_o, ntiles);
    LAUNCH_CHECK();
    ctx->arena.top_rewind(mk);
    plan->rowstart = rowstart; plan->ot_j = ot_j; plan->ot_q = ot_q; plan->ot_o = ot_o; plan->ntiles = ntiles;
    return GPCC_OK;
}

// ------------------------------------------------------------------ the two phases of a convolution
namespace {

struct PlanV {
    const uint32_t *rowstart; const int32_t *ot_j; const uint32_t *ot_q, *ot_o, *ntiles;
    int n, K; uint32_t tcap, pcap;
};
inline PlanV plan_view(const PairPlan &p) { return PlanV{p.rowstart, p.ot_j, p.ot_q, p.ot_o, p.ntiles, (int)p.n, p.K, (uint32_t)p.tcap, (uint32_t)p.pcap}; }

// PRODUCTS.  Per tile the 16 x 32 products of its pairs -- 16 MFMAs from a zero accumulator, the transposed product
// D^T = W^T X^T of the asm conv loop, so lane (g, e) holds four physically consecutive channels of tile row e per accumulator
// -- go to the pairs' rows of P.  Two tiles at a time (their loads in flight together).
// in: the layer's input rows (written by other workgroups a grid barrier ago: no __restrict__ / const promises to the compiler).
struct TileHdr { uint32_t oc, j, q; };   // offset | count << 16; neighbour row and P row of this lane's entry (clamped)
__device__ __forceinline__ TileHdr tile_hdr(const PlanV &pl, uint32_t t, int e)
{
    TileHdr h;
    h.oc = pl.ot_o[t];
    h.j = min((uint32_t)pl.ot_j[(size_t)t * 16 + e], (uint32_t)pl.n - 1u);
    h.q = min(pl.ot_q[(size_t)t * 16 + e], pl.pcap - 1u);
    return h;
}
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ void tile_pair(const PlanV &pl, const TileHdr &ha, const TileHdr &hb, bool two, const float *in, const float *__restrict__ wt, float *P, int lane)
{
    const int e = lane & 15, g = lane >> 4;
    const uint32_t km1 = (uint32_t)pl.K - 1u;
    const float *inl = in + 4 * g;
    const float *wl = wt + lane * 4;
    const float *pa = inl + (size_t)ha.j * 32, *pb = inl + (size_t)hb.j * 32;
    const float *wa = wl + (size_t)min(ha.oc & 0xFFFFu, km1) * 1024, *wb = wl + (size_t)min(hb.oc & 0xFFFFu, km1) * 1024;
    const float4 xa0 = ld4(pa), xa1 = ld4(pa + 16), xb0 = ld4(pb), xb1 = ld4(pb + 16);
    const float4 wa0 = ld4(wa), wa1 = ld4(wa + 256), wa2 = ld4(wa + 512), wa3 = ld4(wa + 768);
    const float4 wb0 = ld4(wb), wb1 = ld4(wb + 256), wb2 = ld4(wb + 512), wb3 = ld4(wb + 768);
    {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        MF(c0, wa0.x, xa0.x); MF(c1, wa2.x, xa0.x);
        MF(c0, wa0.y, xa0.y); MF(c1, wa2.y, xa0.y);
        MF(c0, wa0.z, xa0.z); MF(c1, wa2.z, xa0.z);
        MF(c0, wa0.w, xa0.w); MF(c1, wa2.w, xa0.w);
        MF(c0, wa1.x, xa1.x); MF(c1, wa3.x, xa1.x);
        MF(c0, wa1.y, xa1.y); MF(c1, wa3.y, xa1.y);
        MF(c0, wa1.z, xa1.z); MF(c1, wa3.z, xa1.z);
        MF(c0, wa1.w, xa1.w); MF(c1, wa3.w, xa1.w);
        if ((uint32_t)e < (ha.oc >> 16)) {
            float *dst = P + (size_t)ha.q * 32 + 4 * g;
            *reinterpret_cast<float4 *>(dst) = make_float4(c0[0], c0[1], c0[2], c0[3]);
            *reinterpret_cast<float4 *>(dst + 16) = make_float4(c1[0], c1[1], c1[2], c1[3]);
        }
    }
    if (two) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        MF(c0, wb0.x, xb0.x); MF(c1, wb2.x, xb0.x);
        MF(c0, wb0.y, xb0.y); MF(c1, wb2.y, xb0.y);
        MF(c0, wb0.z, xb0.z); MF(c1, wb2.z, xb0.z);
        MF(c0, wb0.w, xb0.w); MF(c1, wb2.w, xb0.w);
        MF(c0, wb1.x, xb1.x); MF(c1, wb3.x, xb1.x);
        MF(c0, wb1.y, xb1.y); MF(c1, wb3.y, xb1.y);
        MF(c0, wb1.z, xb1.z); MF(c1, wb3.z, xb1.z);
        MF(c0, wb1.w, xb1.w); MF(c1, wb3.w, xb1.w);
        if ((uint32_t)e < (hb.oc >> 16)) {
            float *dst = P + (size_t)hb.q * 32 + 4 * g;
            *reinterpret_cast<float4 *>(dst) = make_float4(c0[0], c0[1], c0[2], c0[3]);
            *reinterpret_cast<float4 *>(dst + 16) = make_float4(c1[0], c1[1], c1[2], c1[3]);
        }
    }
}
#undef MF

// wave gw of NW takes tiles gw, gw + NW, ... starting with tile number `first` of its sequence (the ones before are cached)
__device__ __forceinline__ void phase_products(const PlanV &pl, const float *in, const float *__restrict__ wt, float *P, uint32_t gw, uint32_t NW, int lane, uint32_t first = 0u)
{
    const uint32_t nt = min((uint32_t)__builtin_amdgcn_readfirstlane((int)*pl.ntiles), pl.tcap);
    const int e = lane & 15;
    for (uint32_t t = gw + first * NW; t < nt; t += 2u * NW) {
        const bool two = t + NW < nt;                    // wave-uniform
        const TileHdr ha = tile_hdr(pl, t, e), hb = tile_hdr(pl, two ? t + NW : t, e);
        tile_pair(pl, ha, hb, two, in, wt, P, lane);
    }
}

// The persistent kernels run up to 18 convolutions on ONE plan: the headers of a wave's first two tiles and the P ranges of a
// thread's first two (row, quad) items are loaded once per launch and stay in registers -- a phase then starts with the
// loads that depend on the previous phase (gathered rows / product rows), not with a round trip for its own work list.
struct PlanCache {
    TileHdr ha, hb; uint32_t ntw;        // cached tiles of this wave (0, 1 or 2); more tiles than 2 NW: the rest streams
    uint32_t q0[2], q1[2];               // P ranges of items tid, tid + 1024 of this workgroup's rows
};
__device__ __forceinline__ void sum_range(const PlanV &pl, int r, uint32_t *q0, uint32_t *q1)
{
    const uint32_t pm1 = pl.pcap - 1u;
    *q0 = min(pl.rowstart[r], pm1);
    *q1 = min(min(pl.rowstart[r + 1], pm1), *q0 + (uint32_t)pl.K);   // (a plan built from a corrupt level: stay inside P)
}
__device__ __forceinline__ PlanCache plan_cache(const PlanV &pl, uint32_t gw, uint32_t NW, int lane, int row0, int row1, int tid, int nthreads)
{
    PlanCache c;
    const uint32_t nt = min((uint32_t)__builtin_amdgcn_readfirstlane((int)*pl.ntiles), pl.tcap);
    c.ntw = gw < nt ? (gw + NW < nt ? 2u : 1u) : 0u;
    const int e = lane & 15;
    c.ha = tile_hdr(pl, c.ntw ? gw : 0u, e);
    c.hb = tile_hdr(pl, c.ntw == 2u ? gw + NW : (c.ntw ? gw : 0u), e);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = tid + i * nthreads;
        c.q0[i] = c.q1[i] = 0u;
        if (it < (row1 - row0) * 8) sum_range(pl, row0 + (it >> 3), &c.q0[i], &c.q1[i]);
    }
    return c;
}
__device__ __forceinline__ void phase_products_cached(const PlanV &pl, const PlanCache &c, const float *in, const float *__restrict__ wt, float *P, uint32_t gw, uint32_t NW, int lane)
{
    if (c.ntw) tile_pair(pl, c.ha, c.hb, c.ntw == 2u, in, wt, P, lane);
    if (c.ntw == 2u) phase_products(pl, in, wt, P, gw, NW, lane, 2u);
}

// SUMS: rows [row0, row1) of the level, a thread per (row, 16-byte channel quad): the row's P rows added in ascending offset
// order (= the order they lie in P), then + residual, ReLU.  Sixteen loads in flight (a dense small level has 40-77 pairs per
// row: the adds are a sequential chain by specification, the loads need not be).
__device__ __forceinline__ void sum_item(const PlanV &pl, const float *P, const float *res, float *out, int relu, int r, int c4, uint32_t q0, uint32_t q1)
{
    const uint32_t pm1 = pl.pcap - 1u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *Pc = P + c4 * 4;
    const size_t at = (size_t)r * 32 + c4 * 4;
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = ld4(res + at);
    for (uint32_t q = q0; q < q1; q += 16u) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld4(Pc + (size_t)min(q + (uint32_t)u, pm1) * 32);   // unconditional: sixteen loads in flight
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (q + (uint32_t)u < q1) { acc.x = acc.x + v[u].x; acc.y = acc.y + v[u].y; acc.z = acc.z + v[u].z; acc.w = acc.w + v[u].w; }
    }
    if (res) { acc.x = acc.x + rv.x; acc.y = acc.y + rv.y; acc.z = acc.z + rv.z; acc.w = acc.w + rv.w; }
    if (relu) { acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f; acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f; }
    *reinterpret_cast<float4 *>(out + at) = acc;
}
// items first, first + 1, ... of the sequence tid, tid + nthreads, ... (the ones before `first` are the caller's cached ones)
__device__ __forceinline__ void phase_sums(const PlanV &pl, const float *P, const float *res, float *out, int relu, int row0, int row1, int tid, int nthreads, int first = 0)
{
    for (int it = tid + first * nthreads; it < (row1 - row0) * 8; it += nthreads) {
        uint32_t q0, q1;
        sum_range(pl, row0 + (it >> 3), &q0, &q1);
        sum_item(pl, P, res, out, relu, row0 + (it >> 3), it & 7, q0, q1);
    }
}
__device__ __forceinline__ void phase_sums_cached(const PlanV &pl, const PlanCache &c, const float *P, const float *res, float *out, int relu, int row0, int row1, int tid, int nthreads)
{
    const int items = (row1 - row0) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int it = tid + i * nthreads;
        if (it < items) sum_item(pl, P, res, out, relu, row0 + (it >> 3), it & 7, c.q0[i], c.q1[i]);
    }
    if (items > 2 * nthreads) phase_sums(pl, P, res, out, relu, row0, row1, tid, nthreads, 2);
}

__global__ __launch_bounds__(256) void k_plan_products(PlanV pl, const float *in, const float *wt, float *P)
{
    phase_products(pl, in, wt, P, blockIdx.x * 4u + (threadIdx.x >> 6), gridDim.x * 4u, (int)(threadIdx.x & 63));
}
__global__ __launch_bounds__(1024) void k_plan_sum(PlanV pl, const float *P, const float *res, float *out, int relu)
{
    const int row0 = (int)blockIdx.x * 128;
    phase_sums(pl, P, res, out, relu, row0, min(pl.n, row0 + 128), (int)threadIdx.x, 1024);
}

}  // namespace

int plan_conv(hipStream_t st, const PairPlan &plan, const ConvJob &job, float *P, int relu)
{
    if (plan.n <= 0) return GPCC_OK;
    const PlanV pl = plan_view(plan);
    // (the number of tiles lives on the device: a grid for the bound of a SPARSE level -- most small levels -- would be mostly idle
    // waves; 4 waves per workgroup, at most 512 workgroups, the waves stride over the tiles)
    const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512, cdiv(plan.tcap, 8)));
    k_plan_products<<<g, 256, 0, st>>>(pl, job.in, job.w + (size_t)plan.K * 1024, P);
    LAUNCH_CHECK();
    k_plan_sum<<<(unsigned)cdiv(plan.n, 128), 1024, 0, st>>>(pl, P, job.res, job.out, relu);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// ------------------------------------------------------------------ grid barrier (tools/ubench/grid_sync.hip: the `xcd` form)
namespace {

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ uint32_t ld_rlx(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t add_rlx(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every spin is bounded: ~2^21 polls of ~0.3 us.  A launch whose workgroups are not all resident (another context's persistent
// launch holding the CUs, a device reset under way) sets the context's sticky timeout word and every workgroup leaves at its
// next poll; the host then decodes again on the launch-per-layer path.
constexpr uint32_t FUSE_SPIN_MAX = 1u << 21;
__device__ __forceinline__ bool poll_ge(uint32_t *p, uint32_t want, uint32_t *tmo)
{
    for (uint32_t s = 0; s < FUSE_SPIN_MAX; ++s) {
        if (ld_rlx(p) >= want) return true;
        if ((s & 63u) == 63u && ld_rlx(tmo)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    st_rlx(tmo, 1u);
    return false;
}

struct BarCtx {
    FusedBar *b; uint32_t *tmo; uint32_t xcc, epoch;
#ifdef FUSED_TIMING
    unsigned long long t_start, t_bar, t_poll;   // developer build: 100 MHz stamps of workgroup 0 (launch start, time inside barriers, of it polling)
#endif
};

// start of a launch: who runs where (the workgroup -> XCC placement is observed, not promised: counted, not assumed), and the
// other barrier block zeroed for the next launch of this context (stream-ordered behind this one)
__device__ __forceinline__ bool bar_begin(BarCtx &bc, FusedBar *b, FusedBar *b_next, uint32_t *tmo)
{
    __shared__ uint32_t ok_s;
    bc.b = b; bc.tmo = tmo; bc.xcc = xcc_id(); bc.epoch = 0;
#ifdef FUSED_TIMING
    bc.t_start = __builtin_amdgcn_s_memrealtime(); bc.t_bar = 0; bc.t_poll = 0;
#endif
    if (blockIdx.x == 0) {
        uint32_t *z = reinterpret_cast<uint32_t *>(b_next);
        for (uint32_t i = threadIdx.x; i < sizeof(FusedBar) / 4; i += blockDim.x) z[i] = 0u;
    }
    if (threadIdx.x == 0) {
        bool ok = ld_rlx(tmo) == 0u;
        // the members add must be visible to whoever sees the census complete (the two words live on different cache lines, possibly
        // different L2 channels, and nothing orders relaxed adds of one workgroup): the census add is a RELEASE, the reader acquires
        add_rlx(&b->members[bc.xcc * 32], 1u);
        __hip_atomic_fetch_add(&b->census[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && poll_ge(&b->census[0], gridDim.x, tmo);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (ok && blockIdx.x == 0) {
            uint32_t nx = 0;
            for (uint32_t x = 0; x < 8u; ++x) nx += ld_rlx(&b->members[x * 32]) ? 1u : 0u;
            st_rlx(&b->nxcc[0], nx);
        }
        __hip_atomic_fetch_add(&b->census[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (workgroup 0: nxcc before its second census add)
        ok = ok && poll_ge(&b->census[1], gridDim.x, tmo);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_s = ok ? 1u : 0u;
    }
    __syncthreads();
    return ok_s != 0u;
}

// every thread of every workgroup of the launch; false = the launch timed out (leave)
__device__ __forceinline__ bool grid_barrier(BarCtx &bc)
{
    __shared__ uint32_t ok_s;
#ifdef FUSED_TIMING
    const unsigned long long tb0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tp0 = 0, tp1 = 0;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its own stores have reached the L2
    __syncthreads();
    const uint32_t epoch = ++bc.epoch;
    if (threadIdx.x == 0) {
        FusedBar *b = bc.b;
        const uint32_t m = ld_rlx(&b->members[bc.xcc * 32]);
        const uint32_t old = add_rlx(&b->xcc_count[bc.xcc * 32], 1u);
        if (old + 1u == epoch * m) {                       // last of this XCC: its L2 holds everything the XCC's workgroups wrote
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t nx = ld_rlx(&b->nxcc[0]);
            const uint32_t t = add_rlx(&b->top[0], 1u);
            if (t + 1u == epoch * nx)
                for (uint32_t x = 0; x < 8u; ++x) st_rlx(&b->xcc_gen[x * 32], epoch);
        }
#ifdef FUSED_TIMING
        tp0 = __builtin_amdgcn_s_memrealtime();
#endif
        const bool ok = poll_ge(&b->xcc_gen[bc.xcc * 32], epoch, bc.tmo);
#ifdef FUSED_TIMING
        tp1 = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_s = ok ? 1u : 0u;
    }
    __syncthreads();
#ifdef FUSED_TIMING
    bc.t_bar += __builtin_amdgcn_s_memrealtime() - tb0; bc.t_poll += tp1 - tp0;
#endif
    return ok_s != 0u;
}

// ------------------------------------------------------------------ the persistent kernels
struct FusedK {
    PlanV pl;
    FusedBar *bar, *bar_next;
    uint32_t *tmo;
    float *P;
    // CHILD: parent features, structure, buffers x / a / b / u, the container; PARENT: occupancy in, buffers x (= F) / a / b
    const float *pA; const uint32_t *parent; const uint64_t *rkey; const uint32_t *m2r;
    const uint32_t *m2s, *cpos;      // symbol slot of every node (one scene: its raster rank = m2r) and, for a batch's merged level, its CDF row slot
    float *x, *a, *b, *u;
    const float *w[13];              // CHILD: conv[5 .. 17]; PARENT: conv[0 .. 4] -- transposed fragments (conv + K * 1024)
    const float *temb, *semb[3], *hfrag[4], *prior_emb;
    uint16_t *cdf; uint8_t *sym[4];
    uint8_t *occ;                    // CHILD: out; PARENT: in
    const uint8_t *bytes; const RcChunk *chunks; uint32_t nlanes; int llog; uint32_t rdw[4];
    uint32_t desert;                 // test hook (GAUSPCC_FUSED_TEST_DESERT): the launch's last workgroup leaves before the census -- the others must time out
};

// GAUSPCC_FUSED_TEST_DESERT=N: the N-th persistent launch of the process loses a workgroup (tests/test_gpu_robustness.py: the bounded spins, the
// sticky timeout word and the decoder's retry on the block-tile kernels are a recovery path that must be seen working)
uint32_t fused_test_desert()
{
    static const int nth = env_int("GAUSPCC_FUSED_TEST_DESERT", 0);
    static std::atomic<int> count{0};
    return nth > 0 && ++count == nth ? 1u : 0u;
}

struct WgMap { uint32_t G, wg, gw, NW, gtid, NT; int row0, row1; int lane, wave; };

__device__ __forceinline__ bool conv_phases(const FusedK &k, BarCtx &bc, const WgMap &m, const PlanCache &pc, const float *in, const float *wt, const float *res, float *out, int relu, bool barrier_after)
{
    phase_products_cached(k.pl, pc, in, wt, k.P, m.gw, m.NW, m.lane);
    if (!grid_barrier(bc)) return false;
    phase_sums_cached(k.pl, pc, k.P, res, out, relu, m.row0, m.row1, (int)threadIdx.x, FUSE_THREADS);
    if (barrier_after && !grid_barrier(bc)) return false;
    return true;
}

template <int M>
__device__ __forceinline__ void head_rows(const FusedK &k, const WgMap &m, int stage, float *lds)
{
    // the rows this workgroup has just summed: visible to its own waves after a workgroup barrier (one CU, one L1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HeadArgs ha = {};
    ha.x = k.b; ha.n = k.pl.n; ha.stage_m = M; ha.frag = k.hfrag[stage]; ha.m2r = k.m2r; ha.cdf = k.cdf; ha.mode = 1; ha.chunk_log2 = k.llog; ha.nch = k.nlanes; ha.stage = stage; ha.pos = k.cpos;
    if (m.wave < FUSE_HEAD_WAVES)
        for (int r = m.row0 + 64 * m.wave; r < m.row1; r += 64 * FUSE_HEAD_WAVES)
            head_wave<M, 1>(ha, r, m.row1, m.lane, lds + m.wave * HEAD_LDS_FLOATS, 0u);
}

template <int LP, int CODER>
__device__ __forceinline__ void rc_rows(const FusedK &k, const WgMap &m, int stage, uint32_t *lds)
{
    // lanes [wg * LPW, (wg + 1) * LPW) of the stream: 3- / 5-entry rows on wave 0, 17-entry rows four coder lanes to a wave
    const uint32_t LPW = (k.nlanes + m.G - 1u) / m.G;
    const int c0 = (int)(m.wg * LPW), cend = (int)min(k.nlanes, (m.wg + 1u) * LPW);
    if (c0 >= cend) return;
    const RcChunk *ch = k.chunks + (size_t)stage * k.nlanes;
    const uint32_t rdw = k.rdw[stage];
    if (LP == 17) {
        const int cw = c0 + 4 * m.wave;
        if (cw < cend) rc_decode17_lds_wave<4, false, CODER>(k.cdf, k.bytes, ch, cend, cw, m.lane, rdw, k.sym[stage], lds + (size_t)m.wave * 4u * rdw);
    } else if (m.wave == 0) {
        rc_decode_lds_wave<LP == 17 ? 3 : LP, 4, false, CODER>(k.cdf, k.bytes, ch, cend, c0, m.lane, (int)LPW, rdw, k.sym[stage], lds);
    }
}

enum { FUSED_CHILD = 0, FUSED_PARENT = 1 };

// CODER: the lanes' coder of the container being decoded (CHILD only)
template <int MODE, int CODER>
__global__ __launch_bounds__(FUSE_THREADS) void k_level_fused(FusedK k)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    BarCtx bc;
    if (k.desert && gridDim.x > 1 && blockIdx.x == gridDim.x - 1) return;
    if (!bar_begin(bc, k.bar, k.bar_next, k.tmo)) return;
    WgMap m;
    m.G = gridDim.x; m.wg = blockIdx.x; m.lane = (int)(threadIdx.x & 63); m.wave = (int)(threadIdx.x >> 6);
    m.gw = m.wg * (FUSE_THREADS / 64) + (uint32_t)m.wave; m.NW = m.G * (FUSE_THREADS / 64);
    m.gtid = m.wg * FUSE_THREADS + threadIdx.x; m.NT = m.G * FUSE_THREADS;
    const int n = k.pl.n;
    const int rpw = (n + (int)m.G - 1) / (int)m.G;
    m.row0 = min(n, (int)m.wg * rpw); m.row1 = min(n, m.row0 + rpw);
    const int64_t n8 = (int64_t)n * 8;
    const PlanCache pc = plan_cache(k.pl, m.gw, m.NW, m.lane, m.row0, m.row1, (int)threadIdx.x, FUSE_THREADS);   // (the plan is a previous launch's output)
    if (MODE == FUSED_PARENT) {
        // F = Emb256[occ]                                                   (pcc_utils.py:99)
        const float4 *emb = reinterpret_cast<const float4 *>(k.prior_emb);
        for (int64_t t = m.gtid; t < n8; t += m.NT) reinterpret_cast<float4 *>(k.x)[t] = emb[(size_t)k.occ[t >> 3] * 8 + (t & 7)];
        if (!grid_barrier(bc)) return;
    } else {
        // X[i] = F[parent[i]] + Emb8[octant(i)]                             (kit/nn.py:77-98,108-117)
        const float4 *F = reinterpret_cast<const float4 *>(k.pA), *te = reinterpret_cast<const float4 *>(k.temb);
        for (int64_t t = m.gtid; t < n8; t += m.NT) {
            const int64_t i = t >> 3;
            const int g = (int)(t & 7);
            const uint64_t kk = k.rkey[i];
            const int q = (int)((rk_x(kk) & 1) | ((rk_y(kk) & 1) << 1) | ((rk_z(kk) & 1) << 2));
            const float4 f = F[(size_t)k.parent[i] * 8 + g], e = te[q * 8 + g];
            reinterpret_cast<float4 *>(k.x)[t] = make_float4(f.x + e.x, f.y + e.y, f.z + e.z, f.w + e.w);
        }
        if (!grid_barrier(bc)) return;
    }
    // Conv-ReLU-ResNet-ResNet (network_ue_4stage_conv.py:17-33; codec.hip: run_trunk) -> a
    if (!conv_phases(k, bc, m, pc, k.x, k.w[0], nullptr, k.a, 1, true)) return;
    if (!conv_phases(k, bc, m, pc, k.a, k.w[1], nullptr, k.b, 1, true)) return;
    if (!conv_phases(k, bc, m, pc, k.b, k.w[2], k.a, k.x, 1, true)) return;
    if (!conv_phases(k, bc, m, pc, k.x, k.w[3], nullptr, k.b, 1, true)) return;
    if (!conv_phases(k, bc, m, pc, k.b, k.w[4], k.x, k.a, 1, MODE == FUSED_CHILD)) return;
#ifdef FUSED_TIMING
    if (MODE == FUSED_PARENT && blockIdx.x == 0 && threadIdx.x == 0)
        printf("[fused] parent n %d G %u: %u barriers, total %.1f us, in barriers %.1f us (polling %.1f us)\n", n, gridDim.x, bc.epoch, (__builtin_amdgcn_s_memrealtime() - bc.t_start) / 100.0, bc.t_bar / 100.0, bc.t_poll / 100.0);
#endif
    if (MODE == FUSED_PARENT) return;
    // the four stages (pcc_utils.py:313-366): input, conv-ReLU-conv, head -> CDF rows, range decoder
    for (int s = 0; s < 4; ++s) {
        const float *xin = k.a;
        if (s) {
            const float4 *X = reinterpret_cast<const float4 *>(k.a), *emb = reinterpret_cast<const float4 *>(k.semb[s - 1]);
            for (int64_t t = m.gtid; t < n8; t += m.NT) {
                const uint32_t r = k.m2s[t >> 3];
                uint32_t prev = k.sym[0][r];
                if (s >= 2) prev = prev * 2 + k.sym[1][r];
                if (s >= 3) prev = prev * 4 + k.sym[2][r];
                const float4 xx = X[t], e = emb[prev * 8 + (t & 7)];
                reinterpret_cast<float4 *>(k.u)[t] = make_float4(xx.x + e.x, xx.y + e.y, xx.z + e.z, xx.w + e.w);
            }
            if (!grid_barrier(bc)) return;
            xin = k.u;
        }
        if (!conv_phases(k, bc, m, pc, xin, k.w[5 + 2 * s], nullptr, k.x, 1, true)) return;
        if (!conv_phases(k, bc, m, pc, k.x, k.w[6 + 2 * s], nullptr, k.b, 0, false)) return;
        if (s < 2) head_rows<2>(k, m, s, reinterpret_cast<float *>(lds));
        else if (s == 2) head_rows<4>(k, m, s, reinterpret_cast<float *>(lds));
        else head_rows<16>(k, m, s, reinterpret_cast<float *>(lds));
        if (!grid_barrier(bc)) return;
        if (s < 2) rc_rows<3, CODER>(k, m, s, lds);
        else if (s == 2) rc_rows<5, CODER>(k, m, s, lds);
        else rc_rows<17, CODER>(k, m, s, lds);
        if (!grid_barrier(bc)) return;
    }
    // occupancy byte from the four symbol arrays (raster order) -> Morton order (pcc_utils.py:369)
    for (int64_t i = m.gtid; i < n; i += m.NT) {
        const uint32_t r = k.m2s[i];
        k.occ[i] = (uint8_t)(k.sym[0][r] * 128 + k.sym[1][r] * 64 + k.sym[2][r] * 16 + k.sym[3][r]);
    }
#ifdef FUSED_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("[fused] child  n %d G %u: %u barriers, total %.1f us, in barriers %.1f us (polling %.1f us)\n", n, gridDim.x, bc.epoch, (__builtin_amdgcn_s_memrealtime() - bc.t_start) / 100.0, bc.t_bar / 100.0, bc.t_poll / 100.0);
#endif
}

// Workgroups of a persistent launch (16 waves each).  A phase costs its barrier (2.2 / 2.7 / 3.9 us at 64 / 128 / 256 workgroups,
// tools/ubench/grid_sync.hip) plus one or two memory round trips, so the grid is the smallest that gives every wave at most ~2 tiles:
// a sparse level (most small levels: 2-5 neighbours per node) has ~n / 4 tiles, a dense one (the levels right below the base:
// 40-77 neighbours per node) ~4 n.  The host knows the density only through the growth of the level (children per parent node).
int fused_grid(int64_t n, int64_t np)
{
    static const int forced = dev_env_int("GAUSPCC_FUSED_GRID", 0);
    if (forced >= 1 && forced <= 256) return forced;
    const bool dense = np > 0 && n > 3 * np;
    const int64_t tiles = dense ? 5 * n : n / 2 + 64;
    const int64_t g = cdiv(tiles, 2 * (FUSE_THREADS / 64));   // two tiles per wave
    // never more workgroups than the device holds at once (the grid barrier needs them all resident): 256 CUs x one workgroup on a
    // whole MI355X, fewer on a partitioned (CPX / DPX) part
    const int64_t cap = std::max<int64_t>(16, std::min<int64_t>(256, fused_device_cap() / 16 * 16));
    return (int)std::min<int64_t>(cap, std::max<int64_t>(16, (g + 15) / 16 * 16));
}

// Two persistent launches whose workgroups spin on grid barriers must not share the device: each may hold CUs the other's missing
// workgroups need (observed with two scenes in flight: both spin until their bounded polls give up, ~2 s, and their contexts fall
// back to the launch-per-layer path).  Contexts of one process therefore CHAIN their persistent launches per device: a launch
// waits for the event behind the previous context's launch.  One context alone pays nothing (no event calls).  Other
// processes on the same GPU are beyond this gate: for them the bounded spins and the fallback remain.
struct FusedGate {
    static std::mutex &mu() { static std::mutex m; return m; }
    static hipEvent_t &last(int dev) { static hipEvent_t e[64] = {}; return e[dev & 63]; }
    static int &users(int dev) { static int n[64] = {}; return n[dev & 63]; }
    std::unique_lock<std::mutex> lk;
    gpcc_ctx *ctx; hipStream_t st; bool chain;
    FusedGate(gpcc_ctx *c, hipStream_t s) : lk(mu()), ctx(c), st(s), chain(users(c->device) > 1) {}
    int begin()
    {
        if (!chain) return GPCC_OK;
        hipEvent_t prev = last(ctx->device);
        if (prev && prev != ctx->fused_ev) HIP_TRY(hipStreamWaitEvent(st, prev, 0));
        return GPCC_OK;
    }
    int end()
    {
        if (!chain) return GPCC_OK;
        HIP_TRY(hipEventRecord(ctx->fused_ev, st));
        last(ctx->device) = ctx->fused_ev;
        return GPCC_OK;
    }
};

// barrier blocks + sticky timeout word of a context: [FusedBar][FusedBar][32 words]
int fused_state(gpcc_ctx *ctx, hipStream_t st, FusedBar **cur, FusedBar **next, uint32_t **tmo)
{
    if (!ctx->fused_state) {
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, 2 * sizeof(FusedBar) + 128));
        HIP_TRY(hipMemsetAsync(p, 0, 2 * sizeof(FusedBar) + 128, st));
        ctx->fused_state = p;
        ctx->fused_flip = 0;
        HIP_TRY(hipEventCreateWithFlags(&ctx->fused_ev, hipEventDisableTiming));
        std::lock_guard<std::mutex> g(FusedGate::mu());
        FusedGate::users(ctx->device) += 1;
    }
    FusedBar *b = static_cast<FusedBar *>(ctx->fused_state);
    *cur = b + ctx->fused_flip; *next = b + (ctx->fused_flip ^ 1);
    *tmo = reinterpret_cast<uint32_t *>(b + 2);
    ctx->fused_flip ^= 1;
    return GPCC_OK;
}

}  // namespace

int fused_device_cap()
{
    static std::mutex mu;
    static int cap[64];
    static bool have[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> g(mu);
    if (!have[dev & 63]) {
        hipDeviceProp_t p;
        int occ = 0;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_level_fused<FUSED_CHILD, RC_CODER_CARRY>, FUSE_THREADS, FUSE_LDS_BYTES) != hipSuccess) return 0;
        cap[dev & 63] = std::max(0, occ * p.multiProcessorCount - 2);   // two CUs of slack for whatever else runs
        have[dev & 63] = true;
    }
    return cap[dev & 63];
}

void fused_ctx_release(gpcc_ctx *ctx)
{
    if (!ctx->fused_state) return;
    std::lock_guard<std::mutex> g(FusedGate::mu());
    FusedGate::users(ctx->device) -= 1;
    if (FusedGate::last(ctx->device) == ctx->fused_ev) FusedGate::last(ctx->device) = nullptr;
    if (ctx->fused_ev) { (void)hipEventDestroy(ctx->fused_ev); ctx->fused_ev = nullptr; }
}

uint32_t *fused_timeout_word(gpcc_ctx *ctx) { return ctx->fused_state ? reinterpret_cast<uint32_t *>(static_cast<FusedBar *>(ctx->fused_state) + 2) : nullptr; }
int fused_reset(gpcc_ctx *ctx, hipStream_t st)
{
    if (ctx->fused_state) HIP_TRY(hipMemsetAsync(ctx->fused_state, 0, 2 * sizeof(FusedBar) + 128, st));
    ctx->fused_flip = 0;
    return GPCC_OK;
}

int fused_child_level(gpcc_ctx *ctx, hipStream_t st, const gpcc_model *m, const PairPlan &plan, const FusedChild &a)
{
    FusedK k = {};
    k.pl = plan_view(plan);
    GP_TRY(fused_state(ctx, st, &k.bar, &k.bar_next, &k.tmo));
    k.desert = fused_test_desert();
    k.P = a.P;
    k.pA = a.pA; k.parent = a.parent; k.rkey = a.rkey; k.m2r = a.m2r;
    k.m2s = a.spos ? a.spos : a.m2r; k.cpos = a.cpos;
    k.x = a.cX; k.a = a.cA; k.b = a.cB; k.u = a.cU;
    for (int i = 0; i < 13; ++i) k.w[i] = m->conv[5 + i] + (size_t)m->K * 1024;
    k.temb = m->temb;
    for (int i = 0; i < 3; ++i) k.semb[i] = m->semb[i];
    for (int i = 0; i < 4; ++i) { k.hfrag[i] = m->hfrag[i]; k.sym[i] = a.sym[i]; k.rdw[i] = (uint32_t)rc_window_dwords(a.win_bytes[i]); }
    k.cdf = a.cdf; k.occ = a.occ; k.bytes = a.bytes; k.chunks = a.chunks; k.nlanes = a.nlanes; k.llog = a.llog;
    const int G = fused_grid(plan.n, a.np);
    FusedGate gate(ctx, st);
    GP_TRY(gate.begin());
    if (a.coder == RC_CODER_CARRY) k_level_fused<FUSED_CHILD, RC_CODER_CARRY><<<(unsigned)G, FUSE_THREADS, FUSE_LDS_BYTES, st>>>(k);
    else k_level_fused<FUSED_CHILD, RC_CODER_CARRYLESS><<<(unsigned)G, FUSE_THREADS, FUSE_LDS_BYTES, st>>>(k);
    LAUNCH_CHECK();
    return gate.end();
}

// can the range-decoder phases of a level keep their byte windows in the fused kernel's LDS?
bool fused_windows_fit(int64_t n, int64_t np, uint32_t nlanes, const uint32_t win_bytes[4])
{
    const uint32_t G = (uint32_t)fused_grid(n, np), LPW = (nlanes + G - 1u) / G;
    if (LPW > 64u) return false;   // (a wave decodes its workgroup's lanes of the 3- / 5-entry streams: one lane each)
    for (int s = 0; s < 4; ++s) {
        const uint64_t rdw = rc_window_dwords(win_bytes[s]);
        const uint64_t lanes = s == 3 ? (uint64_t)((LPW + 3u) / 4u) * 4u : LPW;
        if (s == 3 && (LPW + 3u) / 4u > (uint32_t)(FUSE_THREADS / 64)) return false;
        if (lanes * rdw * 4u > FUSE_LDS_BYTES) return false;
    }
    return true;
}

int fused_parent_trunk(gpcc_ctx *ctx, hipStream_t st, const gpcc_model *m, const PairPlan &plan, int64_t np, const uint8_t *occ, float *pF, float *pA, float *pB, float *P)
{
    FusedK k = {};
    k.pl = plan_view(plan);
    GP_TRY(fused_state(ctx, st, &k.bar, &k.bar_next, &k.tmo));
    k.desert = fused_test_desert();
    k.P = P;
    k.x = pF; k.a = pA; k.b = pB;
    for (int i = 0; i < 5; ++i) k.w[i] = m->conv[i] + (size_t)m->K * 1024;
    k.prior_emb = m->prior_emb;
    k.occ = const_cast<uint8_t *>(occ);
    const int G = fused_grid(plan.n, np);
    FusedGate gate(ctx, st);
    GP_TRY(gate.begin());
    k_level_fused<FUSED_PARENT, RC_CODER_CARRYLESS><<<(unsigned)G, FUSE_THREADS, FUSE_LDS_BYTES, st>>>(k);
    LAUNCH_CHECK();
    return gate.end();
}

}  // namespace gpcc
