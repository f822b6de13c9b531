// octree.hip -- octree build / expansion / neighbour maps on gfx950.
// All kernels here are HBM-bound integer work: coalesced 4/8-byte streams, no LDS tiling needed
// except the level histogram.  See octree.hpp for the data model.
#include "octree.hpp"
#include "primitives.hpp"

namespace gpcc {

constexpr int TB = 256;
static inline unsigned nblk(int64_t n, int per = TB) { return (unsigned)cdiv(n, per); }

// ------------------------------------------------------------------ leaves
__global__ __launch_bounds__(TB) void k_bbox(const int32_t *__restrict__ xyz, int64_t n, int32_t *__restrict__ bbox)
{
    int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int v = xyz[3 * i + a];
            mn[a] = min(mn[a], v);
            mx[a] = max(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = min(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = max(mx[a], __shfl_xor(mx[a], d, 64));
        }
    }
    // one set of atomics per block (the six words are a serial hot spot: 4 waves x 1024 blocks of them cost 0.25 ms)
    __shared__ int red[TB / 64][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (int w = 1; w < TB / 64; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], v);
        else atomicMax(&bbox[threadIdx.x], v);
    }
}

__global__ __launch_bounds__(TB) void k_leaf_keys(const int32_t *__restrict__ xyz, int64_t n, Bias3 b, uint64_t *__restrict__ mkey)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    mkey[i] = morton3((uint32_t)((int64_t)xyz[3 * i] + b.v[0]), (uint32_t)((int64_t)xyz[3 * i + 1] + b.v[1]), (uint32_t)((int64_t)xyz[3 * i + 2] + b.v[2]));
}

// counts[l] (l = 0..21): number of sorted leaves whose highest differing Morton triple vs the
// previous leaf is l (leaf 0 counts as 21); counts[22] = number of duplicates.
__global__ __launch_bounds__(TB) void k_leaf_levels(const uint64_t *__restrict__ mkey, int64_t n, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t c[24];
    if (threadIdx.x < 24) c[threadIdx.x] = 0;
    __syncthreads();
    // grid-stride: a workgroup's ~16 global adds land on 24 words of ONE cache line, and device-scope atomics on a word serialise at ~10 ns -- with a
    // workgroup per 256 leaves (3 907 of them at 1 M points) the kernel took 46 us, 40 of them waiting for that line
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        int l;
        if (i == 0) l = 21;
        else {
            uint64_t d = mkey[i] ^ mkey[i - 1];
            l = d == 0 ? 22 : (63 - __clzll((long long)d)) / 3;
        }
        atomicAdd(&c[l], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 24 && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], c[threadIdx.x]);
}

// ------------------------------------------------------------------ bottom-up level build
__global__ __launch_bounds__(TB) void k_level_flags(const uint64_t *__restrict__ key, int64_t n, uint32_t *__restrict__ flag)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || (key[i] >> 3) != (key[i - 1] >> 3)) ? 1u : 0u;
}

// one child of the level below: its parent index, and for the first child of a parent the parent's node (Morton key, raster
// key, first child, occupancy).  pos = exclusive scan of the head flags at i.
__device__ __forceinline__ void level_build_at(const uint64_t *__restrict__ key, int64_t n, int64_t i, uint32_t pos, uint64_t *__restrict__ key_up, uint64_t *__restrict__ rkey_up,
                                               uint32_t *__restrict__ cstart_up, uint8_t *__restrict__ occ_up, uint32_t *__restrict__ parent_lo)
{
    const uint64_t k = key[i], pk = k >> 3;
    const bool head = i == 0 || (key[i - 1] >> 3) != pk;
    const uint32_t p = pos - (head ? 0u : 1u);
    if (parent_lo) parent_lo[i] = p;
    if (head) {
        uint32_t occ = 0;
        for (int j = 0; j < 8 && i + j < n; ++j) {
            uint64_t kj = key[i + j];
            if ((kj >> 3) != pk) break;
            occ |= 1u << (kj & 7);
        }
        key_up[p] = pk;
        rkey_up[p] = rkey3(compact1by2(pk), compact1by2(pk >> 1), compact1by2(pk >> 2));
        cstart_up[p] = (uint32_t)i;
        occ_up[p] = (uint8_t)occ;
    }
    if (i == n - 1) cstart_up[p + 1] = (uint32_t)n;  // sentinel: cstart[n_up] = n_lo
}

__global__ __launch_bounds__(TB) void k_level_build(const uint64_t *__restrict__ key, int64_t n, const uint32_t *__restrict__ pos,
                                                    uint64_t *__restrict__ key_up, uint64_t *__restrict__ rkey_up, uint32_t *__restrict__ cstart_up,
                                                    uint8_t *__restrict__ occ_up, uint32_t *__restrict__ parent_lo)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    level_build_at(key, n, i, pos[i], key_up, rkey_up, cstart_up, occ_up, parent_lo);
}

// Levels of at most LEVEL_SINGLE_MAX children: head flags, their scan and the build in ONE single-workgroup launch (the upper
// half of a tree is launch-latency-bound: four launches per level otherwise).  The workgroup walks the children in tiles of
// 1024 with the running count in a register.
// (round 6: 1 024 threads -- sixteen waves' partial counts chained through LDS -- so that 8 k children are two tiles, not eight: a tile is a chain of
// dependent round trips, ~4 us whatever its width; the four levels of S1M that take this path 30 + 21 + 24 + 17 us -> see profiles/r06_timeline.txt)
constexpr int LS_T = 1024, LS_E = 4, LS_TILE = LS_T * LS_E;
constexpr int64_t LEVEL_SINGLE_MAX = 2 * LS_TILE;   // (8 k children, as before; above: the flags / scan / build launches)
__global__ __launch_bounds__(LS_T) void k_level_up_single(const uint64_t *__restrict__ key, int64_t n, uint64_t *__restrict__ key_up, uint64_t *__restrict__ rkey_up,
                                                         uint32_t *__restrict__ cstart_up, uint8_t *__restrict__ occ_up, uint32_t *__restrict__ parent_lo)
{
    constexpr int NWV = LS_T / 64;
    __shared__ uint32_t lds[NWV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < n; b0 += LS_TILE) {
        const int64_t base = b0 + (int64_t)threadIdx.x * LS_E;
        uint32_t f[LS_E], s = 0;
#pragma unroll
        for (int e = 0; e < LS_E; ++e) {
            const int64_t i = base + e;
            f[e] = i < n ? (uint32_t)(i == 0 || (key[i] >> 3) != (key[i - 1] >> 3)) : 0u;
            s += f[e];
        }
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += v;
        }
        __syncthreads();
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const uint32_t v = lds[w]; before += w < wave ? v : 0u; all += v; }
        uint32_t ex = carry + before + inc - s;
#pragma unroll
        for (int e = 0; e < LS_E; ++e) {
            if (base + e < n) level_build_at(key, n, base + e, ex, key_up, rkey_up, cstart_up, occ_up, parent_lo);
            ex += f[e];
        }
        carry += all;
    }
}

// ------------------------------------------------------------------ raster ranks
__global__ __launch_bounds__(TB) void k_rank_keys(const uint64_t *__restrict__ rkey, int64_t n, int hb, uint64_t *__restrict__ skey, uint32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = rkey[i], msk = (1ull << hb) - 1;
    skey[i] = ((uint64_t)(rk_z(k) & msk) << (2 * hb)) | ((uint64_t)(rk_y(k) & msk) << hb) | (rk_x(k) & msk);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(TB) void k_invert_perm(const uint32_t *__restrict__ r2m, int64_t n, uint32_t *__restrict__ m2r)
{
    int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    m2r[r2m[r]] = (uint32_t)r;
}

int level_raster_rank(gpcc_ctx *ctx, hipStream_t st, Level *lv, int hb_level)
{
    const int64_t n = lv->n;
    if (hb_level < 1) hb_level = 1;
    if (hb_level > 21) hb_level = 21;
    size_t mk = ctx->arena.mark();
    TAKE(ka, uint64_t, n);
    TAKE(kb, uint64_t, n);
    TAKE(vb, uint32_t, n);
    uint32_t *va = lv->r2m;
    k_rank_keys<<<nblk(n), TB, 0, st>>>(lv->rkey, n, hb_level, ka, va);
    LAUNCH_CHECK();
    uint64_t *k0 = ka, *k1 = kb;
    uint32_t *v0 = va, *v1 = vb;
    GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, n, 3 * hb_level));
    if (v0 != lv->r2m) HIP_TRY(hipMemcpyAsync(lv->r2m, v0, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st));
    k_invert_perm<<<nblk(n), TB, 0, st>>>(lv->r2m, n, lv->m2r);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

// ------------------------------------------------------------------ raster ranks of a level from its parent's (no sort)
// Raster order is (z, y, x) = (pz, dz, py, dy, px, dx) for the child (dx, dy, dz) of the parent (px, py, pz): walking the
// parents in THEIR raster order, slab by slab (equal pz) and inside a slab row by row (equal py), the children come out in
// raster order when every slab is walked twice (dz = 0, 1) and inside it every row twice (dy = 0, 1).  So the raster rank
// of a child is a prefix sum over that walk of "children of parent r with this (dz, dy)" (0..2 each) -- one scan over
// 4 n_parent small counts placed at their position in the walk -- plus 1 for the dx = 1 child behind an existing dx = 0
// sibling.  Slab / row boundaries come from two flag scans over the parents.  ~200 B of traffic per parent against
// ceil(3 hb / 8) = 5..6 radix passes of 24 B x 2 per child.
__device__ __forceinline__ void rr_flags_at(const uint64_t *__restrict__ rkey, const uint32_t *__restrict__ r2m, int64_t r, uint32_t &s, uint32_t &w)
{
    const uint64_t k = rkey[r2m[r]];
    s = 1; w = 1;
    if (r > 0) {
        const uint64_t kp = rkey[r2m[r - 1]];
        s = rk_z(k) != rk_z(kp);
        w = s | (uint32_t)(rk_y(k) != rk_y(kp));
    }
}

__global__ __launch_bounds__(TB) void k_rr_flags(const uint64_t *__restrict__ rkey, const uint32_t *__restrict__ r2m, int64_t n, uint32_t *__restrict__ fs, uint32_t *__restrict__ fw)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    uint32_t s, w;
    rr_flags_at(rkey, r2m, r, s, w);
    fs[r] = s; fw[r] = w;
}

// es / ew: exclusive scans of fs / fw; segment id of r = e[r] + f[r] - 1
__device__ __forceinline__ void rr_starts_at(uint32_t fs, uint32_t fw, uint32_t es, uint32_t ew, int64_t r, int64_t n, uint32_t *__restrict__ sstart, uint32_t *__restrict__ wstart)
{
    const uint32_t sid = es + fs - 1u, wid = ew + fw - 1u;
    if (fs) sstart[sid] = (uint32_t)r;
    if (fw) wstart[wid] = (uint32_t)r;
    if (r == n - 1) { sstart[sid + 1] = (uint32_t)n; wstart[wid + 1] = (uint32_t)n; }
}

__global__ __launch_bounds__(TB) void k_rr_starts(const uint32_t *__restrict__ fs, const uint32_t *__restrict__ fw, const uint32_t *__restrict__ es, const uint32_t *__restrict__ ew,
                                                  int64_t n, uint32_t *__restrict__ sstart, uint32_t *__restrict__ wstart)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    rr_starts_at(fs[r], fw[r], es[r], ew[r], r, n, sstart, wstart);
}

__device__ __forceinline__ void rr_counts_at(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m, uint32_t sid, uint32_t wid, const uint32_t *__restrict__ sstart,
                                             const uint32_t *__restrict__ wstart, int64_t r, uint32_t *__restrict__ cnt4, uint4 *__restrict__ walk)
{
    const uint32_t S = sstart[sid], Send = sstart[sid + 1], W = wstart[wid], Wend = wstart[wid + 1];
    const uint32_t posbase = 4u * S + 2u * (W - S) + ((uint32_t)r - W), shalf = 2u * (Send - S), rlen = Wend - W;
    const uint32_t m = r2m[r];
    const uint32_t o = occ[m];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) cnt4[posbase + dz * shalf + dy * rlen] = (uint32_t)__popc(o & (3u << (2 * dy + 4 * dz)));
    walk[r] = make_uint4(posbase, shalf, rlen, m);
}

__global__ __launch_bounds__(TB) void k_rr_counts(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m, const uint32_t *__restrict__ fs, const uint32_t *__restrict__ fw,
                                                  const uint32_t *__restrict__ es, const uint32_t *__restrict__ ew, const uint32_t *__restrict__ sstart,
                                                  const uint32_t *__restrict__ wstart, int64_t n, uint32_t *__restrict__ cnt4, uint4 *__restrict__ walk)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    rr_counts_at(occ, r2m, es[r] + fs[r] - 1u, ew[r] + fw[r] - 1u, sstart, wstart, r, cnt4, walk);
}

__device__ __forceinline__ void rr_assign_at(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ cstart, const uint4 *__restrict__ walk, const uint32_t *__restrict__ base4,
                                             int64_t t, int64_t nc, uint32_t *__restrict__ m2r_c, uint32_t *__restrict__ r2m_c)
{
    const int64_t r = t >> 3;
    const int q = (int)(t & 7);
    const uint4 w = walk[r];
    const uint32_t o = occ[w.w];
    if (!((o >> q) & 1u)) return;
    const uint32_t ci = cstart[w.w] + (uint32_t)__popc(o & ((1u << q) - 1u));
    const int dx = q & 1, dy = (q >> 1) & 1, dz = q >> 2;
    const uint32_t rank = base4[w.x + dz * w.y + dy * w.z] + (uint32_t)(dx && ((o >> (q - 1)) & 1u));
    if ((int64_t)ci >= nc || (int64_t)rank >= nc) return;   // a header that understates the level (the decoder reports it at its final sync)
    m2r_c[ci] = rank;
    r2m_c[rank] = ci;
}

__global__ __launch_bounds__(TB) void k_rr_assign(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ cstart, const uint4 *__restrict__ walk,
                                                  const uint32_t *__restrict__ base4, int64_t n, int64_t nc, uint32_t *__restrict__ m2r_c, uint32_t *__restrict__ r2m_c)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if ((t >> 3) >= n) return;
    rr_assign_at(occ, cstart, walk, base4, t, nc, m2r_c, r2m_c);
}

__device__ __forceinline__ void expand_at(const uint64_t *__restrict__ rkey, const uint8_t *__restrict__ occ, const uint32_t *__restrict__ cstart, int64_t t,
                                          uint64_t *__restrict__ rkey_c, uint32_t *__restrict__ parent_c, int64_t nc_cap)
{
    const int64_t p = t >> 3;
    const int q = (int)(t & 7);
    const uint32_t o = occ[p];
    if (!((o >> q) & 1u)) return;
    const uint32_t idx = cstart[p] + (uint32_t)__popc(o & ((1u << q) - 1u));
    if ((int64_t)idx >= nc_cap) return;   // a header that understates the level: the caller reports it at its next sync
    const uint64_t k = rkey[p];
    rkey_c[idx] = rkey3(2 * rk_x(k) + (q & 1), 2 * rk_y(k) + ((q >> 1) & 1), 2 * rk_z(k) + ((q >> 2) & 1));
    parent_c[idx] = (uint32_t)p;
}

// ------------------------------------------------------------------ small levels: expansion + ranks in ONE single-workgroup launch
// A decode spends ~12 launches of 4-5 us on the structure of a level (population counts, scan, expansion, the rank
// derivation above with its three scans); for the first levels that chain, not the five parent convolutions beside it, is
// what the level waits for.  One workgroup of 1024 threads runs the same steps with workgroup barriers between them: every
// thread owns a contiguous run of each array, scans are (thread run -> wave shuffle -> 16 wave totals).  Measured (MI355X):
// 6 / 8 / 15 us for 8 / 64 / 511 parents against ~55 us of launches; from ~2 k parents on the per-thread runs of dependent
// loads make the single workgroup SLOWER than the launches (170 us at 8 k parents), hence the limits.
constexpr int SL_T = 1024;
constexpr int64_t SMALL_PAR_MAX = 1024, SMALL_CHI_MAX = 8192;

// exclusive scan of f(0..len) into out (may alias what f reads at the same index); returns the total to every thread
template <typename F>
__device__ __forceinline__ uint32_t block_exscan(F f, uint32_t *__restrict__ out, int len, uint32_t *lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = (len + SL_T - 1) / SL_T, b = min(len, tid * C), e = min(len, b + C);
    uint32_t s = 0;
    for (int i = b; i < e; ++i) s += f(i);
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += v;
    }
    __syncthreads();   // lds free (a previous call's readers are done)
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SL_T / 64; ++w) { const uint32_t v = lds[w]; wbase += w < wave ? v : 0u; total += v; }
    uint32_t run = wbase + inc - s;
    for (int i = b; i < e; ++i) { const uint32_t v = f(i); out[i] = run; run += v; }
    __syncthreads();   // out complete and visible to the workgroup
    return total;
}

struct SmallLevelArgs {
    const uint64_t *rkey; const uint8_t *occ; const uint32_t *r2m; uint32_t *cstart; int n;   // parent level (cstart is written when EXPAND)
    uint64_t *rkey_c; uint32_t *parent_c, *m2r_c, *r2m_c; int nc;                              // child level (rkey_c / parent_c written when EXPAND)
    uint32_t *total_dev;                                                                          // EXPAND: number of children the occupancy expands to
    uint32_t *ef, *sstart, *wstart, *cnt4; uint4 *walk;                                         // scratch: n, n + 1, n + 1, 4 n, n
};

template <bool EXPAND>
__global__ __launch_bounds__(SL_T) void k_small_level(SmallLevelArgs a)
{
    __shared__ uint32_t lds[SL_T / 64];
    const int tid = threadIdx.x, n = a.n, nc = a.nc;
    if (EXPAND) {
        const uint32_t total = block_exscan([&](int i) { return (uint32_t)__popc((uint32_t)a.occ[i]); }, a.cstart, n, lds);
        if (tid == 0) { a.cstart[n] = total; if (a.total_dev) *a.total_dev = total; }
        // a lying container header may leave children unwritten: keep every index valid underneath
        for (int i = tid; i < nc; i += SL_T) a.parent_c[i] = 0;
    }
    for (int i = tid; i < nc; i += SL_T) { a.m2r_c[i] = 0; a.r2m_c[i] = 0; }
    __syncthreads();
    if (EXPAND)
        for (int t = tid; t < 8 * n; t += SL_T) expand_at(a.rkey, a.occ, a.cstart, t, a.rkey_c, a.parent_c, nc);
    // slab / row flags of the parents in raster order, both scans in one (n < 2^16: a 16-bit field each)
    auto flags = [&](int r) -> uint32_t { uint32_t s, w; rr_flags_at(a.rkey, a.r2m, r, s, w); return s | (w << 16); };
    block_exscan(flags, a.ef, n, lds);
    for (int r = tid; r < n; r += SL_T) {
        const uint32_t f = flags(r), e = a.ef[r];
        rr_starts_at(f & 0xFFFFu, f >> 16, e & 0xFFFFu, e >> 16, r, n, a.sstart, a.wstart);
    }
    __syncthreads();
    for (int r = tid; r < n; r += SL_T) {
        const uint32_t f = flags(r), e = a.ef[r];
        rr_counts_at(a.occ, a.r2m, (e & 0xFFFFu) + (f & 0xFFFFu) - 1u, (e >> 16) + (f >> 16) - 1u, a.sstart, a.wstart, r, a.cnt4, a.walk);
    }
    __syncthreads();
    block_exscan([&](int i) { return a.cnt4[i]; }, a.cnt4, 4 * n, lds);
    for (int t = tid; t < 8 * n; t += SL_T) rr_assign_at(a.occ, a.cstart, a.walk, a.cnt4, t, nc, a.m2r_c, a.r2m_c);
}

// par (ranks known) -> chi: ranks of the children, and with `expand` also par->cstart, chi->rkey / parent and the child count
int small_level(gpcc_ctx *ctx, hipStream_t st, Level *par, Level *chi, bool expand, uint32_t *total_dev)
{
    const int64_t n = par->n;
    if (n > SMALL_PAR_MAX || chi->n > SMALL_CHI_MAX) return fail(GPCC_ERR_ARG, "internal: small_level on %lld -> %lld nodes", (long long)n, (long long)chi->n);
    const size_t mk = ctx->arena.mark();
    TAKE(ef, uint32_t, n); TAKE(sstart, uint32_t, n + 1); TAKE(wstart, uint32_t, n + 1); TAKE(cnt4, uint32_t, 4 * n); TAKE(walk, uint4, n);
    SmallLevelArgs a;
    a.rkey = par->rkey; a.occ = par->occ; a.r2m = par->r2m; a.cstart = par->cstart; a.n = (int)n;
    a.rkey_c = chi->rkey; a.parent_c = chi->parent; a.m2r_c = chi->m2r; a.r2m_c = chi->r2m; a.nc = (int)chi->n;
    a.total_dev = total_dev; a.ef = ef; a.sstart = sstart; a.wstart = wstart; a.cnt4 = cnt4; a.walk = walk;
    if (expand) k_small_level<true><<<1, SL_T, 0, st>>>(a);
    else k_small_level<false><<<1, SL_T, 0, st>>>(a);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

bool small_level_fits(const Level *par, const Level *chi) { return par && par->n <= SMALL_PAR_MAX && chi->n <= SMALL_CHI_MAX; }

int level_ranks_from_parent(gpcc_ctx *ctx, hipStream_t st, const Level *par, Level *chi)
{
    const int64_t n = par->n, nc = chi->n;
    if (4 * n >= (int64_t)1 << 32) return fail(GPCC_ERR_ARG, "level too large");
    const size_t mk = ctx->arena.mark();
    TAKE(fs, uint32_t, n); TAKE(fw, uint32_t, n); TAKE(es, uint32_t, n); TAKE(ew, uint32_t, n);
    TAKE(sstart, uint32_t, n + 1); TAKE(wstart, uint32_t, n + 1);
    TAKE(cnt4, uint32_t, 4 * n); TAKE(walk, uint4, n);
    k_rr_flags<<<nblk(n), TB, 0, st>>>(par->rkey, par->r2m, n, fs, fw);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_pair_u32(ctx, st, fs, es, fw, ew, n));
    k_rr_starts<<<nblk(n), TB, 0, st>>>(fs, fw, es, ew, n, sstart, wstart);
    LAUNCH_CHECK();
    k_rr_counts<<<nblk(n), TB, 0, st>>>(par->occ, par->r2m, fs, fw, es, ew, sstart, wstart, n, cnt4, walk);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_u32(ctx, st, cnt4, cnt4, 4 * n, nullptr));
    // (a decoder's child arrays are zeroed by level_expand_rank: a corrupt stream may leave ranks unwritten; an encoder's tree writes them all)
    k_rr_assign<<<nblk(n * 8), TB, 0, st>>>(par->occ, par->cstart, walk, cnt4, n, nc, chi->m2r, chi->r2m);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

// sort below this size (one single-workgroup launch), derive above it
constexpr int64_t RANK_SORT_MAX = 1024;
static bool no_fuse()   // cross-check knob: the one-launch-per-step path for every level
{
    static const bool v = dev_env_int("GAUSPCC_SMALL_FUSE", 1) == 0;
    return v;
}

int rank_level(gpcc_ctx *ctx, hipStream_t st, const Level *par, Level *chi, int hb_level)
{
    static const bool force_sort = dev_env_int("GAUSPCC_RANK_SORT", 0) != 0;   // cross-check knob
    if (!par || force_sort) return level_raster_rank(ctx, st, chi, hb_level);
    if (small_level_fits(par, chi) && !no_fuse()) return small_level(ctx, st, const_cast<Level *>(par), chi, false, nullptr);
    if (chi->n <= RANK_SORT_MAX) return level_raster_rank(ctx, st, chi, hb_level);
    return level_ranks_from_parent(ctx, st, par, chi);
}

// ------------------------------------------------------------------ encode-side tree build
static inline int bitlen(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

int tree_pick_bias(const int32_t mn[3], const int32_t mx[3], int64_t bias[3])
{
    bool inside = true;
    int64_t ext = 0;
    for (int a = 0; a < 3; ++a) {
        inside = inside && mn[a] >= -CLIM && mx[a] < CLIM;
        ext = std::max<int64_t>(ext, (int64_t)mx[a] - (int64_t)mn[a]);
    }
    if (inside) { bias[0] = bias[1] = bias[2] = CB; return GPCC_OK; }
    // Anywhere in int32: the tree has at most max(1, bitlen(extent)) levels (a level whose cells are wider than half the
    // extent has at most 27 nodes, and the FOG loop stops below 64), so an origin that is a multiple of A = 2^that keeps
    // every floor-halving exact.  Internal coordinates then lie in [0, A + extent] and must fit 21 bits: extent < 2^20.
    const int hbE = std::max(1, bitlen((uint64_t)ext));
    if (hbE > 20)
        return fail(GPCC_ERR_RANGE, "coordinate out of range: a cloud that leaves (-2^20, 2^20) must have an extent below 2^20 (it spans %lld)", (long long)ext);
    const int64_t A = (int64_t)1 << hbE;
    for (int a = 0; a < 3; ++a) {
        const int64_t m = mn[a];
        const int64_t fl = (m >= 0 ? m / A : -((-m + A - 1) / A)) * A;   // floor to a multiple of A
        bias[a] = -fl;
    }
    return GPCC_OK;
}

static int level_alloc(gpcc_ctx *ctx, Level *lv, int64_t n, int lvl)
{
    lv->n = n; lv->lvl = lvl;
    TAKE(rkey, uint64_t, n); TAKE(occ, uint8_t, n); TAKE(cstart, uint32_t, n + 1); TAKE(parent, uint32_t, n);
    TAKE(m2r, uint32_t, n); TAKE(r2m, uint32_t, n);
    lv->rkey = rkey; lv->occ = occ; lv->cstart = cstart; lv->parent = parent; lv->m2r = m2r; lv->r2m = r2m;
    return GPCC_OK;
}

double tree_alg_bytes(const Tree &T)
{
    double b = (double)T.npts * (12 + 8) + (double)cdiv(3 * T.hb, 8) * (double)T.npts * 16 + (double)T.npts * 8;
    int64_t n_lo = T.npts;
    for (int l = 1; l <= T.L; ++l) {
        const int64_t n_up = T.lv[T.L - l].n;
        const int hbl = std::max(1, T.hb - l);
        b += (double)n_lo * (8 + 4) + (double)n_up * (25 + 16) + (double)cdiv(3 * hbl, 8) * (double)n_up * 24 + (double)n_up * 8;
        n_lo = n_up;
    }
    return b;
}

int tree_build(gpcc_ctx *ctx, hipStream_t st, const int32_t *xyz, int64_t n, Tree *T)
{
    if (n <= 0) return fail(GPCC_ERR_ARG, "empty point cloud");
    if (n >= (int64_t)1 << 31) return fail(GPCC_ERR_ARG, "point count must be < 2^31");
    GP_TRY(ctx->hstage.reserve(4096));
    int32_t *hb32 = reinterpret_cast<int32_t *>(ctx->hstage.p);
    TAKE(dsmall, int32_t, 64);
    // bbox: one sync
    for (int a = 0; a < 3; ++a) { hb32[a] = INT32_MAX; hb32[3 + a] = INT32_MIN; }
    HIP_TRY(hipMemcpyAsync(dsmall, hb32, 24, hipMemcpyHostToDevice, st));
    k_bbox<<<(unsigned)std::min<int64_t>(cdiv(n, TB), 512), TB, 0, st>>>(xyz, n, dsmall);
    LAUNCH_CHECK();
    HIP_TRY(hipMemcpyAsync(hb32, dsmall, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int hb = 1;
    GP_TRY(tree_pick_bias(hb32, hb32 + 3, T->bias));
    for (int a = 0; a < 3; ++a) hb = std::max(hb, bitlen((uint64_t)((int64_t)hb32[a] + T->bias[a]) ^ (uint64_t)((int64_t)hb32[3 + a] + T->bias[a])));
    T->hb = hb; T->npts = n;
    // sorted Morton keys of the leaves
    TAKE(mk0, uint64_t, n);
    TAKE(mk1, uint64_t, n);
    k_leaf_keys<<<nblk(n), TB, 0, st>>>(xyz, n, Bias3{{T->bias[0], T->bias[1], T->bias[2]}}, mk0);
    LAUNCH_CHECK();
    uint64_t *ka = mk0, *kb = mk1;
    GP_TRY(radix_sort_u64(ctx, st, &ka, &kb, nullptr, nullptr, n, 3 * hb));
    T->leaf_mkey = ka;
    // level sizes: one sync
    uint32_t *dcounts = reinterpret_cast<uint32_t *>(dsmall);
    HIP_TRY(hipMemsetAsync(dcounts, 0, 24 * 4, st));
    k_leaf_levels<<<std::min<unsigned>(nblk(n), 512u), TB, 0, st>>>(ka, n, dcounts);
    LAUNCH_CHECK();
    uint32_t *hc = reinterpret_cast<uint32_t *>(ctx->hstage.p);
    HIP_TRY(hipMemcpyAsync(hc, dcounts, 24 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hc[22]) return fail(GPCC_ERR_DUPLICATE, "input has %u duplicate point(s); the octree occupancy code needs unique voxels", hc[22]);
    int64_t nl[24] = {0};
    {
        int64_t acc = 0;
        for (int l = 21; l >= 1; --l) { acc += hc[l]; nl[l] = acc; }
    }
    int L = 1;
    while (L < 21 && nl[L] >= 64) ++L;  // pcc_utils.py:83-89: stop at the first level with < 64 nodes
    T->L = L;
    // bottom-up: level l from level l-1 (Morton keys); stored depth d = L - l
    const uint64_t *key_lo = ka;
    int64_t n_lo = n;
    Level *lo = nullptr;
    for (int l = 1; l <= L; ++l) {
        Level *up = &T->lv[L - l];
        GP_TRY(level_alloc(ctx, up, nl[l], l));
        TAKE(key_up, uint64_t, nl[l]);
        if (n_lo <= LEVEL_SINGLE_MAX) {
            k_level_up_single<<<1, LS_T, 0, st>>>(key_lo, n_lo, key_up, up->rkey, up->cstart, up->occ, lo ? lo->parent : nullptr);
            LAUNCH_CHECK();
        } else {
            size_t mk = ctx->arena.mark();
            TAKE(flag, uint32_t, n_lo);
            k_level_flags<<<nblk(n_lo), TB, 0, st>>>(key_lo, n_lo, flag);
            LAUNCH_CHECK();
            GP_TRY(exclusive_scan_u32(ctx, st, flag, flag, n_lo, nullptr));
            k_level_build<<<nblk(n_lo), TB, 0, st>>>(key_lo, n_lo, flag, key_up, up->rkey, up->cstart, up->occ, lo ? lo->parent : nullptr);
            LAUNCH_CHECK();
            ctx->arena.rewind(mk);
        }
        key_lo = key_up; n_lo = nl[l]; lo = up;
    }
    return GPCC_OK;
}

// raster ranks of every level, top-down: small levels by a sort of their (z, y, x) keys, the others from their parent's
// ranks.  Nothing of the network needs them (the convolutions run in Morton order); the encoder runs this on its second
// stream beside the tile lists and the first convolutions.
int tree_ranks(gpcc_ctx *ctx, hipStream_t st, Tree *T)
{
    for (int d = 0; d < T->L; ++d) {
        Level *lv = &T->lv[d];
        GP_TRY(rank_level(ctx, st, d ? &T->lv[d - 1] : nullptr, lv, T->hb - lv->lvl));
    }
    return GPCC_OK;
}

// ------------------------------------------------------------------ decode-side expansion
__global__ __launch_bounds__(TB) void k_popc(const uint8_t *__restrict__ occ, int64_t n, uint32_t *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) cnt[i] = (uint32_t)__popc((uint32_t)occ[i]);
}

__global__ __launch_bounds__(TB) void k_expand(const uint64_t *__restrict__ rkey, const uint8_t *__restrict__ occ, const uint32_t *__restrict__ cstart,
                                               int64_t n, uint64_t *__restrict__ rkey_c, uint32_t *__restrict__ parent_c, int64_t nc_cap)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if ((t >> 3) >= n) return;
    expand_at(rkey, occ, cstart, t, rkey_c, parent_c, nc_cap);
}

// expansion + ranks of the child level: one launch for small levels, the two chains otherwise
int level_expand_rank(gpcc_ctx *ctx, hipStream_t st, Level *par, Level *chi, uint32_t *total_dev, int hb_level)
{
    // The child arrays are sized from the container header and the occupancy they are expanded from is whatever the range
    // decoder produced: a corrupt stream (or a lying header) leaves their tail unwritten.  Ranks are INDICES (rows of the
    // CDF and symbol arrays are addressed through m2r / r2m): stale arena bytes there become out-of-range addresses -- a
    // memory fault at 10^6 nodes, where the small clouds of the corruption tests never left mapped memory.  Zero = valid.
    // One memset: the decoder carves a level's arrays from the arena back to back and records the span (codec.hip: alloc_level,
    // Level::span0 / span_bytes); occ and cstart are written in full later (assemble_occ, the next level's popcount scan).
    {
        if (chi->span0 && chi->span_bytes) HIP_TRY(hipMemsetAsync(chi->span0, 0, chi->span_bytes, st));   // the span alloc_level recorded: nothing else lives in it
        else {   // (a caller with its own layout)
            HIP_TRY(hipMemsetAsync(chi->m2r, 0, 4 * (size_t)chi->n, st));
            HIP_TRY(hipMemsetAsync(chi->r2m, 0, 4 * (size_t)chi->n, st));
            HIP_TRY(hipMemsetAsync(chi->rkey, 0, 8 * (size_t)chi->n, st));
            HIP_TRY(hipMemsetAsync(chi->parent, 0, 4 * (size_t)chi->n, st));
        }
    }
    if (small_level_fits(par, chi) && !no_fuse()) return small_level(ctx, st, par, chi, true, total_dev);
    GP_TRY(level_expand(ctx, st, par, chi, total_dev, true));
    return rank_level(ctx, st, par, chi, hb_level);
}

int level_expand(gpcc_ctx *ctx, hipStream_t st, Level *par, Level *chi, uint32_t *total_dev, bool chi_zeroed)
{
    const int64_t n = par->n;
    k_popc<<<nblk(n), TB, 0, st>>>(par->occ, n, par->cstart);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_u32(ctx, st, par->cstart, par->cstart, n, par->cstart + n));
    if (total_dev) HIP_TRY(hipMemcpyAsync(total_dev, par->cstart + n, 4, hipMemcpyDeviceToDevice, st));
    if (chi) {
        // the child arrays were sized from the container header, which is verified only at the caller's next sync: keep
        // every parent index valid (0) where a lying header leaves children unwritten
        if (!chi_zeroed) HIP_TRY(hipMemsetAsync(chi->parent, 0, 4 * (size_t)chi->n, st));
        k_expand<<<nblk(n * 8), TB, 0, st>>>(par->rkey, par->occ, par->cstart, n, chi->rkey, chi->parent, chi->n);
        LAUNCH_CHECK();
    }
    return GPCC_OK;
}

// ------------------------------------------------------------------ outputs
__global__ __launch_bounds__(TB) void k_popc_raster(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m, int64_t n, uint32_t *__restrict__ cnt)
{
    int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r < n) cnt[r] = (uint32_t)__popc((uint32_t)occ[r2m[r]]);
}

__global__ __launch_bounds__(TB) void k_leaves_out(const uint64_t *__restrict__ rkey, const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m,
                                                   const uint32_t *__restrict__ start_r, int64_t n, Bias3 b, int32_t *__restrict__ xyz, int64_t cap)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t r = t >> 3;
    const int q = (int)(t & 7);
    if (r >= n) return;
    const uint32_t m = r2m[r];
    const uint32_t o = occ[m];
    if (!((o >> q) & 1u)) return;
    const int64_t idx = (int64_t)start_r[r] + __popc(o & ((1u << q) - 1u));
    if (idx >= cap) return;   // a container whose header undercounts the leaves: the caller compares the counts after its sync
    const uint64_t k = rkey[m];
    xyz[3 * idx] = (int32_t)((int64_t)(2 * rk_x(k) + (q & 1)) - b.v[0]);
    xyz[3 * idx + 1] = (int32_t)((int64_t)(2 * rk_y(k) + ((q >> 1) & 1)) - b.v[1]);
    xyz[3 * idx + 2] = (int32_t)((int64_t)(2 * rk_z(k) + ((q >> 2) & 1)) - b.v[2]);
}

int leaves_reference_order(gpcc_ctx *ctx, hipStream_t st, const Level *last, const int64_t bias[3], int32_t *xyz_out, int64_t npts)
{
    if (last->lvl != 1) return fail(GPCC_ERR_ARG, "leaves_reference_order: level is not the parents of the leaves");
    const int64_t n = last->n;
    size_t mk = ctx->arena.mark();
    TAKE(cnt, uint32_t, n);
    k_popc_raster<<<nblk(n), TB, 0, st>>>(last->occ, last->r2m, n, cnt);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_u32(ctx, st, cnt, cnt, n, nullptr));
    k_leaves_out<<<nblk(n * 8), TB, 0, st>>>(last->rkey, last->occ, last->r2m, cnt, n, Bias3{{bias[0], bias[1], bias[2]}}, xyz_out, npts);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_level_to_raster(const uint64_t *__restrict__ rkey, const uint8_t *__restrict__ occ, const uint32_t *__restrict__ m2r,
                                                        int64_t n, Bias3 b, int32_t *__restrict__ xyz, uint8_t *__restrict__ occ_out)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    const uint64_t k = rkey[i];
    xyz[3 * (int64_t)r] = (int32_t)((int64_t)rk_x(k) - b.v[0]);
    xyz[3 * (int64_t)r + 1] = (int32_t)((int64_t)rk_y(k) - b.v[1]);
    xyz[3 * (int64_t)r + 2] = (int32_t)((int64_t)rk_z(k) - b.v[2]);
    occ_out[r] = occ[i];
}

int level_to_raster(gpcc_ctx *ctx, hipStream_t st, const Level *lv, const int64_t bias[3], int32_t *xyz_out_dev, uint8_t *occ_out_dev)
{
    // the bias is a multiple of 2^L >= 2^lvl: the arithmetic shift is exact
    k_level_to_raster<<<nblk(lv->n), TB, 0, st>>>(lv->rkey, lv->occ, lv->m2r, lv->n, Bias3{{bias[0] >> lv->lvl, bias[1] >> lv->lvl, bias[2] >> lv->lvl}}, xyz_out_dev, occ_out_dev);
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace gpcc
