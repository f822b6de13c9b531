// forest.hip -- the merged octree of a batch of scenes (forest.hpp): per-scene bounding boxes and level histograms, the
// scene-major level build, the root level's cell map, per-scene CDF slots and leaf output.  HBM-bound integer work like
// octree.hip; what is new is only that a row finds its scene (a key's top bits, or a binary search over <= 256 row bounds).
#include "forest.hpp"

#include <algorithm>

#include "network.hpp"
#include "primitives.hpp"

namespace gpcc {

namespace {

constexpr int TB = 256;
inline unsigned nblk(int64_t n, int per = TB) { return (unsigned)cdiv(n, per); }
inline int bitlen(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

struct FPts { const int32_t *xyz; int64_t first; };   // K + 1 entries: first point of every scene in the concatenation, then the total

// scene of element i: the last entry with first <= i
__device__ __forceinline__ int scene_of_point(const FPts *__restrict__ tab, int K, int64_t i)
{
    int lo = 0, hi = K;   // first[lo] <= i < first[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tab[mid].first <= i) lo = mid; else hi = mid; }
    return lo;
}

// blockIdx.y = scene; the blocks of a row stride over the scene's points
__global__ __launch_bounds__(TB) void k_fbbox(const FPts *__restrict__ tab, int32_t *__restrict__ bbox)
{
    const int q = blockIdx.y;
    const int32_t *xyz = tab[q].xyz;
    const int64_t n = tab[q + 1].first - tab[q].first;
    int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const int v = xyz[3 * i + a]; mn[a] = min(mn[a], v); mx[a] = max(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], d, 64)); mx[a] = max(mx[a], __shfl_xor(mx[a], d, 64)); }
    __shared__ int red[TB / 64][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (int w = 1; w < TB / 64; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[6 * q + threadIdx.x], v);
        else atomicMax(&bbox[6 * q + threadIdx.x], v);
    }
}

// composite key: scene << keybits | Morton code of the scene-local coordinates
__global__ __launch_bounds__(TB) void k_fleaf_keys(const FPts *__restrict__ tab, int K, int64_t total, const int64_t *__restrict__ bias, int keybits, uint64_t *__restrict__ mkey)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= total) return;
    const int q = scene_of_point(tab, K, i);
    const int32_t *p = tab[q].xyz + 3 * (i - tab[q].first);
    const uint64_t m = morton3((uint32_t)((int64_t)p[0] + bias[3 * q]), (uint32_t)((int64_t)p[1] + bias[3 * q + 1]), (uint32_t)((int64_t)p[2] + bias[3 * q + 2]));
    mkey[i] = ((uint64_t)q << keybits) | m;
}

// counts[q][l] (l = 0..21): sorted leaves of scene q whose highest differing Morton triple against the previous leaf of the
// scene is l (a scene's first leaf counts as 21); counts[q][22] = duplicates (octree.hip: k_leaf_levels, per scene)
__global__ __launch_bounds__(TB) void k_fleaf_levels(const uint64_t *__restrict__ mkey, int64_t n, int keybits, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t c[24];
    if (threadIdx.x < 24) c[threadIdx.x] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * TB, i = b0 + threadIdx.x;
    const uint32_t home = (uint32_t)(mkey[min(b0, n - 1)] >> keybits);
    if (i < n) {
        const uint64_t k = mkey[i];
        const uint32_t q = (uint32_t)(k >> keybits);
        int l;
        if (i == 0) l = 21;
        else {
            const uint64_t d = k ^ mkey[i - 1];
            l = d == 0 ? 22 : (d >> keybits) ? 21 : (63 - __clzll((long long)d)) / 3;
        }
        if (q == home) atomicAdd(&c[l], 1u);
        else atomicAdd(&counts[24 * q + l], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 24 && c[threadIdx.x]) atomicAdd(&counts[24 * home + threadIdx.x], c[threadIdx.x]);
}

// One scene's part in the build of leaf-aligned level l (l halvings from the points) from level l - 1.  Indexed by the scene
// id in the key (the caller's order).
struct FLevelScene {
    uint32_t lo0, up0;        // first element of the scene in the leaf-aligned arrays of level l - 1 / level l
    uint32_t row_lo, row_up;  // first row of the scene in the MERGED levels those nodes belong to
    int32_t part;             // the scene has a level l (l <= L + 1: its stored levels and its root level)
    uint32_t ztr;             // z translation in units of the upper level
    int32_t fix;              // the upper level is the scene's last one: its child starts are cstart_fix (no next level)
    uint32_t cstart_fix;
    uint64_t *rkey_up; uint8_t *occ_up; uint32_t *cstart_up;
    uint32_t *parent_lo;      // parent rows of the lower level's nodes (nullptr: the lower level is the points)
};

__global__ __launch_bounds__(TB) void k_flevel_flags(const uint64_t *__restrict__ key, int64_t n, int sh, const FLevelScene *__restrict__ tab, uint32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    const uint32_t q = (uint32_t)(k >> sh);
    flag[i] = (tab[q].part && (i == 0 || (key[i - 1] >> 3) != (k >> 3))) ? 1u : 0u;
}

__global__ __launch_bounds__(TB) void k_flevel_build(const uint64_t *__restrict__ key, int64_t n, int sh, const FLevelScene *__restrict__ tab, const uint32_t *__restrict__ pos,
                                                     uint64_t *__restrict__ key_up)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i], pk = k >> 3;
    const uint32_t q = (uint32_t)(k >> sh);
    const FLevelScene S = tab[q];
    if (!S.part) return;
    const bool head = i == 0 || (key[i - 1] >> 3) != pk;
    const uint32_t p = pos[i] - (head ? 0u : 1u);
    const uint32_t ju = p - S.up0, jl = (uint32_t)i - S.lo0;
    if (S.parent_lo) S.parent_lo[S.row_lo + jl] = S.row_up + ju;
    if (!head) return;
    uint32_t occ = 0;
    for (int j = 0; j < 8 && i + j < n; ++j) {
        const uint64_t kj = key[i + j];
        if ((kj >> 3) != pk) break;
        occ |= 1u << (kj & 7);
    }
    key_up[p] = pk;
    const uint64_t m = pk & ((1ull << (sh - 3)) - 1ull);   // the Morton part (the scene id sits above it)
    const uint32_t row = S.row_up + ju;
    S.rkey_up[row] = rkey3(compact1by2(m), compact1by2(m >> 1), compact1by2(m >> 2) + S.ztr);
    S.cstart_up[row] = S.fix ? S.cstart_fix : S.row_lo + jl;
    S.occ_up[row] = (uint8_t)occ;
}

struct FSentinels { int n; uint32_t *at[MAXLV + 2]; uint32_t v[MAXLV + 2]; };
__global__ void k_fsentinels(FSentinels s) { if ((int)threadIdx.x < s.n) *s.at[threadIdx.x] = s.v[threadIdx.x]; }

// root level: the cells within +-PR of every root node, searched among the root nodes of its scene (at most 63)
__global__ __launch_bounds__(TB) void k_froot_cells(const uint64_t *__restrict__ rkey, int64_t n, const uint32_t *__restrict__ root0, int K, int PR, int32_t *__restrict__ cell)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = K;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)root0[mid] <= i) lo = mid; else hi = mid; }
    const uint32_t a = root0[lo], b = root0[lo + 1];
    const int PW = 2 * PR + 1, NP = PW * PW * PW;
    for (int c = 0; c < NP; ++c) cell[(int64_t)c * n + i] = -1;
    const uint64_t ki = rkey[i];
    for (uint32_t j = a; j < b; ++j) {
        const uint64_t kj = rkey[j];
        const int dx = (int)rk_x(kj) - (int)rk_x(ki), dy = (int)rk_y(kj) - (int)rk_y(ki), dz = (int)rk_z(kj) - (int)rk_z(ki);
        if (dx >= -PR && dx <= PR && dy >= -PR && dy <= PR && dz >= -PR && dz <= PR) cell[(int64_t)((dx + PR) + PW * (dy + PR) + PW * PW * (dz + PR)) * n + i] = (int32_t)j;
    }
}

__device__ __forceinline__ int seg_of_rank(const ForestSeg *__restrict__ seg, int nseg, uint32_t r)
{
    int lo = 0, hi = nseg;   // seg[lo].row0 <= r < seg[hi].row0 (seg[nseg] = the sentinel)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg[mid].row0 <= r) lo = mid; else hi = mid; }
    return lo;
}

// decoder: CDF row slot (lane-interleaved over ALL lanes of the level) and symbol slot (the scene's padded block) of every node
__global__ __launch_bounds__(TB) void k_fcdf_pos(const ForestSeg *__restrict__ seg, int nseg, const uint32_t *__restrict__ m2r, int64_t n, uint32_t nch, uint32_t *__restrict__ pos,
                                                 uint32_t *__restrict__ spos)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    const ForestSeg s = seg[seg_of_rank(seg, nseg, r)];
    const uint32_t rl = r - s.row0;
    pos[i] = s.lane0 + min(rl >> s.llog, s.nlanes - 1u) + (rl & ((1u << s.llog) - 1u)) * nch;
    spos[i] = s.base + rl;
}

// a scene's nodes of one level expand to exactly its share of the next: child start at the scene's first row == its first row below
__global__ void k_fbounds(const uint32_t *__restrict__ cstart, const ForestSeg *__restrict__ seg_par, const ForestSeg *__restrict__ seg_chi, int nchi, uint32_t *__restrict__ bad)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 1 && q < nchi && cstart[seg_par[q].row0] != seg_chi[q].row0) atomicOr(bad, 1u);
}

__global__ __launch_bounds__(TB) void k_fpopc_raster(const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m, int64_t first, int64_t n, uint32_t *__restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r < n) cnt[r] = (uint32_t)__popc((uint32_t)occ[r2m[first + r]]);
}

__global__ __launch_bounds__(TB) void k_fleaves_out(const uint64_t *__restrict__ rkey, const uint8_t *__restrict__ occ, const uint32_t *__restrict__ r2m, const uint32_t *__restrict__ start,
                                                    int64_t first, int64_t n, const ForestLeafScene *__restrict__ sc, int ns)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    const int64_t rr = t >> 3;
    const int oct = (int)(t & 7);
    if (rr >= n) return;
    const uint32_t r = (uint32_t)(first + rr);
    int lo = 0, hi = ns;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sc[mid].rank0 <= r) lo = mid; else hi = mid; }
    const ForestLeafScene S = sc[lo];
    const uint32_t m = r2m[r];
    const uint32_t o = occ[m];
    if (!((o >> oct) & 1u)) return;
    const int64_t idx = (int64_t)start[rr] - (int64_t)start[S.rank0 - first] + __popc(o & ((1u << oct) - 1u));
    if (idx < 0 || idx >= S.cap) return;   // a container whose header undercounts the leaves: the caller compares the counts after its sync
    const uint64_t k = rkey[m];
    S.xyz[3 * idx] = (int32_t)((int64_t)(2 * rk_x(k) + (oct & 1)) - S.bias[0]);
    S.xyz[3 * idx + 1] = (int32_t)((int64_t)(2 * rk_y(k) + ((oct >> 1) & 1)) - S.bias[1]);
    S.xyz[3 * idx + 2] = (int32_t)((int64_t)(2 * rk_z(k) + ((oct >> 2) & 1)) - S.bias[2]);
}

__global__ void k_fleaf_counts(const uint32_t *__restrict__ start, const uint32_t *__restrict__ total, int64_t first, const ForestLeafScene *__restrict__ sc, int ns, uint32_t *__restrict__ counts)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ns) return;
    const uint32_t a = start[sc[q].rank0 - first];
    counts[q] = (q + 1 < ns ? start[sc[q + 1].rank0 - first] : *total) - a;
}

int level_take(gpcc_ctx *ctx, Level *lv, int64_t n, int lvl)
{
    lv->n = n; lv->lvl = lvl;
    TAKE(rkey, uint64_t, n); TAKE(occ, uint8_t, n); TAKE(cstart, uint32_t, n + 1); TAKE(parent, uint32_t, n);
    TAKE(m2r, uint32_t, n); TAKE(r2m, uint32_t, n);
    lv->rkey = rkey; lv->occ = occ; lv->cstart = cstart; lv->parent = parent; lv->m2r = m2r; lv->r2m = r2m;
    return GPCC_OK;
}

}  // namespace

bool forest_place(int L, int64_t zlo, int64_t zhi, int64_t *zcur, int64_t *tz)
{
    int64_t t = *zcur - zlo;
    if (t & 1) ++t;            // even: a base node's parity (its octant under the root level) is its solo one
    *tz = t;
    *zcur = zhi + t + 1;
    // the scene's finest stored level (L - 1 doublings of the base) must keep its coordinates below 2^21
    return L >= 1 && (*zcur << (L - 1)) <= ((int64_t)1 << 21);
}

int forest_root_cells(gpcc_ctx *ctx, hipStream_t st, const Level *root, const uint32_t *root0_dev, int K, int kernel_size, int32_t *cell_root)
{
    (void)ctx;
    const int PR = (kernel_size / 2 + 1) / 2;
    k_froot_cells<<<nblk(root->n), TB, 0, st>>>(root->rkey, root->n, root0_dev, K, PR, cell_root);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int forest_ranks(gpcc_ctx *ctx, hipStream_t st, Forest *F)
{
    for (int d = 0; d < F->L; ++d) {
        Level *lv = &F->T.lv[d];
        if (d == 0) { GP_TRY(rank_level(ctx, st, nullptr, lv, F->hb0)); continue; }
        const Level pv = forest_parent_view(*F, d - 1);
        GP_TRY(rank_level(ctx, st, &pv, lv, F->hb0 + d));
    }
    return GPCC_OK;
}

int forest_upload_segs(gpcc_ctx *ctx, hipStream_t st, Forest *F, const std::vector<ForestSeg> seg[MAXLV], uint8_t *pinned, size_t pinned_bytes)
{
    size_t total = 0;
    for (int d = 0; d < F->L; ++d) total += seg[d].size();
    if (total * sizeof(ForestSeg) > pinned_bytes) return fail(GPCC_ERR_HIP, "internal: scene records exceed their staging area");
    TAKE(dev, ForestSeg, std::max<size_t>(total, 1));
    ForestSeg *h = reinterpret_cast<ForestSeg *>(pinned);
    size_t at = 0;
    for (int d = 0; d < F->L; ++d) {
        F->seg_dev[d] = dev + at;
        for (const ForestSeg &s : seg[d]) h[at++] = s;
    }
    if (total) HIP_TRY(hipMemcpyAsync(dev, h, total * sizeof(ForestSeg), hipMemcpyHostToDevice, st));
    return GPCC_OK;
}

int forest_cdf_pos(hipStream_t st, const ForestSeg *seg_dev, int nseg, const uint32_t *m2r, int64_t n, uint32_t nch_total, uint32_t *pos, uint32_t *spos)
{
    k_fcdf_pos<<<nblk(n), TB, 0, st>>>(seg_dev, nseg, m2r, n, nch_total, pos, spos);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int forest_check_bounds(hipStream_t st, const uint32_t *cstart_par, const ForestSeg *seg_par, const ForestSeg *seg_chi, int nchi, uint32_t *bad_dev)
{
    if (nchi < 2) return GPCC_OK;
    k_fbounds<<<(unsigned)cdiv(nchi, 64), 64, 0, st>>>(cstart_par, seg_par, seg_chi, nchi, bad_dev);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int forest_leaves(gpcc_ctx *ctx, hipStream_t st, const Level *lv, int64_t first_rank, const ForestLeafScene *scenes_dev, int nscenes, uint32_t *counts_dev)
{
    const int64_t n = lv->n - first_rank;
    if (n <= 0 || nscenes <= 0) return GPCC_OK;
    const size_t mk = ctx->arena.mark();
    TAKE(cnt, uint32_t, n + 1);
    k_fpopc_raster<<<nblk(n), TB, 0, st>>>(lv->occ, lv->r2m, first_rank, n, cnt);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_u32(ctx, st, cnt, cnt, n, cnt + n));
    k_fleaves_out<<<nblk(n * 8), TB, 0, st>>>(lv->rkey, lv->occ, lv->r2m, cnt, first_rank, n, scenes_dev, nscenes);
    LAUNCH_CHECK();
    k_fleaf_counts<<<(unsigned)cdiv(nscenes, 64), 64, 0, st>>>(cnt, cnt + n, first_rank, scenes_dev, nscenes, counts_dev);
    LAUNCH_CHECK();
    ctx->arena.rewind(mk);
    return GPCC_OK;
}

// pinned staging the build needs (the caller reserves it ONCE, before anything asynchronous reads from it)
size_t forest_build_pinned(int K) { return (size_t)K * (24 + 96 + 16 + 24) + 16 + (size_t)(MAXLV + 2) * K * sizeof(FLevelScene) + 4 * (size_t)(K + 1) + 256; }

int forest_build(gpcc_ctx *ctx, hipStream_t st, const int32_t *const *xyz, const int64_t *n, int K, int kernel_size, Forest *F, int *bad_scene)
{
    if (K < 1 || K > FOREST_MAX_SCENES) return fail(GPCC_ERR_ARG, "a batch holds 1..%d scenes", FOREST_MAX_SCENES);
    int64_t total = 0;
    for (int q = 0; q < K; ++q) {
        if (!xyz[q] || n[q] <= 0) { if (bad_scene) *bad_scene = q; return fail(GPCC_ERR_ARG, "scene %d: empty point cloud", q); }
        total += n[q];
    }
    if (total >= (int64_t)1 << 31) return FOREST_UNFIT;
    GP_TRY(ctx->hbatch.reserve(forest_build_pinned(K)));
    uint8_t *pin = ctx->hbatch.p;
    int32_t *hbox = reinterpret_cast<int32_t *>(pin);
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(pin + 24 * (size_t)K);
    FPts *htab = reinterpret_cast<FPts *>(pin + (24 + 96) * (size_t)K);
    int64_t *hbias = reinterpret_cast<int64_t *>(pin + (24 + 96 + 16) * (size_t)K + 16);
    FLevelScene *hlev = reinterpret_cast<FLevelScene *>(pin + (((24 + 96 + 16 + 24) * (size_t)K + 16 + 63) & ~(size_t)63));
    uint32_t *hroot0 = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(hlev) + (size_t)(MAXLV + 2) * K * sizeof(FLevelScene));
    // ---- bounding boxes: one sync
    {
        int64_t acc = 0;
        for (int q = 0; q < K; ++q) { htab[q] = FPts{xyz[q], acc}; acc += n[q]; for (int a = 0; a < 3; ++a) { hbox[6 * q + a] = INT32_MAX; hbox[6 * q + 3 + a] = INT32_MIN; } }
        htab[K] = FPts{nullptr, acc};
    }
    TAKE(dtab, FPts, K + 1);
    TAKE(dbox, int32_t, 6 * K);
    HIP_TRY(hipMemcpyAsync(dtab, htab, sizeof(FPts) * (size_t)(K + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dbox, hbox, 24 * (size_t)K, hipMemcpyHostToDevice, st));
    {
        int64_t nmax = 0;
        for (int q = 0; q < K; ++q) nmax = std::max(nmax, n[q]);
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(nmax, TB * 4), std::max(1, 1024 / K)));
        k_fbbox<<<dim3(bx, (unsigned)K), TB, 0, st>>>(dtab, dbox);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(hbox, dbox, 24 * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // ---- per scene: a frame of its own (octree.hip: tree_pick_bias, the branch for clouds anywhere in int32 -- any origin that
    // is a multiple of 2^L gives the reference's tree), and the width of the Morton part of the sort key
    std::vector<ForestScene> sc((size_t)K);
    int hbmax = 1;
    for (int q = 0; q < K; ++q) {
        ForestScene &s = sc[(size_t)q];
        s.user = q; s.npts = n[q];
        int64_t ext = 0;
        for (int a = 0; a < 3; ++a) ext = std::max<int64_t>(ext, (int64_t)hbox[6 * q + 3 + a] - (int64_t)hbox[6 * q + a]);
        const int hbE = std::max(1, bitlen((uint64_t)ext));
        if (hbE > 20) return FOREST_UNFIT;   // (solo: GPCC_ERR_RANGE unless the cloud lies inside (-2^20, 2^20))
        const int64_t A = (int64_t)1 << hbE;
        for (int a = 0; a < 3; ++a) {
            const int64_t m = hbox[6 * q + a];
            s.bias[a] = -((m >= 0 ? m / A : -((-m + A - 1) / A)) * A);
            hbias[3 * q + a] = s.bias[a];
            hbmax = std::max(hbmax, bitlen((uint64_t)((int64_t)hbox[6 * q + 3 + a] + s.bias[a])));
        }
    }
    const int hbkey = hbmax + 1;                  // the root level (L + 1 <= hbkey halvings) still shifts inside the Morton part
    const int keybits = 3 * hbkey, sbits = K > 1 ? bitlen((uint64_t)(K - 1)) : 0;
    if (keybits + sbits > 64) return FOREST_UNFIT;
    // ---- sorted composite keys of all leaves
    TAKE(dbias, int64_t, 3 * K);
    HIP_TRY(hipMemcpyAsync(dbias, hbias, 24 * (size_t)K, hipMemcpyHostToDevice, st));
    TAKE(mk0, uint64_t, total);
    TAKE(mk1, uint64_t, total);
    k_fleaf_keys<<<nblk(total), TB, 0, st>>>(dtab, K, total, dbias, keybits, mk0);
    LAUNCH_CHECK();
    uint64_t *ka = mk0, *kb = mk1;
    GP_TRY(radix_sort_u64(ctx, st, &ka, &kb, nullptr, nullptr, total, keybits + sbits));
    // ---- level sizes of every scene: one sync
    TAKE(dcounts, uint32_t, 24 * K);
    HIP_TRY(hipMemsetAsync(dcounts, 0, 96 * (size_t)K, st));
    k_fleaf_levels<<<nblk(total), TB, 0, st>>>(ka, total, keybits, dcounts);
    LAUNCH_CHECK();
    HIP_TRY(hipMemcpyAsync(hcnt, dcounts, 96 * (size_t)K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<std::vector<int64_t>> nl((size_t)K, std::vector<int64_t>(24, 0));   // nl[q][l]: nodes of scene q after l halvings
    for (int q = 0; q < K; ++q) {
        const uint32_t *hc = hcnt + 24 * (size_t)q;
        if (hc[22]) { if (bad_scene) *bad_scene = q; return fail(GPCC_ERR_DUPLICATE, "scene %d has %u duplicate point(s); the octree occupancy code needs unique voxels", q, hc[22]); }
        int64_t acc = 0;
        for (int l = 21; l >= 1; --l) { acc += hc[l]; nl[(size_t)q][(size_t)l] = acc; }
        nl[(size_t)q][0] = n[q];
        nl[(size_t)q][22] = 1;
        int L = 1;
        while (L < 21 && nl[(size_t)q][(size_t)L] >= 64) ++L;   // pcc_utils.py:83-89: stop at the first level with < 64 nodes
        if (L > 20) return FOREST_UNFIT;
        ForestScene &s = sc[(size_t)q];
        s.L = L;
        for (int d = 0; d < L; ++d) s.n[d] = nl[(size_t)q][(size_t)(L - d)];
        s.nroot = nl[(size_t)q][(size_t)(L + 1)];
    }
    // ---- internal order, slabs, row tables
    std::vector<int> order((size_t)K);
    for (int q = 0; q < K; ++q) order[(size_t)q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sc[(size_t)a].L > sc[(size_t)b].L; });
    std::vector<int> internal_of((size_t)K);
    F->K = K; F->sc.resize((size_t)K);
    for (int qi = 0; qi < K; ++qi) { F->sc[(size_t)qi] = sc[(size_t)order[(size_t)qi]]; internal_of[(size_t)order[(size_t)qi]] = qi; }
    const int Lm = F->sc[0].L;
    F->L = Lm;
    int64_t zcur = 0, xymax = 0;
    std::vector<int64_t> tz((size_t)K);
    for (int qi = 0; qi < K; ++qi) {
        ForestScene &s = F->sc[(size_t)qi];
        const int u = s.user;
        const int64_t zlo = ((int64_t)hbox[6 * u + 2] + s.bias[2]) >> s.L, zhi = ((int64_t)hbox[6 * u + 5] + s.bias[2]) >> s.L;
        if (!forest_place(s.L, zlo, zhi, &zcur, &tz[(size_t)qi])) return FOREST_UNFIT;
        for (int a = 0; a < 2; ++a) xymax = std::max(xymax, ((int64_t)hbox[6 * u + 3 + a] + s.bias[a]) >> s.L);
    }
    F->hb0 = std::max(1, bitlen((uint64_t)std::max(zcur, xymax)));
    F->npts = total;
    for (int d = 0; d <= Lm + 1; ++d) F->Kd[d] = 0;
    for (int qi = 0; qi < K; ++qi) for (int d = 0; d < F->sc[(size_t)qi].L; ++d) F->Kd[d] = qi + 1;
    for (int d = 0; d < Lm; ++d) {
        F->row0[d].assign((size_t)F->Kd[d] + 1, 0u);
        int64_t acc = 0;
        for (int qi = 0; qi < F->Kd[d]; ++qi) { F->row0[d][(size_t)qi] = (uint32_t)acc; acc += F->sc[(size_t)qi].n[d]; }
        if (acc >= (int64_t)1 << 31) return FOREST_UNFIT;
        F->row0[d][(size_t)F->Kd[d]] = (uint32_t)acc;
    }
    F->root0.assign((size_t)K + 1, 0u);
    {
        int64_t acc = 0;
        for (int qi = 0; qi < K; ++qi) { F->root0[(size_t)qi] = (uint32_t)acc; acc += F->sc[(size_t)qi].nroot; }
        F->root0[(size_t)K] = (uint32_t)acc;
    }
    // ---- merged levels
    Tree *T = &F->T;
    T->L = Lm; T->npts = total; T->hb = std::min(21, F->hb0 + Lm); T->leaf_mkey = ka;
    T->bias[0] = T->bias[1] = T->bias[2] = 0;
    for (int d = 0; d < Lm; ++d) GP_TRY(level_take(ctx, &T->lv[d], (int64_t)F->row0[d][(size_t)F->Kd[d]], Lm - d));
    GP_TRY(level_take(ctx, &F->root, (int64_t)F->root0[(size_t)K], Lm + 1));
    // per (leaf-aligned level, scene) records: all levels in one upload
    auto cnt_at = [&](int u, int l) -> int64_t { return l == 0 ? n[u] : nl[(size_t)u][(size_t)l]; };
    std::vector<int64_t> nla((size_t)Lm + 2, 0);
    for (int l = 0; l <= Lm + 1; ++l) {
        std::vector<int64_t> off((size_t)K, 0);   // first element of scene u in leaf-aligned level l (key order = the caller's order)
        int64_t acc = 0;
        for (int u = 0; u < K; ++u) { off[(size_t)u] = acc; if (l <= sc[(size_t)u].L + 1) acc += cnt_at(u, l); }
        nla[(size_t)l] = acc;
        for (int u = 0; u < K; ++u) {
            if (l >= 1) hlev[(size_t)l * K + u].up0 = (uint32_t)off[(size_t)u];
            if (l <= Lm) hlev[(size_t)(l + 1) * K + u].lo0 = (uint32_t)off[(size_t)u];
        }
    }
    for (int l = 1; l <= Lm + 1; ++l)
        for (int u = 0; u < K; ++u) {
            FLevelScene &S = hlev[(size_t)l * K + u];
            const int qi = internal_of[(size_t)u];
            const ForestScene &s = F->sc[(size_t)qi];
            S.part = l <= s.L + 1 ? 1 : 0;
            if (!S.part) { S.row_lo = S.row_up = 0; S.ztr = 0; S.fix = 0; S.cstart_fix = 0; S.rkey_up = nullptr; S.occ_up = nullptr; S.cstart_up = nullptr; S.parent_lo = nullptr; continue; }
            const int d_up = s.L - l, d_lo = d_up + 1;
            Level *up = d_up >= 0 ? &T->lv[d_up] : &F->root;
            S.row_up = d_up >= 0 ? F->row0[d_up][(size_t)qi] : F->root0[(size_t)qi];
            S.row_lo = l == 1 ? S.lo0 : F->row0[d_lo][(size_t)qi];
            S.ztr = (uint32_t)(d_up >= 0 ? tz[(size_t)qi] * ((int64_t)1 << d_up) : tz[(size_t)qi] / 2);   // (tz is even and may be negative)
            S.fix = l == 1 ? 1 : 0;
            S.cstart_fix = l == 1 ? (d_up + 1 < Lm ? (uint32_t)T->lv[d_up + 1].n : (uint32_t)total) : 0u;
            S.rkey_up = up->rkey; S.occ_up = up->occ; S.cstart_up = up->cstart;
            S.parent_lo = l == 1 ? nullptr : T->lv[d_lo].parent;
        }
    for (int qi = 0; qi < K; ++qi) F->sc[(size_t)qi].bias[2] += tz[(size_t)qi] * ((int64_t)1 << F->sc[(size_t)qi].L);
    TAKE(dlev, FLevelScene, (size_t)(Lm + 2) * K);
    HIP_TRY(hipMemcpyAsync(dlev, hlev, sizeof(FLevelScene) * (size_t)(Lm + 2) * K, hipMemcpyHostToDevice, st));
    const uint64_t *key_lo = ka;
    for (int l = 1; l <= Lm + 1; ++l) {
        const int64_t n_lo = nla[(size_t)l - 1], n_up = nla[(size_t)l];
        const int sh = keybits - 3 * (l - 1);   // 3 <= sh <= 63: the scene id of a level-(l - 1) key sits above its Morton part
        TAKE(key_up, uint64_t, std::max<int64_t>(n_up, 1));
        const size_t mk = ctx->arena.mark();
        TAKE(flag, uint32_t, n_lo);
        k_flevel_flags<<<nblk(n_lo), TB, 0, st>>>(key_lo, n_lo, sh, dlev + (size_t)l * K, flag);
        LAUNCH_CHECK();
        GP_TRY(exclusive_scan_u32(ctx, st, flag, flag, n_lo, nullptr));
        k_flevel_build<<<nblk(n_lo), TB, 0, st>>>(key_lo, n_lo, sh, dlev + (size_t)l * K, flag, key_up);
        LAUNCH_CHECK();
        ctx->arena.rewind(mk);
        key_lo = key_up;
    }
    {
        FSentinels s = {};
        for (int d = 0; d < Lm; ++d) { s.at[s.n] = T->lv[d].cstart + T->lv[d].n; s.v[s.n] = d + 1 < Lm ? (uint32_t)T->lv[d + 1].n : (uint32_t)total; ++s.n; }
        s.at[s.n] = F->root.cstart + F->root.n; s.v[s.n] = (uint32_t)T->lv[0].n; ++s.n;
        k_fsentinels<<<1, 64, 0, st>>>(s);
        LAUNCH_CHECK();
    }
    // ---- the root's cell map
    const int NPc = cell_map_entries(kernel_size);
    TAKE(cr, int32_t, (int64_t)NPc * F->root.n);
    F->cell_root = cr;
    TAKE(droot0, uint32_t, K + 1);
    for (int qi = 0; qi <= K; ++qi) hroot0[qi] = F->root0[(size_t)qi];
    HIP_TRY(hipMemcpyAsync(droot0, hroot0, 4 * (size_t)(K + 1), hipMemcpyHostToDevice, st));
    GP_TRY(forest_root_cells(ctx, st, &F->root, droot0, K, kernel_size, cr));
    return GPCC_OK;
}

}  // namespace gpcc
