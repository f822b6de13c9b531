// tiles.hip -- the kernel map of spnn.Conv3d (network_ue_4stage_conv.py:17-62) as the conv kernels consume it: per block of
// H Morton-ordered rows and per kernel offset, the (output row, neighbour row) pairs packed 16 to a tile.
//
// There is no dense [k^3][n] neighbour map any more (round 1 wrote 500 B per node and read it back twice).  A level keeps
// only its CELL MAP: for every node the indices of the (2 PR + 1)^3 nodes around it (PR = (k/2 + 1) / 2: 27 entries for k = 3
// and 5), which is all the next level needs.  The k^3 neighbours of a child follow from its parent's cells by index
// arithmetic: the voxel at offset delta from child c lies in the cell floor((c + delta) / 2) - parent(c), its octant bit in
// that cell's occupancy says whether it exists, the cell's child start + popcount of the lower octant bits where.  One wave
// takes one block: it stages the cells of the block's parents in LDS (per 64-row chunk the parents are a contiguous run of at
// most 64 nodes), then walks the k^3 offsets with up to four rows per lane, ballots the rows that have the neighbour and
// either counts (pass 1: tiles per block, pairs, the level's own cell map) or writes the compacted tiles (pass 2).
// Both passes recompute the neighbours -- ~20 integer instructions per (row, offset) -- instead of storing them.
// HBM-bound by design, in practice latency / issue-bound: ~250 B of traffic per node and pass.
#include "network.hpp"
#include "octree.hpp"
#include "primitives.hpp"

namespace gpcc {

namespace {

struct LevelTilesArgs {
    // the level whose tiles are built (Morton order)
    const uint64_t *rkey_c; const uint32_t *parent_c; int64_t nc;
    // its parent level: cell map [NP][np] (local indices, -1 = absent), occupancy, child starts
    const int32_t *cell_p; int64_t np; const uint8_t *occ_p; const uint32_t *cstart_p;
    int H;                      // rows per block
    uint32_t blk0;              // global id of the level's first block
    // pass 1
    int32_t *cell_c;            // [NP][nc] the level's own cell map, nullable (nobody below needs it)
    uint32_t *per_block;        // [global blocks] tiles of each block
    unsigned long long *pairs;  // += (row, neighbour) pairs of the level
    // pass 2
    const uint32_t *first;      // [global blocks + 1] first tile of each block
    int32_t *tj; uint8_t *tr; uint32_t *toc;
};

// Compaction of one kernel offset: Q rows per lane (row q * 64 + lane of the wave's span), the rows with a neighbour are
// packed in row order.  A wave covers one block of up to 64 Q rows, or -- blocks of 16 / 32 rows -- 64 / H blocks side by
// side as lane segments (segmask selects a lane's own segment; ls = lane inside the segment).
template <int Q, bool FILL>
struct Packer {
    uint64_t segmask, below;   // lanes of my segment; those of them below me
    int ls;                    // lane index inside the segment
    uint32_t t;                // running tile index of my block (FILL: global; else count)
    uint32_t npairs = 0;       // pairs seen by the whole wave (uniform)
    const LevelTilesArgs *a;

    __device__ __forceinline__ void step(int o, const int (&j)[Q])
    {
        uint64_t b[Q];
        uint32_t cnt = 0, all = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            b[q] = __ballot(j[q] >= 0);
            cnt += (uint32_t)__popcll(b[q] & segmask);
            all += (uint32_t)__popcll(b[q]);
        }
        if (all == 0) return;           // wave-uniform
        npairs += all;
        const uint32_t nt = (cnt + 15u) >> 4;
        if (FILL) {
            uint32_t base = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (j[q] >= 0) {
                    const uint32_t p = base + (uint32_t)__popcll(b[q] & below);
                    const size_t at = (size_t)(t + (p >> 4)) * 16 + (p & 15);
                    a->tj[at] = j[q];
                    a->tr[at] = (uint8_t)(q * 64 + ls + 1);     // LDS slot of the output row inside its block: row + 1
                }
                base += (uint32_t)__popcll(b[q] & segmask);
            }
            if ((uint32_t)ls < nt * 16u - cnt) {                // padding entries: neighbour row 0 into the dummy slot
                const uint32_t p = cnt + (uint32_t)ls;
                const size_t at = (size_t)(t + (p >> 4)) * 16 + (p & 15);
                a->tj[at] = 0;
                a->tr[at] = 0;
            }
            if ((uint32_t)ls < nt) a->toc[t + (uint32_t)ls] = (uint32_t)o | (min(16u, cnt - 16u * (uint32_t)ls) << 16);
        }
        t += nt;
    }
};

template <int Q, bool FILL>
__device__ __forceinline__ void packer_init(Packer<Q, FILL> &P, const LevelTilesArgs &a, int lane, uint32_t *blk_out, bool *blk_live, int64_t w0)
{
    const int H = a.H;
    const bool multi = Q == 1 && (H == 16 || H == 32);   // 4 or 2 blocks side by side in one wave
    const int Hs = multi ? H : 64;                        // segment width
    const int seg = lane / Hs;
    P.ls = lane - seg * Hs;
    P.segmask = Hs == 64 ? ~0ull : (((1ull << Hs) - 1ull) << (seg * Hs));
    P.below = P.segmask & (lane == 0 ? 0ull : (~0ull >> (64 - lane)));
    P.a = &a;
    const int nseg = 64 / Hs;
    const uint32_t blk = a.blk0 + (multi ? (uint32_t)(blockIdx.x * nseg + seg) : (uint32_t)blockIdx.x);
    *blk_out = blk;
    *blk_live = multi ? (w0 + (int64_t)seg * Hs < a.nc) : true;
    P.t = (FILL && *blk_live) ? a.first[blk] : 0u;
}

template <int KS, int Q, bool FILL>
__global__ __launch_bounds__(64) void k_level_tiles(LevelTilesArgs a)
{
    constexpr int r = KS / 2, PR = (r + 1) / 2, PW = 2 * PR + 1, NP = PW * PW * PW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *cst = reinterpret_cast<uint32_t *>(smem);     // [Q][64 NP] child start of every staged cell
    uint8_t *coc = smem + (size_t)Q * 64 * NP * 4;          // [Q][64 NP] occupancy (0 = the cell does not exist)
    const int lane = threadIdx.x;
    const int64_t span = (a.H == 16 || a.H == 32) ? 64 : a.H;    // rows of this wave: one block, or 64 rows of 16- / 32-row blocks
    const int64_t w0 = (int64_t)blockIdx.x * span, wend = min(a.nc, w0 + span);
    int cx[Q], cy[Q], cz[Q];
    uint32_t mine[Q];
    bool live[Q];
    int64_t row[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t c0 = w0 + 64 * q;
        live[q] = false; mine[q] = 0; cx[q] = cy[q] = cz[q] = 0; row[q] = 0;
        if (c0 >= wend) continue;                           // wave-uniform
        const int64_t cend = min(wend, c0 + 64);
        const int64_t i = min(c0 + lane, cend - 1);
        const uint32_t my_parent = a.parent_c[i];
        const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.parent_c[c0]);
        const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.parent_c[cend - 1]);
        const uint32_t npar = min(p_hi - p_lo, 63u) + 1u;    // a header that lies about a level cannot overrun the staging area
        uint32_t *cs = cst + (size_t)q * 64 * NP;
        uint8_t *co = coc + (size_t)q * 64 * NP;
        // lane = parent: the cell map is read in coalesced rows (one per cell), the gathers of a cell's child start and
        // occupancy hit neighbouring nodes for neighbouring parents; CB cells' loads are in flight together
        {
            const uint32_t p = p_lo + (uint32_t)lane;
            const bool pl = (uint32_t)lane < npar && p < (uint32_t)a.np;
            constexpr int CBATCH = 9;
            for (int c0c = 0; c0c < NP; c0c += CBATCH) {
                int32_t pn[CBATCH];
#pragma unroll
                for (int u = 0; u < CBATCH; ++u) pn[u] = (pl && c0c + u < NP) ? a.cell_p[(int64_t)(c0c + u) * a.np + p] : -1;
                uint32_t sv[CBATCH], ov[CBATCH];
#pragma unroll
                for (int u = 0; u < CBATCH; ++u) {
                    sv[u] = 0; ov[u] = 0;
                    if (pn[u] >= 0) { sv[u] = a.cstart_p[pn[u]]; ov[u] = a.occ_p[pn[u]]; }
                }
#pragma unroll
                for (int u = 0; u < CBATCH; ++u)
                    if (pl && c0c + u < NP) { cs[lane * NP + c0c + u] = sv[u]; co[lane * NP + c0c + u] = (uint8_t)ov[u]; }
            }
        }
        live[q] = c0 + lane < cend;
        row[q] = i;
        const uint64_t kc = a.rkey_c[i];
        cx[q] = (int)(rk_x(kc) & 1); cy[q] = (int)(rk_y(kc) & 1); cz[q] = (int)(rk_z(kc) & 1);
        mine[q] = min(my_parent - p_lo, 63u) * (uint32_t)NP;
    }
    __syncthreads();
    Packer<Q, FILL> P;
    uint32_t blk; bool blk_live;
    packer_init<Q, FILL>(P, a, lane, &blk, &blk_live, w0);
    int o = 0;
    for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy)
#pragma unroll
            for (int dx = -r; dx <= r; ++dx, ++o) {
                int j[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int tx = cx[q] + dx, ty = cy[q] + dy, tz = cz[q] + dz;
                    const int cq = ((tx >> 1) + PR) + PW * ((ty >> 1) + PR) + PW * PW * ((tz >> 1) + PR);   // floor halves: the parent cell
                    const int tq = (tx & 1) | ((ty & 1) << 1) | ((tz & 1) << 2);
                    const uint32_t oc = coc[(size_t)q * 64 * NP + mine[q] + cq];
                    const uint32_t s = cst[(size_t)q * 64 * NP + mine[q] + cq];
                    int32_t res = (oc >> tq) & 1u ? (int32_t)(s + (uint32_t)__popc(oc & ((1u << tq) - 1u))) : -1;
                    if ((int64_t)res >= a.nc) res = -1;      // only when a container header understates the level (reported at the decoder's final sync)
                    j[q] = live[q] ? res : -1;
                }
                if (!FILL && a.cell_c && dx >= -PR && dx <= PR && dy >= -PR && dy <= PR && dz >= -PR && dz <= PR) {
                    const int c = (dx + PR) + PW * (dy + PR) + PW * PW * (dz + PR);
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        if (live[q]) a.cell_c[(int64_t)c * a.nc + row[q]] = j[q];
                }
                P.step(o, j);
            }
    if (!FILL) {
        if (P.ls == 0 && blk_live) a.per_block[blk] = P.t;
        if (lane == 0 && a.pairs && P.npairs) atomicAdd(a.pairs, (unsigned long long)P.npairs);
    }
}

// The base level (< 64 nodes, no parent): neighbours by search over the level's raster keys.
template <bool FILL>
__global__ __launch_bounds__(64) void k_base_tiles(LevelTilesArgs a, int k)
{
    __shared__ uint64_t keys[64];
    const int lane = threadIdx.x;
    const int n = (int)a.nc, r = k / 2, PR = (r + 1) / 2, PW = 2 * PR + 1;
    keys[lane] = lane < n ? a.rkey_c[lane] : ~0ull;
    __syncthreads();
    // one wave per block, or per 64 rows of 16- / 32-row blocks (as in k_level_tiles)
    const int span = (a.H == 16 || a.H == 32) ? 64 : a.H;
    const int w0 = (int)blockIdx.x * span;
    const int me = w0 + lane;                                  // my row
    const bool live = lane < span && me < n;
    const uint64_t ki = keys[min(me, n - 1)];
    Packer<1, FILL> P;
    uint32_t blk; bool blk_live;
    packer_init<1, FILL>(P, a, lane, &blk, &blk_live, w0);
    int o = 0;
    for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx, ++o) {
                const int tx = (int)rk_x(ki) + dx, ty = (int)rk_y(ki) + dy, tz = (int)rk_z(ki) + dz;
                int res = -1;
                if (live && tx >= 0 && ty >= 0 && tz >= 0) {
                    const uint64_t tgt = rkey3((uint32_t)tx, (uint32_t)ty, (uint32_t)tz);
                    for (int jj = 0; jj < n; ++jj)
                        if (keys[jj] == tgt) res = jj;
                }
                if (!FILL && a.cell_c && live && dx >= -PR && dx <= PR && dy >= -PR && dy <= PR && dz >= -PR && dz <= PR)
                    a.cell_c[(int64_t)((dx + PR) + PW * (dy + PR) + PW * PW * (dz + PR)) * n + me] = res;
                const int j[1] = {res};
                P.step(o, j);
            }
    if (!FILL) {
        if (P.ls == 0 && blk_live) a.per_block[blk] = P.t;
        if (lane == 0 && a.pairs && P.npairs) atomicAdd(a.pairs, (unsigned long long)P.npairs);
    }
}

template <int KS, bool FILL>
int launch_level(hipStream_t st, const LevelTilesArgs &a)
{
    constexpr int r = KS / 2, PR = (r + 1) / 2, PW = 2 * PR + 1, NP = PW * PW * PW;
    const int Q = a.H <= 64 ? 1 : (a.H + 63) / 64;
    const int64_t span = (a.H == 16 || a.H == 32) ? 64 : a.H;
    const unsigned grid = (unsigned)cdiv(a.nc, span);
    const size_t lds = (size_t)Q * 64 * NP * 5;
    if (lds > 64 * 1024) return fail(GPCC_ERR_ARG, "internal: block height %d with kernel size %d", a.H, KS);
    switch (Q) {
    case 1: k_level_tiles<KS, 1, FILL><<<grid, 64, lds, st>>>(a); break;
    case 2: k_level_tiles<KS, 2, FILL><<<grid, 64, lds, st>>>(a); break;
    case 3: k_level_tiles<KS, 3, FILL><<<grid, 64, lds, st>>>(a); break;
    case 4: k_level_tiles<KS, 4, FILL><<<grid, 64, lds, st>>>(a); break;
    default: return fail(GPCC_ERR_ARG, "internal: block height %d", a.H);
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

template <bool FILL>
int run_level(hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int k, LevelTilesArgs a)
{
    a.rkey_c = chi->rkey; a.parent_c = chi->parent; a.nc = chi->n;
    if (!par) {
        if (chi->n > 64) return fail(GPCC_ERR_ARG, "internal: base level with %lld nodes", (long long)chi->n);
        k_base_tiles<FILL><<<(unsigned)cdiv(chi->n, (a.H == 16 || a.H == 32) ? 64 : a.H), 64, 0, st>>>(a, k);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    if (a.H < 16 || a.H > CONV_R_MAX) return fail(GPCC_ERR_ARG, "internal: block height %d", a.H);
    a.cell_p = cell_par; a.np = par->n; a.occ_p = par->occ; a.cstart_p = par->cstart;
    switch (k) {
    case 3: return launch_level<3, FILL>(st, a);
    case 5: return launch_level<5, FILL>(st, a);
    case 7: return launch_level<7, FILL>(st, a);
    default: return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
    }
}

__global__ __launch_bounds__(256) void k_order_keys(const uint32_t *__restrict__ first, uint32_t b0, int nblk, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    key[b] = 0xFFFFull - (uint64_t)min(first[b0 + b + 1] - first[b0 + b], 0xFFFFu);  // ascending sort of this = descending tile count (16 bits: 2 radix passes)
    idx[b] = b0 + (uint32_t)b;
}

// CONV_HDR_PAD zeroed tiles behind the list, whose length lives on the device
__global__ __launch_bounds__(64) void k_pad_tiles(const uint32_t *__restrict__ total, int32_t *__restrict__ tj, uint32_t *__restrict__ tr4, uint32_t *__restrict__ toc)
{
    const uint32_t t = *total;
    for (int i = threadIdx.x; i < CONV_HDR_PAD * 16; i += 64) tj[(size_t)t * 16 + i] = 0;
    for (int i = threadIdx.x; i < CONV_HDR_PAD * 4; i += 64) tr4[(size_t)t * 4 + i] = 0;
    for (int i = threadIdx.x; i < CONV_HDR_PAD; i += 64) toc[t + i] = 0;
}

}  // namespace

int cell_map_entries(int k) { const int PW = 2 * ((k / 2 + 1) / 2) + 1; return PW * PW * PW; }

int tiles_build(gpcc_ctx *ctx, hipStream_t st, const TileLevel *lv, int nlv, int k, int R, int H, TilePool *pool, unsigned long long *pairs_dev)
{
    if (nlv < 1 || nlv > MAXLV) return fail(GPCC_ERR_ARG, "internal: %d levels", nlv);
    const int K = k * k * k;
    if (k == 7 && H > 64) return fail(GPCC_ERR_ARG, "internal: kernel size 7 needs blocks of at most 64 rows (125 staged cells per parent)");
    pool->R = R; pool->H = H; pool->K = K; pool->nlv = nlv;
    int64_t nblk = 0;
    for (int l = 0; l < nlv; ++l) {
        if (lv[l].lv->n >= (int64_t)1 << 31) return fail(GPCC_ERR_ARG, "level too large");
        pool->lv_blk0[l] = (uint32_t)nblk;
        pool->lv_rows[l] = (uint32_t)lv[l].lv->n;
        nblk += cdiv(lv[l].lv->n, H);
    }
    if (nblk >= (int64_t)1 << 31) return fail(GPCC_ERR_ARG, "too many blocks");
    pool->lv_blk0[nlv] = (uint32_t)nblk;
    pool->nblk = nblk;
    TAKE(first, uint32_t, nblk + 1);
    pool->first = first;
    LevelTilesArgs a = {};
    a.H = H; a.per_block = first;
    for (int l = 0; l < nlv; ++l) {
        a.blk0 = pool->lv_blk0[l]; a.cell_c = lv[l].cell_own; a.pairs = pairs_dev ? pairs_dev + l : nullptr;
        GP_TRY(run_level<false>(st, lv[l].par, lv[l].cell_par, lv[l].lv, k, a));
    }
    GP_TRY(exclusive_scan_u32(ctx, st, first, first, nblk, first + nblk));
    // 16-row blocks hold at most one tile per kernel offset: the list is sized by that bound and built without the host ever
    // learning its length (the small levels of a decode are launch-bound; every sync removed lets the host run ahead).
    // Taller blocks are sized exactly: one sync.
    int64_t cap;
    if (H <= 16) cap = nblk * K + CONV_HDR_PAD;
    else {
        uint32_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, first + nblk, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        cap = (int64_t)total + CONV_HDR_PAD;   // the conv kernel streams whole header batches: zeroed padding (row 0, offset 0)
    }
    TAKE(tj, int32_t, cap * 16);
    TAKE(tr, uint8_t, cap * 16);
    TAKE(toc, uint32_t, cap);
    pool->tj = tj; pool->tr = tr; pool->toc = toc;
    k_pad_tiles<<<1, 64, 0, st>>>(first + nblk, tj, reinterpret_cast<uint32_t *>(tr), toc);
    LAUNCH_CHECK();
    a.first = first; a.tj = tj; a.tr = tr; a.toc = toc; a.cell_c = nullptr; a.pairs = nullptr;
    for (int l = 0; l < nlv; ++l) {
        a.blk0 = pool->lv_blk0[l];
        GP_TRY(run_level<true>(st, lv[l].par, lv[l].cell_par, lv[l].lv, k, a));
    }
    return GPCC_OK;
}

int tiles_view(gpcc_ctx *ctx, hipStream_t st, const TilePool &pool, int l0, int l1, const int64_t *row_base, ConvTiles *T)
{
    if (l0 < 0 || l1 > pool.nlv || l0 >= l1) return fail(GPCC_ERR_ARG, "internal: level range [%d, %d)", l0, l1);
    T->tj = pool.tj; T->tr = pool.tr; T->toc = pool.toc; T->first = pool.first;
    T->R = pool.R; T->H = pool.H; T->K = pool.K;
    T->nlv = l1 - l0;
    for (int l = l0; l <= l1; ++l) T->lv_blk0[l - l0] = pool.lv_blk0[l];
    for (int l = l0; l < l1; ++l) { T->lv_rows[l - l0] = pool.lv_rows[l]; T->lv_row0[l - l0] = (uint32_t)row_base[l - l0]; }
    const uint32_t b0 = pool.lv_blk0[l0];
    const int64_t nblk = (int64_t)pool.lv_blk0[l1] - b0;
    T->nblk = nblk;
    // dispatch order: longest blocks first, so the tail of the launch is made of short blocks (LPT scheduling).  When all
    // blocks are resident at once the order cannot matter: identity, no sort.
    TAKE(order, uint32_t, nblk);
    {
        const size_t mk = ctx->arena.mark();
        TAKE(ka, uint64_t, nblk); TAKE(kb, uint64_t, nblk); TAKE(vb, uint32_t, nblk);
        k_order_keys<<<(unsigned)cdiv(nblk, 256), 256, 0, st>>>(pool.first, b0, (int)nblk, ka, order);
        LAUNCH_CHECK();
        if (nblk > 2048) {
            uint64_t *k0 = ka, *k1 = kb; uint32_t *v0 = order, *v1 = vb;
            GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, nblk, 16));   // keys are 0xFFFF - min(tiles, 0xFFFF)
            if (v0 != order) HIP_TRY(hipMemcpyAsync(order, v0, 4 * (size_t)nblk, hipMemcpyDeviceToDevice, st));
        }
        ctx->arena.rewind(mk);
    }
    T->order = order;
    return GPCC_OK;
}

}  // namespace gpcc
