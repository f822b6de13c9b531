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
    int paired;                 // blocks taller than 64 rows MAY be paired: every (block, offset) run padded to an EVEN number of tiles (the
                                // pair-step conv loop) -- decided block by block from the run lengths (k_block_sum -> pflag)
    uint8_t *pflag;             // [pool blocks] 1 = this block's list is paired
    uint32_t blk0;              // pool id of the level's first block
    // pass 1
    int32_t *cell_c;            // [NP][nc] the level's own cell map, nullable (nobody below needs it)
    uint32_t *per_block;        // [pool blocks] tiles of each block (blocks of at most 64 rows; taller ones: k_block_sum)
    unsigned long long *pairs;  // [64] += (row, neighbour) pairs of the level, spread over 64 counters by wave index
    uint8_t *cnt_oq;            // [pool blocks][K][4] pairs per (offset, 64-row chunk) of the blocks taller than 64 rows
    // pass 2
    const uint32_t *first;      // [pool blocks + 1] first tile of each block
    int32_t *tj; uint8_t *tr; uint32_t *toc;
    // DENSE count pass (small levels, fused.hip: the pair plan): the whole neighbour map and, per (offset, row), the rank of the
    // offset among the row's present offsets; rowcnt[row] = its present offsets
    int32_t *nbr; uint16_t *rk; uint32_t *rowcnt;
};

// How a 64-lane wave maps to blocks.  H = 16 / 32: the wave takes 64 consecutive rows = 4 / 2 whole blocks side by side as
// lane segments.  H <= 64 otherwise: one block.  H > 64: one 64-row CHUNK q of a block (nq = ceil(H / 64) waves per block).
struct WaveMap {
    bool multi; int nq, q; uint32_t blk; bool blk_live;
    int64_t r0, rend;          // my rows [r0, rend) of the level
    int ls; uint64_t segmask, below;
};
__device__ __forceinline__ WaveMap wave_map(int H, int64_t nc, uint32_t blk0, int lane, uint32_t bid)
{
    WaveMap m;
    m.multi = H == 16 || H == 32;
    m.nq = H > 64 ? (H + 63) >> 6 : 1;
    const int Hs = m.multi ? H : 64;
    const int seg = lane / Hs;
    m.ls = lane - seg * Hs;
    m.segmask = Hs == 64 ? ~0ull : (((1ull << Hs) - 1ull) << (seg * Hs));
    m.below = m.segmask & (lane == 0 ? 0ull : (~0ull >> (64 - lane)));
    if (m.multi) {
        m.q = 0;
        m.r0 = (int64_t)bid * 64; m.rend = min(nc, m.r0 + 64);
        m.blk = blk0 + (uint32_t)(bid * (64 / Hs) + seg);
        m.blk_live = m.r0 + (int64_t)seg * Hs < nc;
    } else {
        const uint32_t b = bid / (uint32_t)m.nq;
        m.q = (int)(bid - b * (uint32_t)m.nq);
        const int64_t b0 = (int64_t)b * H, bend = min(nc, b0 + H);
        m.r0 = min(bend, b0 + 64 * (int64_t)m.q); m.rend = min(bend, m.r0 + 64);
        m.blk = blk0 + b;
        m.blk_live = true;
    }
    return m;
}

// tiles of a run of `tot` pairs: ceil(tot / 16), rounded up to an even number in a paired pool (the padding tile is empty)
__host__ __device__ __forceinline__ uint32_t run_tiles(uint32_t tot, int paired) { return paired ? ((tot + 31u) >> 5) << 1 : (tot + 15u) >> 4; }

// One kernel offset of a wave whose block(s) fit the wave (multi mode or a block of at most 64 rows): the rows with a
// neighbour are packed in row order behind the block's running tile counter.
template <bool FILL>
__device__ __forceinline__ void pack_local(const LevelTilesArgs &a, const WaveMap &m, int o, int j, uint32_t &t, uint32_t &npairs)
{
    const uint64_t b = __ballot(j >= 0);
    if (b == 0) return;                 // wave-uniform
    npairs += (uint32_t)__popcll(b);
    const uint32_t cnt = (uint32_t)__popcll(b & m.segmask);
    const uint32_t nt = (cnt + 15u) >> 4;           // (blocks that fit a wave are never paired)
    if (FILL) {
        if (j >= 0) {
            const uint32_t p = (uint32_t)__popcll(b & m.below);
            const size_t at = (size_t)(t + (p >> 4)) * 16 + (p & 15);
            a.tj[at] = j;
            a.tr[at] = (uint8_t)(m.ls + 1);                      // LDS slot of the output row inside its block: row + 1
        }
        if ((uint32_t)m.ls < nt * 16u - cnt) {                   // padding entries: neighbour row 0 into the dummy slot
            const uint32_t p = cnt + (uint32_t)m.ls;
            const size_t at = (size_t)(t + (p >> 4)) * 16 + (p & 15);
            a.tj[at] = 0;
            a.tr[at] = 0;
        }
        if ((uint32_t)m.ls < nt) a.toc[t + (uint32_t)m.ls] = (uint32_t)o | ((cnt > 16u * (uint32_t)m.ls ? min(16u, cnt - 16u * (uint32_t)m.ls) : 0u) << 16);
    }
    t += nt;
}

// Tables of a block taller than 64 rows for the fill pass (from the count pass's per-chunk counts): per offset the pairs of
// the whole block, the block-relative first tile, and the pairs of the chunks in front of mine.
struct ChunkTables { uint16_t *tot, *obase, *cbase; };
__device__ __forceinline__ void chunk_tables(const uint8_t *__restrict__ cnt_blk, int K, int q, int lane, ChunkTables T, int paired)
{
    uint32_t carry = 0;
    for (int o0 = 0; o0 < K; o0 += 64) {
        const int o = o0 + lane;
        uint32_t c4 = 0;
        if (o < K) c4 = *reinterpret_cast<const uint32_t *>(cnt_blk + 4 * (size_t)o);
        const uint32_t c[4] = {c4 & 255u, (c4 >> 8) & 255u, (c4 >> 16) & 255u, c4 >> 24};
        const uint32_t tot = c[0] + c[1] + c[2] + c[3];
        const uint32_t cb = (q > 0 ? c[0] : 0u) + (q > 1 ? c[1] : 0u) + (q > 2 ? c[2] : 0u);
        const uint32_t nt = run_tiles(tot, paired);
        uint32_t inc = nt;                                        // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += v;
        }
        if (o < K) { T.tot[o] = (uint16_t)tot; T.obase[o] = (uint16_t)(carry + inc - nt); T.cbase[o] = (uint16_t)cb; }
        carry += (uint32_t)__shfl((int)inc, 63, 64);
    }
}

// TALL: the wave is one 64-row chunk of a block taller than 64 rows; else its block(s) fit the wave.
// The k^2 neighbours of a (dy, dx) plane are computed as one batch (independent LDS reads and integer chains in flight
// together -- a wave alone on its SIMD has nothing else to hide their latency behind), then compacted one by one.
template <int KS, bool FILL, bool TALL, bool DENSE = false>
__device__ __forceinline__ void chunk_tiles(const LevelTilesArgs &a, const uint32_t bid)
{
    static_assert(!DENSE || (!FILL && !TALL), "the dense map is a count-pass output of blocks that fit a wave");
    constexpr int r = KS / 2, K = KS * KS * KS, PR = (r + 1) / 2, PW = 2 * PR + 1, NP = PW * PW * PW;
    __shared__ uint32_t cst[64 * NP];     // child start of every staged cell
    __shared__ uint8_t coc[64 * NP];      // its occupancy (0 = the cell does not exist)
    __shared__ uint16_t tab[3 * K + 2];   // fill pass, tall blocks: tot | obase | cbase
    const int lane = threadIdx.x;
#ifdef TILES_TIMING
    const long long tc0 = clock64();
#endif
    const WaveMap m = wave_map(a.H, a.nc, a.blk0, lane, bid);
    if (m.r0 >= m.rend) return;           // a chunk behind the end of its (last, short) block: its counts were zeroed by the host
    const ChunkTables T = {tab, tab + K + 1, tab + 2 * K + 2};
    if (FILL && TALL) chunk_tables(a.cnt_oq + (size_t)m.blk * K * 4, K, m.q, lane, T, a.pflag ? (int)a.pflag[m.blk] : 0);
    // ---- stage the cells of my rows' parents: lane = parent, the cell map is read in coalesced rows (one per cell), the
    // gathers of a cell's child start and occupancy hit neighbouring nodes for neighbouring parents
    const int64_t i = min(m.r0 + lane, m.rend - 1);
    const bool live = m.r0 + lane < m.rend;
    const uint32_t my_parent = a.parent_c[i];
    const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.parent_c[m.r0]);
    const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.parent_c[m.rend - 1]);
    const uint32_t npar = min(p_hi - p_lo, 63u) + 1u;     // a header that lies about a level cannot overrun the staging area
    {
        // three dependent round trips in all (parents -> cell rows -> the cells' child start / occupancy): every load of a
        // round is in flight before the first is used.  NP > 32 (kernel size 7): in batches, registers are finite.
        const uint32_t p = p_lo + (uint32_t)lane;
        const bool pl = (uint32_t)lane < npar && p < (uint32_t)a.np;
        constexpr int CBATCH = NP <= 32 ? NP : 25;
        for (int c0 = 0; c0 < NP; c0 += CBATCH) {
            int32_t pn[CBATCH];
#pragma unroll
            for (int u = 0; u < CBATCH; ++u) pn[u] = (pl && c0 + u < NP) ? a.cell_p[(int64_t)(c0 + u) * a.np + p] : -1;
            uint32_t sv[CBATCH], ov[CBATCH];
#pragma unroll
            for (int u = 0; u < CBATCH; ++u) {
                // unconditional gathers (row 0 is always there): no branch between the loads.  Clamped from above too: the cell map
                // of a level decoded from a corrupt stream is built from whatever occupancy came out of the coder
                const int32_t q = min(max(pn[u], 0), (int32_t)min(a.np - 1, (int64_t)INT32_MAX));
                sv[u] = a.cstart_p[q]; ov[u] = a.occ_p[q];
            }
#pragma unroll
            for (int u = 0; u < CBATCH; ++u)
                if (c0 + u < NP) {   // every slot is written: rows of an inconsistent level (a corrupt stream) may point at a parent outside the run, and what they
                                     // read there must be the same in the count pass and in the fill pass -- not whatever the LDS held
                    const bool have = pl && pn[u] >= 0;
                    cst[lane * NP + c0 + u] = have ? sv[u] : 0u; coc[lane * NP + c0 + u] = have ? (uint8_t)ov[u] : (uint8_t)0;
                }
        }
    }
    const uint64_t kc = a.rkey_c[i];
    const int cx = (int)(rk_x(kc) & 1), cy = (int)(rk_y(kc) & 1), cz = (int)(rk_z(kc) & 1);
    const uint32_t mine = min(my_parent - p_lo, 63u) * (uint32_t)NP;
    __syncthreads();
#ifdef TILES_TIMING
    const long long tc1 = clock64();
#endif
    uint32_t t = (FILL && m.blk_live) ? a.first[m.blk] : 0u;   // running tile of my block (blocks that fit the wave)
    const uint32_t t0 = t;
    uint32_t npairs = 0;
    uint32_t dcount = 0;   // DENSE: present offsets of my row so far
    const int nc32 = (int)min(a.nc, (int64_t)INT32_MAX);
    for (int dz = -r; dz <= r; ++dz) {
        const int tz = cz + dz;
        const int cqz = PW * PW * ((tz >> 1) + PR) + PR, tqz = (tz & 1) << 2;
        int jv[KS * KS];
        {
            // both LDS reads of every offset of the plane first, unconditionally (the compiler otherwise reads the child start
            // under a branch on the occupancy bit and waits for each of the 2 k^2 reads in turn), then the integer part
            uint32_t ocv[KS * KS], scv[KS * KS];
            int tqv[KS * KS];
#pragma unroll
            for (int iy = 0; iy < KS; ++iy) {
                const int ty = cy + iy - r;
                const int cqy = PW * ((ty >> 1) + PR) + cqz, tqy = ((ty & 1) << 1) | tqz;
#pragma unroll
                for (int ix = 0; ix < KS; ++ix) {
                    const int tx = cx + ix - r;
                    const int cq = (tx >> 1) + cqy;                  // floor halves: the parent cell
                    tqv[iy * KS + ix] = (tx & 1) | tqy;
                    ocv[iy * KS + ix] = coc[mine + cq];
                    scv[iy * KS + ix] = cst[mine + cq];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KS * KS; ++u) {
                const uint32_t bit = (ocv[u] >> tqv[u]) & 1u;
                const int32_t idx = (int32_t)(scv[u] + (uint32_t)__popc(ocv[u] & ((1u << tqv[u]) - 1u)));
                // res >= nc: only when a container header understates the level (reported at the decoder's final sync)
                jv[u] = (bit != 0u && idx < nc32 && live) ? idx : -1;
            }
        }
        if (!FILL && a.cell_c && live && dz >= -PR && dz <= PR) {
#pragma unroll
            for (int iy = r - PR; iy <= r + PR; ++iy)
#pragma unroll
                for (int ix = r - PR; ix <= r + PR; ++ix)
                    a.cell_c[(int64_t)((ix - r + PR) + PW * (iy - r + PR) + PW * PW * (dz + PR)) * a.nc + i] = jv[iy * KS + ix];
        }
        const int ob = (dz + r) * KS * KS;
        if constexpr (DENSE) {
            if (live) {
#pragma unroll
                for (int u = 0; u < KS * KS; ++u) {
                    a.nbr[(int64_t)(ob + u) * a.nc + i] = jv[u];
                    a.rk[(int64_t)(ob + u) * a.nc + i] = (uint16_t)dcount;
                    dcount += jv[u] >= 0 ? 1u : 0u;
                }
            }
            continue;
        }
        if (!TALL) {
#pragma unroll
            for (int u = 0; u < KS * KS; ++u) pack_local<FILL>(a, m, ob + u, jv[u], t, npairs);
            continue;
        }
        // a chunk of a tall block: its pairs go behind those of the chunks in front of it
        if (!FILL) {
            // the plane's counts collect in lanes 0 .. k^2 - 1 of one register (one v_writelane per offset), one store per plane
            int acc = 0;
#pragma unroll
            for (int u = 0; u < KS * KS; ++u)
            {
                const int cnt = __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(jv[u] >= 0)));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(cnt), "n"(u));
            }
            if (lane < KS * KS) { a.cnt_oq[((size_t)m.blk * K + ob + lane) * 4 + m.q] = (uint8_t)acc; npairs += (uint32_t)acc; }
            continue;
        }
        // (pads and offset words of the tiles are written by k_tile_words from the same counts)
        uint32_t obase[KS * KS], cbase[KS * KS];
#pragma unroll
        for (int u = 0; u < KS * KS; ++u) { obase[u] = T.obase[ob + u]; cbase[u] = T.cbase[ob + u]; }
#pragma unroll
        for (int u = 0; u < KS * KS; ++u) {
            const int j = jv[u];
            const uint64_t b = __ballot(j >= 0);
            if (j >= 0) {
                const uint32_t p = cbase[u] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                const uint32_t at = (t0 + obase[u]) * 16u + p;       // (tile, slot) = (p >> 4, p & 15) behind the offset's first tile
                a.tj[at] = j;
                a.tr[at] = (uint8_t)(m.q * 64 + lane + 1);
            }
        }
    }
#ifdef TILES_TIMING
    const long long tc2 = clock64();
#endif
    if constexpr (DENSE) {
        if (live) a.rowcnt[i] = dcount;
        return;
    }
    if (!FILL) {
        if (!TALL && m.ls == 0 && m.blk_live) a.per_block[m.blk] = t;
        // pairs: tall chunks hold partial sums in lanes 0 .. k^2 - 1; one atomic per wave on one of 64 counters (a single
        // hot word serialises ~12 ns per wave: 180 us for the finest level of a 1 M-point cloud)
        if (TALL) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) npairs += (uint32_t)__shfl_xor((int)npairs, d, 64);
        }
        if (lane == 0 && a.pairs && npairs) atomicAdd(a.pairs + (bid & 63u), (unsigned long long)npairs);
    }
#ifdef TILES_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    const long long tc3 = clock64();
    if (lane == 0 && (bid == 0 || bid == gridDim.x / 2))
        printf("[tiles] fill %d tall %d grid %u wave %u: stage %lld loop %lld tail %lld cycles\n", (int)FILL, (int)TALL, gridDim.x, bid, tc1 - tc0, tc2 - tc1, tc3 - tc2);
#endif
}

template <int KS, bool FILL, bool TALL>
__global__ __launch_bounds__(64) void k_chunk_tiles(LevelTilesArgs a) { chunk_tiles<KS, FILL, TALL>(a, blockIdx.x); }
template <int KS>
__global__ __launch_bounds__(64) void k_chunk_dense(LevelTilesArgs a) { chunk_tiles<KS, false, false, true>(a, blockIdx.x); }

// The fill pass of every level of a pool in ONE launch (an encode builds the tile lists of its whole tree; the passes of
// different levels are independent once the counts are scanned, and ten of its fifteen levels are launch-latency-bound).
struct SetTilesArgs {
    int nlv;
    uint32_t g0[MAXLV + 1];     // first workgroup of each level in the grid
    LevelTilesArgs common;      // H, first, tj / tr / toc, cnt_oq
    struct Lv { const uint64_t *rkey_c; const uint32_t *parent_c; const int32_t *cell_p; const uint8_t *occ_p; const uint32_t *cstart_p; int64_t nc, np; uint32_t blk0; } lv[MAXLV];
};
template <int KS, bool TALL>
__global__ __launch_bounds__(64) void k_fill_tiles_set(SetTilesArgs S)
{
    int l = 0;
    for (int q = 1; q < S.nlv; ++q) l = blockIdx.x >= S.g0[q] ? q : l;
    LevelTilesArgs a = S.common;
    a.rkey_c = S.lv[l].rkey_c; a.parent_c = S.lv[l].parent_c; a.nc = S.lv[l].nc; a.cell_p = S.lv[l].cell_p; a.np = S.lv[l].np;
    a.occ_p = S.lv[l].occ_p; a.cstart_p = S.lv[l].cstart_p; a.blk0 = S.lv[l].blk0;
    chunk_tiles<KS, true, TALL>(a, blockIdx.x - S.g0[l]);
}

// tiles of the blocks taller than 64 rows from the per-chunk counts: a wave per block, lanes over the offsets
// Pairing pays when runs are long: a pair step of two full tiles costs ~1350 cycles against 2 x 750, the half step that ends
// an odd run ~800 against 750, and the empty tile it reads is header traffic.  Measured per level (1 M-point cloud, TFLOP/s
// paired / unpaired): 25 pairs per row 78.3 / 71.9, 22.6: 76.9 / 75.0, 13.5: 64.6 / 66.5, 6.8: 50.6 / 55.4.  So the choice is
// made per block: paired when its runs average at least 1.5 tiles.
__global__ __launch_bounds__(256) void k_block_sum(const uint8_t *__restrict__ cnt_oq, uint32_t blk0, int nblk, int K, uint32_t *__restrict__ per_block, int paired,
                                                   uint8_t *__restrict__ pflag)   // paired: 0, or 100 x the least average run length (tiles) of a paired block
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const uint32_t *c = reinterpret_cast<const uint32_t *>(cnt_oq + (size_t)(blk0 + b) * K * 4);
    uint32_t tu = 0, tp = 0, runs = 0;
    for (int o = lane; o < K; o += 64) {
        const uint32_t c4 = c[o];
        const uint32_t tot = (c4 & 255u) + ((c4 >> 8) & 255u) + ((c4 >> 16) & 255u) + (c4 >> 24);
        tu += run_tiles(tot, 0); tp += run_tiles(tot, 1); runs += tot ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { tu += (uint32_t)__shfl_xor((int)tu, d, 64); tp += (uint32_t)__shfl_xor((int)tp, d, 64); runs += (uint32_t)__shfl_xor((int)runs, d, 64); }
    const bool pair = paired && 100u * tu >= (uint32_t)paired * runs;
    if (lane == 0) { per_block[blk0 + b] = pair ? tp : tu; if (pflag) pflag[blk0 + b] = pair ? 1 : 0; }
}

// The base level (< 64 nodes, no parent): neighbours by search over the level's raster keys.
template <bool FILL>
__global__ __launch_bounds__(64) void k_base_tiles(LevelTilesArgs a, int k)
{
    __shared__ uint64_t keys[64];
    __shared__ uint8_t nb[343 * 64];   // [offset][lane]: neighbour row + 1 of the lane's row, 0 = none
    const int lane = threadIdx.x;
    const int n = (int)a.nc, r = k / 2, PR = (r + 1) / 2, PW = 2 * PR + 1, K = k * k * k;
    keys[lane] = lane < n ? a.rkey_c[lane] : ~0ull;
    __syncthreads();
    const WaveMap m = wave_map(min(a.H, 64), a.nc, a.blk0, lane, blockIdx.x);   // n < 64: a block taller than 64 rows is the whole level
    const int me = (int)m.r0 + lane;                                  // my row
    const bool live = m.r0 + lane < m.rend;
    const uint64_t ki = keys[min(me, n - 1)];
    // every node of the level against my row once (n < 64 steps) instead of a search per offset (k^3 n steps: 58 us)
    for (int q = 0; q < K; ++q) nb[q * 64 + lane] = 0;
    for (int jj = 0; jj < n; ++jj) {
        const uint64_t kj = keys[jj];
        const int dx = (int)rk_x(kj) - (int)rk_x(ki), dy = (int)rk_y(kj) - (int)rk_y(ki), dz = (int)rk_z(kj) - (int)rk_z(ki);
        if (live && dx >= -r && dx <= r && dy >= -r && dy <= r && dz >= -r && dz <= r) nb[((dx + r) + k * (dy + r) + k * k * (dz + r)) * 64 + lane] = (uint8_t)(jj + 1);
    }
    uint32_t t = (FILL && m.blk_live) ? a.first[m.blk] : 0u, npairs = 0;
    int o = 0;
    for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx, ++o) {
                const int res = (int)nb[o * 64 + lane] - 1;
                if (!FILL && a.cell_c && live && dx >= -PR && dx <= PR && dy >= -PR && dy <= PR && dz >= -PR && dz <= PR)
                    a.cell_c[(int64_t)((dx + PR) + PW * (dy + PR) + PW * PW * (dz + PR)) * n + me] = res;
                pack_local<FILL>(a, m, o, res, t, npairs);
            }
    if (!FILL) {
        if (m.ls == 0 && m.blk_live) a.per_block[m.blk] = t;
        if (lane == 0 && a.pairs && npairs) atomicAdd(a.pairs, (unsigned long long)npairs);
    }
}

// pads and offset words of the tiles of blocks taller than 64 rows: one thread per (block, offset)
__global__ __launch_bounds__(256) void k_tile_words(const uint8_t *__restrict__ cnt_oq, const uint32_t *__restrict__ first, uint32_t blk0, int nblk, int K,
                                                    int32_t *__restrict__ tj, uint8_t *__restrict__ tr, uint32_t *__restrict__ toc, const uint8_t *__restrict__ pflag)
{
    // a wave per block: lanes scan the offsets' tile counts, then every lane finishes its own offsets
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const uint32_t *c = reinterpret_cast<const uint32_t *>(cnt_oq + (size_t)(blk0 + b) * K * 4);
    uint32_t carry = first[blk0 + b];
    const int paired = pflag ? (int)pflag[blk0 + b] : 0;
    for (int o0 = 0; o0 < K; o0 += 64) {
        const int o = o0 + lane;
        const uint32_t c4 = o < K ? c[o] : 0u;
        const uint32_t tot = (c4 & 255u) + ((c4 >> 8) & 255u) + ((c4 >> 16) & 255u) + (c4 >> 24);
        const uint32_t nt = run_tiles(tot, paired);
        uint32_t inc = nt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += v;
        }
        const uint32_t tb = carry + inc - nt;
        for (uint32_t i = 0; i < nt; ++i) toc[tb + i] = (uint32_t)o | ((tot > 16u * i ? min(16u, tot - 16u * i) : 0u) << 16);   // a paired run's padding tile: 0 entries
        for (uint32_t p = tot; p < nt * 16u; ++p) { tj[(size_t)tb * 16 + p] = 0; tr[(size_t)tb * 16 + p] = 0; }
        carry += (uint32_t)__shfl((int)inc, 63, 64);
    }
}

template <int KS, bool FILL>
int launch_level(hipStream_t st, const LevelTilesArgs &a, bool tail)
{
    const int H = a.H;
    const bool multi = H == 16 || H == 32;
    const int nq = H > 64 ? (H + 63) / 64 : 1;
    const unsigned grid = multi ? (unsigned)cdiv(a.nc, 64) : (unsigned)(cdiv(a.nc, H) * nq);
    if (nq > 1) k_chunk_tiles<KS, FILL, true><<<grid, 64, 0, st>>>(a);
    else k_chunk_tiles<KS, FILL, false><<<grid, 64, 0, st>>>(a);
    LAUNCH_CHECK();
    if (nq > 1 && tail) {
        const int nblk = (int)cdiv(a.nc, H);
        if (!FILL) k_block_sum<<<(unsigned)cdiv(nblk, 4), 256, 0, st>>>(a.cnt_oq, a.blk0, nblk, KS * KS * KS, a.per_block, a.paired, a.pflag);
        else k_tile_words<<<(unsigned)cdiv(nblk, 4), 256, 0, st>>>(a.cnt_oq, a.first, a.blk0, nblk, KS * KS * KS, a.tj, a.tr, a.toc, a.pflag);
        LAUNCH_CHECK();
    }
    return GPCC_OK;
}

// tail: also the per-level pass over the blocks taller than 64 rows (k_block_sum / k_tile_words); a pool of several levels
// runs that once over all its blocks
template <bool FILL>
int run_level(hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int k, LevelTilesArgs a, bool tail = true)
{
    a.rkey_c = chi->rkey; a.parent_c = chi->parent; a.nc = chi->n;
    if (a.H < 16 || a.H > CONV_R_MAX) return fail(GPCC_ERR_ARG, "internal: block height %d", a.H);
    if (!par) {
        if (chi->n >= 64) return fail(GPCC_ERR_ARG, "internal: base level with %lld nodes", (long long)chi->n);
        const int Hb = std::min(a.H, 64);
        k_base_tiles<FILL><<<(unsigned)cdiv(chi->n, (Hb == 16 || Hb == 32) ? 64 : Hb), 64, 0, st>>>(a, k);
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    a.cell_p = cell_par; a.np = par->n; a.occ_p = par->occ; a.cstart_p = par->cstart;
    switch (k) {
    case 3: return launch_level<3, FILL>(st, a, tail);
    case 5: return launch_level<5, FILL>(st, a, tail);
    case 7: return launch_level<7, FILL>(st, a, tail);
    default: return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
    }
}

__global__ __launch_bounds__(64) void k_fold_pairs(const unsigned long long *__restrict__ spread, int nlv, unsigned long long *__restrict__ pairs)
{
    for (int l = 0; l < nlv; ++l) {
        unsigned long long v = spread[l * 64 + threadIdx.x];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (threadIdx.x == 0) pairs[l] += v;
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
    static const bool pair_on = dev_env_int("GAUSPCC_CONV_PAIR", 1) != 0;
    static const int pair_min = 150;   // x 0.01 tiles per run (sweep 100 .. 250: profiles/r03_*, DESIGN history round 3)
    pool->paired = (H > 64 && R == CONV_R_MAX && pair_on) ? std::max(pair_min, 1) : 0;   // tall blocks of the wave-serial class may be paired (block by block)
    pool->pflag = nullptr;
    // Three arrays that start zeroed -- the pair flags (the base level's block, built by k_base_tiles, is never paired), the per-chunk counts of
    // the tall blocks (blocks of 2 or 3 chunks leave the other columns untouched) and the spread pair counters -- carved back to back and
    // zeroed by ONE memset (until round 6: one runtime fill each, per tile pool = per decoded level)
    unsigned long long *spread = nullptr;
    {
        const size_t b_sp = pairs_dev ? 8 * (size_t)nlv * 64 : 0;
        const size_t b_cq = H > 64 ? (((size_t)nblk * K * 4 + 15) & ~(size_t)15) : 0;
        const size_t b_pf = pool->paired ? (((size_t)nblk + 15) & ~(size_t)15) : 0;
        if (b_sp + b_cq + b_pf) {
            TAKE(z, unsigned long long, (b_sp + b_cq + b_pf) / 8);
            uint8_t *zb = reinterpret_cast<uint8_t *>(z);
            HIP_TRY(hipMemsetAsync(zb, 0, b_sp + b_cq + b_pf, st));
            if (b_sp) spread = z;
            if (b_cq) a.cnt_oq = zb + b_sp;
            if (b_pf) pool->pflag = zb + b_sp + b_cq;
        }
    }
    a.H = H; a.paired = pool->paired; a.pflag = pool->pflag; a.per_block = first;
    // a pool of several levels (the encoder's whole tree; only its first level may be a base level without a parent): the
    // count passes stay one launch per level (a level reads the cell map its parent's pass wrote), everything that is
    // independent across levels -- the sums of the tall blocks, the fill pass, the tile words -- is one launch over the pool
    const bool batch = nlv > 1;
    for (int l = 1; l < nlv; ++l) if (!lv[l].par) return fail(GPCC_ERR_ARG, "internal: level %d of a tile pool has no parent", l);
    const int lp = lv[0].par ? 0 : 1;   // first level with a parent
    for (int l = 0; l < nlv; ++l) {
        a.blk0 = pool->lv_blk0[l]; a.cell_c = lv[l].cell_own; a.pairs = spread ? spread + (size_t)l * 64 : nullptr;
        GP_TRY(run_level<false>(st, lv[l].par, lv[l].cell_par, lv[l].lv, k, a, !batch));
    }
    if (batch && H > 64 && lp < nlv) {
        const uint32_t b1 = pool->lv_blk0[lp];
        k_block_sum<<<(unsigned)cdiv(nblk - b1, 4), 256, 0, st>>>(a.cnt_oq, b1, (int)(nblk - b1), K, a.per_block, a.paired, a.pflag);
        LAUNCH_CHECK();
    }
    if (spread) { k_fold_pairs<<<1, 64, 0, st>>>(spread, nlv, pairs_dev); LAUNCH_CHECK(); }
    GP_TRY(exclusive_scan_u32(ctx, st, first, first, nblk, first + nblk));
    // 16-row blocks hold at most one tile per kernel offset: the list is sized by that bound and built without the host ever
    // learning its length (the small levels of a decode are launch-bound; every sync removed lets the host run ahead).
    // Taller blocks are sized exactly: one sync.
    {
        const int NPc = cell_map_entries(k);
        double b = 0.0;
        for (int l = 0; l < nlv; ++l)
            b += (double)lv[l].lv->n * (12 + (lv[l].cell_own ? 4.0 * NPc : 0.0)) + (lv[l].par ? (double)lv[l].par->n * 9.0 * NPc : 0.0);
        pool->alg_bytes = b;
    }
    int64_t cap;
    int64_t rows_total = 0;
    for (int l = 0; l < nlv; ++l) rows_total += lv[l].lv->n;
    // (blocks of 32 / 64 rows of a small level -- the cooperative kernel's taller classes -- hold at most H / 16 tiles per offset)
    if (H <= 16) cap = nblk * K + CONV_HDR_PAD;
    else if (H <= 64 && H % 16 == 0 && conv_is_coop(rows_total, R)) cap = nblk * K * (H / 16) + CONV_HDR_PAD;
    else {
        // (A single level COULD be sized by a bound as well -- a (block, offset) run holds at most ceil(H / 16) tiles: ~660 B per node at 255 rows --
        // so that the host never waits for the count pass.  Measured in round 4: dec_ms 26.93 with the bound, 26.95 with the five syncs -- they
        // fall on the second stream while the first runs a parent trunk.  The exact size keeps the workspace smaller; the bound left the tree.)
        {
            uint32_t total = 0;
            HIP_TRY(hipMemcpyAsync(&total, first + nblk, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            cap = (int64_t)total + CONV_HDR_PAD;   // the conv kernel streams whole header batches: zeroed padding (row 0, offset 0)
            pool->alg_bytes += 84.0 * total;
        }
    }
    if (cap >= (int64_t)1 << 28) return fail(GPCC_ERR_ARG, "too many conv tiles (%lld): clouds beyond ~10^8 points need 64-bit tile addressing", (long long)cap);
    TAKE(tj, int32_t, cap * 16);
    TAKE(tr, uint8_t, cap * 16);
    TAKE(toc, uint32_t, cap);
    pool->tj = tj; pool->tr = tr; pool->toc = toc;
    k_pad_tiles<<<1, 64, 0, st>>>(first + nblk, tj, reinterpret_cast<uint32_t *>(tr), toc);
    LAUNCH_CHECK();
    a.first = first; a.tj = tj; a.tr = tr; a.toc = toc; a.cell_c = nullptr; a.pairs = nullptr;
    if (!batch) {
        a.blk0 = pool->lv_blk0[0];
        GP_TRY(run_level<true>(st, lv[0].par, lv[0].cell_par, lv[0].lv, k, a));
        return GPCC_OK;
    }
    if (lp == 1) { a.blk0 = pool->lv_blk0[0]; GP_TRY(run_level<true>(st, nullptr, nullptr, lv[0].lv, k, a)); }
    {
        SetTilesArgs S = {};
        S.common = a;
        const bool multi = H == 16 || H == 32;
        const int nq = H > 64 ? (H + 63) / 64 : 1;
        uint32_t g = 0;
        for (int l = lp; l < nlv; ++l) {
            auto &d = S.lv[l - lp];
            d.rkey_c = lv[l].lv->rkey; d.parent_c = lv[l].lv->parent; d.nc = lv[l].lv->n; d.cell_p = lv[l].cell_par; d.np = lv[l].par->n;
            d.occ_p = lv[l].par->occ; d.cstart_p = lv[l].par->cstart; d.blk0 = pool->lv_blk0[l];
            S.g0[l - lp] = g;
            g += multi ? (uint32_t)cdiv(d.nc, 64) : (uint32_t)(cdiv(d.nc, H) * nq);
        }
        S.nlv = nlv - lp; S.g0[S.nlv] = g;
        switch (k) {
        case 3: if (nq > 1) k_fill_tiles_set<3, true><<<g, 64, 0, st>>>(S); else k_fill_tiles_set<3, false><<<g, 64, 0, st>>>(S); break;
        case 5: if (nq > 1) k_fill_tiles_set<5, true><<<g, 64, 0, st>>>(S); else k_fill_tiles_set<5, false><<<g, 64, 0, st>>>(S); break;
        case 7: if (nq > 1) k_fill_tiles_set<7, true><<<g, 64, 0, st>>>(S); else k_fill_tiles_set<7, false><<<g, 64, 0, st>>>(S); break;
        default: return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
        }
        LAUNCH_CHECK();
        if (nq > 1) {
            const uint32_t b1 = pool->lv_blk0[lp];
            k_tile_words<<<(unsigned)cdiv(nblk - b1, 4), 256, 0, st>>>(a.cnt_oq, a.first, b1, (int)(nblk - b1), K, tj, tr, toc, a.pflag);
            LAUNCH_CHECK();
        }
    }
    return GPCC_OK;
}

// Small levels (fused.hip): the level's whole neighbour map nbr[K][n] (-1 = absent), rk[K][n] = rank of the offset among the
// row's present offsets, rowcnt[n] -- and, as the count pass of tiles_build does, the level's own cell map for the level below.
int tiles_dense_map(hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int32_t *cell_own, int k, int32_t *nbr, uint16_t *rk, uint32_t *rowcnt)
{
    if (!par) return fail(GPCC_ERR_ARG, "internal: dense map of a level without a parent");
    LevelTilesArgs a = {};
    a.H = 64; a.blk0 = 0; a.cell_c = cell_own; a.nbr = nbr; a.rk = rk; a.rowcnt = rowcnt;
    a.rkey_c = chi->rkey; a.parent_c = chi->parent; a.nc = chi->n;
    a.cell_p = cell_par; a.np = par->n; a.occ_p = par->occ; a.cstart_p = par->cstart;
    const unsigned grid = (unsigned)cdiv(chi->n, 64);
    switch (k) {
    case 3: k_chunk_dense<3><<<grid, 64, 0, st>>>(a); break;
    case 5: k_chunk_dense<5><<<grid, 64, 0, st>>>(a); break;
    case 7: k_chunk_dense<7><<<grid, 64, 0, st>>>(a); break;
    default: return fail(GPCC_ERR_ARG, "kernel_size must be 3, 5 or 7");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

int tiles_view(gpcc_ctx *ctx, hipStream_t st, const TilePool &pool, int l0, int l1, const int64_t *row_base, ConvTiles *T)
{
    if (l0 < 0 || l1 > pool.nlv || l0 >= l1) return fail(GPCC_ERR_ARG, "internal: level range [%d, %d)", l0, l1);
    T->tj = pool.tj; T->tr = pool.tr; T->toc = pool.toc; T->first = pool.first;
    T->R = pool.R; T->H = pool.H; T->K = pool.K; T->paired = pool.paired; T->pflag = pool.pflag;
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
