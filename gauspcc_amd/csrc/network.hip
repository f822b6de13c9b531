// network.hip -- GausPcgc context network kernels for gfx950.
//
// k_sparse_conv   submanifold 3-D convolution, C = 32, exact fp32 on v_mfma_f32_16x16x4_f32.
//                 A submanifold conv on an octree level is ~90 % empty (7..25 of the 125 taps exist),
//                 so the work is the list of (output row, neighbour row) PAIRS, not rows x offsets.
//                 Each wave owns up to 255 consecutive Morton-ordered output rows (spatially compact) and
//                 keeps their fp32 sums in wave-private LDS.  Per kernel offset the block's pairs were
//                 compacted (k_conv_tiles, once per level) into tiles of 16 rows; per tile the wave gathers
//                 the 16 neighbour rows, loads the pre-swizzled 4 KiB weight fragment of the offset, issues
//                 16 MFMAs from a zero accumulator and adds the 16 x 32 products to the rows' sums in LDS.
//                 Offsets are visited in ascending order: acc = acc + (fma chain over k from 0) per offset,
//                 exactly the oracle's order, so results are bit-exact no matter how rows are packed into
//                 tiles.  The asm tile loop (conv_loop_gfx950.inc) computes the product transposed; see
//                 tools/gen_conv_loop.py and HISTORY.md section 4 for why.
//                 Bound: fp32 MFMA (2*32*32 flop per pair); gathers come from L2/MALL.
// k_head          Linear-ReLU-Linear-softmax-cumsum-integerise, one node per lane, weights through
//                 the scalar cache.  Negligible next to the convolutions.
#include <algorithm>
#include <utility>
#include <vector>

#include "network_dev.hpp"
#include "octree.hpp"
#include "primitives.hpp"
#include "rangecoder.hpp"
#ifndef CONV_LOOP_INC
#define CONV_LOOP_INC "conv_loop_gfx950.inc"   // generated: tools/gen_conv_loop.py (tools/build_variants.sh substitutes ablations)
#endif
#include CONV_LOOP_INC
#include "conv_loop2_gfx950.inc"               // generated: tools/gen_conv_loop2.py (the pair-step loop of paired tile lists)
// (Two more loops were built, proven bit-exact and measured slower everywhere in round 4 -- the pair step on v_mfma_f32_32x32x2_f32, -7 %, and
// half-channel waves, -6 .. -35 % -- and a 4-row tail packing on v_mfma_f32_4x4x1_16B_f32 failed its gate in round 6: their code left the tree with
// the kernel's freeze, their numbers are profiles/r04_conv_quad_ablation.txt and profiles/r06_conv_tail_gate.txt.)
static_assert(CONV_LOOP2_ROW_BYTES == CONV_LDS_ROW_BYTES, "both asm loops address the running sums at one LDS row pitch");

namespace gpcc {

#ifndef CONV_SC_WAVES_N
#define CONV_SC_WAVES_N 4
#endif
constexpr int SC_WAVES = CONV_SC_WAVES_N;   // waves per workgroup of k_sparse_conv (every wave works alone on its own block)
// LDS floats per wave: slot 0 = the dummy row (padding entries, tiles past the end of a list), slots 1..R = the block's rows,
// then the tile-header ring: 32 slots + 16 mirror slots (copies of slots 0..15, so that the asm loop reaches every slot
// the eight steps of an iteration read -- up to 15 tiles ahead -- by immediate offsets): neighbour rows 48 x 16 dwords |
// output slots 48 x 4 dwords (16 bytes a tile) | kernel offsets 48 dwords
constexpr int HDR_R = 768, HDR_O = 960, HDR_DWORDS = 1008;
// A row's 32 sums take CONV_LDS_ROW_BYTES (conv_loop_gfx950.inc) = 144 bytes of LDS, not 128: a tile's 16 rows are mostly
// consecutive slots, and at a 128-byte pitch lane e's 16 bytes of row e fall on the same four banks for every e -- an
// 8-way conflict on each 8-lane group of ds_write_b128 (64 LDS cycles an instruction instead of 8), 4-5-way on
// ds_read_b128.  One 16-byte pad per row rotates consecutive rows over the bank quads.
constexpr int ROWF = CONV_LDS_ROW_BYTES / 4;
static_assert(ROWF >= 32 && ROWF % 4 == 0, "row pitch: whole 16-byte units");
__host__ __device__ constexpr int conv_lds_wave_floats(int R) { return (R + 1) * ROWF + HDR_DWORDS; }

// ------------------------------------------------------------------ block policy
static int conv_rows_forced()
{
    static const int forced = [] {
        const int f = dev_env_int("GAUSPCC_CONV_R", 0);
        return (f != 0 && f != 16 && f != 32 && f != 64 && f != 96 && f != 128 && f != 255) ? 0 : f;
    }();
    return forced;
}
static int64_t conv_tall_min()
{
    static const int64_t tall_min = 64 * 256 + 1;   // 245 64-row workgroups: 22.2 us against 23.1 for 977 16-row waves at 15.6 k nodes
    return tall_min;
}

// Levels the cooperative kernel takes (up to 16 k nodes; one workgroup per block, one workgroup per CU): the block height
// is the smallest of 16 / 32 / 64 rows that runs the level as ONE round of at most 256 workgroups -- 263 16-row blocks
// take two rounds (38 us per convolution on the 4 k-node level of the 1 M-point cloud), 132 32-row blocks one.  Taller blocks have
// better-filled tiles (a 16-row block of a k = 5 level has ~100 tiles of 2-3 pairs) but more of them per block, so the
// height only grows when the rounds of workgroups shrink.  (k = 7: the headers of a 64-row block do not fit the LDS; 16.)
int conv_coop_rows(int64_t n, int k)
{
    static const int tall = dev_env_int("GAUSPCC_COOP_TALL", 1);
    if (!tall || k > 5) return 16;
    return n <= 16 * 256 ? 16 : n <= 32 * 256 ? 32 : 64;
}
bool conv_is_coop(int64_t n, int R)
{
    static const int use_coop = dev_env_int("GAUSPCC_CONV_COOP", 1) != 0;
    if (!use_coop) return false;
    if (R == 16) return true;
    return (R == 32 || R == 64) && !conv_rows_forced() && n < conv_tall_min() && n <= 64 * 256;
}

int conv_pick_rows(int64_t n, int k)
{
    if (conv_rows_forced()) return conv_rows_forced();
    // every level the cooperative kernel does not take (it wins up to ~16 k nodes): the 255-row class at one wave per SIMD (4 x 35 KiB of LDS per CU),
    // with the block height set by conv_pick_height -- up to 1024 blocks run as ONE round of equal blocks, one per SIMD
    // (measured against 2-3 waves per SIMD on 32..128-row blocks: 34 k nodes 40 vs 56 us, 92 k 75 vs 88, 251 k 185 vs 199,
    // 540 k 322 vs 377).  The asm loop addresses rows with 32-bit offsets: n < 2^25.
    if (n >= conv_tall_min() && n < ((int64_t)1 << 25)) return 255;
    if (n >= 192 * 1024) return 128;
    if (n >= 96 * 1024) return 64;
    if (n >= 24 * 1024) return 32;
    return conv_coop_rows(n, k);
}

// A launch runs nblk single-wave blocks on a fixed number of wave slots (LDS and registers allow 1 wave per SIMD for the
// 255-row class, 2 below); with a handful of blocks per slot the last round is mostly idle slots
// (3.06 blocks per slot take as long as 4).  The rows are therefore cut into slots x k equal blocks, k the smallest
// number of rounds the class allows -- slightly lower blocks, every round full.
int conv_pick_height(int64_t n, int R)
{
    static const int balance = dev_env_int("GAUSPCC_CONV_BALANCE", 2);
    if (!balance || R <= 16 || conv_is_coop(n, R)) return R;
    const int64_t slots = 1024 * (int64_t)(R >= 255 ? 1 : 2);
    const int64_t k = cdiv(n, slots * R);
    // measured (MI355X, 1 M-point cloud): with the longest-first order a last round that is mostly full costs nothing
    // (3.6 blocks per slot at 255 rows beat 4.0 at 231), a nearly empty one does (2.07 -> 3.0 rounds of 176 rows: -19 %)
    if (balance == 2 && (double)(k * slots * R - n) < 0.2 * (double)(k * slots * R)) return R;
    const int64_t H = cdiv(n, slots * k);
    return (int)std::min<int64_t>(R, std::max<int64_t>(H, 16));
}

// ------------------------------------------------------------------ convolution
#ifdef CONV_TIMING
// developer build (tools/conv_timing.sh): shader-clock stamps around the phases of every wave, summed per launch
__device__ unsigned long long g_conv_timing[16];
#define CT_STAMP(v) const long long v = clock64()
#else
#define CT_STAMP(v)
#endif

// ASM = true: the tile loop is the hand-scheduled gfx950 instruction stream of conv_loop_gfx950.inc (row offsets are
// 32-bit there: n < 2^25).  ASM = false: the same loop in HIP C++ (any n; also the readable statement of the schedule).
// CONV_ONE_PER_SIMD (experiment, with CONV_SC_WAVES_N=1): one-wave workgroups whose register allocation exceeds half the
// file, so that a SIMD can hold only one of them -- the dispatcher has to place a new workgroup on the SIMD that just
// became free, and a wave's LDS is released when IT finishes, not when the slowest of four does.
#ifdef CONV_ONE_PER_SIMD
#define CONV_MIN_WAVES(R, ASM) 1
#else
#define CONV_MIN_WAVES(R, ASM) ((ASM) || (R) >= 128 ? 2 : (R) >= 96 ? 3 : 4)
#endif
// PAIR (with ASM): the pool holds paired lists (tiles.hip: every run an even number of tiles) for the blocks whose runs are long
// enough (T.pflag); those run the pair-step loop of conv_loop2_gfx950.inc -- two tiles of one kernel offset per step, one
// weight fragment for both -- the others the one-tile loop.
template <int R, int DIST, bool ASM, int PAIR = 0>
__global__ __launch_bounds__(64 * SC_WAVES, CONV_MIN_WAVES(R, ASM)) void k_sparse_conv(ConvBatch jobs, ConvTiles T, int n, int relu, int njobs)
{
#ifdef CONV_ONE_PER_SIMD
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23",
                 "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47",
                 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71",
                 "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
#endif
    constexpr int CONV_LDS_WAVE = conv_lds_wave_floats(R);  // R rows + 1 dummy row for padding entries + the tile-header ring
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    // everything derived from the wave index is wave-uniform: keep it in SGPRs (scalar loads for the tile
    // headers, scalar address arithmetic for the weight fragments)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Work items are (block, job) pairs, one per wave: w = slot * njobs + job with slot running over the blocks longest
    // first (neighbouring waves share a block's tile list across the jobs of a batched launch).  No block-wide barrier
    // anywhere: every wave works on its own LDS slice.
    const uint32_t total = (uint32_t)T.nblk * (uint32_t)njobs;
    const uint32_t w = (uint32_t)(blockIdx.x * SC_WAVES + wave);
    if (w >= total) return;
    {
    CT_STAMP(ct0);
#ifdef CONV_TIMING
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t job = w % (uint32_t)njobs, slot = w / (uint32_t)njobs;
    ConvJob J = jobs.job[0];
    if (job == 1) J = jobs.job[1];
    if (job == 2) J = jobs.job[2];
    if (job == 3) J = jobs.job[3];
    const int blk = __builtin_amdgcn_readfirstlane((int)T.order[slot]);   // block id in the tile pool
    // the level of the block (scalar search over <= 24 entries of the kernel argument): its rows in the set's feature arrays
    int lvi = 0;
    for (int i = 1; i < T.nlv; ++i) lvi = blk >= (int)T.lv_blk0[i] ? i : lvi;
    const int lrow0 = (blk - (int)T.lv_blk0[lvi]) * T.H;                   // first row of the block inside its level
    const int nrows = min(T.H, (int)T.lv_rows[lvi] - lrow0);
    const int row0 = (int)T.lv_row0[lvi] + lrow0;                          // ... and in the set
    J.in += (size_t)T.lv_row0[lvi] * 32;                                   // tile entries are row indices inside the level
    float *acc = lds + wave * CONV_LDS_WAVE;
    float4 *acc4 = reinterpret_cast<float4 *>(acc);
    // tile-header ring: 32 slots = two batches of 16 tiles (neighbour rows 32 x 16 dwords, output rows 32 x 4 dwords,
    // offsets 32 dwords).  The tile list of a block is streamed from HBM exactly once, so its latency is the full
    // DRAM latency: a batch is fetched with three wide loads a whole batch ahead of its first use.
    int32_t *hdr = reinterpret_cast<int32_t *>(acc + (R + 1) * ROWF);
    constexpr int RQ = ROWF / 4;                // 16-byte units per LDS row
#pragma unroll
    for (int it = 0; it < (R * RQ + 63) / 64; ++it)
        if (it * 64 + lane < R * RQ) acc4[RQ + it * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);   // slots 1..R
    const int e = lane & 15, g = lane >> 4;
    // column of this lane inside an accumulator row (physical order) for output halves 0 / 1
    const int col0 = 4 * (e & 3) + (e >> 2), col1 = col0 + 16;
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk]), t1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk + 1]);
    const float *__restrict__ in = J.in + 4 * g;
    const float *__restrict__ wf = J.w + lane * 4;
    const int32_t *hj = hdr + e;            // + slot * 16
    const int32_t *hr = hdr + HDR_R + g;    // + slot * 4
    const int32_t *ho = hdr + HDR_O;        // + slot
    struct AB { float4 a0, a1, b00, b01, b10, b11; };
    auto load_ab = [&](int j, uint32_t o) -> AB {
        const float *p = in + (size_t)(uint32_t)j * 32;
        const float *w = wf + (size_t)o * 1024;
        AB r;
        r.a0 = ld4(p); r.a1 = ld4(p + 16);
        r.b00 = ld4(w); r.b01 = ld4(w + 256); r.b10 = ld4(w + 512); r.b11 = ld4(w + 768);
        return r;
    };
    // Per tile: 16 MFMAs build the tile's 16 x 32 partial products from a ZERO accumulator (an fma chain over
    // k = 0..31 per output element), then 8 ds_add_f32 per lane add them to the rows' running sums in LDS.  Nothing
    // in the MFMA block depends on LDS, and the LDS adds of a wave execute in program order, so consecutive
    // tiles may share output rows without any wait; the adds of tile t are issued behind the MFMAs of tile
    // t + 1 (their operands have long left the matrix pipe by then).
    // Per tile: 16 MFMAs build the tile's 16 x 32 partial products from a ZERO accumulator (an fma chain over
    // k = 0..31 per output element); the partial products are then added to the rows' running sums in LDS
    // (read - add - write).  Nothing in the MFMA block depends on LDS, and the LDS operations of a wave execute
    // in program order, so consecutive tiles may share output rows.  The read-add-write of tile t is slotted
    // between the MFMAs of tile t + 1: reads behind the first MFMAs, adds and writes six MFMAs later, when the
    // LDS data has long arrived -- the matrix pipe never waits for LDS.
    struct PT { f32x4 c0, c1; uint32_t r4; };
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
    auto step = [&](const AB &v, const PT &p, PT &q) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        float s0[4], s1[4];
        int row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            row[k] = (int)((p.r4 >> (8 * k)) & 255u) * ROWF;
            s0[k] = acc[row[k] + col0];
            s1[k] = acc[row[k] + col1];
        }
        MF(c0, v.a0.x, v.b00.x); MF(c1, v.a0.x, v.b10.x);
        MF(c0, v.a0.y, v.b00.y); MF(c1, v.a0.y, v.b10.y);
        MF(c0, v.a0.z, v.b00.z); MF(c1, v.a0.z, v.b10.z);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[row[k] + col0] = s0[k] + p.c0[k];
            acc[row[k] + col1] = s1[k] + p.c1[k];
        }
        MF(c0, v.a0.w, v.b00.w); MF(c1, v.a0.w, v.b10.w);
        MF(c0, v.a1.x, v.b01.x); MF(c1, v.a1.x, v.b11.x);
        __builtin_amdgcn_sched_barrier(0);
        MF(c0, v.a1.y, v.b01.y); MF(c1, v.a1.y, v.b11.y);
        MF(c0, v.a1.z, v.b01.z); MF(c1, v.a1.z, v.b11.z);
        MF(c0, v.a1.w, v.b01.w); MF(c1, v.a1.w, v.b11.w);
        q.c0 = c0; q.c1 = c1;
    };
#undef MF
    auto accumulate = [&](const PT &p) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = (int)((p.r4 >> (8 * k)) & 255u) * ROWF;
            acc[row + col0] = acc[row + col0] + p.c0[k];
            acc[row + col1] = acc[row + col1] + p.c1[k];
        }
    };
    if (t0 < t1) {
        // Software pipeline inside one instruction stream: a ring of RING = DIST + 1 register sets, RING tiles
        // per (fully unrolled) iteration.  While tile u computes out of set u % RING the gathered rows and the
        // weight fragment of tile u + DIST land in set (u + DIST) % RING, and the header words of tile
        // u + DIST + 1 are read from the LDS ring -- no register rotation of the big sets, so nothing forces a
        // load to complete inside the step that issued it.  The body has no data-dependent branch: steps past the
        // last tile work on the headers that follow in memory (the next block's, or the zeroed padding of the
        // list) and accumulate into the dummy LDS row.
        constexpr int RING = DIST + 1;
        constexpr uint32_t DUMMY4 = 0u;   // four times the dummy slot
        const uint32_t nt = t1 - t0;
        const int4 *__restrict__ gtj = reinterpret_cast<const int4 *>(T.tj + (size_t)t0 * 16) + lane;
        const uint32_t *__restrict__ gtr = reinterpret_cast<const uint32_t *>(T.tr + (size_t)t0 * 16) + lane;
        const uint32_t *__restrict__ gto = T.toc + t0 + (lane & 15);
        int4 *sj = reinterpret_cast<int4 *>(hdr) + lane;   // + (batch & 1) * 64
        int32_t *sr = hdr + HDR_R + lane;                  // + (batch & 1) * 64
        int32_t *so = hdr + HDR_O + (lane & 15);           // + (batch & 1) * 16
        {   // batches 0 and 1 straight into the ring (and batch 0 into its mirror behind slot 31)
            const int4 a0 = gtj[0], a1 = gtj[64];
            const uint32_t b0 = gtr[0], b1 = gtr[64], c0 = gto[0], c1 = gto[16];
            sj[0] = a0; sj[64] = a1;
            sr[0] = (int32_t)b0; sr[64] = (int32_t)b1;
            so[0] = (int32_t)(c0 & 0xFFFFu); so[16] = (int32_t)(c1 & 0xFFFFu);
            sj[128] = a0;                          // mirror of batch 0 behind slot 31
            sr[128] = (int32_t)b0;
            so[32] = (int32_t)(c0 & 0xFFFFu);
        }
        CT_STAMP(ct1);
        if constexpr (ASM) {
            uint32_t su, st0, st1;
            const uint32_t acc_lds = __builtin_amdgcn_groupstaticsize() + (uint32_t)(wave * CONV_LDS_WAVE * 4);
            const uint32_t hdr_lds = acc_lds + (uint32_t)((R + 1) * ROWF * 4);
#ifdef CONV_LOOP_STAMPS
            uint32_t sw0, sw1, sw2, sw3;
            asm volatile(CONV_LOOP_ASM
                         : [u] "=&s"(su), [t0] "=&s"(st0), [t1] "=&s"(st1), [w0] "=&s"(sw0), [w1] "=&s"(sw1), [w2] "=&s"(sw2), [w3] "=&s"(sw3)
                         : [in] "s"(J.in), [w] "s"(J.w + (size_t)T.K * 1024), [tj] "s"(T.tj + (size_t)t0 * 16), [tr] "s"(T.tr + (size_t)t0 * 16), [toc] "s"(T.toc + t0),
                           [nt] "s"(nt), [acc] "s"(acc_lds), [hdr] "s"(hdr_lds), [lane] "v"(lane)
                         : CONV_LOOP_CLOBBERS);
            if (lane == 0) {
                atomicAdd(&g_conv_timing[12], (unsigned long long)sw0); atomicAdd(&g_conv_timing[13], (unsigned long long)sw1);
                atomicAdd(&g_conv_timing[14], (unsigned long long)sw2); atomicAdd(&g_conv_timing[15], (unsigned long long)sw3);
            }
#else
            if (PAIR == 1 && __builtin_amdgcn_readfirstlane((int)T.pflag[blk]))
                asm volatile(CONV_LOOP2_ASM
                             : [u] "=&s"(su), [t0] "=&s"(st0), [t1] "=&s"(st1)
                             : [in] "s"(J.in), [w] "s"(J.w + (size_t)T.K * 1024), [tj] "s"(T.tj + (size_t)t0 * 16), [tr] "s"(T.tr + (size_t)t0 * 16), [toc] "s"(T.toc + t0),
                               [nt] "s"(nt), [acc] "s"(acc_lds), [hdr] "s"(hdr_lds), [lane] "v"(lane)
                             : CONV_LOOP2_CLOBBERS);
            else
                asm volatile(CONV_LOOP_ASM
                             : [u] "=&s"(su), [t0] "=&s"(st0), [t1] "=&s"(st1)
                             : [in] "s"(J.in), [w] "s"(J.w + (size_t)T.K * 1024), [tj] "s"(T.tj + (size_t)t0 * 16), [tr] "s"(T.tr + (size_t)t0 * 16), [toc] "s"(T.toc + t0),
                               [nt] "s"(nt), [acc] "s"(acc_lds), [hdr] "s"(hdr_lds), [lane] "v"(lane)
                             : CONV_LOOP_CLOBBERS);
#endif
        } else {
        int4 st_j = make_int4(0, 0, 0, 0);
        uint32_t st_r = 0, st_o = 0;
        AB ring[RING];
        uint32_t r4r[RING];
        // prologue: tiles 0 .. DIST-1 in flight, headers of tile DIST in registers
#pragma unroll
        for (int s = 0; s < DIST; ++s) {
            const uint32_t r4_ld = (uint32_t)hr[s * 4];
            r4r[s] = (uint32_t)s < nt ? r4_ld : DUMMY4;
            ring[s] = load_ab(hj[s * 16], (uint32_t)ho[s] & 0xFFFFu);
        }
        int j_n = hj[DIST * 16];
        const uint32_t r4_ld0 = (uint32_t)hr[DIST * 4];
        uint32_t r4_n = (uint32_t)DIST < nt ? r4_ld0 : DUMMY4, o_n = (uint32_t)ho[DIST] & 0xFFFFu;
        PT prev;
        prev.c0 = f32x4{0.f, 0.f, 0.f, 0.f}; prev.c1 = prev.c0; prev.r4 = DUMMY4;
        for (uint32_t u = 0; u < nt; u += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                const uint32_t us = u + (uint32_t)s;
                // header batch (us / 16) + 1: fetched at the first step of batch us / 16, moved into the ring
                // eight steps later (the half of the ring it replaces was last read before step us)
                if ((us & 15u) == 0u && us != 0u) {
                    const uint32_t bo = (us >> 4) + 1u;
                    st_j = gtj[bo * 64]; st_r = gtr[bo * 64]; st_o = gto[bo * 16];
                }
                if ((us & 15u) == 8u && us > 8u) {
                    const uint32_t bb = ((us >> 4) + 1u) & 1u;
                    sj[bb * 64] = st_j; sr[bb * 64] = (int32_t)st_r; so[bb * 16] = (int32_t)(st_o & 0xFFFFu);
                }
                const int sl = (s + DIST) % RING;            // set that receives tile us + DIST
                const uint32_t tq = us + (uint32_t)(DIST + 1);
                const uint32_t slot = tq & 31u;
                const int j_nn = hj[slot * 16];
                const uint32_t r4_ld = (uint32_t)hr[slot * 4], o_nn = (uint32_t)ho[slot] & 0xFFFFu;
                __builtin_amdgcn_sched_barrier(0);  // header reads first: the next step needs them before the big loads
                const uint32_t r4_cur = r4r[s];
                r4r[sl] = r4_n;
                ring[sl] = load_ab(j_n, o_n);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this tile's MFMAs (hipcc sinks it otherwise)
                PT cur;
                step(ring[s], prev, cur);
                cur.r4 = r4_cur;
                prev = cur;
                const uint32_t r4_nn = tq < nt ? r4_ld : DUMMY4;
                __builtin_amdgcn_sched_barrier(0);
                j_n = j_nn; r4_n = r4_nn; o_n = o_nn;
            }
        }
        accumulate(prev);
        }
#ifdef CONV_TIMING
        const long long ct2 = clock64();
        if (lane == 0) { atomicAdd(&g_conv_timing[0], (unsigned long long)(ct1 - ct0)); atomicAdd(&g_conv_timing[1], (unsigned long long)(ct2 - ct1)); atomicAdd(&g_conv_timing[3], (unsigned long long)nt); }
#endif
    }
    CT_STAMP(ct3);
#ifdef CONV_TIMING
    __builtin_amdgcn_s_waitcnt(0);   // what the tile loop left in flight (the prefetches past the last tile, the last LDS sums)
    const long long ct3a = clock64();
#endif
    // epilogue: the block's rows are contiguous in the output and already in the physical channel order -> straight 16-byte
    // copies, one wave instruction per 8 rows (64 float4).  ONE code path for full and short blocks: loads run on clamped
    // addresses (always valid memory), LDS reads are unconditional (a slot past the block's rows is LDS of this wave), only
    // the stores are predicated per lane.  All residual rows of the block are requested before anything else: memory reads
    // and writes complete out of order with respect to each other, so a load issued behind stores can only be waited for
    // with vmcnt(0) -- a batch loop that loads its residuals per batch waits for the previous batch's stores every time
    // (measured: 33 k cycles per 255-row block with a residual, 18 k without -- the short last batch went through a
    // dword-by-dword predicated path; now ~5 k / ~3 k).
#ifdef CONV_NO_EPILOGUE   // developer ablation (WRONG results; HISTORY.md section 4, round 4): what the copy-out costs -- one store keeps the block's work alive
    if (lane == 0 && nrows > 0) J.out[(size_t)row0 * 32] = acc[ROWF];
    if (false)
#endif
    {
        const int nvec = nrows * 8;   // float4 elements of this block
        const bool has_res = J.res != nullptr;
        const float4 *__restrict__ res4 = reinterpret_cast<const float4 *>(J.res) + (size_t)row0 * 8;
        float4 *__restrict__ out4 = reinterpret_cast<float4 *>(J.out) + (size_t)row0 * 8;
        const float4 *src = acc4 + RQ + (lane >> 3) * RQ + (lane & 7);   // slot 1 = row 0; a wave instruction copies 8 rows
        constexpr int NG = (R * 8 + 63) / 64;       // 8-row groups of a full-height block
        constexpr int NB = (NG + 7) / 8;            // batches of 8 groups
        float4 r[NG];
        if (has_res) {
#pragma unroll
            for (int g8 = 0; g8 < NG; ++g8) r[g8] = res4[min(g8 * 64 + lane, nvec - 1)];
        }
#ifdef CONV_TIMING
        long long te[NB + 1];
        te[0] = clock64();
#endif
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            if (bt * 512 < nvec) {   // wave-uniform
                float4 v[8];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (bt * 8 + b < NG) v[b] = src[(bt * 8 + b) * 8 * RQ];
                if (has_res) {
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (bt * 8 + b < NG) { const float4 q = r[bt * 8 + b]; v[b].x = v[b].x + q.x; v[b].y = v[b].y + q.y; v[b].z = v[b].z + q.z; v[b].w = v[b].w + q.w; }
                }
                if (relu) {
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (bt * 8 + b < NG) { v[b].x = v[b].x > 0.f ? v[b].x : 0.f; v[b].y = v[b].y > 0.f ? v[b].y : 0.f; v[b].z = v[b].z > 0.f ? v[b].z : 0.f; v[b].w = v[b].w > 0.f ? v[b].w : 0.f; }
                }
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (bt * 8 + b < NG && (bt * 8 + b) * 64 + lane < nvec) out4[(bt * 8 + b) * 64 + lane] = v[b];
            }
#ifdef CONV_TIMING
            te[bt + 1] = clock64();
#endif
        }
#ifdef CONV_TIMING
#ifndef CONV_LOOP_STAMPS
        if (lane == 0 && NB == 4) {
            atomicAdd(&g_conv_timing[12], (unsigned long long)(te[1] - te[0])); atomicAdd(&g_conv_timing[13], (unsigned long long)(te[2] - te[1]));
            atomicAdd(&g_conv_timing[14], (unsigned long long)(te[3] - te[2])); atomicAdd(&g_conv_timing[15], (unsigned long long)(te[4] - te[3]));
        }
#endif
#endif
    }
#ifdef CONV_TIMING
    const long long ct3b = clock64();
    __builtin_amdgcn_s_waitcnt(0);
    const long long ct4 = clock64();
    if (lane == 0) { atomicAdd(&g_conv_timing[10], (unsigned long long)(ct3a - ct3)); atomicAdd(&g_conv_timing[11], (unsigned long long)(ct3b - ct3a)); }
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        atomicAdd(&g_conv_timing[2], (unsigned long long)(ct4 - ct3)); atomicAdd(&g_conv_timing[4], 1ull); atomicAdd(&g_conv_timing[5], (unsigned long long)(ct4 - ct0));
        atomicAdd(&g_conv_timing[6], rt1 - rt0); atomicMin(&g_conv_timing[7], rt0); atomicMax(&g_conv_timing[8], rt1); atomicMax(&g_conv_timing[9], rt1 - rt0);
    }
#endif
    }
}



// Small levels (16-row blocks: every (block, offset) pair is exactly one tile, a block's list is a serial chain of up to
// 125 tiles, and the whole level is a few hundred blocks): latency, not throughput.  One 16-wave workgroup per block
// breaks the chain: 32 tiles at a time, phase A -- every wave turns two tiles into their 16 x 32 partial products
// (MFMAs from a zero accumulator, exactly as in the big kernel) and parks them in LDS; phase B -- one thread per output
// element adds the products of the tiles that contain its row, in tile (= offset) order.  Same sums in the same
// order as the wave-serial kernel, so the results are bit-identical; the chain per block drops from ~125 tile
// latencies to 4 rounds.
constexpr int COOP_TILES = 32;
#ifndef COOP_WAVES_N
#define COOP_WAVES_N 16
#endif
constexpr int COOP_WAVES = COOP_WAVES_N;
constexpr int COOP_TPW = COOP_TILES / COOP_WAVES;   // tiles per wave and round
// LDS: products [32][16][32] f32, row->entry map [ROWS][32] u8, then the block's tile headers (at most K x ROWS / 16 tiles: 16 + 4 + 1 dwords each)
// ROWS = 16 / 32 / 64 rows per block (conv_coop_rows); a block of ROWS rows has at most ROWS / 16 tiles per kernel offset.
static inline size_t coop_lds_bytes(int K, int rows) { return (size_t)COOP_TILES * 16 * 32 * 4 + (size_t)rows * COOP_TILES + (size_t)K * (rows / 16) * 84; }

template <int ROWS>
__global__ __launch_bounds__(64 * COOP_WAVES) void k_sparse_conv_coop(ConvBatch jobs, ConvTiles T, int n, int relu)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NTMAX = T.K * (ROWS / 16);                              // the most tiles a block can have
    float *P = lds;                                                   // [tile][entry][channel (physical)]
    uint8_t *inv = reinterpret_cast<uint8_t *>(lds + COOP_TILES * 16 * 32);   // [row][tile] -> entry of that row in the tile, 255 = absent
    int32_t *hj = reinterpret_cast<int32_t *>(inv + ROWS * COOP_TILES);  // [tile][16] neighbour rows
    uint32_t *hr = reinterpret_cast<uint32_t *>(hj + (size_t)NTMAX * 16); // [tile][4]  output rows (bytes)
    uint32_t *ho = hr + (size_t)NTMAX * 4;                                // [tile]     offset | count << 16
#ifdef COOP_TIMING
    const long long q0 = clock64();
#endif
    const ConvJob J = jobs.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = (int)T.lv_blk0[0] + (int)blockIdx.x;   // block id in the tile pool (a set's blocks are consecutive)
    int lvi = 0;
    for (int i = 1; i < T.nlv; ++i) lvi = blk >= (int)T.lv_blk0[i] ? i : lvi;
    const int lrow0 = (blk - (int)T.lv_blk0[lvi]) * ROWS;
    const int nrows = min(ROWS, (int)T.lv_rows[lvi] - lrow0);
    const int row0 = (int)T.lv_row0[lvi] + lrow0;
    const int e = lane & 15, g = lane >> 4;
    const int col0 = 4 * (e & 3) + (e >> 2), col1 = col0 + 16;
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk]), t1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk + 1]);
    const int nt = (int)min(t1 - t0, (uint32_t)NTMAX);   // a block has at most ROWS / 16 tiles per kernel offset
    // the whole header list of the block in one sweep (three latencies instead of one per round)
    for (int i = tid; i < nt * 16; i += 64 * COOP_WAVES) hj[i] = T.tj[(size_t)t0 * 16 + i];
    for (int i = tid; i < nt * 4; i += 64 * COOP_WAVES) hr[i] = reinterpret_cast<const uint32_t *>(T.tr + (size_t)t0 * 16)[i];
    for (int i = tid; i < nt; i += 64 * COOP_WAVES) ho[i] = T.toc[t0 + i];
    __syncthreads();
#ifdef COOP_TIMING
    const long long q1 = clock64();
#endif
    const float *__restrict__ in = J.in + (size_t)T.lv_row0[lvi] * 32 + 4 * g;   // tile entries are row indices inside the level
    const float *__restrict__ wf = J.w + lane * 4;
    constexpr int NEL = ROWS * 32;                                                        // output elements of the block
    constexpr int COOP_EPT = NEL / (64 * COOP_WAVES) > 0 ? NEL / (64 * COOP_WAVES) : 1;   // output elements per thread in phase B
    float acc[COOP_EPT];
#pragma unroll
    for (int u = 0; u < COOP_EPT; ++u) acc[u] = 0.0f;
    struct AB { float4 a0, a1, b00, b01, b10, b11; };
    const uint32_t nlv_rows = max(T.lv_rows[lvi], 1u);
    auto fetch = [&](int t) -> AB {   // gathered rows + weight fragment of tile t of this block (t < nt); clamped: see k_conv_products
        const float *p = in + (size_t)min((uint32_t)hj[t * 16 + e], nlv_rows - 1u) * 32;
        const float *w = wf + (size_t)min(ho[t] & 0xFFFFu, (uint32_t)T.K - 1u) * 1024;
        AB r;
        r.a0 = ld4(p); r.a1 = ld4(p + 16);
        r.b00 = ld4(w); r.b01 = ld4(w + 256); r.b10 = ld4(w + 512); r.b11 = ld4(w + 768);
        return r;
    };
    auto products = [&](const AB &v, int t, int tl) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 c0 = z, c1 = z;
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
        MF(c0, v.a0.x, v.b00.x); MF(c1, v.a0.x, v.b10.x);
        MF(c0, v.a0.y, v.b00.y); MF(c1, v.a0.y, v.b10.y);
        MF(c0, v.a0.z, v.b00.z); MF(c1, v.a0.z, v.b10.z);
        MF(c0, v.a0.w, v.b00.w); MF(c1, v.a0.w, v.b10.w);
        MF(c0, v.a1.x, v.b01.x); MF(c1, v.a1.x, v.b11.x);
        MF(c0, v.a1.y, v.b01.y); MF(c1, v.a1.y, v.b11.y);
        MF(c0, v.a1.z, v.b01.z); MF(c1, v.a1.z, v.b11.z);
        MF(c0, v.a1.w, v.b01.w); MF(c1, v.a1.w, v.b11.w);
#undef MF
        float *dst = P + (size_t)tl * 512;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[(4 * g + k) * 32 + col0] = c0[k];
            dst[(4 * g + k) * 32 + col1] = c1[k];
        }
        if (lane < ROWS) inv[lane * COOP_TILES + tl] = 255;
        const uint32_t cnt = ho[t] >> 16;
        const uint32_t r = (hr[t * 4 + (e >> 2)] >> (8 * (e & 3))) & 255u;
        if (lane < 16 && (uint32_t)lane < cnt && r >= 1u && r <= (uint32_t)ROWS) inv[(r - 1u) * COOP_TILES + tl] = (uint8_t)lane;   // tr holds row + 1; same wave: ordered behind the 255s
    };
    // this wave's two tiles of a round: base + 2 wave + {0, 1}; the next round's operands are requested before this
    // round's products are summed, so a round costs MFMAs + two barriers, not a memory latency
    AB cur[COOP_TPW], nxt[COOP_TPW];
#pragma unroll
    for (int q = 0; q < COOP_TPW; ++q)
        if (wave * COOP_TPW + q < nt) cur[q] = fetch(wave * COOP_TPW + q);
    for (int base = 0; base < nt; base += COOP_TILES) {
#pragma unroll
        for (int q = 0; q < COOP_TPW; ++q) {
            const int t = base + COOP_TILES + wave * COOP_TPW + q;
            if (t < nt) nxt[q] = fetch(t);
        }
#pragma unroll
        for (int q = 0; q < COOP_TPW; ++q) {
            const int t = base + wave * COOP_TPW + q;
            if (t < nt) products(cur[q], t, wave * COOP_TPW + q);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < COOP_EPT; ++u) {   // ordered sum over the tiles of this round, one output element at a time
            const int el = tid + u * 64 * COOP_WAVES;
            if (el < NEL) {
                const int orow = el >> 5, och = el & 31;
                const int ntl = min(COOP_TILES, nt - base);
                const uint4 w0 = *reinterpret_cast<const uint4 *>(inv + orow * COOP_TILES), w1 = *reinterpret_cast<const uint4 *>(inv + orow * COOP_TILES + 16);
                const uint32_t wd[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                float pv[COOP_TILES];
#pragma unroll
                for (int tl = 0; tl < COOP_TILES; ++tl) {
                    const uint32_t i = (wd[tl >> 2] >> (8 * (tl & 3))) & 255u;
                    // absent (or past the end of the list): + 0.0f, which leaves the sum unchanged (the sum is never -0).
                    // The read itself is unconditional (a valid address either way) so that all 32 are in flight together;
                    // under the condition the compiler branches around each read and waits for them one by one.
                    const float pvv = P[(size_t)tl * 512 + (i & 15u) * 32 + och];
                    pv[tl] = (tl < ntl && i != 255u) ? pvv : 0.0f;
                }
#pragma unroll
                for (int tl = 0; tl < COOP_TILES; ++tl) acc[u] = acc[u] + pv[tl];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < COOP_TPW; ++q) cur[q] = nxt[q];
    }
#ifdef COOP_TIMING
    const long long q2 = clock64();
#endif
#pragma unroll
    for (int u = 0; u < COOP_EPT; ++u) {
        const int el = tid + u * 64 * COOP_WAVES;
        const int grow = row0 + (el >> 5), och = el & 31;
        if (el < NEL && (el >> 5) < nrows) {
            float v = acc[u];
            if (J.res) v = v + J.res[(size_t)grow * 32 + och];
            if (relu) v = v > 0.f ? v : 0.f;
            J.out[(size_t)grow * 32 + och] = v;
        }
    }
#ifdef COOP_TIMING
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) printf("[coop] blocks %u nt %d: headers %lld rounds %lld epilogue %lld cycles\n", gridDim.x, nt, q1 - q0, q2 - q1, clock64() - q2);
#endif
}

// Small levels, spread over the whole chip: TWO launches per convolution instead of a chain inside one CU.
// The cooperative kernel above gives a 16-row block to one workgroup; a dense small level (40 .. 77 neighbours per node)
// has ~100 tiles per block, so a block is ~10^4 matrix cycles on ONE CU while a level of 500 nodes leaves 220 CUs idle
// (17 .. 39 us per convolution, 131 of them per decode).  Here
//   k_conv_products   a wave per tile, the tiles of a block spread over up to 32 workgroups: the 16 x 32 products of the
//                     tile's pairs -- MFMAs from a zero accumulator, the transposed product of the asm loop, so a lane holds
//                     four physically consecutive channels of one tile row -- go to P[tile][entry][32] in global memory
//                     (valid entries only);
//   k_conv_sum        a workgroup per block, a thread per output element: the products of the tiles that contain its row,
//                     added in tile (= offset) order.  The same sums in the same order as every other conv kernel.
// P lives in the context (gpcc_ctx::conv_products), 2 KiB per tile of the list's bound; traffic is 128 B per pair each way.
constexpr int PROD_WAVES = 4;
__global__ __launch_bounds__(64 * PROD_WAVES) void k_conv_products(ConvJob J, ConvTiles T, float *__restrict__ P)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = (int)T.lv_blk0[0] + (int)blockIdx.x;
    int lvi = 0;
    for (int i = 1; i < T.nlv; ++i) lvi = blk >= (int)T.lv_blk0[i] ? i : lvi;
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk]), t1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.first[blk + 1]);
    const int e = lane & 15, g = lane >> 4;
    const float *__restrict__ in = J.in + (size_t)T.lv_row0[lvi] * 32 + 4 * g;   // tile entries are row indices inside the level
    const float *__restrict__ wt = J.w + (size_t)T.K * 1024 + lane * 4;          // the transposed fragments (conv_weight_fragments_t)
    const uint32_t nlv_rows = max(T.lv_rows[lvi], 1u);
    P -= (size_t)T.first[T.lv_blk0[0]] * 512;                                    // P[0] = the set's first tile (the set's blocks are consecutive in the pool)
    const uint32_t tend = t0 + min(t1 - t0, (uint32_t)T.K);   // a 16-row block has at most one tile per offset (a decoder fed a lying header builds lists from garbage: stay inside them)
    for (uint32_t t = t0 + blockIdx.y * PROD_WAVES + (uint32_t)wave; t < tend; t += gridDim.y * PROD_WAVES) {
        // (row and offset clamped: the lists of a level whose container header lies are built from stale rows, and round 3
        // found unwritten tiles in the last block of such a level -- garbage in, garbage out, but inside the arrays)
        const uint32_t oc = T.toc[t];
        const uint32_t cnt = oc >> 16;
        const float *p = in + (size_t)min((uint32_t)T.tj[(size_t)t * 16 + e], nlv_rows - 1u) * 32;
        const float *w = wt + (size_t)min(oc & 0xFFFFu, (uint32_t)T.K - 1u) * 1024;
        const float4 x0 = ld4(p), x1 = ld4(p + 16);
        const float4 w0 = ld4(w), w1 = ld4(w + 256), w2 = ld4(w + 512), w3 = ld4(w + 768);
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
        MF(c0, w0.x, x0.x); MF(c1, w2.x, x0.x);
        MF(c0, w0.y, x0.y); MF(c1, w2.y, x0.y);
        MF(c0, w0.z, x0.z); MF(c1, w2.z, x0.z);
        MF(c0, w0.w, x0.w); MF(c1, w2.w, x0.w);
        MF(c0, w1.x, x1.x); MF(c1, w3.x, x1.x);
        MF(c0, w1.y, x1.y); MF(c1, w3.y, x1.y);
        MF(c0, w1.z, x1.z); MF(c1, w3.z, x1.z);
        MF(c0, w1.w, x1.w); MF(c1, w3.w, x1.w);
#undef MF
        if ((uint32_t)e < cnt) {
            float *dst = P + ((size_t)t * 16 + e) * 32 + 4 * g;
            *reinterpret_cast<float4 *>(dst) = make_float4(c0[0], c0[1], c0[2], c0[3]);
            *reinterpret_cast<float4 *>(dst + 16) = make_float4(c1[0], c1[1], c1[2], c1[3]);
        }
    }
}

// developer check (GAUSPCC_DEBUG_LAUNCH): every tile of the set inside its bounds?
__global__ void k_check_tiles(ConvTiles T, int64_t n, unsigned long long total_cap, int *bad, int koff)
{
    // T.K = the most tiles a block may have, koff = kernel offsets
    const int blk = (int)T.lv_blk0[0] + (int)blockIdx.x;
    const uint32_t t0 = T.first[blk], t1 = T.first[blk + 1];
    if (threadIdx.x == 0 && (t1 < t0 || t1 - t0 > (uint32_t)T.K || t1 > total_cap)) { if (atomicAdd(bad, 1) < 8) printf("[check] block %d: tiles %u .. %u (at most %d, cap %llu)\n", blk, t0, t1, T.K, total_cap); }
    const uint32_t tend = t0 + min(t1 - t0, (uint32_t)T.K);
    for (uint32_t t = t0 + threadIdx.x / 16; t < tend && t < total_cap; t += blockDim.x / 16) {
        const int e = threadIdx.x & 15;
        const int32_t j = T.tj[(size_t)t * 16 + e];
        const uint32_t oc = T.toc[t];
        const uint32_t r = T.tr[(size_t)t * 16 + e];
        if (j < 0 || j >= n || (oc & 0xFFFFu) >= (uint32_t)koff || (oc >> 16) > 16u || r > (uint32_t)T.H) { if (atomicAdd(bad, 1) < 8) printf("[check] block %d tile %u entry %d: j %d (n %lld) toc %08x slot %u\n", blk, t, e, j, (long long)n, oc, r); }
    }
}

constexpr int SUM_BATCH = 32;   // products a thread has in flight
__global__ __launch_bounds__(512) void k_conv_sum(ConvJob J, ConvTiles T, const float *__restrict__ P, int relu)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t inv[];   // [16][ntp]: entry of row r in tile t of the block, 255 = absent
    const int tid = threadIdx.x;
    const int blk = (int)T.lv_blk0[0] + (int)blockIdx.x;
    int lvi = 0;
    for (int i = 1; i < T.nlv; ++i) lvi = blk >= (int)T.lv_blk0[i] ? i : lvi;
    const int lrow0 = (blk - (int)T.lv_blk0[lvi]) * 16;
    const int nrows = min(16, (int)T.lv_rows[lvi] - lrow0);
    const int row0 = (int)T.lv_row0[lvi] + lrow0;
    const uint32_t t0 = T.first[blk], t1 = T.first[blk + 1];
    const int nt = (int)min(t1 - t0, (uint32_t)T.K);          // a 16-row block has at most one tile per kernel offset
    const int ntp = (T.K + SUM_BATCH - 1) / SUM_BATCH * SUM_BATCH;   // row pitch of inv: whole batches, read unconditionally
    for (int i = tid; i < 16 * ntp; i += 512) inv[i] = 255;
    __syncthreads();
    for (int i = tid; i < nt * 16; i += 512) {
        const int t = i >> 4, en = i & 15;
        const uint32_t cnt = T.toc[t0 + t] >> 16;
        const uint32_t r = T.tr[(size_t)(t0 + t) * 16 + en];      // LDS slot = row + 1, 0 = padding
        if ((uint32_t)en < cnt && r >= 1u && r <= 16u) inv[(r - 1u) * ntp + t] = (uint8_t)en;
    }
    __syncthreads();
    const int r = tid >> 5, ch = tid & 31;
    const float *__restrict__ Pb = P + (size_t)(t0 - T.first[T.lv_blk0[0]]) * 512 + ch;
    const uint8_t *iv = inv + r * ntp;
    float acc = 0.0f;
    for (int tb = 0; tb < nt; tb += SUM_BATCH) {
        float pv[SUM_BATCH];
#pragma unroll
        for (int u = 0; u < SUM_BATCH; ++u) {
            // every load is issued whether the row is in the tile or not (a valid address either way: tiles past the block's
            // last read the first tile again): SUM_BATCH loads in flight instead of one branch and one wait per tile
            const int t = tb + u;
            const uint32_t en = iv[t];
            const float v = Pb[(size_t)(t < nt ? t : 0) * 512 + (en & 15u) * 32];
            pv[u] = (t < nt && en != 255u) ? v : 0.0f;             // absent: + 0.0f leaves the sum unchanged (it is never -0)
        }
#pragma unroll
        for (int u = 0; u < SUM_BATCH; ++u) acc = acc + pv[u];
    }
    if (r < nrows) {
        const size_t at = (size_t)(row0 + r) * 32 + ch;
        float v = acc;
        if (J.res) v = v + J.res[at];
        if (relu) v = v > 0.f ? v : 0.f;
        J.out[at] = v;
    }
}

int prof_event(gpcc_ctx *ctx, hipStream_t st, int *idx)
{
    Prof &p = ctx->prof;
    if (p.used == (int)p.pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        p.pool.push_back(e);
    }
    *idx = p.used++;
    HIP_TRY(hipEventRecord(p.pool[(size_t)*idx], st));
    return GPCC_OK;
}

int sparse_conv(gpcc_ctx *ctx, int level, hipStream_t st, const ConvBatch &jobs, int njobs, const ConvTiles &T, int64_t n, int relu)
{
    if (n <= 0) return GPCC_OK;
    if (n >= (int64_t)1 << 31) return fail(GPCC_ERR_ARG, "level too large");
    if (jobs.C != 0 && jobs.C != 32) return any_sparse_conv(st, jobs, njobs, T, n, relu, jobs.C);   // (not timed as k_sparse_conv: another kernel family)
    const bool chained = ctx && ctx->prof.on && ctx->prof.chain_open;
    const bool prof = ctx && ctx->prof.on && !chained;
    ConvRec rec = {0, 0, level, njobs, T.R, T.H, (long long)n, (long long)T.nblk, 1};
    if (chained) { ConvRec &c = ctx->prof.chain; c.level = level; c.njobs = njobs; c.R = T.R; c.H = T.H; c.n = (long long)n; c.nblk = (long long)T.nblk; c.launches += 1; }
    if (prof) GP_TRY(prof_event(ctx, st, &rec.e0));
    static const int use_asm = dev_env_int("GAUSPCC_CONV_ASM", 1) != 0;
    static PerDeviceOnce lds_attr;
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    GP_TRY(lds_attr.run(cur_dev, [&]() -> int {  // 128-row blocks need more LDS per workgroup than the 64 KiB default cap
        const int bytes = SC_WAVES * conv_lds_wave_floats(128) * 4;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv<128, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv<255, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SC_WAVES * conv_lds_wave_floats(255) * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv<255, 1, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, SC_WAVES * conv_lds_wave_floats(255) * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv<255, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SC_WAVES * conv_lds_wave_floats(255) * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv<128, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv_coop<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes(343, 16)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv_coop<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes(125, 32)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sparse_conv_coop<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes(125, 64)));
        return GPCC_OK;
    }));
    static const int use_split = dev_env_int("GAUSPCC_CONV_SPLIT", 1) != 0;
    const bool use_coop = conv_is_coop(n, T.R) && T.H == T.R;
    // 16-row blocks.  Up to 64 blocks (a level of at most 1 k nodes): products over the whole chip + ordered sums, two
    // launches -- measured 11.7 / 15.2 us against 18.8 / 26.2 for the one-workgroup-per-block kernel at 4 / 32 blocks.
    // Beyond that the two launches cost what they save (26.9 vs 25.8 us at 160 blocks): the cooperative kernel.
    // (A second cooperative kernel -- 8 waves, two workgroups per CU, transposed products parked with 16-byte swizzled
    // stores, operands one tile ahead -- was built and measured in round 3: 25.1 / 34.3 / 24.7 us at 160 / 263 / 315 blocks
    // against 25.8 / 36.5 / 22.5: a block's ~100 tiles move ~750 KB through one CU's L1 whatever the schedule.  Dropped.)
    static const int split_max = 64;
    const size_t prod_floats = ((size_t)T.nblk * (size_t)T.K + CONV_HDR_PAD) * 512;
    if (T.R == 16 && use_coop && use_split && ctx && T.K <= 343 && T.nblk <= split_max && prod_floats * 4 <= ((size_t)768 << 20)) {
        if (ctx->conv_products_cap < prod_floats) {
            HIP_TRY(hipStreamSynchronize(st));   // an earlier convolution of this context may still read the old buffer
            if (ctx->conv_products) HIP_TRY(hipFree(ctx->conv_products));
            ctx->conv_products = nullptr; ctx->conv_products_cap = 0;
            const size_t want = std::max(prod_floats, (size_t)16 << 20);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->conv_products), want * 4));
            ctx->conv_products_cap = want;
        }
        // tiles of a block over `split` workgroups of 4 waves: about 2 k workgroups in all, at most one tile per wave
        // (requesting the headers, then the operands of four tiles per wave together and 128 products per thread in the sum
        // was built and measured: slower at every size -- 30 / 45 / 41 us at 160 / 263 / 315 blocks)
        const unsigned split = (unsigned)std::max<int64_t>(1, std::min<int64_t>({(int64_t)32, (int64_t)cdiv(T.K, PROD_WAVES), (int64_t)cdiv(2048, T.nblk)}));
        const int ntp = (T.K + SUM_BATCH - 1) / SUM_BATCH * SUM_BATCH;
        static const bool dbg = getenv("GAUSPCC_DEBUG_LAUNCH") != nullptr;
        for (int j = 0; j < njobs; ++j) {
            if (dbg) {
                int *bad = nullptr, hb = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&bad), 4));
                HIP_TRY(hipMemsetAsync(bad, 0, 4, st));
                k_check_tiles<<<(unsigned)T.nblk, 256, 0, st>>>(T, n, (unsigned long long)((size_t)T.nblk * T.K + CONV_HDR_PAD), bad, T.K);
                HIP_TRY(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                (void)hipFree(bad);
                fprintf(stderr, "[dbg] tile check: %d bad\n", hb);
            }
            if (dbg) { fprintf(stderr, "[dbg] products level %d n %lld nblk %lld split %u K %d cap %zu need %zu\n", level, (long long)n, (long long)T.nblk, split, T.K, ctx->conv_products_cap, prod_floats); fflush(stderr); }
            k_conv_products<<<dim3((unsigned)T.nblk, split), 64 * PROD_WAVES, 0, st>>>(jobs.job[j], T, ctx->conv_products);
            LAUNCH_CHECK();
            if (dbg) { HIP_TRY(hipStreamSynchronize(st)); fprintf(stderr, "[dbg] sum\n"); fflush(stderr); }
            k_conv_sum<<<(unsigned)T.nblk, 512, (size_t)16 * ntp, st>>>(jobs.job[j], T, ctx->conv_products, relu);
            LAUNCH_CHECK();
            if (dbg) { HIP_TRY(hipStreamSynchronize(st)); fprintf(stderr, "[dbg] done\n"); fflush(stderr); }
        }
        if (prof) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
        return GPCC_OK;
    }
    if (use_coop) {
        if (T.K > 343 || (T.R > 16 && T.K > 125)) return fail(GPCC_ERR_ARG, "kernel size > 7 is not supported");
        const dim3 cg((unsigned)T.nblk, (unsigned)njobs);
        const size_t cl = coop_lds_bytes(T.K, T.R);
        if (T.R == 16) k_sparse_conv_coop<16><<<cg, 64 * COOP_WAVES, cl, st>>>(jobs, T, (int)n, relu);
        else if (T.R == 32) k_sparse_conv_coop<32><<<cg, 64 * COOP_WAVES, cl, st>>>(jobs, T, (int)n, relu);
        else k_sparse_conv_coop<64><<<cg, 64 * COOP_WAVES, cl, st>>>(jobs, T, (int)n, relu);
        LAUNCH_CHECK();
        if (prof) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
        return GPCC_OK;
    }
    if (debug_sync_on()) {   // developer: every tile of the set inside its bounds?
        int *bad = nullptr, hb = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&bad), 4));
        HIP_TRY(hipMemsetAsync(bad, 0, 4, st));
        ConvTiles Tc = T; Tc.K = T.K * ((T.H + 15) / 16);   // tiles a block may have
        k_check_tiles<<<(unsigned)T.nblk, 256, 0, st>>>(Tc, n, ~0ull, bad, T.K);
        HIP_TRY(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(bad);
        fflush(stdout);
        fprintf(stderr, "[conv] level %d n %lld R %d H %d blocks %lld paired %d: tile check %d bad\n", level, (long long)n, T.R, T.H, (long long)T.nblk, T.paired, hb); fflush(stderr);
        if (hb) return fail(GPCC_ERR_FORMAT, "developer check: %d tile entries out of bounds at level %d", hb, level);
    }
    dim3 grid((unsigned)cdiv(T.nblk * njobs, SC_WAVES), 1u);   // work items = blocks x jobs, one per wave
    const size_t lds_bytes = (size_t)SC_WAVES * conv_lds_wave_floats(T.R) * 4;
    const bool asm_ok = use_asm && n < ((int64_t)1 << 25);   // the asm loop addresses rows with 32-bit byte offsets
#define CONV_LAUNCH(RR, DD, AA) k_sparse_conv<RR, DD, AA><<<grid, 64 * SC_WAVES, lds_bytes, st>>>(jobs, T, (int)n, relu, njobs)
    switch (T.R) {
    case 16: if (asm_ok) CONV_LAUNCH(16, 1, true); else CONV_LAUNCH(16, 1, false); break;
    case 32: if (asm_ok) CONV_LAUNCH(32, 1, true); else CONV_LAUNCH(32, 1, false); break;
    case 64: if (asm_ok) CONV_LAUNCH(64, 1, true); else CONV_LAUNCH(64, 1, false); break;
    case 96: if (asm_ok) CONV_LAUNCH(96, 1, true); else CONV_LAUNCH(96, 1, false); break;
    case 255:
        if (asm_ok && T.paired) k_sparse_conv<255, 1, true, 1><<<grid, 64 * SC_WAVES, lds_bytes, st>>>(jobs, T, (int)n, relu, njobs);
        else if (asm_ok) CONV_LAUNCH(255, 1, true);
        else CONV_LAUNCH(255, 1, false);
        break;
    default:
        if (asm_ok) CONV_LAUNCH(128, 1, true);
        else CONV_LAUNCH(128, 1, false);
        break;
    }
#undef CONV_LAUNCH
    LAUNCH_CHECK();
    if (prof) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
#ifdef CONV_TIMING
    if (prof) {
        unsigned long long h[16], z[16] = {0};
        z[7] = ~0ull;
        HIP_TRY(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof.pool[(size_t)rec.e0], ctx->prof.pool[(size_t)rec.e1]));
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_timing), sizeof h));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_timing), z, sizeof z));
        const double waves = (double)h[4], tiles = (double)h[3];
        fprintf(stderr, "[conv] R %3d H %3d jobs %d n %8lld blocks %6.0f tiles %9.0f  %.1f us | per wave: setup %.0f loop %.0f epilogue %.0f total %.0f cyc | loop %.0f cyc/tile | clock %.0f MHz | span %.1f us, wave time / 1024 slots %.1f us, longest block %.1f us\n",
                T.R, T.H, njobs, (long long)n, waves, tiles, ms * 1e3, h[0] / waves, h[1] / waves, h[2] / waves, h[5] / waves, h[1] / tiles, 100.0 * h[5] / (double)h[6],
                (h[8] - h[7]) / 100.0, h[6] / 100.0 / 1024.0, h[9] / 100.0);
        fprintf(stderr, "[conv]    epilogue: loop drain %.0f, copy-out issue %.0f, store drain %.0f cycles; residual %d relu %d\n", h[10] / waves, h[11] / waves,
                (h[2] - h[10] - h[11]) / waves, jobs.job[0].res != nullptr, relu);
        if (h[12] + h[13] + h[14] + h[15])
#ifdef CONV_LOOP_STAMPS
            fprintf(stderr, "[conv]    per tile: step-start wait %.0f, first pair + VALU burst + LDS issue %.0f, second pair + loads issue %.0f, remaining 12 MFMAs %.0f cycles\n",
                    h[12] / tiles, h[13] / tiles, h[14] / tiles, h[15] / tiles);
#else
            fprintf(stderr, "[conv]    copy-out batches of 64 rows: %.0f %.0f %.0f %.0f cycles\n", h[12] / waves, h[13] / waves, h[14] / waves, h[15] / waves);
#endif
    }
#endif
    return GPCC_OK;
}

// a run of convolutions on the same level with nothing else enqueued between them: one pair of events for all
int conv_chain_begin(gpcc_ctx *ctx, hipStream_t st)
{
    if (!ctx || !ctx->prof.on) return GPCC_OK;
#ifdef CONV_TIMING
    return GPCC_OK;   // the developer timing build reads its counters back after every launch
#endif
    ctx->prof.chain = ConvRec{};
    GP_TRY(prof_event(ctx, st, &ctx->prof.chain.e0));
    ctx->prof.chain_open = true;
    return GPCC_OK;
}

int conv_chain_end(gpcc_ctx *ctx, hipStream_t st)
{
    if (!ctx || !ctx->prof.on || !ctx->prof.chain_open) return GPCC_OK;
    ctx->prof.chain_open = false;
    GP_TRY(prof_event(ctx, st, &ctx->prof.chain.e1));
    if (ctx->prof.chain.launches > 0) ctx->prof.recs.push_back(ctx->prof.chain);
    return GPCC_OK;
}

int prof_collect(gpcc_ctx *ctx, const unsigned long long *pairs, int nlevels)
{
    Prof &p = ctx->prof;
    p.chain_open = false;
    static const int log = env_int("GAUSPCC_CONV_LOG", 0);
    for (const ConvRec &r : p.recs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.pool[(size_t)r.e0], p.pool[(size_t)r.e1]));
        if (log)
            for (int q = 0; q < r.launches; ++q)
                fprintf(stderr, "[conv] level %2d n %8lld R %3d H %3d blocks %6lld jobs %d  %8.1f us pairs %llu\n", r.level, r.n, r.R, r.H, r.nblk, r.njobs, ms * 1e3 / r.launches,
                        (r.level >= 0 && r.level < nlevels) ? pairs[r.level] : 0ull);
        if (r.fused) {
            p.fused_ms += ms;
            p.fused_launches += r.launches;
            if (r.level >= 0 && r.level < nlevels) p.fused_pair_jobs += (int64_t)pairs[r.level] * r.njobs * r.launches;
            continue;
        }
        p.conv_ms += ms;
        p.conv_launches += r.launches;
        if (r.level >= 0 && r.level < nlevels) p.conv_pair_jobs += (int64_t)pairs[r.level] * r.njobs * r.launches;
    }
    // the convolution brackets of this call on a common time axis (events of any stream against the first event of the pool):
    // a stage bracket's critical time is what lies outside all of them
    std::vector<std::pair<float, float>> busy;
    if (!p.srecs.empty() && p.used > 0) {
        for (const ConvRec &r : p.recs) {
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, p.pool[0], p.pool[(size_t)r.e0]) != hipSuccess || hipEventElapsedTime(&b, p.pool[0], p.pool[(size_t)r.e1]) != hipSuccess) { busy.clear(); break; }
            busy.emplace_back(a, b);
        }
        std::sort(busy.begin(), busy.end());
        size_t w = 0;
        for (size_t i = 0; i < busy.size(); ++i) {   // merge
            if (w && busy[i].first <= busy[w - 1].second) busy[w - 1].second = std::max(busy[w - 1].second, busy[i].second);
            else busy[w++] = busy[i];
        }
        busy.resize(w);
    }
    p.recs.clear();
    for (const Prof::StageRec &r : p.srecs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.pool[(size_t)r.e0], p.pool[(size_t)r.e1]));
        p.stage_ms[r.id] += ms;
        p.stage_n[r.id] += 1;
        float a = 0.f, b = 0.f, cov = 0.f;
        if (p.used > 0 && hipEventElapsedTime(&a, p.pool[0], p.pool[(size_t)r.e0]) == hipSuccess && hipEventElapsedTime(&b, p.pool[0], p.pool[(size_t)r.e1]) == hipSuccess) {
            for (const auto &iv : busy) cov += std::max(0.f, std::min(b, iv.second) - std::max(a, iv.first));
            p.stage_crit_ms[r.id] += std::max(0.f, (b - a) - cov);
        } else p.stage_crit_ms[r.id] += ms;
    }
    p.srecs.clear();
    p.used = 0;
    return GPCC_OK;
}

// ------------------------------------------------------------------ row-wise elementwise kernels
constexpr int TB = 256;
static inline unsigned nblk(int64_t n) { return (unsigned)cdiv(n, TB); }

__global__ __launch_bounds__(TB) void k_embed_occ(const float4 *__restrict__ emb, const uint8_t *__restrict__ occ, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    out[t] = emb[(size_t)occ[i] * 8 + (t & 7)];
}
int embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out, int C)
{
    if (C != 32) return any_embed_occ(st, emb, occ, n, out, C);
    k_embed_occ<<<nblk(n * 8), TB, 0, st>>>((const float4 *)emb, occ, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(TB) void k_child_features(const float4 *__restrict__ F, const uint32_t *__restrict__ parent, const uint64_t *__restrict__ rkey_c,
                                                       const float4 *__restrict__ temb, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const int g = (int)(t & 7);
    const uint64_t k = rkey_c[i];
    const int q = (int)((rk_x(k) & 1) | ((rk_y(k) & 1) << 1) | ((rk_z(k) & 1) << 2));
    out[t] = add4(F[(size_t)parent[i] * 8 + g], temb[q * 8 + g]);
}
int child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out, int C)
{
    if (C != 32) return any_child_features(st, F, parent, rkey_c, temb, n, out, C);
    k_child_features<<<nblk(n * 8), TB, 0, st>>>((const float4 *)F, parent, rkey_c, (const float4 *)temb, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// the encoder's three stage inputs in one pass over X (one read of the trunk output instead of three)
__global__ __launch_bounds__(TB) void k_stage_inputs_gt(const float4 *__restrict__ X, const float4 *__restrict__ e1, const float4 *__restrict__ e2, const float4 *__restrict__ e3,
                                                        const uint8_t *__restrict__ occ, int64_t n, float4 *__restrict__ o1, float4 *__restrict__ o2, float4 *__restrict__ o3)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const uint32_t o = occ[i];
    const float4 x = X[t];
    const int c = (int)(t & 7);
    o1[t] = add4(x, e1[((o >> 7) & 1u) * 8 + c]);    // pcc_utils.py:121,128,136
    o2[t] = add4(x, e2[((o >> 6) & 3u) * 8 + c]);
    o3[t] = add4(x, e3[((o >> 4) & 15u) * 8 + c]);
}
int stage_inputs_gt(hipStream_t st, const float *X, const float *const emb[3], const uint8_t *occ, int64_t n, float *const out[3], int C)
{
    if (C != 32) return any_stage_inputs_gt(st, X, emb, occ, n, out, C);
    k_stage_inputs_gt<<<nblk(n * 8), TB, 0, st>>>((const float4 *)X, (const float4 *)emb[0], (const float4 *)emb[1], (const float4 *)emb[2], occ, n,
                                                  (float4 *)out[0], (float4 *)out[1], (float4 *)out[2]);
    LAUNCH_CHECK();
    return GPCC_OK;
}

struct SymPtrs { const uint8_t *s[4]; };

__global__ __launch_bounds__(TB) void k_stage_input_dec(const float4 *__restrict__ X, const float4 *__restrict__ emb, SymPtrs sp, const uint32_t *__restrict__ m2r,
                                                        int stage, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    uint32_t prev = sp.s[0][r];
    if (stage >= 2) prev = prev * 2 + sp.s[1][r];
    if (stage >= 3) prev = prev * 4 + sp.s[2][r];
    out[t] = add4(X[t], emb[prev * 8 + (t & 7)]);
}
int stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out, int C)
{
    if (C != 32) return any_stage_input_dec(st, X, emb, sym_r, m2r, stage, n, out, C);
    SymPtrs sp = {{sym_r[0], sym_r[1], sym_r[2], nullptr}};
    k_stage_input_dec<<<nblk(n * 8), TB, 0, st>>>((const float4 *)X, (const float4 *)emb, sp, m2r, stage, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_assemble_occ(SymPtrs sp, const uint32_t *__restrict__ m2r, int64_t n, uint8_t *__restrict__ occ)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    occ[i] = (uint8_t)(sp.s[0][r] * 128 + sp.s[1][r] * 64 + sp.s[2][r] * 16 + sp.s[3][r]);
}
int assemble_occ(hipStream_t st, const uint8_t *const sym_r[4], const uint32_t *m2r, int64_t n, uint8_t *occ)
{
    SymPtrs sp = {{sym_r[0], sym_r[1], sym_r[2], sym_r[3]}};
    k_assemble_occ<<<nblk(n), TB, 0, st>>>(sp, m2r, n, occ);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_rows_permute(const float *__restrict__ in, float *__restrict__ out, int64_t n, int to_physical)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * 32) return;
    const int64_t i = t >> 5;
    const int c = (int)(t & 31);
    if (to_physical) out[i * 32 + phys_of(c)] = in[t];
    else out[t] = in[i * 32 + phys_of(c)];
}
int rows_permute(hipStream_t st, const float *in, float *out, int64_t n, int to_physical)
{
    k_rows_permute<<<nblk(n * 32), TB, 0, st>>>(in, out, n, to_physical);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// ------------------------------------------------------------------ heads
template <int M, int MODE>
__global__ __launch_bounds__(TB) void k_head(HeadArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= a.n) return;
    float x[32];
    {
        const float4 *__restrict__ px = reinterpret_cast<const float4 *>(a.x + (size_t)i * 32);
        float raw[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 v = px[g];
            raw[4 * g] = v.x; raw[4 * g + 1] = v.y; raw[4 * g + 2] = v.z; raw[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) x[c] = MODE == 2 ? raw[c] : raw[phys_of(c)];
    }
    const float *__restrict__ W1 = a.w1;
    const float *__restrict__ B1 = a.b1;
    const float *__restrict__ W2 = a.w2;
    const float *__restrict__ B2 = a.b2;
    float hdn[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float acc = B1[c];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(x[k], W1[c * 32 + k], acc);
        hdn[c] = acc > 0.0f ? acc : 0.0f;
    }
    float z[M];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float acc = B2[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(hdn[k], W2[j * 32 + k], acc);
        z[j] = acc;
        mx = acc > mx ? acc : mx;
    }
    head_tail<M, MODE>(a, i, z, mx, blockIdx.x);
}

// Encode / decode heads on the matrix pipe: network_dev.hpp: head_wave (a wave takes 64 nodes as four 16-row tiles).
template <int M, int MODE>
__global__ __launch_bounds__(TB) void k_head_mfma(HeadArgs a)
{
    __shared__ __attribute__((aligned(16))) float sm[(TB / 64) * HEAD_LDS_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    head_wave<M, MODE>(a, ((int64_t)blockIdx.x * (TB / 64) + wave) * 64, a.n, lane, sm + wave * HEAD_LDS_FLOATS, blockIdx.x);
}

template <int MODE>
static int head_launch(hipStream_t st, const HeadArgs &a)
{
    if (MODE != 2 && a.frag) {
        const unsigned g = (unsigned)cdiv(a.n, TB);   // 64 nodes per wave
        switch (a.stage_m) {
        case 2: k_head_mfma<2, MODE><<<g, TB, 0, st>>>(a); break;
        case 4: k_head_mfma<4, MODE><<<g, TB, 0, st>>>(a); break;
        case 16: k_head_mfma<16, MODE><<<g, TB, 0, st>>>(a); break;
        default: return fail(GPCC_ERR_ARG, "head width must be 2, 4 or 16");
        }
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    const unsigned g = nblk(a.n);
    switch (a.stage_m) {
    case 2: k_head<2, MODE><<<g, TB, 0, st>>>(a); break;
    case 4: k_head<4, MODE><<<g, TB, 0, st>>>(a); break;
    case 16: k_head<16, MODE><<<g, TB, 0, st>>>(a); break;
    default: return fail(GPCC_ERR_ARG, "head width must be 2, 4 or 16");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

int head_cdf(hipStream_t st, const HeadArgs &a)
{
    if (a.n <= 0) return GPCC_OK;
    if (a.C != 0 && a.C != 32) return any_head_cdf(st, a, a.C);
    if (a.mode == 0) return head_launch<0>(st, a);
    if (a.mode == 1) return head_launch<1>(st, a);
    return head_launch<2>(st, a);
}

}  // namespace gpcc
