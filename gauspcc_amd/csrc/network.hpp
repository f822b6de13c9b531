// network.hpp -- the GausPcgc context network on gfx950 (network_ue_4stage_conv.py:15-94):
// embeddings, submanifold sparse convolution on fp32 MFMA, prediction heads + CDF integerisation.
//
// Feature rows are (n, 32) fp32 in a PHYSICAL channel order chosen so that the MFMA A-operand
// of v_mfma_f32_16x16x4_f32 is two 16-byte loads per lane that are contiguous across the four
// lane groups: lane group g = lane/16 supplies k = 4*kk + g for the 8 k-steps kk; physical position
// 4*g + kk (kk < 4) or 16 + 4*g + (kk - 4) holds logical channel 4*kk + g, so the four lanes that
// gather one row read its first 64 bytes with the first load and the second 64 with the second.
// The accumulation chain is still the logical order k = 0..31 (oracle/gpcc_oracle.c
// "NORMATIVE NUMERICS").
#pragma once
#include "common.hpp"

struct gpcc_model {
    int C = 32, k = 5, K = 125;
    float *slab = nullptr;         // one allocation
    const float *prior_emb = nullptr;   // (256, 32) physical order
    const float *conv[18] = {0};        // (K, 2 halves, 64 lanes, 8) MFMA B-fragment order, followed by the same in the transposed (A-fragment) order and in the 32x32x2 A order
    const float *temb = nullptr;        // (8, 32) physical
    const float *hw1[4] = {0}, *hb1[4] = {0}, *hw2[4] = {0}, *hb2[4] = {0};  // upstream layouts (logical)
    const float *hfrag[4] = {0};        // per head: W1 as 2 x 512 B-fragment floats, W2 (columns padded to 16) as 512, b1 (32), b2 padded (16)
    const float *semb[3] = {0};         // ({2,4,16}, 32) physical
};

namespace gpcc {

__host__ __device__ __forceinline__ int phys_of(int c)   // logical -> physical
{
    const int kk = c >> 2, g = c & 3;
    return kk < 4 ? 4 * g + kk : 16 + 4 * g + (kk - 4);
}
__host__ __device__ __forceinline__ int logical_of(int p)
{
    const int kk = 4 * (p >> 4) + (p & 3), g = (p & 15) >> 2;
    return 4 * kk + g;
}

// (K, C, C) upstream kernel -> B fragments, 4 x 1 KiB fully coalesced loads per offset:
// [o][half][q][lane][r] = W[o][4*(4q + r) + lane/16][16*half + lane%16]     (k-step kk = 4q + r)
inline void conv_weight_fragments(const float *W, int K, float *out)
{
    for (int o = 0; o < K; ++o)
        for (int hh = 0; hh < 2; ++hh)
            for (int q = 0; q < 2; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r)
                        out[((((size_t)o * 2 + hh) * 2 + q) * 64 + lane) * 4 + r] = W[((size_t)o * 32 + (4 * (4 * q + r) + (lane >> 4))) * 32 + 16 * hh + (lane & 15)];
}

// The same kernel as the A operand of the TRANSPOSED product D^T = W^T X^T (the asm tile loop): MFMA row m of output half
// hh is logical output channel 16*hh + 4*(m%4) + m/4, so that the four accumulator registers of lane (g = lane/16, e) are
// four physically consecutive channels of tile row e -- one 16-byte LDS access per half instead of four 4-byte ones.
// [o][half][q][lane][r] = W[o][4*(4q + r) + lane/16][16*half + 4*((lane%16)%4) + (lane%16)/4]
inline void conv_weight_fragments_t(const float *W, int K, float *out)
{
    for (int o = 0; o < K; ++o)
        for (int hh = 0; hh < 2; ++hh)
            for (int q = 0; q < 2; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int m = lane & 15;
                        out[((((size_t)o * 2 + hh) * 2 + q) * 64 + lane) * 4 + r] = W[((size_t)o * 32 + (4 * (4 * q + r) + (lane >> 4))) * 32 + 16 * hh + 4 * (m & 3) + (m >> 2)];
                    }
}

constexpr int STAGE_M[4] = {2, 2, 4, 16};

struct ConvJob {
    const float *in;   // (n,32) physical
    const float *w;    // K x 1024 floats in B-fragment order, then K x 1024 in the transposed order (conv_weight_fragments_t)
    const float *res;  // nullable, physical
    float *out;        // physical
};
struct ConvBatch { ConvJob job[4]; int C = 0; };   // C: channels (0 = 32, the MFMA kernels; 16 / 64: network_any.hip, weights (K, C, C) and rows in logical order)

// Compacted work list of a SET of levels (one level in the decoder; all parent levels / all coded levels in the encoder,
// which batches them into one launch per layer): every wave owns a block of up to H consecutive (Morton-ordered) rows of ONE
// level; for each kernel offset the (output row, neighbour row) pairs of the block are packed into tiles of 16 rows (one
// v_mfma_f32_16x16x4_f32 M-tile), offsets ascending.  Built once per tree (tiles.hip: straight from the parent level's
// structure, no dense neighbour map), used by all 5 / 13 convolutions that run on a level.  Neighbour rows are indices
// inside the block's own level; lv_row0 places a level in the set's feature arrays.  R is the capacity class (LDS rows per
// wave, selects the kernel: 16 = the cooperative kernel, 255 = the wave-serial kernel at one wave per SIMD; 32 / 64 / 96 /
// 128 remain for the variant tests), H <= R the block height (conv_pick_rows / conv_pick_height, HISTORY.md section 4).
constexpr int CONV_R_MAX = 255;
constexpr int CONV_HDR_PAD = 48;  // tiles a wave may read past the end of its block's list (two header batches + look-ahead)
struct ConvTiles {
    int32_t *tj = nullptr;     // [tiles][16] neighbour row inside the level (padding: 0, a valid row whose result is discarded)
    uint8_t *tr = nullptr;     // [tiles][16] LDS slot of the output row = row inside the block + 1 (padding: 0 = the dummy slot)
    uint32_t *toc = nullptr;   // [tiles]     offset | valid entries << 16
    uint32_t *first = nullptr; // [pool blocks + 1]  tile range of each block, indexed by the block's id in the pool
    uint32_t *order = nullptr; // [nblk]      the set's block ids sorted by tile count, longest first (dispatch order)
    int64_t nblk = 0;          // blocks of the set
    int R = CONV_R_MAX;        // capacity class of the blocks (LDS rows per wave; selects the kernel)
    int H = CONV_R_MAX;        // rows per block, <= R
    int K = 0;                 // kernel offsets
    int paired = 0;            // blocks may hold PAIRED lists: every (block, offset) run an even number of tiles (odd runs end in an empty
                               // tile) for the pair-step loop; pflag[block id in the pool] says which do
    const uint8_t *pflag = nullptr;
    int nlv = 0;               // levels of the set
    uint32_t lv_blk0[MAXLV + 1] = {0};  // pool id of each level's first block; lv_blk0[nlv] = one past the set's last block
    uint32_t lv_row0[MAXLV] = {0};      // first row of each level in the set's feature arrays
    uint32_t lv_rows[MAXLV] = {0};      // rows of each level
};
int conv_pick_rows(int64_t n, int k = 5);  // policy (env GAUSPCC_CONV_R overrides)
bool conv_is_coop(int64_t n, int R);     // does a level of n nodes at block class R run the cooperative kernel (16 / 32 / 64-row blocks, H = R)?
int conv_pick_height(int64_t n, int R);   // rows per block for capacity class R (env GAUSPCC_CONV_BALANCE=0: H = R)

// Tile lists of several levels in one pool (tiles.hip).  Level l is built from its parent level's cell map (par == nullptr:
// a base level of < 64 nodes, searched directly); cell_own, when not null, receives the level's own cell map
// [cell_map_entries(k)][n] for the level below it.
struct Level;
struct TileLevel { const Level *lv; const Level *par; const int32_t *cell_par; int32_t *cell_own; };
struct TilePool {
    int32_t *tj = nullptr; uint8_t *tr = nullptr; uint32_t *toc = nullptr; uint32_t *first = nullptr;
    int64_t nblk = 0;
    int R = 0, H = 0, K = 0, nlv = 0, paired = 0;
    uint8_t *pflag = nullptr;
    uint32_t lv_blk0[MAXLV + 1] = {0}, lv_rows[MAXLV] = {0};
    // algorithmic HBM traffic of the build (gpcc_profile stages): per node 12 B of its own structure + its cell map written
    // (4 B a cell) + per parent its cell map row read with child start / occupancy (9 B a cell) + 84 B per tile written;
    // both passes recompute, the figure counts ONE pass.  Tiles are counted when the pool was sized by a sync.
    double alg_bytes = 0.0;
};
int cell_map_entries(int k);
// count pass per level -> scan -> (one stream sync unless H <= 16) -> fill pass per level.  pairs_dev (nullable): [nlv]
// accumulators, += the (row, present neighbour) pairs of each level.  Arrays come from the arena.
int tiles_build(gpcc_ctx *ctx, hipStream_t st, const TileLevel *lv, int nlv, int k, int R, int H, TilePool *pool, unsigned long long *pairs_dev);
// small levels (fused.hip): dense neighbour map + per-(offset, row) ranks + per-row counts; writes the level's own cell map too
int tiles_dense_map(hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int32_t *cell_own, int k, int32_t *nbr, uint16_t *rk, uint32_t *rowcnt);
// the levels [l0, l1) of a pool as the work list of one set; row_base[l - l0] = first row of level l in the set's arrays
int tiles_view(gpcc_ctx *ctx, hipStream_t st, const TilePool &pool, int l0, int l1, const int64_t *row_base, ConvTiles *T);

// out = conv(in) (+res) (relu); up to 4 independent jobs on the same tile list in one launch
// ctx/level: when ctx->prof.on the launch is bracketed by HIP events tagged with `level` (bench.py roofline)
int sparse_conv(gpcc_ctx *ctx, int level, hipStream_t st, const ConvBatch &jobs, int njobs, const ConvTiles &T, int64_t n, int relu);
// profiling only: the sparse_conv calls between the two share one pair of timing events (same level, nothing else enqueued between)
int conv_chain_begin(gpcc_ctx *ctx, hipStream_t st);
int conv_chain_end(gpcc_ctx *ctx, hipStream_t st);
// fold the recorded events into ctx->prof (call after the stream is synchronised); pairs[level] = present neighbours
int prof_collect(gpcc_ctx *ctx, const unsigned long long *pairs_per_level, int nlevels);

// F[i] = Emb256[occ[i]]                                    (pcc_utils.py:99)
int embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out, int C = 32);
// X[i] = F[parent[i]] + Emb8[octant(i)]                    (kit/nn.py:77-98,108-117)
int child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out, int C = 32);
// stage input X + Emb_s[prev bits]; prev from ground-truth occupancy (encode) ...
int stage_inputs_gt(hipStream_t st, const float *X, const float *const emb[3], const uint8_t *occ, int64_t n, float *const out[3], int C = 32);   // stages 1..3 in one pass
// ... or from the symbols decoded so far (raster order, looked up through m2r)
int stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out, int C = 32);

// Heads.  mode 0: encode -> lohi[pos] = c_low | (c_high-1) << 16 for the ground-truth symbol of `stage`
//         mode 1: decode -> compact cdf row (interior values only, rc_row_stride u16) at row pos
//         mode 2: test   -> prob (n,m) and cdf (n,Lp) in input order, x in LOGICAL channel order
// pos = rc_interleaved(m2r[i], chunk_log2, nch): raster rank -> chunk-interleaved slot of the stream
constexpr int HEAD_FRAG_FLOATS = 1024 + 512 + 32 + 16;
struct HeadArgs {
    const float *x; int64_t n; int stage_m;
    const float *w1, *b1, *w2, *b2;
    const float *frag;   // modes 0 / 1: the head's MFMA fragments (gpcc_model::hfrag); mode 2 uses w1 .. b2
    const uint32_t *m2r; const uint8_t *occ; int stage;
    uint32_t *lohi; uint16_t *cdf; float *prob; int mode;
    int chunk_log2; uint32_t nch;
    // mode 0 over several concatenated levels: per-row slot of stage 0 and per-row stride between stages
    // (then m2r / chunk_log2 / nch are unused); mode 1 with pos: the row slot of every node (a batch's merged level)
    const uint32_t *pos; const uint32_t *slots;
    // mode 0, optional: 16 accumulators of sum clamp(-log2(p_gt + 1e-10), 0, 50) (network_ue_4stage_conv.py:176-179)
    double *bits;
    int C;   // channels (0 = 32); other widths: x in logical order, w1 .. b2 used (network_any.hip)
};
int head_cdf(hipStream_t st, const HeadArgs &a);

// occupancy byte from the four decoded symbol arrays (raster order) -> Morton order (pcc_utils.py:369)
int assemble_occ(hipStream_t st, const uint8_t *const sym_r[4], const uint32_t *m2r, int64_t n, uint8_t *occ);

// channel counts other than 32 (network_any.hip): the same arithmetic as plain kernels in logical channel order
int any_embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out, int C);
int any_child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out, int C);
int any_stage_inputs_gt(hipStream_t st, const float *X, const float *const emb[3], const uint8_t *occ, int64_t n, float *const out[3], int C);
int any_stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out, int C);
int any_sparse_conv(hipStream_t st, const ConvBatch &jobs, int njobs, const ConvTiles &T, int64_t n, int relu, int C);
int any_head_cdf(hipStream_t st, const HeadArgs &a, int C);

// logical <-> physical row conversion (test entry points)
int rows_permute(hipStream_t st, const float *in, float *out, int64_t n, int to_physical);

}  // namespace gpcc
