// network.hpp -- the GausPcgc context network on gfx950 (network_ue_4stage_conv.py:15-94):
// embeddings, submanifold sparse convolution on fp32 MFMA, prediction heads + CDF integerisation.
//
// Feature rows are (n, 32) fp32 in a PHYSICAL channel order chosen so that the MFMA A-operand
// of v_mfma_f32_32x32x2_f32 is a contiguous 64-byte load per lane: physical position p holds
// logical channel 2p (p < 16) or 2(p-16)+1 (p >= 16).  The accumulation chain is still the
// logical order k = 0..31 (see oracle/gpcc_oracle.c "NORMATIVE NUMERICS").
#pragma once
#include "common.hpp"

struct gpcc_model {
    int C = 32, k = 5, K = 125;
    float *slab = nullptr;         // one allocation
    const float *prior_emb = nullptr;   // (256, 32) physical order
    const float *conv[18] = {0};        // (K, 64 lanes, 16) MFMA B-fragment order
    const float *temb = nullptr;        // (8, 32) physical
    const float *hw1[4] = {0}, *hb1[4] = {0}, *hw2[4] = {0}, *hb2[4] = {0};  // upstream layouts (logical)
    const float *semb[3] = {0};         // ({2,4,16}, 32) physical
};

namespace gpcc {

__host__ __device__ __forceinline__ int phys_of(int c) { return (c >> 1) + 16 * (c & 1); }   // logical -> physical
__host__ __device__ __forceinline__ int logical_of(int p) { return p < 16 ? 2 * p : 2 * (p - 16) + 1; }

constexpr int STAGE_M[4] = {2, 2, 4, 16};

struct ConvJob {
    const float *in;   // (n,32) physical
    const float *w;    // B-fragment order
    const float *res;  // nullable, physical
    float *out;        // physical
};
struct ConvBatch { ConvJob job[4]; };

// out = conv(in) (+res) (relu); up to 4 independent jobs on the same neighbour map in one launch
// ctx/level: when ctx->prof.on the launch is bracketed by HIP events tagged with `level` (bench.py roofline)
int sparse_conv(gpcc_ctx *ctx, int level, hipStream_t st, const ConvBatch &jobs, int njobs, const int32_t *nbrT, int64_t n, int K, int relu);
// fold the recorded events into ctx->prof (call after the stream is synchronised); pairs[level] = present neighbours
int prof_collect(gpcc_ctx *ctx, const unsigned long long *pairs_per_level, int nlevels);

// F[i] = Emb256[occ[i]]                                    (pcc_utils.py:99)
int embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out);
// X[i] = F[parent[i]] + Emb8[octant(i)]                    (kit/nn.py:77-98,108-117)
int child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out);
// stage input X + Emb_s[prev bits]; prev from ground-truth occupancy (encode) ...
int stage_input_gt(hipStream_t st, const float *X, const float *emb, const uint8_t *occ, int stage, int64_t n, float *out);
// ... or from the symbols decoded so far (raster order, looked up through m2r)
int stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out);

// Heads.  mode 0: encode -> lohi[m2r[i]] = c_low | (c_high-1) << 16 for the ground-truth symbol of `stage`
//         mode 1: decode -> cdf rows (Lp u16) at raster position m2r[i]
//         mode 2: test   -> prob (n,m) and cdf (n,Lp) in input order, x in LOGICAL channel order
struct HeadArgs {
    const float *x; int64_t n; int stage_m;
    const float *w1, *b1, *w2, *b2;
    const uint32_t *m2r; const uint8_t *occ; int stage;
    uint32_t *lohi; uint16_t *cdf; float *prob; int mode;
};
int head_cdf(hipStream_t st, const HeadArgs &a);

// occupancy byte from the four decoded symbol arrays (raster order) -> Morton order (pcc_utils.py:369)
int assemble_occ(hipStream_t st, const uint8_t *const sym_r[4], const uint32_t *m2r, int64_t n, uint8_t *occ);

// logical <-> physical row conversion (test entry points)
int rows_permute(hipStream_t st, const float *in, float *out, int64_t n, int to_physical);

}  // namespace gpcc
