// codec_shared.hpp -- what gpcc_encode / gpcc_decode (codec.hip) and their batched forms (codec_batch.hip) have in common: the
// Conv-ReLU-ResNet-ResNet trunk, the per-row metadata of the encoder's two level sets, little-endian header fields.
#pragma once
#include "fused.hpp"
#include "network.hpp"
#include "octree.hpp"
#include "primitives.hpp"

namespace gpcc {

struct Trunk { float *x, *a, *b; };

// Conv-ReLU-ResNet-ResNet (network_ue_4stage_conv.py:17-33; kit/nn.py:18-22).  Result in t.a.
static inline int run_trunk(gpcc_ctx *ctx, int level, hipStream_t st, const gpcc_model *m, int conv0, const Trunk &t, const ConvTiles &tiles, int64_t n, const PairPlan *plan = nullptr,
              float *P = nullptr)
{
    ConvBatch cb = {}; cb.C = m->C;
    auto one = [&](const float *in, int ci, const float *res, float *out) {
        cb.job[0] = ConvJob{in, m->conv[ci], res, out};
        if (plan) return plan_conv(st, *plan, cb.job[0], P, 1);
        return sparse_conv(ctx, level, st, cb, 1, tiles, n, 1);
    };
    GP_TRY(conv_chain_begin(ctx, st));
    GP_TRY(one(t.x, conv0, nullptr, t.a));
    GP_TRY(one(t.a, conv0 + 1, nullptr, t.b));
    GP_TRY(one(t.b, conv0 + 2, t.a, t.x));
    GP_TRY(one(t.x, conv0 + 3, nullptr, t.b));
    GP_TRY(one(t.b, conv0 + 4, t.x, t.a));
    GP_TRY(conv_chain_end(ctx, st));
    return GPCC_OK;
}

static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t get32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// encode keeps the whole tree resident: ~3 nodes per point, per node its cell map (27 x 4 B) + tile lists (~150 B) + ~12
// feature rows of 128 B + level arrays; grown and retried when a cloud needs more
static inline size_t arena_estimate(int64_t n, int K) { return (size_t)n * 3 * (size_t)(4 * 125 + 300 + 12 * 128 + 96) + (size_t)n * 64 + (size_t)K * 4096 + ((size_t)64 << 20); }

// Per-row metadata of the two concatenated sets, every level in one launch.  Level d lives in the prior set P at rows
// pb[d].. (d <= L-2) and in the target set C at rows cbase[d].. (d >= 1).
struct SetLevels {
    int L;
    uint32_t n[MAXLV], pb[MAXLV], cbase[MAXLV], lohi_base[MAXLV], slots[MAXLV], nch[MAXLV];
    int clog[MAXLV];
    const uint8_t *occ[MAXLV];
    const uint64_t *rkey[MAXLV];
    const uint32_t *parent[MAXLV], *m2r[MAXLV];
};

// occupancy of both sets, raster keys and global parent rows of C: what the network needs (no ranks)
static __global__ __launch_bounds__(256) void k_set_rows(SetLevels S, int64_t nP, int64_t nC, uint8_t *__restrict__ occP, uint8_t *__restrict__ occC, uint64_t *__restrict__ rkeyC,
                                                  uint32_t *__restrict__ parentC)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nP) {
        int d = 0;
        for (int q = 1; q + 1 < S.L; ++q) d = i >= (int64_t)S.pb[q] ? q : d;
        occP[i] = S.occ[d][i - S.pb[d]];
    }
    if (i < nC) {
        int d = 1;
        for (int q = 2; q < S.L; ++q) d = i >= (int64_t)S.cbase[q] ? q : d;
        const int64_t j = i - S.cbase[d];
        occC[i] = S.occ[d][j];
        rkeyC[i] = S.rkey[d][j];
        parentC[i] = S.pb[d - 1] + S.parent[d][j];
    }
}

}  // namespace gpcc
