// fused.hpp -- the decoder's small levels (up to FUSE_MAX_NODES nodes) as ONE persistent launch per chain of layers.
//
// Reference chain being replaced: HAC/utils/pcc_utils.py:283-372 -- per level: FCG + target embedding, target_resnet (5
// convolutions), then for each of the four stages conv-ReLU-conv, the head, the range decoder and the next stage's
// embedding; and prior_resnet (5 convolutions) on the finished level for the level below.  Round 3 ran that as ~32
// dependent launches per level on the first stream (18 of them 14-26 us convolutions on 16/32/64-row blocks whose 16-row
// tiles are 7-30 % full on the sparse small levels): 0.4-0.6 ms per level for 3 % of the nodes.
//
// Here a small level gets
//   * a PAIR PLAN (fused.hip: pairplan_build) instead of block-wise tile lists: the level's (row, neighbour) pairs grouped by
//     kernel offset ACROSS THE WHOLE LEVEL into tiles of 16 (fill ~95 % instead of 7-30 %), every pair with a slot in a
//     product buffer P ordered by (row, offset);
//   * convolution = two phases: PRODUCTS (a wave per tile: 16 MFMAs from a zero accumulator, the 16 x 32 products of the
//     tile's pairs go to their P rows) and SUMS (a thread per output row and channel quad adds the row's P rows in
//     ascending offset order) -- the normative sum, so every byte of every stream is unchanged;
//   * ONE persistent launch for the level's chain (k_level_fused<CHILD>: child features, 5 + 8 convolutions, 4 heads fused
//     into the sums of their convolution, 4 range-decoder phases, stage embeddings, occupancy assembly) and one for the
//     finished level's prior_resnet (k_level_fused<PARENT>), with an XCD-hierarchical grid barrier between phases
//     (tools/ubench/grid_sync.hip: 3.1 / 4.0 / 6.1 us per phase at 64 / 128 / 256 workgroups against 4.4-4.6 us per launch of
//     a minimal kernel in a chain -- a barrier is no cheaper than a launch boundary; the gain is the work per layer).
#pragma once
#include "network.hpp"
#include "octree.hpp"
#include "rangecoder.hpp"

namespace gpcc {

constexpr int64_t FUSE_MAX_NODES = 16384;        // default class boundary (GAUSPCC_FUSED_MAX, developer knob, moves it up to FUSE_HARD_MAX)
constexpr int64_t FUSE_HARD_MAX = 65536;         // what the plan builder and the persistent kernels are written for
#ifndef FUSE_THREADS_N
#define FUSE_THREADS_N 512
#endif
constexpr int FUSE_THREADS = FUSE_THREADS_N;   // 8 waves per workgroup (256 VGPRs a wave: the chain's phases -- MFMA tiles, 16-deep sums, three range decoders, three heads -- share one register allocation; at 16 waves / 128 VGPRs it spilled 93 of them)
constexpr int FUSE_HEAD_WAVES = 8;          // waves of a workgroup that may run a head (LDS: HEAD_LDS_FLOATS each)
constexpr size_t FUSE_LDS_BYTES = (size_t)FUSE_HEAD_WAVES * (512 + 1024) * 4;   // 48 KiB: head buffers / range-decoder byte windows (phases apart)

struct PairPlan {
    int64_t n = 0;
    int K = 0;
    uint32_t *rowstart = nullptr;   // [n + 1] first P row of every output row (its pairs in ascending offset order)
    int32_t *ot_j = nullptr;        // [tcap][16] neighbour row of a tile entry (padding: row 0)
    uint32_t *ot_q = nullptr;       // [tcap][16] P row of a tile entry (padding: the dummy row pcap - 1)
    uint32_t *ot_o = nullptr;       // [tcap] kernel offset | valid entries << 16
    uint32_t *ntiles = nullptr;     // device word: tiles of the level
    int64_t tcap = 0;               // tile capacity (bound: K ceil(n / 16))
    int64_t pcap = 0;               // rows of P incl. the dummy row (bound: n K + 1)
    bool valid() const { return rowstart != nullptr; }
};

// state of the grid barrier of one launch (zeroed by the launch before it: two blocks alternate)
struct FusedBar {
    uint32_t xcc_count[8 * 32];     // one 128-byte line per XCC
    uint32_t xcc_gen[8 * 32];
    uint32_t members[8 * 32];
    uint32_t top[32], census[32], nxcc[32];
};

bool fused_enabled();                                   // GAUSPCC_FUSED (default 1; 0 = the launch-per-layer path of round 3; 2 = plan-based convolutions as separate launches)
int fused_mode();
bool fused_level_ok(int64_t n, int k);                  // does a level of n nodes run fused?

// Plan of level `chi` from its parent's cell map (side stream); also writes chi's own cell map (cell_own, nullable) for the
// level below, as the count pass of tiles_build does.  Arrays from the arena's bottom (kept: the level's chain and its prior
// trunk use them).  pairs_dev (nullable): += the level's (row, neighbour) pairs.
int pairplan_build(gpcc_ctx *ctx, hipStream_t st, const Level *par, const int32_t *cell_par, const Level *chi, int32_t *cell_own, int k, PairPlan *plan,
                   unsigned long long *pairs_dev);

// out = conv(in) (+res) (relu) on the plan, as two launches (products, sums): the debug / cross-check form of what the fused
// kernels do in two phases.  P: plan.pcap x 32 floats of scratch.
int plan_conv(hipStream_t st, const PairPlan &plan, const ConvJob &job, float *P, int relu);

struct FusedChild {
    // inputs
    const float *pA; int64_t np;     // parent level's trunk output (np, 32)
    const uint32_t *parent;          // (n) parent row
    const uint64_t *rkey;            // (n) raster key (octant bits)
    const uint32_t *m2r;             // (n) Morton row -> raster rank
    const uint8_t *bytes;            // the uploaded container
    const RcChunk *chunks;           // [4][nlanes] lane descriptors of the level's four streams
    uint32_t nlanes; int llog;       // lanes per stream, log2 symbols per lane
    // a batch's merged level (forest.hpp): the lanes of all scenes (nlanes = their total, chunks carry their own geometry), and per node
    // its CDF row slot and its symbol slot (nullptr: one scene -- both follow from m2r, nlanes and llog)
    const uint32_t *cpos = nullptr, *spos = nullptr;
    uint32_t win_bytes[4];           // longest byte window of a lane, per stage
    int coder;                       // the lanes' coder (rangecoder_dev.hpp: RC_CODER_*: container version 4 / versions 1-3)
    // work buffers (n, 32) and outputs
    float *cX, *cA, *cB, *cU, *P;
    uint16_t *cdf;                   // rc_rows_capacity(nlanes, 2^llog) * 16 u16
    uint8_t *sym[4];                 // (n + 4) each, raster order
    uint8_t *occ;                    // (n) out: the level's occupancy, Morton order
};
// the whole chain of one coded level (pcc_utils.py:313-372) in one launch; the result cA (the level's target trunk output) is not
// needed afterwards -- only occ is
int fused_child_level(gpcc_ctx *ctx, hipStream_t st, const gpcc_model *m, const PairPlan &plan, const FusedChild &a);
// can the range-decoder phases of the level keep their lanes' byte windows in the fused kernel's LDS?  (always, for containers this
// library wrote: 64-symbol lanes; a foreign or corrupt table may claim more)
bool fused_windows_fit(int64_t n, int64_t np, uint32_t nlanes, const uint32_t win_bytes[4]);
// a context is going away: its event leaves the per-device chain of persistent launches (api.hip: gpcc_ctx_destroy)
void fused_ctx_release(gpcc_ctx *ctx);
// the context's sticky timeout word (device) -- the caller copies it out at its final sync; fused_reset: after a timeout
uint32_t *fused_timeout_word(gpcc_ctx *ctx);
int fused_reset(gpcc_ctx *ctx, hipStream_t st);
// prior_resnet of a finished level (pcc_utils.py:99-101): F = Emb256[occ] -> 5 convolutions; result in pA
// np_of_level: nodes of the level ABOVE it (the density hint of fused_grid; the same value the level's own chain was launched with)
int fused_parent_trunk(gpcc_ctx *ctx, hipStream_t st, const gpcc_model *m, const PairPlan &plan, int64_t np_of_level, const uint8_t *occ, float *pF, float *pA, float *pB, float *P);

}  // namespace gpcc
