// forest.hpp -- K scenes as ONE virtual octree: the batch axis of the codec (gpcc_encode_batch / gpcc_decode_batch).
//
// Reference: the codec's coordinate tensor has a batch column -- coords = [b, x, y, z] (HAC/utils/pcc_utils.py:73 pins b = 0),
// sort_CF orders by batch last (GausPcgc/kit/op.py:17-30) and the stand-alone CLI loops over files
// (GausPcgc/compress_ue_4stage_conv.py:72-75).  A scene is a chain of ~60 dependent launches per level whatever its size, so
// small scenes leave the device idle; here the K chains become one.
//
// Layout.  Scenes are aligned at their BASE levels (solo level d of every scene is merged level d) and ordered by depth,
// deepest first, so that the scenes present at level d are a prefix [0, Kd[d]) and those that end there (their children are
// points, not nodes) a suffix of it.  A merged level stores its scenes one after the other (scene-major), Morton order inside
// a scene.  Every scene keeps its own solo frame in x and y; in z the scenes are stacked in disjoint slabs of BASE units
// (level d coordinates = solo coordinates + (tz << d)), so the merged raster (z, y, x) order is scene-major as well and a
// node's rank inside its scene is its merged rank minus the scene's first row: m2r / r2m map a scene's block onto itself.
// Isolation is structural, not geometric: the base nodes hang under a virtual ROOT level (their would-be parents) whose cell
// map is searched per scene, and every finer level finds its neighbours through its parent's cells (tiles.hip) -- a node
// never sees another scene's nodes, so every scene's convolutions, CDFs and bytes are its solo ones, bit for bit.
#pragma once
#include <vector>

#include "octree.hpp"

namespace gpcc {

constexpr int FOREST_MAX_SCENES = 256;
// internal status: the scenes cannot share one tree (coordinate budget, depth, key width): the caller codes them one by one
constexpr int FOREST_UNFIT = -2000;

struct ForestScene {
    int user = 0;                 // index in the caller's order
    int L = 0;                    // stored levels (base + coded)
    int64_t npts = 0;
    int64_t n[MAXLV] = {0};       // nodes per stored level, base first
    int64_t nroot = 0;
    int64_t bias[3] = {0, 0, 0};  // leaf frame: merged leaf coordinate = c + bias (z: the slab translation included); a multiple of 2^L
};

// per (level, scene) record the kernels look rows up in (device copy: Forest::seg_dev[d], Kd[d] + 1 entries; the last one
// is a sentinel with row0 = N(d))
struct ForestSeg {
    uint32_t row0;        // first row = first raster rank of the scene at this level
    uint32_t lane0;       // coder: first lane of the scene among the level's lanes (one stage)
    uint32_t nlanes;      // coder: lanes of the scene's stream
    uint32_t base;        // encoder: first lohi word of the scene's stage-0 stream; decoder: first symbol slot of the scene
    uint32_t slots;       // encoder: words per stage stream (nlanes << llog)
    int32_t llog;         // lane size log2 (0: one lane holds the stream)
    uint32_t pad0, pad1;
};

struct Forest {
    int K = 0, L = 0;                       // scenes (internal order), merged levels
    std::vector<ForestScene> sc;            // internal order: L descending, the caller's order on a tie
    Tree T;                                 // merged levels (T.L = L)
    Level root;                             // virtual parents of the base nodes (rkey, occ, cstart; no ranks)
    int32_t *cell_root = nullptr;           // [cell_map_entries(k)][root.n]
    int Kd[MAXLV + 2] = {0};                // scenes with more than d levels; Kd[L] = 0
    std::vector<uint32_t> row0[MAXLV];      // [Kd[d] + 1] first row of each scene, then N(d)
    std::vector<uint32_t> root0;            // [K + 1]
    ForestSeg *seg_dev[MAXLV] = {nullptr};  // device copies (uploaded by forest_upload_segs)
    int hb0 = 1;                            // bits of the base level's coordinates (z slabs included)
    int64_t npts = 0;
    // rows of level d that have children in level d + 1 (a prefix); the rest are the last levels of their scenes
    int64_t inner(int d) const { return d + 1 < L ? (int64_t)row0[d][(size_t)Kd[d + 1]] : 0; }
};

// A level seen as the parent of the next one: its inner rows only (the ranks of a prefix of scenes are a prefix of ranks).
inline Level forest_parent_view(const Forest &F, int d) { Level v = F.T.lv[d]; v.n = F.inner(d); return v; }

// slab placement shared by encoder and decoder: scene q's base nodes span [zlo, zhi] (its own frame) -> even translation tz;
// *zcur = first free base-unit z, advanced.  Returns false when the scene's finest level would leave 21 bits.
bool forest_place(int L, int64_t zlo, int64_t zhi, int64_t *zcur, int64_t *tz);

// encode side: per-scene bounding boxes and level histograms (two syncs), then the merged levels bottom-up.  xyz: K device
// pointers (host array).  kernel_size: of the convolutions (the root's cell map).  On GPCC_ERR_DUPLICATE *bad_scene names the scene.
int forest_build(gpcc_ctx *ctx, hipStream_t st, const int32_t *const *xyz, const int64_t *n, int K, int kernel_size, Forest *F, int *bad_scene);
// cell map of the root level: neighbours within +-PR searched inside the scene (bounds: root0 on the device)
int forest_root_cells(gpcc_ctx *ctx, hipStream_t st, const Level *root, const uint32_t *root0_dev, int K, int kernel_size, int32_t *cell_root);
// raster ranks of every merged level (the encoder runs this on its second stream)
int forest_ranks(gpcc_ctx *ctx, hipStream_t st, Forest *F);
// device copies of the per-level scene records
int forest_upload_segs(gpcc_ctx *ctx, hipStream_t st, Forest *F, const std::vector<ForestSeg> seg[MAXLV], uint8_t *pinned, size_t pinned_bytes);

// decoder: pos[i] = CDF row slot of node i of level d: lane0 + (r >> llog) + (r & mask) * nch_total, r = rank inside the scene;
// spos[i] = its symbol slot: the scene's block (ForestSeg::base, a multiple of 16 with 4 bytes of slack behind every scene: the
// staged range decoders store 16 / 4 symbols at a time) + r
int forest_cdf_pos(hipStream_t st, const ForestSeg *seg_dev, int nseg, const uint32_t *m2r, int64_t n, uint32_t nch_total, uint32_t *pos, uint32_t *spos);
// decoder: *bad_dev |= 1 unless every scene's nodes of the parent level expanded to exactly the scene's rows of the child level
int forest_check_bounds(hipStream_t st, const uint32_t *cstart_par, const ForestSeg *seg_par, const ForestSeg *seg_chi, int nchi, uint32_t *bad_dev);
// pinned bytes forest_build stages its tables in (ctx->hbatch; the caller reserves its own needs on top BEFORE the call)
size_t forest_build_pinned(int K);

// leaves of the scenes that end at level d (a suffix of its rows), each scene into its own buffer in the reference's order
struct ForestLeafScene { uint32_t rank0; int32_t *xyz; int64_t cap; int64_t bias[3]; };
int forest_leaves(gpcc_ctx *ctx, hipStream_t st, const Level *lv, int64_t first_rank, const ForestLeafScene *scenes_dev, int nscenes, uint32_t *counts_dev);

}  // namespace gpcc
