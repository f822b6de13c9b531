// octree.hpp -- device octree of a voxelised point set (the FOG/FCG machinery of
// kit/nn.py:25-98 without torchsparse), in Morton order internally, with the
// per-level raster ranks the bitstream order needs (kit/op.py:17-30) and the k^3
// neighbour maps the sparse convolutions consume.
#pragma once
#include "common.hpp"

namespace gpcc {

// 21-bit-per-axis bit interleave, z most significant inside each triple (octant index
// x | y<<1 | z<<2, kit/nn.py:43-45,64-73).
__host__ __device__ __forceinline__ uint64_t part1by2(uint64_t v)
{
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__host__ __device__ __forceinline__ uint32_t compact1by2(uint64_t v)
{
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
    v = (v | (v >> 16)) & 0x1F00000000FFFFull;
    v = (v | (v >> 32)) & 0x1FFFFFull;
    return (uint32_t)v;
}
__host__ __device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    return part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2);
}
// raster key of BIASED coordinates (each < 2^21): z | y | x, 21 bits per field
__host__ __device__ __forceinline__ uint64_t rkey3(uint32_t x, uint32_t y, uint32_t z)
{
    return ((uint64_t)z << 42) | ((uint64_t)y << 21) | (uint64_t)x;
}
__host__ __device__ __forceinline__ uint32_t rk_x(uint64_t k) { return (uint32_t)(k & 0x1FFFFF); }
__host__ __device__ __forceinline__ uint32_t rk_y(uint64_t k) { return (uint32_t)((k >> 21) & 0x1FFFFF); }
__host__ __device__ __forceinline__ uint32_t rk_z(uint64_t k) { return (uint32_t)((k >> 42) & 0x1FFFFF); }

// One stored level, Morton order.  Internal (biased) coordinate b = (c + Tree::bias[axis]) >> lvl >= 0, lvl = the number of
// halvings from the input resolution; the bias is a multiple of 2^L per axis, so b = floor(c / 2^lvl) + (bias >> lvl)
// exactly and the tree is the reference's tree of the absolute coordinates.
struct Level {
    int64_t n = 0;
    int lvl = 0;            // halvings from the leaves (leaves = 0)
    uint64_t *rkey = nullptr;   // (n) raster key of biased coordinates
    uint8_t *occ = nullptr;     // (n) occupancy byte
    uint32_t *cstart = nullptr; // (n) index of the first child in the next finer level
    uint32_t *parent = nullptr; // (n) index of the parent in the next coarser level
    uint32_t *m2r = nullptr;    // (n) Morton index -> raster rank
    uint32_t *r2m = nullptr;    // (n) raster rank  -> Morton index
    // decoder: the byte span that holds ALL of the arrays above when they were carved back to back (codec.hip: alloc_level);
    // nullptr = no such guarantee (level_expand_rank then zeroes array by array)
    char *span0 = nullptr;
    size_t span_bytes = 0;
};

struct Tree {
    int L = 0;             // stored levels: lv[0] base ... lv[L-1] parents of the leaves
    int hb = 0;            // varying low bits per axis at leaf resolution
    Level lv[MAXLV];
    int64_t npts = 0;
    uint64_t *leaf_mkey = nullptr;  // (npts) sorted Morton keys of the input points
    // Origin of the internal 21-bit coordinate frame, per axis: internal = c + bias.  2^20 for clouds inside (-2^20, 2^20);
    // otherwise minus the cloud's minimum rounded down to a multiple of 2^ceil_log2(extent) (>= 2^L), which keeps every
    // level's floor-halving exact -- only the EXTENT has to fit, the position is any int32.
    int64_t bias[3] = {CB, CB, CB};
};
struct Bias3 { int64_t v[3]; };
// bias of a cloud from its bounding box (codec and oracle agree on the tree, not on the bias: any multiple of 2^L serves)
int tree_pick_bias(const int32_t mn[3], const int32_t mx[3], int64_t bias_out[3]);

// algorithmic HBM traffic of tree_build (gpcc_profile stages): input 12 B / point, the leaf key sort (ceil(3 hb / 8) passes
// x 8 B x read + write), per level the pass over the finer keys (8 B) + its arrays (25 B / node), and the raster ranks
// (ceil(3 hb_l / 8) passes x 12 B x 2 + 8 B / node for the inverse permutation)
double tree_alg_bytes(const Tree &T);
// encode side: build every level bottom-up from the raw points (one bbox sync + one counts sync)
int tree_build(gpcc_ctx *ctx, hipStream_t st, const int32_t *xyz_dev, int64_t n, Tree *T);
int tree_ranks(gpcc_ctx *ctx, hipStream_t st, Tree *T);   // Level::m2r / r2m of every level; tree_build leaves them unset

// raster ranks (m2r / r2m) of one level
int level_raster_rank(gpcc_ctx *ctx, hipStream_t st, Level *lv, int hb_level);

// the same ranks without a sort, from the parent level's ranks, occupancy and child starts (octree.hip: one scan over a
// raster-order walk of the parents); rank_level picks: a sort for levels of at most 1024 nodes (one launch), this otherwise
int level_ranks_from_parent(gpcc_ctx *ctx, hipStream_t st, const Level *par, Level *chi);
int rank_level(gpcc_ctx *ctx, hipStream_t st, const Level *par, Level *chi, int hb_level);

// decode side: children of `par` (occupancy known) -> `chi` (rkey, parent; n must be known)
int level_expand(gpcc_ctx *ctx, hipStream_t st, Level *par, Level *chi, uint32_t *total_dev, bool chi_zeroed = false);   // chi_zeroed: the caller has cleared the child arrays
// level_expand + rank_level; small levels (<= 1 k parents, <= 8 k children) in ONE single-workgroup launch (GAUSPCC_SMALL_FUSE=0: never)
int level_expand_rank(gpcc_ctx *ctx, hipStream_t st, Level *par, Level *chi, uint32_t *total_dev, int hb_level);

// leaves of the last level in the reference's decoder order (parents in raster order, octants ascending)
int leaves_reference_order(gpcc_ctx *ctx, hipStream_t st, const Level *last, const int64_t bias[3], int32_t *xyz_out, int64_t npts);
// copy a level to host-visible buffers in raster order: coords (n,3) int32 (un-biased), occ (n)
int level_to_raster(gpcc_ctx *ctx, hipStream_t st, const Level *lv, const int64_t bias[3], int32_t *xyz_out_dev, uint8_t *occ_out_dev);

}  // namespace gpcc
