// rangecoder.hpp -- device range coder launchers (see rangecoder.hip).
#pragma once
#include "common.hpp"

namespace gpcc {

// One chunk = one lane.  Element t of the chunk (packed symbol word on encode, compact CDF row on
// decode) lives at index first + t * stride: chunks of one stream are interleaved so that the 64
// lanes of a wave touch consecutive addresses.
struct RcChunk {
    uint32_t first;     // index of element 0
    uint32_t stride;    // elements between consecutive symbols of this chunk (= chunks in the stream)
    uint32_t n;         // symbols in the chunk
    uint32_t out;       // decode: index of the chunk's first symbol in the (raster-ordered) output
    uint32_t byte_off;  // decode: offset of the chunk's bytes in the uploaded file
    uint32_t nbytes;    // decode: byte count
};

// compact CDF row: only the interior values v[1..Lp-2] are stored (v[0] = 0, v[Lp-1] is never read)
static inline int rc_row_stride(int lp) { return lp == 3 ? 1 : lp == 5 ? 4 : 16; }  // uint16 units
// position of raster rank r inside a stream cut into 2^chunk_log2-symbol chunks (chunk_log2 = 0: one chunk)
__host__ __device__ __forceinline__ uint32_t rc_interleaved(uint32_t r, int chunk_log2, uint32_t nch)
{
    return chunk_log2 ? (r & ((1u << chunk_log2) - 1u)) * nch + (r >> chunk_log2) : r;
}

// The decoder fetches compact rows RC_ROW_LOOKAHEAD symbols ahead without clamping: the row buffer of a stream of
// nch chunks of (at most) S symbols must hold rc_rows_capacity(nch, S) rows (what lies past the last row is never used).
constexpr int RC_ROW_LOOKAHEAD = 8;
static inline int64_t rc_rows_capacity(int64_t nch, int64_t S) { return (S + RC_ROW_LOOKAHEAD) * nch; }

// Container version 2: the chunk size of a level's four streams follows the level's size, so that a small level still
// spreads over many lanes (the decoder's latency per stream is chunk length x ~0.14 us, whatever the level's size):
// 2^clog symbols with clog = clamp(ceil_log2(ceil(n / 256)), 7, chunk_log2) -- about 256 chunks per stream until the
// header's chunk_log2 (the maximum) is reached at n >= 2^(chunk_log2 + 8) nodes.  Version 1 used chunk_log2 everywhere.
static inline int rc_level_chunk_log2(int64_t n, int chunk_log2, int version)
{
    if (chunk_log2 == 0 || version < 2) return chunk_log2;
    const int64_t want = (n + 255) / 256;
    int c = 0;
    while (((int64_t)1 << c) < want) ++c;
    const int lo = chunk_log2 < 7 ? chunk_log2 : 7;
    return c < lo ? lo : (c > chunk_log2 ? chunk_log2 : c);
}

static inline uint32_t rc_scratch_stride(uint32_t max_syms) { return (2u * max_syms + 32u + 15u) & ~15u; }

int rc_encode_launch(hipStream_t st, const uint32_t *lohi, const RcChunk *chunks, int nchunks, uint8_t *scratch, uint32_t stride, uint32_t *cnt);
int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, const uint32_t *gap, int nchunks, uint8_t *payload);
// payload[0 .. *total + extra) -> dst in 16-byte words (both 16-byte aligned; dst may be pinned host memory): the size stays on the device
int rc_to_host_launch(hipStream_t st, const uint8_t *payload, const uint32_t *total, uint32_t extra, uint8_t *dst);
// cdf: compact interleaved rows (rc_row_stride uint16 per row)
int rc_decode_launch(hipStream_t st, const uint16_t *cdf, int lp, const uint8_t *bytes, const RcChunk *chunks, int nchunks, uint8_t *sym);
// full natural-order rows (n, Lp) -> compact interleaved rows; (cdf, sym) -> interleaved packed words
int rc_pack_rows(hipStream_t st, const uint16_t *cdf_full, int lp, int64_t n, int chunk_log2, uint16_t *rows);
int rc_pack_lohi(hipStream_t st, const uint16_t *cdf_full, int lp, const uint8_t *sym, int64_t n, int chunk_log2, uint32_t *lohi);

}  // namespace gpcc
