// rangecoder.hpp -- device range coder launchers (see rangecoder.hip).
#pragma once
#include "common.hpp"

namespace gpcc {

struct RcChunk {
    uint32_t start;     // first symbol (index into the stream-major symbol / lohi / cdf-row arrays)
    uint32_t n;         // symbols in the chunk
    uint32_t byte_off;  // decode: offset of the chunk's bytes in the uploaded file
    uint32_t nbytes;    // decode: byte count
};

static inline uint32_t rc_scratch_stride(uint32_t max_syms) { return (2u * max_syms + 16u + 15u) & ~15u; }

int rc_encode_launch(hipStream_t st, const uint32_t *lohi, const RcChunk *chunks, int nchunks, uint8_t *scratch, uint32_t stride, uint32_t *cnt);
int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, int nchunks, uint8_t *payload);
int rc_decode_launch(hipStream_t st, const uint16_t *cdf, int lp, const uint8_t *bytes, const RcChunk *chunks, int nchunks, uint8_t *sym);

}  // namespace gpcc
