// rangecoder.hpp -- device range coder launchers (see rangecoder.hip).
#pragma once
#include "common.hpp"
#include "rc_format.hpp"

namespace gpcc {

// coder: 0 = torchac's carry-less coder (the reference layout, container versions 1-3), 1 = the carry-propagating coder of container
// version 4 (rangecoder_dev.hpp: RC_CODER_*, rc_coder_of_version)
int rc_encode_launch(hipStream_t st, const uint32_t *lohi, const RcChunk *chunks, int nchunks, uint8_t *scratch, uint32_t stride, uint32_t *cnt, int coder = 0);
// dual_lanes (the device lane descriptors, version 3): the bytes of a stream's odd lanes are written in reverse order (the backwards half of a chunk)
int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, const uint32_t *gap, int nchunks, uint8_t *payload, const RcChunk *dual_lanes = nullptr);
// payload[0 .. *total + extra (+ *extra_dev)) -> dst in 16-byte words (both 16-byte aligned; dst may be pinned host memory): the size stays on the device
int rc_to_host_launch(hipStream_t st, const uint8_t *payload, const uint32_t *total, uint32_t extra, const uint32_t *extra_dev, uint8_t *dst);
// cdf: compact interleaved rows (rc_row_stride uint16 per row)
// max_bytes: the longest byte window of the lanes (rc_parse_table); lanes whose windows fit the LDS are decoded from a staged
// copy, one lane of any size (the reference layout) straight from memory
// dual: the lanes are version-3 chunk halves (some run backwards: staged path only)
int rc_decode_launch(hipStream_t st, const uint16_t *cdf, int lp, const uint8_t *bytes, const RcChunk *chunks, int nchunks, uint32_t max_bytes, bool dual, uint8_t *sym, int coder = 0);
// version-3 streams: the bytes in front of every lane's payload (stream lengths + varint tables) depend on the byte counts
// and are worked out on the device: lane_stream[l] = stream of lane l, stream_first[s] = first lane of stream s (nstreams + 1
// entries), dual: lanes pair up into chunks.  gap[l] and *gap_total come out.
int rc_layout_launch(hipStream_t st, const uint32_t *cnt, const uint32_t *stream_first, int nstreams, const uint32_t *lane_stream, int nlanes, bool dual, uint32_t *gap, uint32_t *gap_total);
// the same for any number of streams (a batch of scenes); extra[s] (nullable): bytes in front of stream s that belong to no stream (a scene's
// header); ssize: nstreams words of scratch (out: the inclusive prefix sums)
int rc_layout_many_launch(hipStream_t st, const uint32_t *cnt, const uint32_t *stream_first, int nstreams, const uint32_t *lane_stream, int nlanes, bool dual, const uint32_t *extra,
                          uint32_t *ssize, uint32_t *gap, uint32_t *gap_total);
// full natural-order rows (n, Lp) -> compact interleaved rows; (cdf, sym) -> interleaved packed words (lanes of 2^lane_log2
// symbols, nlanes of them; lane_log2 = 0: one lane)
int rc_pack_rows(hipStream_t st, const uint16_t *cdf_full, int lp, int64_t n, int lane_log2, uint32_t nlanes, uint16_t *rows);
int rc_pack_lohi(hipStream_t st, const uint16_t *cdf_full, int lp, const uint8_t *sym, int64_t n, int lane_log2, uint32_t nlanes, uint32_t *lohi);

}  // namespace gpcc
