// primitives.hpp -- device-wide exclusive scan and LSD radix sort (hand-written, wave64).
#pragma once
#include "common.hpp"

namespace gpcc {

// out[i] = sum_{j<i} in[j] (uint32).  in == out allowed.  If total_dev != nullptr the grand
// total is stored there.  Workspace comes from ctx->arena (released before return).
int exclusive_scan_pair_u32(gpcc_ctx *ctx, hipStream_t st, const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, int64_t n);   // two scans of one length (one launch when short)
int exclusive_scan_u32(gpcc_ctx *ctx, hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n,
                       uint32_t *total_dev);

// Stable LSD radix sort of (key, val) pairs on the low `bits` bits of key, 8 bits per pass.
// keys/vals are ping-ponged with keys_tmp/vals_tmp; on return *keys_io / *vals_io point at the
// buffers holding the result.  vals may be nullptr (keys only).
int radix_sort_u64(gpcc_ctx *ctx, hipStream_t st, uint64_t **keys_io, uint64_t **keys_tmp_io,
                   uint32_t **vals_io, uint32_t **vals_tmp_io, int64_t n, int bits);

// Device-side faults that must not abort the process: a look-back scan whose predecessors made no progress for ~10 s (status words damaged)
// raises the context's sticky error word and lets the launch run out with garbage instead of trapping.  Every entry point that scans calls this after
// its final sync: GPCC_OK, or GPCC_ERR_HIP once (the word and the scan states are reset: the next call starts clean).
int device_error_check(gpcc_ctx *ctx);

}  // namespace gpcc
