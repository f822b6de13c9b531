// hostcoder.hpp -- the host-side torchac coder as the reference-layout container uses it (hostcoder.hip; HIP-free like that file).
//
// chunk_log2 = 0 writes / reads the reference's container: ONE torchac stream per (level, stage) (HAC/utils/pcc_utils.py:174-177, :198-203).
// A stream is one dependent chain of up to 10^6 symbols: one GPU lane runs it at ~7 Msymbols/s (decode) / ~40 (encode), a host core at
// ~35 / ~80 -- and torchac itself is a CPU coder.  So for this layout only the CODER runs on the host, on the library's own implementation
// (the one behind gauspcc_amd.torchac), while the network stays on the device: the heads' packed coder words / compact CDF rows cross PCIe
// once per stream.  An encode's streams are independent and are coded on a pool of native threads; a decode's are sequentially dependent
// (stage s + 1 of a level needs stage s's symbols) and run on one thread.  The chunked containers (the default) never come here.
#pragma once
#include <cstdint>
#include <vector>

namespace gpcc {

// streams[k]: n[k] packed (c_low | (c_high - 1) << 16) words; out[k]: torchac's bytes of stream k.  threads <= 0: min(16, hardware threads).
int host_encode_streams(const uint32_t *const *streams, const int64_t *n, int nstreams, std::vector<std::vector<uint8_t>> *out, int threads);
// n symbols under compact CDF rows (rc_format.hpp: rc_row_stride(lp) uint16 per row, interior values) from one torchac stream
int host_decode_compact(const uint16_t *rows, int lp, const uint8_t *bytes, int64_t nbytes, int64_t n, uint8_t *sym_out);

}  // namespace gpcc
