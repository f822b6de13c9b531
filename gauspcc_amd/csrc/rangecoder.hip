// rangecoder.hip -- 32-bit range coder (torchac 0.9.3 == arithmetic_kernel.cu:94-163,290-356
// of HAC/submodules/arithmetic.zip), one lane per chunk of symbols.
//
// The coder is serial inside a chunk by construction (every interval update depends on the
// previous one), so the parallel axis is chunks: lane = chunk, 64 chunks per wave.  Encode
// consumes 4 bytes per symbol (c_low | (c_high-1) << 16, produced by the head kernel for the
// ground-truth symbol); decode consumes the full uint16 CDF row of each symbol and searches it.
// Integer / byte work bound by HBM latency per lane, not bandwidth.
#include "rangecoder.hpp"

namespace gpcc {

struct BitW {
    uint8_t *out;
    uint32_t len;
    uint32_t cache;
    uint32_t count;
    __device__ __forceinline__ void put(uint32_t bit)
    {
        cache = (cache << 1) | bit;
        if (++count == 8) { out[len++] = (uint8_t)cache; count = 0; cache = 0; }
    }
    __device__ __forceinline__ void put_pending(uint32_t bit, uint32_t &pending)
    {
        put(bit);
        while (pending) { put(bit ^ 1u); --pending; }
    }
};

__global__ __launch_bounds__(64) void k_rc_encode(const uint32_t *__restrict__ lohi, const RcChunk *__restrict__ chunks, int nchunks,
                                                  uint8_t *__restrict__ scratch, uint32_t stride, uint32_t *__restrict__ cnt)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nchunks) return;
    const RcChunk ch = chunks[c];
    const uint32_t *__restrict__ src = lohi + ch.start;
    BitW w = {scratch + (size_t)c * stride, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
    for (uint32_t i = 0; i < ch.n; ++i) {
        const uint32_t lh = src[i];
        const uint64_t c_low = lh & 0xFFFFu, c_high = (uint64_t)(lh >> 16) + 1u;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) { w.put_pending(0, pending); low <<= 1; high = (high << 1) | 1u; }
            else if (low >= 0x80000000u) { w.put_pending(1, pending); low <<= 1; high = (high << 1) | 1u; }
            else if (low >= 0x40000000u && high < 0xC0000000u) { ++pending; low = (low << 1) & 0x7FFFFFFFu; high = (high << 1) | 0x80000001u; }
            else break;
        }
    }
    ++pending;
    w.put_pending(low < 0x40000000u ? 0u : 1u, pending);
    while (w.count) w.put(0);
    cnt[c] = w.len;
}

// gather the per-chunk scratch rows into one contiguous payload; one block per chunk
__global__ __launch_bounds__(256) void k_rc_compact(const uint8_t *__restrict__ scratch, uint32_t stride, const uint32_t *__restrict__ cnt,
                                                    const uint32_t *__restrict__ off, uint8_t *__restrict__ payload)
{
    const int c = blockIdx.x;
    const uint32_t n = cnt[c];
    const uint8_t *src = scratch + (size_t)c * stride;
    uint8_t *dst = payload + off[c];
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

template <int LP>
__global__ __launch_bounds__(64) void k_rc_decode(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                  int nchunks, uint8_t *__restrict__ sym)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nchunks) return;
    const RcChunk ch = chunks[c];
    const uint8_t *__restrict__ in = bytes + ch.byte_off;
    const uint32_t nbytes = ch.nbytes;
    uint32_t ptr = 0, cache = 0, cached = 0;
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    auto getbit = [&]() {
        if (cached == 0) {
            if (ptr == nbytes) { value <<= 1; return; }
            cache = in[ptr++];
            cached = 8;
        }
        value = (value << 1) | ((cache >> (cached - 1)) & 1u);
        --cached;
    };
    for (int i = 0; i < 32; ++i) getbit();
    constexpr int max_symbol = LP - 2;
    for (uint32_t i = 0; i < ch.n; ++i) {
        const uint16_t *__restrict__ row = cdf + (size_t)(ch.start + i) * LP;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        const uint32_t count = (uint32_t)(((((uint64_t)value - (uint64_t)low + 1u) << 16) - 1u) / span) & 0xFFFFu;
        // largest s in [0, max_symbol] with row[s] <= count (row[0] == 0)
        int s = 0;
        uint32_t c_low = 0, c_high = 0x10000u;
#pragma unroll
        for (int j = 1; j <= max_symbol; ++j) {
            const uint32_t v = row[j];
            if (v <= count) { s = j; c_low = v; }
        }
        if (s != max_symbol) c_high = row[s + 1];
        sym[ch.start + i] = (uint8_t)s;
        high = (low - 1u) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) { low <<= 1; high = (high << 1) | 1u; getbit(); }
            else if (low >= 0x40000000u && high < 0xC0000000u) { low = (low << 1) & 0x7FFFFFFFu; high = (high << 1) | 0x80000001u; value -= 0x40000000u; getbit(); }
            else break;
        }
    }
}

int rc_encode_launch(hipStream_t st, const uint32_t *lohi, const RcChunk *chunks, int nchunks, uint8_t *scratch, uint32_t stride, uint32_t *cnt)
{
    if (nchunks <= 0) return GPCC_OK;
    k_rc_encode<<<(unsigned)cdiv(nchunks, 64), 64, 0, st>>>(lohi, chunks, nchunks, scratch, stride, cnt);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, int nchunks, uint8_t *payload)
{
    if (nchunks <= 0) return GPCC_OK;
    k_rc_compact<<<(unsigned)nchunks, 256, 0, st>>>(scratch, stride, cnt, off, payload);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_decode_launch(hipStream_t st, const uint16_t *cdf, int lp, const uint8_t *bytes, const RcChunk *chunks, int nchunks, uint8_t *sym)
{
    if (nchunks <= 0) return GPCC_OK;
    const unsigned g = (unsigned)cdiv(nchunks, 64);
    switch (lp) {
    case 3: k_rc_decode<3><<<g, 64, 0, st>>>(cdf, bytes, chunks, nchunks, sym); break;
    case 5: k_rc_decode<5><<<g, 64, 0, st>>>(cdf, bytes, chunks, nchunks, sym); break;
    case 17: k_rc_decode<17><<<g, 64, 0, st>>>(cdf, bytes, chunks, nchunks, sym); break;
    default: return fail(GPCC_ERR_ARG, "rc_decode: Lp must be 3, 5 or 17");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace gpcc
