// rangecoder.hip -- 32-bit range coder (torchac 0.9.3 == arithmetic_kernel.cu:94-163,290-356
// of HAC/submodules/arithmetic.zip), one lane per chunk of symbols, bit-identical output.
//
// The coder is serial inside a chunk by construction (every interval update depends on the
// previous one), so the parallel axis is chunks: lane = chunk, 64 chunks per wave.  What the
// device version changes is only HOW the same integers are produced:
//   * the chunks of a stream are interleaved in memory, so a wave reads one coalesced line per step;
//   * the decoder never divides: for count = ((value-low+1)*2^16 - 1) / span the reference picks the
//     largest s with v[s] <= count, and  v <= count  <=>  (span*v) >> 16 <= value - low  exactly, so
//     the search runs on the scaled bounds -- which are also what the interval update needs;
//   * renormalisation shifts out the common leading bits of low/high in one step (clz of low ^ high)
//     and then the run of underflow bits in a second (leading ones of low<<1 / zeros of high<<1);
//   * bits move through a 64-bit reservoir, memory is touched 4 bytes at a time.
// Integer / byte work; latency-bound per lane, not bandwidth-bound.
#include "rangecoder.hpp"

namespace gpcc {

__device__ __forceinline__ int clz32(uint32_t x) { return x ? __clz((int)x) : 32; }

// ------------------------------------------------------------------ encode
struct BitOut {
    uint8_t *out;     // chunk scratch (16-byte aligned)
    uint64_t acc;     // bits collected, MSB first, in the top `n` bits
    uint32_t n;       // valid bits in acc
    uint32_t words;   // 32-bit words already stored
    __device__ __forceinline__ void put(uint32_t bits, uint32_t k)  // k in [1, 32]
    {
        acc |= (uint64_t)bits << (64 - n - k);
        n += k;
        if (n >= 32) {
            reinterpret_cast<uint32_t *>(out)[words++] = __builtin_bswap32((uint32_t)(acc >> 32));
            acc <<= 32;
            n -= 32;
        }
    }
    __device__ __forceinline__ void put_run(uint32_t bit, uint32_t k)  // k copies of bit
    {
        const uint32_t pat = bit ? 0xFFFFFFFFu : 0u;
        while (k >= 32) { put(pat, 32); k -= 32; }
        if (k) put(pat >> (32 - k), k);
    }
};

__global__ __launch_bounds__(64) void k_rc_encode(const uint32_t *__restrict__ lohi, const RcChunk *__restrict__ chunks, int nchunks,
                                                  uint8_t *__restrict__ scratch, uint32_t sstride, uint32_t *__restrict__ cnt)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nchunks) return;
    const RcChunk ch = chunks[c];
    BitOut w = {scratch + (size_t)c * sstride, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
    uint32_t lh = ch.n ? lohi[ch.first] : 0;
    for (uint32_t i = 0; i < ch.n; ++i) {
        const uint32_t cur = lh;
        if (i + 1 < ch.n) lh = lohi[ch.first + (size_t)(i + 1) * ch.stride];  // next symbol's word: independent of the coder state
        const uint64_t c_low = cur & 0xFFFFu, c_high = (uint64_t)(cur >> 16) + 1u;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        const int n1 = clz32(low ^ high);  // leading bits on which low and high agree (< 32: low < high)
        if (n1) {
            const uint32_t bits = low >> (32 - n1);
            const uint32_t b = bits >> (n1 - 1);
            w.put(b, 1);
            if (pending) { w.put_run(b ^ 1u, pending); pending = 0; }
            if (n1 > 1) w.put(bits & ((1u << (n1 - 1)) - 1u), (uint32_t)n1 - 1u);
            low <<= n1;
            high = (high << n1) | ((1u << n1) - 1u);
        }
        // underflow run: low = 01.., high = 10..  ->  drop the second bit n2 times
        const int n2 = min(min(clz32(~(low << 1)), clz32(high << 1)), 31);
        if (n2) {
            pending += (uint32_t)n2;
            low = (low << n2) & 0x7FFFFFFFu;
            high = (high << n2) | 0x80000000u | ((1u << n2) - 1u);
        }
    }
    pending += 1;
    const uint32_t b = low < 0x40000000u ? 0u : 1u;
    w.put(b, 1);
    w.put_run(b ^ 1u, pending);
    // flush the tail, zero padded to a byte boundary
    const uint32_t tail_bytes = (w.n + 7u) >> 3;
    for (uint32_t k = 0; k < tail_bytes; ++k) w.out[4 * w.words + k] = (uint8_t)(w.acc >> (56 - 8 * k));
    cnt[c] = 4 * w.words + tail_bytes;
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

// ------------------------------------------------------------------ decode
// Bit reservoir over a chunk's bytes.  The next 32 bits of the stream are always already loaded (or in
// flight) in `nw`, so a refill is a handful of ALU ops plus the ISSUE of one unaligned dword load whose
// result is not needed before the following refill -- no wait on the decoder's critical path.  Bytes past
// the end of the chunk read as zero (arithmetic_kernel.cu:244-262); the byte buffer is padded so the
// load itself never leaves the allocation.
struct BitIn {
    const uint8_t *p;   // address of the word held in nw
    int32_t rem;        // bytes of the chunk at and after p (may go negative)
    uint32_t nw;        // big-endian word at p, zero-masked beyond the chunk
    uint64_t buf;       // next bits in the top `n` bits
    uint32_t n;
    __device__ __forceinline__ static uint32_t fetch(const uint8_t *q, int32_t rem)
    {
        uint32_t w;
        __builtin_memcpy(&w, q, 4);  // one global_load_dword (unaligned access is legal on gfx950)
        w = __builtin_bswap32(w);
        const uint32_t keep = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * rem));
        return w & keep;
    }
    __device__ __forceinline__ void init(const uint8_t *base, uint32_t nbytes)
    {
        p = base; rem = (int32_t)nbytes; buf = 0; n = 0;
        nw = fetch(p, rem);
    }
    __device__ __forceinline__ uint32_t take(uint32_t k)  // k in [0, 32]
    {
        if (n <= 32) {
            buf |= (uint64_t)nw << (32 - n);
            n += 32;
            p += 4; rem -= 4;
            nw = fetch(p, rem);
        }
        const uint32_t r = (uint32_t)((buf >> 1) >> (63 - k));  // == buf >> (64 - k), and 0 for k == 0
        buf <<= k;
        n -= k;
        return r;
    }
};

__device__ __forceinline__ uint32_t scale(uint64_t span, uint32_t v) { return (uint32_t)((span * (uint64_t)v) >> 16); }

template <int LP>
__global__ __launch_bounds__(64) void k_rc_decode(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                  int nchunks, uint8_t *__restrict__ sym)
{
    constexpr int NV = LP - 2;                              // interior CDF values per row
    constexpr int RS = LP == 3 ? 1 : LP == 5 ? 4 : 16;      // row stride in uint16
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= nchunks) return;
    const RcChunk ch = chunks[c];
    BitIn in;
    in.init(bytes + ch.byte_off, ch.nbytes);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = in.take(32);
    uint16_t v[16];
    uint16_t vn[16];
    auto load_row = [&](uint32_t i, uint16_t *dst) {
        const uint16_t *row = cdf + ((size_t)ch.first + (size_t)i * ch.stride) * RS;
        if (LP == 3) dst[0] = row[0];
        else if (LP == 5) { const uint2 q = *reinterpret_cast<const uint2 *>(row); dst[0] = (uint16_t)q.x; dst[1] = (uint16_t)(q.x >> 16); dst[2] = (uint16_t)q.y; }
        else {
            const uint4 q0 = reinterpret_cast<const uint4 *>(row)[0], q1 = reinterpret_cast<const uint4 *>(row)[1];
            const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) { dst[2 * k] = (uint16_t)w[k]; dst[2 * k + 1] = (uint16_t)(w[k] >> 16); }
        }
    };
    if (ch.n) load_row(0, vn);
    for (uint32_t i = 0; i < ch.n; ++i) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = vn[k];
        if (i + 1 < ch.n) load_row(i + 1, vn);  // rows do not depend on decoded symbols: prefetch
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        const uint32_t x = value - low;
        // largest s in [0, NV] with scaled(v[s]) <= x, v[0] = 0; lo / hi = scaled bounds of the symbol
        uint32_t s = 0, lo = 0, hi;
        if (LP == 3) {
            const uint32_t t1 = scale(span, v[0]);
            const bool ge = t1 <= x;
            s = ge; lo = ge ? t1 : 0u;
            hi = ge ? (uint32_t)span : t1;  // span == 2^32 wraps to 0: high = low - 1 + 2^32 (mod 2^32), as the reference
        } else if (LP == 5) {
            const uint32_t t1 = scale(span, v[0]), t2 = scale(span, v[1]), t3 = scale(span, v[2]);
            s = (uint32_t)(t1 <= x) + (uint32_t)(t2 <= x) + (uint32_t)(t3 <= x);
            lo = s == 0 ? 0u : s == 1 ? t1 : s == 2 ? t2 : t3;
            hi = s == 0 ? t1 : s == 1 ? t2 : s == 2 ? t3 : (uint32_t)span;
        } else {
            // binary search over v[1..15] (array index k holds v[k+1]) with register selects
            const bool b8 = scale(span, v[7]) <= x;
            const uint16_t m4 = b8 ? v[11] : v[3];
            const bool b4 = scale(span, m4) <= x;
            const uint16_t m2 = b8 ? (b4 ? v[13] : v[9]) : (b4 ? v[5] : v[1]);
            const bool b2 = scale(span, m2) <= x;
            const uint32_t base = (b8 ? 8u : 0u) + (b4 ? 4u : 0u) + (b2 ? 2u : 0u);  // s in {base, base+1}
            // candidates: v[base] (0 when base == 0), v[base+1], v[base+2] (span when base + 2 == 16)
            uint16_t cm1 = 0, c0 = 0, cp1 = 0;
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                if ((uint32_t)k + 1u == base) cm1 = v[k];
                if ((uint32_t)k == base) c0 = v[k];
                if ((uint32_t)k == base + 1u) cp1 = v[k];
            }
            const uint32_t tm1 = base ? scale(span, cm1) : 0u, t0 = scale(span, c0);
            const uint32_t tp1 = base + 2u == 16u ? (uint32_t)span : scale(span, cp1);
            const bool b1 = t0 <= x;
            s = base + (uint32_t)b1;
            lo = b1 ? t0 : tm1;
            hi = b1 ? tp1 : t0;
        }
        sym[ch.out + i] = (uint8_t)s;
        high = (low - 1u) + hi;
        low = low + lo;
        // branch-free renormalisation: shifts by zero are no-ops
        const int n1 = clz32(low ^ high);  // < 32: low < high
        low <<= n1;
        high = (high << n1) | ((1u << n1) - 1u);
        value = (value << n1) | in.take((uint32_t)n1);
        const int n2 = min(min(clz32(~(low << 1)), clz32(high << 1)), 31);
        low = (low << n2) & (n2 ? 0x7FFFFFFFu : 0xFFFFFFFFu);
        high = (high << n2) | (n2 ? 0x80000000u : 0u) | ((1u << n2) - 1u);
        value = ((value << n2) ^ (n2 ? 0x80000000u : 0u)) | in.take((uint32_t)n2);
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

// ------------------------------------------------------------------ layout helpers for the stage-level API
__global__ __launch_bounds__(256) void k_rc_pack_rows(const uint16_t *__restrict__ full, int lp, int64_t n, int chunk_log2, uint32_t nch, int rs, uint16_t *__restrict__ rows)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint16_t *dst = rows + (size_t)rc_interleaved((uint32_t)r, chunk_log2, nch) * rs;
    for (int k = 0; k < lp - 2; ++k) dst[k] = full[r * lp + k + 1];
}

__global__ __launch_bounds__(256) void k_rc_pack_lohi(const uint16_t *__restrict__ full, int lp, const uint8_t *__restrict__ sym, int64_t n, int chunk_log2, uint32_t nch,
                                                      uint32_t *__restrict__ lohi)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int s = sym[r];
    const uint32_t lo = full[r * lp + s];
    const uint32_t hi = s == lp - 2 ? 0x10000u : full[r * lp + s + 1];
    lohi[rc_interleaved((uint32_t)r, chunk_log2, nch)] = lo | ((hi - 1u) << 16);
}

int rc_pack_rows(hipStream_t st, const uint16_t *cdf_full, int lp, int64_t n, int chunk_log2, uint16_t *rows)
{
    const uint32_t nch = chunk_log2 ? (uint32_t)cdiv(n, (int64_t)1 << chunk_log2) : 1u;
    k_rc_pack_rows<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(cdf_full, lp, n, chunk_log2, nch, rc_row_stride(lp), rows);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_pack_lohi(hipStream_t st, const uint16_t *cdf_full, int lp, const uint8_t *sym, int64_t n, int chunk_log2, uint32_t *lohi)
{
    const uint32_t nch = chunk_log2 ? (uint32_t)cdiv(n, (int64_t)1 << chunk_log2) : 1u;
    k_rc_pack_lohi<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(cdf_full, lp, sym, n, chunk_log2, nch, lohi);
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace gpcc
