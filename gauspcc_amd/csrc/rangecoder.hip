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
// (gap, optional: bytes the container puts in front of chunk c on top of the chunk bytes before it -- stream headers and
// chunk tables -- so that the payload lands in its final place and leaves the device in one copy)
__global__ __launch_bounds__(256) void k_rc_compact(const uint8_t *__restrict__ scratch, uint32_t stride, const uint32_t *__restrict__ cnt,
                                                    const uint32_t *__restrict__ off, const uint32_t *__restrict__ gap, uint8_t *__restrict__ payload)
{
    const int c = blockIdx.x;
    const uint32_t n = cnt[c];
    const uint8_t *src = scratch + (size_t)c * stride;
    uint8_t *dst = payload + off[c] + (gap ? gap[c] : 0u);
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// ------------------------------------------------------------------ decode
// Bit window over a chunk's bytes, addressed by the absolute bit position `bp` of the next unread bit.  The
// eight bytes around bp are fetched as soon as bp is known (end of a symbol) and consumed at the end of the
// next symbol, so the load hides behind the symbol search; there is no reservoir state and no refill branch.
// Bits past the end of the chunk read as zero (arithmetic_kernel.cu:244-262); the byte buffer is padded so
// the clamped load never leaves the allocation.
struct BitWin {
    const uint8_t *base;
    uint32_t nbytes, nbits;   // chunk size
    uint32_t bp;              // bits consumed
    uint2 raw;                // the 8 bytes at base + min(bp / 8, nbytes)
    __device__ __forceinline__ void fetch()
    {
        __builtin_memcpy(&raw, base + min(bp >> 3, nbytes), 8);  // one global_load_dwordx2 (unaligned access is legal on gfx950)
    }
    __device__ __forceinline__ void init(const uint8_t *b, uint32_t n) { base = b; nbytes = n; nbits = 8u * n; bp = 0; fetch(); }
    __device__ __forceinline__ uint32_t peek32() const   // the next 32 bits, zero past the end
    {
        const uint64_t w = ((uint64_t)__builtin_bswap32(raw.x) << 32) | (uint64_t)__builtin_bswap32(raw.y);
        const uint32_t top = (uint32_t)((w << (bp & 7u)) >> 32);
        // keep the top min(max(nbits - bp, 0), 32) bits: an arithmetic shift of FFFFFFFF'00000000 drags them in
        int32_t valid = (int32_t)(nbits - bp);
        valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
        return top & (uint32_t)((int64_t)0xFFFFFFFF00000000ll >> valid);
    }
    // k in [0, 31]; the caller re-fetches.  Masked shifts keep garbage (lanes past their chunk, corrupt input) defined.
    __device__ __forceinline__ uint32_t take(uint32_t k)
    {
        const uint32_t t = peek32();
        bp += k;
        return (t >> 1) >> ((31u - k) & 31u);   // == t >> (32 - k), and 0 for k == 0
    }
    __device__ __forceinline__ uint32_t take32() { const uint32_t t = peek32(); bp += 32u; return t; }
};

__device__ __forceinline__ uint32_t scale(uint64_t span, uint32_t v) { return (uint32_t)((span * (uint64_t)v) >> 16); }
// the same on d = span - 1 (fits 32 bits): (d + 1) * v = d * v + v -> one v_mad_u64_u32 and one v_alignbit
__device__ __forceinline__ uint32_t scale_d(uint32_t d, uint32_t v) { return (uint32_t)(((uint64_t)d * (uint64_t)v + (uint64_t)v) >> 16); }

// One lane per chunk.  The compact CDF rows of a chunk do not depend on decoded symbols, so they are fetched
// DEPTH symbols ahead through a register ring (the loop is unrolled over the ring, nothing rotates): the
// serial part of a symbol is then ALU only.  The loop bound is the longest chunk of the wave, shorter lanes
// run on with clamped row addresses and masked stores, so the body is free of divergent control flow except
// the bit-reservoir refill.  Symbols leave four at a time.
template <int LP>
__global__ __launch_bounds__(64) void k_rc_decode(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                  int nchunks, uint8_t *__restrict__ sym)
{
    static_assert(LP == 3 || LP == 5, "17-entry rows are decoded by k_rc_decode17");
    constexpr int RS = LP == 3 ? 1 : 4;                     // row stride in uint16
    constexpr int DEPTH = 8;                                // rows in flight per lane (multiple of 4)
    struct Row { uint32_t w[LP == 3 ? 1 : 2]; };
    const int c = blockIdx.x * 64 + threadIdx.x;
    RcChunk ch = {0, 0, 0, 0, 0, 0};
    if (c < nchunks) ch = chunks[c];
    uint32_t nmax = ch.n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    if (nmax == 0) return;
    BitWin in;
    in.init(bytes + ch.byte_off, ch.nbytes);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = in.take32();
    in.fetch();
    static_assert(DEPTH <= RC_ROW_LOOKAHEAD, "row look-ahead exceeds the capacity contract (rc_rows_capacity)");
    const uint16_t *rowp = cdf + (size_t)ch.first * RS;      // row of the next fetch; runs DEPTH rows ahead, unclamped
    const size_t rstep = (size_t)ch.stride * RS;
    auto load_row = [&]() -> Row {
        const uint16_t *row = rowp;
        rowp += rstep;
        Row r;
        if (LP == 3) r.w[0] = row[0];
        else { const uint2 q = *reinterpret_cast<const uint2 *>(row); r.w[0] = q.x; r.w[1] = q.y; }
        return r;
    };
    Row ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = load_row();
    const bool wide = (ch.out & 3u) == 0u;
    uint32_t pack = 0;
    for (uint32_t i0 = 0; i0 < nmax; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint32_t i = i0 + (uint32_t)d;
            const Row rw = ring[d];
            ring[d] = load_row();
            const uint32_t dd = high - low;   // span - 1
            const uint32_t x = value - low;
            // largest s in [0, LP-2] with scaled(v[s]) <= x, v[0] = 0; lo / hi = scaled bounds of the symbol
            uint32_t s = 0, lo = 0, hi;
            if (LP == 3) {
                const uint32_t t1 = scale_d(dd, rw.w[0] & 0xFFFFu);
                const bool ge = t1 <= x;
                s = ge; lo = ge ? t1 : 0u;
                hi = ge ? dd + 1u : t1;  // span == 2^32 wraps to 0: high = low - 1 + 2^32 (mod 2^32), as the reference
            } else {
                const uint32_t t1 = scale_d(dd, rw.w[0] & 0xFFFFu), t2 = scale_d(dd, rw.w[0] >> 16), t3 = scale_d(dd, rw.w[1] & 0xFFFFu);
                s = (uint32_t)(t1 <= x) + (uint32_t)(t2 <= x) + (uint32_t)(t3 <= x);
                lo = s == 0 ? 0u : s == 1 ? t1 : s == 2 ? t2 : t3;
                hi = s == 0 ? t1 : s == 1 ? t2 : s == 2 ? t3 : dd + 1u;
            }
            pack |= s << (8 * (d & 3));
            if ((d & 3) == 3) {
                if (wide && i < ch.n) *reinterpret_cast<uint32_t *>(sym + ch.out + i - 3u) = pack;
                else if (i - 3u < ch.n) {
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q)
                        if (i - 3u + q < ch.n) sym[ch.out + i - 3u + q] = (uint8_t)(pack >> (8 * q));
                }
                pack = 0;
            }
            high = (low - 1u) + hi;
            low = low + lo;
            // Branch-free renormalisation, both phases at once.  Phase 1 drops the n1 leading bits on which low and
            // high agree; then low = 0..., high = 1... and phase 2 drops the n2 underflow bits (low = 01^n2.., high =
            // 10^n2..) behind the top bit.  A symbol narrows the interval by at most 2^-16 (+1), so k = n1 + n2 <= 19
            // for any valid CDF row; the shifts are masked to 5 bits so that garbage (lanes running past their
            // chunk, corrupt input) stays defined.
            const uint32_t n1 = (uint32_t)clz32(low ^ high) & 31u;  // low < high
            const uint32_t l1 = low << n1, h1 = ~((~high) << n1);
            const uint32_t n2 = (uint32_t)min(min(clz32(~(l1 << 1)), clz32(h1 << 1)), 31);
            const uint32_t flip = n2 ? 0x80000000u : 0u;
            const uint32_t k = n1 + n2;
            low = (l1 << n2) & ~flip;
            high = ~((~h1) << n2) | flip;
            value = (((value << n1) << n2) | in.take(k)) ^ flip;
            in.fetch();
        }
    }
}

// 17-entry rows (the 16-way last stage): a 16-lane group per chunk, lane k holds CDF entry k.  The symbol search of the
// lane-per-chunk kernel is four dependent scale-and-compare rounds (the longest part of its 160 instructions per symbol);
// here every lane scales its own entry, one ballot gives the symbol (the scaled bounds are monotone, so the lanes with
// t <= x form a prefix) and two lane reads give its bounds.  The interval update then runs redundantly on the 16 lanes.
// Four chunks per wave; the rows of consecutive chunks are adjacent in memory (chunk-interleaved layout).
__global__ __launch_bounds__(64) void k_rc_decode17(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                    int nchunks, uint8_t *__restrict__ sym)
{
    constexpr int DEPTH = 8;
    static_assert(DEPTH <= RC_ROW_LOOKAHEAD, "row look-ahead exceeds the capacity contract (rc_rows_capacity)");
    const int lane = threadIdx.x, grp = lane >> 4, k = lane & 15;
    const int c = blockIdx.x * 4 + grp;
    RcChunk ch = {0, 0, 0, 0, 0, 0};
    if (c < nchunks) ch = chunks[c];
    uint32_t nmax = ch.n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    if (nmax == 0) return;
    BitWin in;
    in.init(bytes + ch.byte_off, ch.nbytes);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = in.take32();
    in.fetch();
    // compact row: v[1..15] at [0..14]; lane 0 stands for v[0] = 0 and reads the unused slot 15
    const uint16_t *rowp = cdf + (size_t)ch.first * 16 + (k ? k - 1 : 15);
    const size_t rstep = (size_t)ch.stride * 16;
    uint32_t ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ring[d] = *rowp; rowp += rstep; }
    const bool wide = (ch.out & 3u) == 0u;
    uint32_t pack = 0;
    for (uint32_t i0 = 0; i0 < nmax; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint32_t i = i0 + (uint32_t)d;
            const uint32_t v = ring[d];
            ring[d] = *rowp; rowp += rstep;
            const uint32_t dd = high - low;   // span - 1
            const uint32_t x = value - low;
            const uint32_t t = k ? scale_d(dd, v) : 0u;
            const uint64_t bal = __ballot(t <= x);
            const uint32_t half = (grp & 2) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
            const uint32_t bits = (half >> ((grp & 1) * 16)) & 0xFFFFu;      // this group's lanes with t <= x: lanes 0..s
            const uint32_t s = (uint32_t)__popc(bits) - 1u;
            const uint32_t lo = (uint32_t)__shfl((int)t, (grp << 4) + (int)s);
            const uint32_t nx = (uint32_t)__shfl((int)t, (grp << 4) + (int)min(s + 1u, 15u));
            const uint32_t hi = s == 15u ? dd + 1u : nx;
            pack |= s << (8 * (d & 3));
            if ((d & 3) == 3) {
                if (k == 0) {
                    if (wide && i < ch.n) *reinterpret_cast<uint32_t *>(sym + ch.out + i - 3u) = pack;
                    else if (i - 3u < ch.n) {
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q)
                            if (i - 3u + q < ch.n) sym[ch.out + i - 3u + q] = (uint8_t)(pack >> (8 * q));
                    }
                }
                pack = 0;
            }
            high = (low - 1u) + hi;
            low = low + lo;
            const uint32_t n1 = (uint32_t)clz32(low ^ high) & 31u;
            const uint32_t l1 = low << n1, h1 = ~((~high) << n1);
            const uint32_t n2 = (uint32_t)min(min(clz32(~(l1 << 1)), clz32(h1 << 1)), 31);
            const uint32_t flip = n2 ? 0x80000000u : 0u;
            const uint32_t kk = n1 + n2;
            low = (l1 << n2) & ~flip;
            high = ~((~h1) << n2) | flip;
            value = (((value << n1) << n2) | in.take(kk)) ^ flip;
            in.fetch();
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

int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, const uint32_t *gap, int nchunks, uint8_t *payload)
{
    if (nchunks <= 0) return GPCC_OK;
    k_rc_compact<<<(unsigned)nchunks, 256, 0, st>>>(scratch, stride, cnt, off, gap, payload);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(256) void k_rc_to_host(const uint4 *__restrict__ src, const uint32_t *__restrict__ total, uint32_t extra, uint4 *__restrict__ dst)
{
    const uint32_t words = (*total + extra + 15u) >> 4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) dst[i] = src[i];
}

int rc_to_host_launch(hipStream_t st, const uint8_t *payload, const uint32_t *total, uint32_t extra, uint8_t *dst)
{
    k_rc_to_host<<<512, 256, 0, st>>>(reinterpret_cast<const uint4 *>(payload), total, extra, reinterpret_cast<uint4 *>(dst));
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
    case 17: k_rc_decode17<<<(unsigned)cdiv(nchunks, 4), 64, 0, st>>>(cdf, bytes, chunks, nchunks, sym); break;
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
