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
#include "rangecoder_dev.hpp"
#include <type_traits>

namespace gpcc {


// ------------------------------------------------------------------ rows ahead of the coder: three register sets
// CDF rows (decoders) and packed symbol words (encoder) do not depend on coded symbols, so they are fetched ahead of the
// serial chain: three sets of RING_PHASE = 16 rows in REGISTERS, filled by ordinary global loads.  The symbol loop is
// unrolled over one trip of the three sets (48 symbols), so every row has a fixed register, nothing rotates, and the set
// just consumed is refilled at the end of its phase -- two phases (32 symbols, ~4 us) before it is read again.  The waits
// are the compiler's own counted `s_waitcnt vmcnt` (ordinary loads retire in order), the 16 symbols of a phase leave in one
// 16-byte store.
//
// Round 3 first built this ring in LDS, filled by LDS-DMA loads (`global_load_lds_*`, hand-counted waits), because a
// ROLLED register ring had cost a memory round trip per trip (the compiler clusters the loads, renames the ring across the
// back edge and waits with vmcnt(0)).  That version passed every parity test and decoded wrong symbols in 2-16 % of the
// decodes as soon as a second scene shared the GPU (tools/inflight_check.py; HISTORY.md section 4 has the whole story):
// under load LDS-DMA loads do not retire in issue order, `s_waitcnt vmcnt(0)` can release a wave before the data is visible
// to its own ds_read, the upper half of the dword a 16-bit LDS-DMA load writes is not reliably zero, and a load still in
// flight at s_endpgm lands in LDS that may belong to another workgroup.  A three-thirds LDS ring with a whole phase between
// a retiring vmcnt(0) and the first read was clean in 4 000 scene-steps -- and the register sets below are as fast
// (51.6 vs 53.1 us per binary launch, 74.2 vs 77.7 4-ary) on documented semantics only.  The LDS-DMA code is gone.
constexpr int RING_PHASE = RC_RING_DEPTH / RING_NPH;
static_assert(RING_PHASE == 16, "a phase's symbols leave as one 16-byte store");

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

// Carry into the bits already collected (container version 4, the carry-propagating coder): + 1 at the last bit of the string;
// a carry out of the accumulator ripples through the words already stored (big-endian bit strings).
__device__ __forceinline__ void bitout_carry(BitOut &w)
{
    if (w.n) {
        const uint64_t unit = 1ull << (64u - w.n);
        const uint64_t a2 = w.acc + unit;
        const bool ovf = a2 < w.acc;
        w.acc = a2;
        if (!ovf) return;
    }
    uint32_t *mem = reinterpret_cast<uint32_t *>(w.out);
    for (int i = (int)w.words - 1; i >= 0; --i) {
        const uint32_t v = __builtin_bswap32(mem[i]) + 1u;
        mem[i] = __builtin_bswap32(v);
        if (v) break;
    }
}

template <int CODER>
__global__ __launch_bounds__(64) void k_rc_encode(const uint32_t *__restrict__ lohi, const RcChunk *__restrict__ chunks, int nchunks,
                                                  uint8_t *__restrict__ scratch, uint32_t sstride, uint32_t *__restrict__ cnt)
{
    // the packed (c_low | (c_high - 1) << 16) words of the lanes are fetched up to RC_RING_DEPTH symbols ahead (header comment):
    // one load per symbol fetched one symbol ahead was a memory round trip per symbol (0.42 us: the whole encode coder took as
    // long as its longest lane times that)
    constexpr int DEPTH = RC_RING_DEPTH, PH = RING_PHASE;
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    RcChunk ch = {0, 0, 0, 0, 0, 0};
    if (c < nchunks) ch = chunks[c];
    uint32_t nmax = ch.n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    BitOut w = {scratch + (size_t)min(c, nchunks - 1) * sstride, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
    // loads run past a lane's last word: clamp the element index to the lane's own words (a finished lane re-reads its last one)
    const uint32_t last = ch.n ? ch.n - 1u : 0u;
    auto src = [&](uint32_t t) -> const uint32_t * { return lohi + ch.first + (size_t)min(t, last) * ch.stride; };
    uint32_t regs[RING_NPH][PH];
    auto fill = [&](int h, uint32_t first) {
#pragma unroll
        for (int dd = 0; dd < PH; ++dd) regs[h][dd] = *src(first + (uint32_t)dd);
    };
#pragma unroll
    for (int h = 0; h < RING_NPH; ++h) fill(h, (uint32_t)(h * PH));
    for (uint32_t i0 = 0; i0 < nmax; i0 += DEPTH) {
#pragma unroll
        for (int h = 0; h < RING_NPH; ++h) {
#pragma unroll
            for (int dd = 0; dd < PH; ++dd) {
                const uint32_t i = i0 + (uint32_t)(h * PH + dd);
                const uint32_t cur = regs[h][dd];
                if (CODER == RC_CODER_CARRY) {
                    // version 4 (oracle/gpcc_oracle.c: cp_encode_core): `high` holds the RANGE, normalised to [2^31, 2^32)
                    if (i < ch.n) {
                        const uint32_t c_lo = cur & 0xFFFFu, c_hi = (cur >> 16) + 1u;
                        const uint32_t r = high >> 16, add = __umul24(r, c_lo);
                        const uint32_t nl = low + add;
                        if (nl < low) bitout_carry(w);
                        low = nl;
                        high = c_hi == 0x10000u ? high - add : __umul24(r, c_hi - c_lo);
                        const int k = clz32(high | 0x8000u);   // (range >= 2^15 for every valid row)
                        if (k) { w.put(low >> (32 - k), (uint32_t)k); low <<= k; high <<= k; }
                    }
                } else
                if (i < ch.n) {
                    const uint64_t c_low = cur & 0xFFFFu, c_high = (uint64_t)(cur >> 16) + 1u;
                    const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
                    high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
                    low = low + (uint32_t)((span * c_low) >> 16);
                    const int n1 = clz32(low ^ high);  // leading bits on which low and high agree (< 32: low < high)
                    if (n1) {
                        const uint32_t bits = low >> (32 - n1);
                        const uint32_t b = bits >> (n1 - 1);
                        const uint32_t rest = bits & ((1u << (n1 - 1)) - 1u);
                        if ((uint32_t)n1 + pending <= 32u) {
                            // the common case in ONE put: b, `pending` copies of its complement, the other n1 - 1 agreed bits
                            const uint32_t len = (uint32_t)n1 + pending;
                            const uint32_t run = b ? 0u : ((pending >= 32u ? 0u : (1u << pending)) - 1u);
                            w.put((b << (len - 1u)) | (run << (n1 - 1)) | rest, len);
                        } else {
                            w.put(b, 1);
                            w.put_run(b ^ 1u, pending);
                            if (n1 > 1) w.put(rest, (uint32_t)n1 - 1u);
                        }
                        pending = 0;
                    }
                    low <<= n1;                                   // (n1 = 0: no change)
                    high = (high << n1) | ((1u << n1) - 1u);
                    // underflow run: low = 01.., high = 10..  ->  drop the second bit n2 times (low's top bit is 0 and high's 1 here,
                    // so the masks below change nothing when n2 = 0)
                    const int n2 = min(min(clz32(~(low << 1)), clz32(high << 1)), 31);
                    pending += (uint32_t)n2;
                    low = (low << n2) & 0x7FFFFFFFu;
                    high = (high << n2) | 0x80000000u | ((1u << n2) - 1u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i0 + (uint32_t)(DEPTH + h * PH) < nmax) fill(h, i0 + (uint32_t)(DEPTH + h * PH));   // (wave-uniform) words no lane will use are not fetched
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (c >= nchunks) return;
    if (CODER == RC_CODER_CARRY) {
        // flush: the two top bits of the smallest multiple of 2^30 >= low (whatever bits follow decode to the same symbols)
        const uint32_t q = (uint32_t)(((uint64_t)low + 0x3FFFFFFFull) >> 30);   // 0 .. 4
        if (q == 4u) bitout_carry(w);
        w.put(q & 3u, 2);
    } else {
    pending += 1;
    const uint32_t b = low < 0x40000000u ? 0u : 1u;
    w.put(b, 1);
    w.put_run(b ^ 1u, pending);
    }
    // flush the tail, zero padded to a byte boundary
    const uint32_t tail_bytes = (w.n + 7u) >> 3;
    for (uint32_t k = 0; k < tail_bytes; ++k) w.out[4 * w.words + k] = (uint8_t)(w.acc >> (56 - 8 * k));
    cnt[c] = 4 * w.words + tail_bytes;
}

// gather the per-chunk scratch rows into one contiguous payload; one block per chunk
// (gap, optional: bytes the container puts in front of chunk c on top of the chunk bytes before it -- stream headers and
// chunk tables -- so that the payload lands in its final place and leaves the device in one copy)
// (lanes != null, container version 3: lane 2c + 1 of a stream is the backwards half of chunk c -- its bytes go out last byte first, so that
// they read forwards from the chunk's end)
__global__ __launch_bounds__(256) void k_rc_compact(const uint8_t *__restrict__ scratch, uint32_t stride, const uint32_t *__restrict__ cnt,
                                                    const uint32_t *__restrict__ off, const uint32_t *__restrict__ gap, uint8_t *__restrict__ payload,
                                                    const RcChunk *__restrict__ lanes)
{
    const int c = blockIdx.x;
    const uint32_t n = cnt[c];
    const uint8_t *src = scratch + (size_t)c * stride;
    uint8_t *dst = payload + off[c] + (gap ? gap[c] : 0u);
    // the lane's index inside its stream has the parity of its first element (streams start on even slots)
    if (lanes && (lanes[c].first & 1u)) { for (uint32_t i = threadIdx.x; i < n; i += 256) dst[n - 1u - i] = src[i]; }
    else { for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i]; }
}

// Version-3 containers: bytes in front of lane l's payload that are not payload = sum over the streams up to and including
// its own of (4-byte stream length + the stream's varint table).  One workgroup; a wave per stream sums the varint sizes.
__global__ __launch_bounds__(256) void k_rc_layout(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ stream_first, int nstreams,
                                                   const uint32_t *__restrict__ lane_stream, int nlanes, int dual, uint32_t *__restrict__ gap, uint32_t *__restrict__ gap_total)
{
    __shared__ uint32_t pre[4 * MAXLV + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < nstreams; s += 4) {
        const uint32_t l0 = stream_first[s], l1 = stream_first[s + 1];
        uint32_t t = 0;
        if (dual) {   // the version-3 table (rangecoder.hpp: rc_table_size): bits of the Rice-coded differences for every k, the best k wins
            uint32_t bits[RC_TAB_KMAX + 1];
#pragma unroll
            for (int k = 0; k <= RC_TAB_KMAX; ++k) bits[k] = 0u;
            auto chunk_bytes = [&](uint32_t l) { return cnt[l] + (l + 1u < l1 ? cnt[l + 1u] : 0u); };
            for (uint32_t l = l0 + 2u + 2u * (uint32_t)lane; l < l1; l += 128u) {
                const uint32_t z = rc_zigzag(chunk_bytes(l), chunk_bytes(l - 2u));
#pragma unroll
                for (int k = 0; k <= RC_TAB_KMAX; ++k) bits[k] += rc_tab_cost(z, k);
            }
            uint64_t tot[RC_TAB_KMAX + 1];
#pragma unroll
            for (int k = 0; k <= RC_TAB_KMAX; ++k) {
                uint32_t v = bits[k];          // at most 48 bits x 2^25 chunks a stream: no overflow
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
                tot[k] = v;
            }
            t = l1 > l0 ? rc_tab_bytes(chunk_bytes(l0), (l1 - l0 + 1u) / 2u, tot, nullptr) : 0u;
        } else t = 2u * (l1 - l0);
        if (lane == 0) pre[s] = 4u + t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int s = 0; s < nstreams; ++s) { a += pre[s]; pre[s] = a; }
        *gap_total = a;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < nlanes; l += 256) gap[l] = pre[lane_stream[l]];
}

// The same for any number of streams (a batch of scenes: codec_batch.hip), as three launches: per-stream sizes (a wave per
// stream), a one-workgroup scan, the gaps.  extra[s] (nullable): container bytes in front of stream s that belong to no stream
// (the header of the scene the stream opens).
__global__ __launch_bounds__(256) void k_rc_stream_sizes(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ stream_first, int nstreams, int dual,
                                                         const uint32_t *__restrict__ extra, uint32_t *__restrict__ ssize)
{
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nstreams) return;
    const uint32_t l0 = stream_first[s], l1 = stream_first[s + 1];
    uint32_t t = 0;
    if (dual) {
        uint32_t bits[RC_TAB_KMAX + 1];
#pragma unroll
        for (int k = 0; k <= RC_TAB_KMAX; ++k) bits[k] = 0u;
        auto chunk_bytes = [&](uint32_t l) { return cnt[l] + (l + 1u < l1 ? cnt[l + 1u] : 0u); };
        for (uint32_t l = l0 + 2u + 2u * (uint32_t)lane; l < l1; l += 128u) {
            const uint32_t z = rc_zigzag(chunk_bytes(l), chunk_bytes(l - 2u));
#pragma unroll
            for (int k = 0; k <= RC_TAB_KMAX; ++k) bits[k] += rc_tab_cost(z, k);
        }
        uint64_t tot[RC_TAB_KMAX + 1];
#pragma unroll
        for (int k = 0; k <= RC_TAB_KMAX; ++k) {
            uint32_t v = bits[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
            tot[k] = v;
        }
        t = l1 > l0 ? rc_tab_bytes(chunk_bytes(l0), (l1 - l0 + 1u) / 2u, tot, nullptr) : 0u;
    } else t = 2u * (l1 - l0);
    if (lane == 0) ssize[s] = 4u + t + (extra ? extra[s] : 0u);
}

// inclusive scan of the stream sizes in place (one workgroup, tiles of 256 with a carry); *gap_total = the sum
__global__ __launch_bounds__(256) void k_rc_stream_scan(uint32_t *__restrict__ ssize, int nstreams, uint32_t *__restrict__ gap_total)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int s0 = 0; s0 < nstreams; s0 += 256) {
        const int s = s0 + (int)threadIdx.x;
        const uint32_t v = s < nstreams ? ssize[s] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += u; }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
        const uint32_t base = carry + (wave == 0 ? 0u : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2);
        if (s < nstreams) ssize[s] = base + inc;
        carry += w0 + w1 + w2 + w3;
    }
    if (threadIdx.x == 0) *gap_total = carry;
}

__global__ __launch_bounds__(256) void k_rc_lane_gaps(const uint32_t *__restrict__ spre, const uint32_t *__restrict__ lane_stream, int nlanes, uint32_t *__restrict__ gap)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l < nlanes) gap[l] = spre[lane_stream[l]];
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

// ------------------------------------------------------------------ decode from an LDS image of the lanes' bytes
// The kernels above read a lane's bytes through a window in global memory, one 8-byte load per symbol: every lane of the
// wave crosses into a new cache line of ITS chunk at its own time, and each crossing is a miss the whole wave waits for.
// The lanes of a chunked container are short (a few hundred bytes), so the wave first copies the byte windows of all its
// lanes into LDS in one sweep -- big-endian dwords in each lane's READING order, which is where the backwards lanes of
// version-3 chunks stop being special -- and the serial part of a symbol touches memory only through two LDS dwords
// (no miss; the dword index is clamped to the lane's window) and the CDF-row ring.
// State per lane: low, d = high - low, x = value - low, q = bits consumed - 1.  high and value themselves are never needed:
//   d' = (d1 << k) | ones(k),  x' = (x1 << k) | next k bits   (the E3 flip adds 2^31 to low, high and value alike;
//   d rather than the span itself because a span of 2^32 does occur: a symbol of probability 2^-16 renormalises to it),
// k = n1 + n2 as in the kernels above.  ~40 instructions per binary symbol instead of ~75 (profiles/r03_rc_decode_isa.txt).

// The staged decoders (byte windows in LDS, rows in three register sets) live in rangecoder_dev.hpp as functions of one
// wave; these kernels are one wave per workgroup.
template <int LP, int CODER>
__global__ __launch_bounds__(64) void k_rc_decode_lds(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                      int nchunks, int lpw, uint32_t rdw, uint8_t *__restrict__ sym)
{
    extern __shared__ uint32_t win[];                          // [lpw][rdw] byte windows
    rc_decode_lds_wave<LP, RING_PHASE, true, CODER>(cdf, bytes, chunks, nchunks, (int)blockIdx.x * lpw, (int)threadIdx.x, lpw, rdw, sym, win);
}

template <int CODER>
__global__ __launch_bounds__(64) void k_rc_decode17_lds(const uint16_t *__restrict__ cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                        int nchunks, uint32_t rdw, uint8_t *__restrict__ sym)
{
    extern __shared__ uint32_t win[];
    rc_decode17_lds_wave<RING_PHASE, true, CODER>(cdf, bytes, chunks, nchunks, (int)blockIdx.x * 4, (int)threadIdx.x, rdw, sym, win);
}

int rc_encode_launch(hipStream_t st, const uint32_t *lohi, const RcChunk *chunks, int nchunks, uint8_t *scratch, uint32_t stride, uint32_t *cnt, int coder)
{
    if (nchunks <= 0) return GPCC_OK;
    if (coder == RC_CODER_CARRY) k_rc_encode<RC_CODER_CARRY><<<(unsigned)cdiv(nchunks, 64), 64, 0, st>>>(lohi, chunks, nchunks, scratch, stride, cnt);
    else k_rc_encode<RC_CODER_CARRYLESS><<<(unsigned)cdiv(nchunks, 64), 64, 0, st>>>(lohi, chunks, nchunks, scratch, stride, cnt);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_compact_launch(hipStream_t st, const uint8_t *scratch, uint32_t stride, const uint32_t *cnt, const uint32_t *off, const uint32_t *gap, int nchunks, uint8_t *payload, const RcChunk *dual_lanes)
{
    if (nchunks <= 0) return GPCC_OK;
    k_rc_compact<<<(unsigned)nchunks, 256, 0, st>>>(scratch, stride, cnt, off, gap, payload, dual_lanes);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_layout_launch(hipStream_t st, const uint32_t *cnt, const uint32_t *stream_first, int nstreams, const uint32_t *lane_stream, int nlanes, bool dual, uint32_t *gap, uint32_t *gap_total)
{
    if (nstreams > 4 * MAXLV) return fail(GPCC_ERR_ARG, "internal: %d streams", nstreams);
    k_rc_layout<<<1, 256, 0, st>>>(cnt, stream_first, nstreams, lane_stream, nlanes, dual ? 1 : 0, gap, gap_total);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_layout_many_launch(hipStream_t st, const uint32_t *cnt, const uint32_t *stream_first, int nstreams, const uint32_t *lane_stream, int nlanes, bool dual, const uint32_t *extra,
                          uint32_t *ssize, uint32_t *gap, uint32_t *gap_total)
{
    if (nstreams <= 0) return GPCC_OK;
    k_rc_stream_sizes<<<(unsigned)cdiv(nstreams, 4), 256, 0, st>>>(cnt, stream_first, nstreams, dual ? 1 : 0, extra, ssize);
    LAUNCH_CHECK();
    k_rc_stream_scan<<<1, 256, 0, st>>>(ssize, nstreams, gap_total);
    LAUNCH_CHECK();
    if (nlanes > 0) { k_rc_lane_gaps<<<(unsigned)cdiv(nlanes, 256), 256, 0, st>>>(ssize, lane_stream, nlanes, gap); LAUNCH_CHECK(); }
    return GPCC_OK;
}

__global__ __launch_bounds__(256) void k_rc_to_host(const uint4 *__restrict__ src, const uint32_t *__restrict__ total, uint32_t extra, const uint32_t *__restrict__ extra_dev,
                                                    uint4 *__restrict__ dst)
{
    const uint32_t words = (*total + extra + (extra_dev ? *extra_dev : 0u) + 15u) >> 4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) dst[i] = src[i];
}

int rc_to_host_launch(hipStream_t st, const uint8_t *payload, const uint32_t *total, uint32_t extra, const uint32_t *extra_dev, uint8_t *dst)
{
    k_rc_to_host<<<512, 256, 0, st>>>(reinterpret_cast<const uint4 *>(payload), total, extra, extra_dev, reinterpret_cast<uint4 *>(dst));
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_decode_launch(hipStream_t st, const uint16_t *cdf, int lp, const uint8_t *bytes, const RcChunk *chunks, int nchunks, uint32_t max_bytes, bool dual, uint8_t *sym, int coder)
{
    if (nchunks <= 0) return GPCC_OK;
    // (invariants the staged kernels rely on, enforced by gpcc_encode / gpcc_decode's chunk_log2 range of 6..14 -- checked by their
    // callers: lanes of a chunked stream start on multiples of 16 symbols, ch.out is 16-byte aligned, `sym` has n + 4 bytes)
    // staged path: every lane's window (+ the two dwords the reader runs ahead) in LDS; as many lanes per wave as fit
    const uint64_t rdw = rc_window_dwords(max_bytes);
    const uint32_t ring_bytes = rc_ring_bytes(lp);       // the row ring behind the windows
    const uint32_t cap = RC_LDS_CAP - ring_bytes;
    if (rc_window_fits(lp, max_bytes)) {
        const bool cp = coder == RC_CODER_CARRY;
        if (lp == 17) {
            const size_t lds = (size_t)(4u * rdw * 4u) + ring_bytes;
            if (cp) k_rc_decode17_lds<RC_CODER_CARRY><<<(unsigned)cdiv(nchunks, 4), 64, lds, st>>>(cdf, bytes, chunks, nchunks, (uint32_t)rdw, sym);
            else k_rc_decode17_lds<RC_CODER_CARRYLESS><<<(unsigned)cdiv(nchunks, 4), 64, lds, st>>>(cdf, bytes, chunks, nchunks, (uint32_t)rdw, sym);
        } else {
            int lpw = 64;
            while ((uint64_t)lpw * rdw * 4u > cap) lpw >>= 1;
            const unsigned g = (unsigned)cdiv(nchunks, lpw);
            const size_t lds = (size_t)lpw * rdw * 4u + ring_bytes;
            if (lp == 3 && cp) k_rc_decode_lds<3, RC_CODER_CARRY><<<g, 64, lds, st>>>(cdf, bytes, chunks, nchunks, lpw, (uint32_t)rdw, sym);
            else if (lp == 3) k_rc_decode_lds<3, RC_CODER_CARRYLESS><<<g, 64, lds, st>>>(cdf, bytes, chunks, nchunks, lpw, (uint32_t)rdw, sym);
            else if (lp == 5 && cp) k_rc_decode_lds<5, RC_CODER_CARRY><<<g, 64, lds, st>>>(cdf, bytes, chunks, nchunks, lpw, (uint32_t)rdw, sym);
            else if (lp == 5) k_rc_decode_lds<5, RC_CODER_CARRYLESS><<<g, 64, lds, st>>>(cdf, bytes, chunks, nchunks, lpw, (uint32_t)rdw, sym);
            else return fail(GPCC_ERR_ARG, "rc_decode: Lp must be 3, 5 or 17");
        }
        LAUNCH_CHECK();
        return GPCC_OK;
    }
    // lanes too long for LDS (the reference layout: one lane per stream) read their bytes through a window in memory;
    // forwards only (backwards lanes exist in version-3 chunks, which are short by construction)
    if (dual || coder != RC_CODER_CARRYLESS) return fail(GPCC_ERR_FORMAT, "a chunk of %u bytes is beyond the decoder's LDS window", max_bytes);
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

int rc_pack_rows(hipStream_t st, const uint16_t *cdf_full, int lp, int64_t n, int chunk_log2, uint32_t nch, uint16_t *rows)
{
    k_rc_pack_rows<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(cdf_full, lp, n, chunk_log2, nch, rc_row_stride(lp), rows);
    LAUNCH_CHECK();
    return GPCC_OK;
}

int rc_pack_lohi(hipStream_t st, const uint16_t *cdf_full, int lp, const uint8_t *sym, int64_t n, int chunk_log2, uint32_t nch, uint32_t *lohi)
{
    k_rc_pack_lohi<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(cdf_full, lp, sym, n, chunk_log2, nch, lohi);
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace gpcc
