// rangecoder_dev.hpp -- device side of the staged range decoders (rangecoder.hip has the story): the lane decoders as
// functions of ONE wave, so that the stand-alone kernels (one wave per workgroup) and the fused small-level kernel
// (fused.hip: one wave of a 16-wave workgroup) run the same code.
#pragma once
#include "rangecoder.hpp"

namespace gpcc {

__device__ __forceinline__ int clz32(uint32_t x) { return x ? __clz((int)x) : 32; }

constexpr int RING_NPH = 3;                       // register sets of rows fetched ahead of the coder

__device__ __forceinline__ uint32_t scale(uint64_t span, uint32_t v) { return (uint32_t)((span * (uint64_t)v) >> 16); }
// the same on d = span - 1 (fits 32 bits): (d + 1) * v = d * v + v -> one v_mad_u64_u32 and one v_alignbit
// (d + 1) * v >> 16 for a 16-bit v, exactly.  Not `v_mad_u64_u32` (a quarter-rate 64-bit multiply on the symbol's critical
// path): d = dh * 2^16 + dl, so (d v + v) >> 16 = dh v + ((dl + 1) v >> 16), both products below 2^32 and both 24-bit multiplies.
__device__ __forceinline__ uint32_t scale_d(uint32_t d, uint32_t v)
{
#ifdef RC_SCALE_MUL64
    return (uint32_t)(((uint64_t)d * (uint64_t)v + (uint64_t)v) >> 16);
#else
    const uint32_t dh = d >> 16, dl = d & 0xFFFFu;
    return __umul24(dh, v) + ((__umul24(dl, v) + v) >> 16);
#endif
}

__device__ __forceinline__ uint32_t ones_below(uint32_t k) { return (1u << (k & 31u)) - 1u; }

// (Round 4 measured a register-resident window of four dwords -- shifted on a dword crossing, one LDS dword requested per symbol
// and first needed a whole symbol later -- against this two-dword re-read: +6 VALU per symbol and 3-4 % SLOWER on both coders
// (2.71 -> 2.81 ms of coder stage with the version-4 coder): a lone wave is bound by the instructions it issues, not by this LDS
// round trip, which already overlaps the next symbol's compare chain.)
struct LaneWin {
    const uint32_t *w;      // the lane's window in LDS
    uint32_t q;             // bits consumed - 1 (starts at 31: the first dword is the initial value)
    uint32_t w0, w1;        // dwords q / 32 and q / 32 + 1
    uint32_t last;          // the last dword pair a reader may touch: a lane past its symbols, or fed a corrupt stream, runs
                            // on garbage and its position may run anywhere -- an LDS access outside the workgroup's
                            // allocation raises a memory violation that the runtime turns into abort()
    __device__ __forceinline__ void init(const uint32_t *p, uint32_t rdw) { w = p; q = 31u; w0 = p[0]; w1 = p[1]; last = rdw - 2u; }
    // the next 32 unread bits: bits [s, s + 32) of w0:w1 with s = q % 32 + 1 in [1, 32]
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(w0, w1, ~q); }
    __device__ __forceinline__ void advance(uint32_t k) { q += k; const uint32_t i = min(q >> 5, last); w0 = w[i]; w1 = w[i + 1u]; }
};

// stage the byte windows of the workgroup's lanes (thread j * owner_stride holds the descriptor of lane j): win[j * rdw + d] = logical bytes 4d .. 4d + 3 of lane j, first byte in the
// most significant position; zero past the lane's byte count (what the reference's reader supplies past the end of a stream)
__device__ __forceinline__ void stage_windows(uint32_t *win, const uint8_t *__restrict__ bytes, const RcChunk &ch, int nl, uint32_t rdw, int nthreads, int owner_stride, int lane)
{
    const uint32_t total = (uint32_t)nl * rdw;
    const uint32_t nbf = ch.nbytes;
    for (uint32_t f0 = 0; f0 < total; f0 += 4u * (uint32_t)nthreads) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t f = f0 + (uint32_t)u * (uint32_t)nthreads + (uint32_t)lane;
            const uint32_t j = min(f / rdw, (uint32_t)nl - 1u), dq = f - (f / rdw) * rdw;
            const uint32_t off = (uint32_t)__shfl((int)ch.byte_off, (int)j * owner_stride, 64);
            const uint32_t nb = (uint32_t)__shfl((int)nbf, (int)j * owner_stride, 64);
            const bool back = (nb & RC_BACKWARDS) != 0u;
            const uint32_t n = nb & ~RC_BACKWARDS;
            uint32_t raw = 0;
            if (f < total && 4u * dq < n) {
                const uint8_t *src = back ? bytes + (size_t)off - 4u * (size_t)dq - 3u : bytes + (size_t)off + 4u * (size_t)dq;
                __builtin_memcpy(&raw, src, 4);
                if (!back) raw = __builtin_bswap32(raw);
                const uint32_t left = n - 4u * dq;                       // valid bytes of this dword: the top `left` of them
                if (left < 4u) raw &= 0xFFFFFFFFu << (8u * (4u - left));
            }
            v[u] = raw;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t f = f0 + (uint32_t)u * (uint32_t)nthreads + (uint32_t)lane;
            if (f < total) win[f] = v[u];
        }
    }
}

// one symbol of a lane: interval update + renormalisation on (low, d = span - 1, x); `t` = the next 32 unread bits.
// lo / d1 = scaled lower bound and width - 1 of the decoded symbol's slice of [0, span).
__device__ __forceinline__ uint32_t ffbh(uint32_t v) { uint32_t r; asm("v_ffbh_u32_e32 %0, %1" : "=v"(r) : "v"(v)); return r; }   // v != 0: no zero check
__device__ __forceinline__ void rc_renorm(uint32_t &low, uint32_t &d, uint32_t &x, uint32_t lo, uint32_t d1, uint32_t t, uint32_t &k_out)
{
    const uint32_t x1 = x - lo, low1 = low + lo, high1 = low1 + d1;
    const uint32_t n1 = ffbh(low1 ^ high1);                                          // low1 < high1 for every valid row: 0..31
    const uint32_t l1 = low1 << n1, h1 = (high1 << n1) | ~(0xFFFFFFFFu << n1);
    const uint32_t n2 = ffbh((((~l1) | h1) << 1) | 1u);                              // run of (low bit 1, high bit 0) behind the top bit
    const uint32_t k = n1 + n2;                                                      // <= 19 for a valid row (shifts use the low 5 / 6 bits)
    low = (l1 << n2) & 0x7FFFFFFFu;
    d = (d1 << k) | ~(0xFFFFFFFFu << k);                                             // span - 1: a span of 2^32 (a certain symbol's bounds renormalised) fits
    x = (uint32_t)(((((uint64_t)x1) << 32 | (uint64_t)t) << k) >> 32);
    k_out = k;
}

// Container version 4: the lanes' coder is a carry-PROPAGATING range coder (oracle/gpcc_oracle.c: cp_encode_core has the
// definition).  The encoder resolves carries in the bits it has written, so the decoder's state is (range, x = value - low):
// no `low`, no underflow bookkeeping -- a shift, a multiply, a compare, a subtract and one count-leading-zeros per binary symbol.
// lo / r1 = scaled lower bound and width of the decoded symbol's slice; `t` = the next 32 unread bits.
__device__ __forceinline__ void cp_renorm(uint32_t &range, uint32_t &x, uint32_t lo, uint32_t r1, uint32_t t, uint32_t &k_out)
{
    const uint32_t x1 = x - lo;
    const uint32_t k = ffbh(r1 | 0x8000u);   // a valid row leaves r1 >= 2^15; the OR keeps garbage rows (r1 = 0) at a defined shift
    range = r1 << k;
    x = (uint32_t)(((((uint64_t)x1) << 32 | (uint64_t)t) << k) >> 32);
    k_out = k;
}

constexpr int RC_CODER_CARRYLESS = 0, RC_CODER_CARRY = 1;   // torchac's coder (reference layout, versions 1-3) / version 4
__host__ __device__ inline int rc_coder_of_version(int version) { return version >= 4 ? RC_CODER_CARRY : RC_CODER_CARRYLESS; }

// Lanes of 3- and 5-entry rows, byte windows staged in LDS, rows in three register sets (header comment).  A lane's first
// symbol sits on a multiple of 16 (lanes are 2^llog >= 32 symbols) and `sym` has 3 bytes of slack behind the stream for the
// last group of its last lane.
// PH = rows per register set (16: a phase's symbols leave in one 16-byte store; 4: the short lanes of small levels, a quarter
// of the registers).  SOLO: the wave is its whole workgroup.  c0 = first lane (chunk descriptor) of this wave, lane = 0..63,
// win = [lpw][rdw] dwords of LDS for the byte windows.  cdf / sym are NOT __restrict__ const here: inside the fused kernel other
// workgroups wrote them a grid barrier ago, and only vector loads see that.
template <bool SOLO> __device__ __forceinline__ void rc_wave_sync()
{
    if (SOLO) __syncthreads();
    else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
}

template <int LP, int PH, bool SOLO, int CODER = RC_CODER_CARRYLESS>
__device__ __forceinline__ void rc_decode_lds_wave(const uint16_t *cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                   int nchunks, int c0, int lane, int lpw, uint32_t rdw, uint8_t *sym, uint32_t *win)
{
    static_assert(LP == 3 || LP == 5, "17-entry rows are decoded by rc_decode17_lds_wave");
    static_assert(PH == 16 || PH == 4, "phase length");
    constexpr int RS = LP == 3 ? 1 : 4;
    constexpr int DEPTH = RING_NPH * PH;
    const int c = c0 + lane;
    RcChunk ch = {0, 0, 0, 0, 0, 0};
    if (lane < lpw && c < nchunks) ch = chunks[c];
    uint32_t nmax = ch.n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    if (nmax == 0) return;
    stage_windows(win, bytes, ch, lpw, rdw, 64, 1, lane);
    rc_wave_sync<SOLO>();
    LaneWin in;
    in.init(win + (size_t)min(lane, lpw - 1) * rdw, rdw);
    const uint16_t *rowp = cdf + (size_t)ch.first * RS;
    const size_t rstep = (size_t)ch.stride * RS;
    struct Row { uint32_t a, b; };
    Row regs[RING_NPH][PH];
    auto fill = [&](int h) {
#pragma unroll
        for (int dd = 0; dd < PH; ++dd) {
            if (LP == 3) { regs[h][dd].a = rowp[0]; regs[h][dd].b = 0; }
            else { const uint2 q = *reinterpret_cast<const uint2 *>(rowp); regs[h][dd].a = q.x; regs[h][dd].b = q.y; }
            rowp += rstep;
        }
    };
    uint32_t low = 0, d = 0xFFFFFFFFu, x = in.w0, k;
#pragma unroll
    for (int h = 0; h < RING_NPH; ++h) fill(h);
    uint8_t *out = sym + ch.out;
    for (uint32_t i0 = 0; i0 < nmax; i0 += DEPTH) {
#pragma unroll
        for (int h = 0; h < RING_NPH; ++h) {
            uint32_t pack[PH / 4];
#pragma unroll
            for (int dd = 0; dd < PH; ++dd) {
                const uint32_t r0 = regs[h][dd].a, r1 = regs[h][dd].b;
                const uint32_t t = in.peek();
                uint32_t s, lo, d1;
                if constexpr (CODER == RC_CODER_CARRY) {
                    // (d holds the RANGE here; low is unused)
                    const uint32_t r = d >> 16;
                    if (LP == 3) {
                        const uint32_t t1 = __umul24(r, r0);
                        const bool ge = t1 <= x;
                        s = ge; lo = ge ? t1 : 0u;
                        d1 = ge ? d - t1 : t1;
                    } else {
                        const uint32_t t1 = __umul24(r, r0 & 0xFFFFu), t2 = __umul24(r, r0 >> 16), t3 = __umul24(r, r1 & 0xFFFFu);
                        const bool g1 = t1 <= x, g2 = t2 <= x, g3 = t3 <= x;
                        lo = g3 ? t3 : (g2 ? t2 : (g1 ? t1 : 0u));
                        const uint32_t hi = g3 ? d : (g2 ? t3 : (g1 ? t2 : t1));
                        s = (g1 ? 1u : 0u) + (g2 ? 1u : 0u) + (g3 ? 1u : 0u);
                        d1 = hi - lo;
                    }
                    if ((dd & 3) == 0) pack[dd >> 2] = s; else pack[dd >> 2] |= s << (8 * (dd & 3));
                    cp_renorm(d, x, lo, d1, t, k);
                } else {
                if (LP == 3) {
                    const uint32_t t1 = scale_d(d, r0);
                    const bool ge = t1 <= x;
                    s = ge; lo = ge ? t1 : 0u;
                    d1 = ge ? d - t1 : t1 - 1u;
                } else {
                    const uint32_t t1 = scale_d(d, r0 & 0xFFFFu), t2 = scale_d(d, r0 >> 16), t3 = scale_d(d, r1 & 0xFFFFu);
                    const bool g1 = t1 <= x, g2 = t2 <= x, g3 = t3 <= x;
                    lo = g3 ? t3 : (g2 ? t2 : (g1 ? t1 : 0u));
                    const uint32_t hi = g3 ? d + 1u : (g2 ? t3 : (g1 ? t2 : t1));
                    s = (g1 ? 1u : 0u) + (g2 ? 1u : 0u) + (g3 ? 1u : 0u);
                    d1 = hi + ~lo;
                }
                if ((dd & 3) == 0) pack[dd >> 2] = s; else pack[dd >> 2] |= s << (8 * (dd & 3));
                rc_renorm(low, d, x, lo, d1, t, k);
                }
                in.advance(k);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i0 + (uint32_t)(DEPTH + h * PH) < nmax) fill(h);
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t ib = i0 + (uint32_t)(h * PH);
            bool wide = false;
            if constexpr (PH == 16) {
                wide = ib + 16u <= ch.n;
                if (wide) *reinterpret_cast<uint4 *>(out + ib) = make_uint4(pack[0], pack[1], pack[2], pack[3]);
            }
            if (!wide) {
#pragma unroll
                for (int q = 0; q < PH / 4; ++q)
                    if (ib + 4u * (uint32_t)q < ch.n) *reinterpret_cast<uint32_t *>(out + ib + 4 * q) = pack[q];
            }
        }
    }
}

// 17-entry rows: a 16-lane group per coder lane as in k_rc_decode17, the byte windows staged in LDS, every lane's own entry
// of the coming rows in three register sets.
template <int PH, bool SOLO, int CODER = RC_CODER_CARRYLESS>
__device__ __forceinline__ void rc_decode17_lds_wave(const uint16_t *cdf, const uint8_t *__restrict__ bytes, const RcChunk *__restrict__ chunks,
                                                     int nchunks, int c0, int lane, uint32_t rdw, uint8_t *sym, uint32_t *win)
{
    static_assert(PH == 16 || PH == 4, "phase length");
    constexpr int DEPTH = RING_NPH * PH;
    static_assert(2 * DEPTH <= RC_ROW_LOOKAHEAD, "row look-ahead exceeds the capacity contract (rc_rows_capacity)");
    const int grp = lane >> 4, kk = lane & 15;
    const int c = c0 + grp;
    RcChunk ch = {0, 0, 0, 0, 0, 0};
    if (c < nchunks) ch = chunks[c];
    uint32_t nmax = ch.n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    if (nmax == 0) return;
    stage_windows(win, bytes, ch, 4, rdw, 64, 16, lane);   // coder lane j's descriptor lives in thread 16 j
    rc_wave_sync<SOLO>();
    LaneWin in;
    in.init(win + (size_t)grp * rdw, rdw);
    // compact row: v[1..15] at [0..14]; lane 0 stands for v[0] = 0 and reads the unused slot 15
    const uint16_t *rowp = cdf + (size_t)ch.first * 16 + (size_t)(kk ? kk - 1 : 15);
    const size_t rstep = (size_t)ch.stride * 16;
    uint32_t regs[RING_NPH][PH];
    auto fill = [&](int h) {
#pragma unroll
        for (int dd = 0; dd < PH; ++dd) { regs[h][dd] = rowp[0]; rowp += rstep; }
    };
    const int g16 = grp << 4;
    uint32_t low = 0, d = 0xFFFFFFFFu, x = in.w0, k;
#pragma unroll
    for (int h = 0; h < RING_NPH; ++h) fill(h);
    uint8_t *out = sym + ch.out;
    for (uint32_t i0 = 0; i0 < nmax; i0 += DEPTH) {
#pragma unroll
        for (int h = 0; h < RING_NPH; ++h) {
            uint32_t pack[PH / 4];
#pragma unroll
            for (int dd = 0; dd < PH; ++dd) {
                const uint32_t v = regs[h][dd];
                const uint32_t tw = in.peek();
                const uint32_t t = kk ? (CODER == RC_CODER_CARRY ? __umul24(d >> 16, v) : scale_d(d, v)) : 0u;   // (CARRY: d holds the range)
                const uint64_t bal = __ballot(t <= x);
                const uint32_t half = (grp & 2) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                const uint32_t bits = (half >> ((grp & 1) * 16)) & 0xFFFFu;      // this group's lanes with t <= x: lanes 0..s
                const uint32_t s = ((uint32_t)__popc(bits) - 1u) & 15u;
                const uint32_t lo = (uint32_t)__shfl((int)t, g16 + (int)s);
                const uint32_t nx = (uint32_t)__shfl((int)t, g16 + (int)min(s + 1u, 15u));
                if ((dd & 3) == 0) pack[dd >> 2] = s; else pack[dd >> 2] |= s << (8 * (dd & 3));
                if constexpr (CODER == RC_CODER_CARRY) {
                    cp_renorm(d, x, lo, (s == 15u ? d : nx) - lo, tw, k);
                } else {
                    const uint32_t d1 = (s == 15u ? d : nx - 1u) - lo;
                    rc_renorm(low, d, x, lo, d1, tw, k);
                }
                in.advance(k);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i0 + (uint32_t)(DEPTH + h * PH) < nmax) fill(h);   // (wave-uniform) rows no lane will use are not fetched
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t ib = i0 + (uint32_t)(h * PH);
            if (kk == 0) {
                bool wide = false;
                if constexpr (PH == 16) {
                    wide = ib + 16u <= ch.n;
                    if (wide) *reinterpret_cast<uint4 *>(out + ib) = make_uint4(pack[0], pack[1], pack[2], pack[3]);
                }
                if (!wide) {
#pragma unroll
                    for (int q = 0; q < PH / 4; ++q)
                        if (ib + 4u * (uint32_t)q < ch.n) *reinterpret_cast<uint32_t *>(out + ib + 4 * q) = pack[q];
                }
            }
        }
    }
}


}  // namespace gpcc
