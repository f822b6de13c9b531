// rc_format.hpp -- the byte-level format of a coded stream: lane descriptors, how a stream is cut into lanes and chunks per
// container version, the varint / Rice chunk tables and their parser.  Plain C++ (no HIP): the kernels include it through
// rangecoder.hpp, and the host-only sanitizer build (tools/asan_host.sh) compiles the SAME parser that gpcc_decode runs on
// untrusted bytes.
#pragma once
#include <algorithm>
#include <vector>

#include "errors.hpp"

#if defined(__HIPCC__)
#define GPCC_HD __host__ __device__ __forceinline__
#else
#define GPCC_HD inline
#endif

namespace gpcc {

// One chunk = one lane.  Element t of the chunk (packed symbol word on encode, compact CDF row on
// decode) lives at index first + t * stride: chunks of one stream are interleaved so that the 64
// lanes of a wave touch consecutive addresses.
struct RcChunk {
    uint32_t first;     // index of element 0
    uint32_t stride;    // elements between consecutive symbols of this chunk (= chunks in the stream)
    uint32_t n;         // symbols in the chunk
    uint32_t out;       // decode: index of the chunk's first symbol in the (raster-ordered) output
    uint32_t byte_off;  // decode: offset of the lane's FIRST byte in the uploaded file (a backwards lane: its chunk's last byte)
    uint32_t nbytes;    // decode: bytes the lane may read (its chunk's byte count) | RC_BACKWARDS
};
constexpr uint32_t RC_BACKWARDS = 0x80000000u;   // RcChunk::nbytes flag: the lane's bytes run towards lower addresses

// compact CDF row: only the interior values v[1..Lp-2] are stored (v[0] = 0, v[Lp-1] is never read)
static inline int rc_row_stride(int lp) { return lp == 3 ? 1 : lp == 5 ? 4 : 16; }  // uint16 units
// position of raster rank r inside a stream cut into 2^chunk_log2-symbol chunks (chunk_log2 = 0: one chunk)
GPCC_HD uint32_t rc_interleaved(uint32_t r, int chunk_log2, uint32_t nch)
{
    return chunk_log2 ? (r & ((1u << chunk_log2) - 1u)) * nch + (r >> chunk_log2) : r;
}

// The decoder fetches compact rows RC_ROW_LOOKAHEAD symbols ahead without clamping: the row buffer of a stream of
// nch chunks of (at most) S symbols must hold rc_rows_capacity(nch, S) rows (what lies past the last row is never used).
constexpr int RC_ROW_LOOKAHEAD = 96;   // staged kernels: the prologue fetches a whole ring of 48 rows whatever the lane's length; refills stop at the wave's longest lane
static inline int64_t rc_rows_capacity(int64_t nch, int64_t S) { return (S + RC_ROW_LOOKAHEAD) * nch; }

// How a stream of n symbols is cut (by container version).  A LANE is what one coder state covers: 2^llog consecutive
// symbols (raster order), coded independently of every other lane.  A CHUNK is what the container counts bytes for.
//   version 1      chunk = lane = 2^chunk_log2 symbols, u16 byte count per chunk.
//   version 2      as 1 with the chunk size following the level's size: 2^clog, clog = clamp(ceil_log2(ceil(n / 256)), 7,
//                  chunk_log2) -- about 256 lanes per stream until the header's chunk_log2 is reached (the decoder's
//                  latency per stream is lane length x time per symbol, whatever the level's size).
//   version 3      a chunk is TWO lanes sharing one byte count: the first half of the chunk's symbols is coded forwards from
//                  the chunk's first byte, the second half backwards from its last byte (the bytes of that lane are stored
//                  in reverse order).  A coder's flush leaves its last symbols decodable whatever bits follow, so each
//                  lane simply reads on into the other's bytes.  One count per two lanes; the table is the first count as
//                  a LEB128 varint, then (more than one chunk) a byte k in 0..7 and the zigzag differences to the previous
//                  count as Rice codes -- bits MSB first, zero-padded to a byte: q = z >> k < 16: q ones, a zero, the low
//                  k bits of z; otherwise sixteen ones and z in 32 bits; k = the value giving the fewest bits, the smallest
//                  on a tie.  The chunks of a stream differ by a few bytes: ~5 bits of table per 2 lanes instead of 16 per
//                  lane.  Chunk size 2^clog with clog = clamp(ceil_log2(ceil(n / 128)), 7, chunk_log2): the same ~256
//                  lanes per stream as version 2.
//   chunk_log2 = 0 the reference layout: one lane per stream, no table.
struct RcPlan {
    int llog;           // lane size log2 (0 with nlanes == 1: the whole stream)
    uint32_t nlanes;    // coder states of the stream
    uint32_t nchunks;   // byte-counted units of the stream's table
    bool dual;          // version 3: lanes 2c (forwards) and 2c + 1 (backwards) share chunk c
    int64_t lane_syms(int64_t n, uint32_t l) const { const int64_t S = llog || nlanes > 1 ? (int64_t)1 << llog : n, r = n - (int64_t)l * S; return r < 0 ? 0 : (r < S ? r : S); }
};
static inline int rc_ceil_log2(int64_t v) { int c = 0; while (((int64_t)1 << c) < v) ++c; return c; }
static inline int rc_level_chunk_log2(int64_t n, int chunk_log2, int version)
{
    if (chunk_log2 == 0 || version < 2) return chunk_log2;
    const int c = rc_ceil_log2((n + (version >= 3 ? 127 : 255)) / (version >= 3 ? 128 : 256));
    const int lo = chunk_log2 < 7 ? chunk_log2 : 7;
    return c < lo ? lo : (c > chunk_log2 ? chunk_log2 : c);
}
static inline RcPlan rc_plan(int64_t n, int chunk_log2, int version)
{
    RcPlan p = {0, 1u, 1u, false};
    if (chunk_log2 == 0) return p;
    const int clog = rc_level_chunk_log2(n, chunk_log2, version);
    p.dual = version >= 3;
    p.llog = p.dual ? clog - 1 : clog;
    if (p.llog < 4) p.llog = 4;   // the staged decoders store 16 symbols at a time: lanes of at least 16 symbols (the API's chunk_log2 >= 6 never gets here)
    const int64_t nl = (std::max<int64_t>(n, 1) + ((int64_t)1 << p.llog) - 1) >> p.llog;
    p.nlanes = (uint32_t)nl;
    p.nchunks = p.dual ? (uint32_t)((nl + 1) / 2) : (uint32_t)nl;
    return p;
}
// chunk tables: u16 per chunk (versions 1, 2); version 3: LEB128 first count + Rice-coded differences (rc_table_*)
static inline size_t rc_varint_size(uint32_t v) { return v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5; }
static inline size_t rc_varint_put(uint8_t *o, uint32_t v) { size_t k = 0; while (v >= 128u) { o[k++] = (uint8_t)(v | 128u); v >>= 7; } o[k++] = (uint8_t)v; return k; }
// returns bytes read, 0 on a malformed / truncated varint
static inline size_t rc_varint_get(const uint8_t *p, size_t avail, uint32_t *v)
{
    uint32_t r = 0;
    for (size_t k = 0; k < 5 && k < avail; ++k) {
        r |= (uint32_t)(p[k] & 127u) << (7 * k);
        if (!(p[k] & 128u)) { if (k == 4 && p[k] > 15u) return 0; *v = r; return k + 1; }
    }
    return 0;
}
// Version-3 chunk table.  b(c) = byte count of chunk c (both lanes), c < nch.
constexpr int RC_TAB_KMAX = 7;
constexpr uint32_t RC_TAB_ESC = 16u, RC_TAB_ESC_BITS = 48u;
GPCC_HD uint32_t rc_zigzag(uint32_t cur, uint32_t prev) { const int32_t d = (int32_t)(cur - prev); return ((uint32_t)d << 1) ^ (uint32_t)(d >> 31); }
GPCC_HD uint32_t rc_tab_cost(uint32_t z, int k) { const uint32_t q = z >> k; return q < RC_TAB_ESC ? q + 1u + (uint32_t)k : RC_TAB_ESC_BITS; }
GPCC_HD uint32_t rc_tab_bytes(uint32_t first, uint32_t nch, const uint64_t bits[RC_TAB_KMAX + 1], int *kbest)
{
    const uint32_t v = first < (1u << 7) ? 1u : first < (1u << 14) ? 2u : first < (1u << 21) ? 3u : first < (1u << 28) ? 4u : 5u;
    int k = 0;
    for (int j = 1; j <= RC_TAB_KMAX; ++j) k = bits[j] < bits[k] ? j : k;
    if (kbest) *kbest = k;
    return nch < 2u ? v : v + 1u + (uint32_t)((bits[k] + 7u) >> 3);
}
template <class F> static inline size_t rc_table_size(F b, uint32_t nch, int *kbest = nullptr)
{
    uint64_t bits[RC_TAB_KMAX + 1] = {0};
    if (!nch) return 0;
    for (uint32_t c = 1; c < nch; ++c) {
        const uint32_t z = rc_zigzag(b(c), b(c - 1));
        for (int k = 0; k <= RC_TAB_KMAX; ++k) bits[k] += rc_tab_cost(z, k);
    }
    return rc_tab_bytes(b(0), nch, bits, kbest);
}
template <class F> static inline size_t rc_table_put(uint8_t *o, F b, uint32_t nch)
{
    int k = 0;
    if (!nch) return 0;
    (void)rc_table_size(b, nch, &k);
    size_t pos = rc_varint_put(o, b(0));
    if (nch < 2u) return pos;
    o[pos++] = (uint8_t)k;
    uint64_t acc = 0; int na = 0;   // MSB first
    auto put = [&](uint32_t v, int n) { acc = (acc << n) | v; na += n; while (na >= 8) { o[pos++] = (uint8_t)(acc >> (na - 8)); na -= 8; } };
    for (uint32_t c = 1; c < nch; ++c) {
        const uint32_t z = rc_zigzag(b(c), b(c - 1)), q = z >> k;
        if (q < RC_TAB_ESC) { put((1u << (q + 1u)) - 2u, (int)q + 1); if (k) put(z & ((1u << k) - 1u), k); }
        else { put(0xFFFFu, 16); put(z >> 16, 16); put(z & 0xFFFFu, 16); }
    }
    if (na) put(0u, 8 - na);
    return pos;
}
// reads the table of nch chunks into cb; returns its bytes, 0 when malformed / truncated
static inline size_t rc_table_get(const uint8_t *p, size_t avail, uint32_t *cb, uint32_t nch)
{
    size_t pos = rc_varint_get(p, avail, &cb[0]);
    if (!pos || nch < 2u) return pos;
    if (pos >= avail || p[pos] > (uint8_t)RC_TAB_KMAX) return 0;
    const int k = p[pos++];
    uint64_t bit = (uint64_t)pos * 8u;
    const uint64_t end = (uint64_t)avail * 8u;
    auto get = [&](uint32_t &dst) -> bool { if (bit >= end) return false; dst = (p[bit >> 3] >> (7u - (bit & 7u))) & 1u; ++bit; return true; };
    for (uint32_t c = 1; c < nch; ++c) {
        uint32_t q = 0, one = 1, z = 0;
        while (q < RC_TAB_ESC) { if (!get(one)) return 0; if (!one) break; ++q; }
        if (q < RC_TAB_ESC) { for (int i = 0; i < k; ++i) { if (!get(one)) return 0; z = z << 1 | one; } z |= q << k; }
        else for (int i = 0; i < 32; ++i) { if (!get(one)) return 0; z = z << 1 | one; }
        const int64_t v = (int64_t)cb[c - 1] + ((int64_t)(z >> 1) ^ -(int64_t)(z & 1u));
        if (v < 0 || v > 0x7FFFFFFF) return 0;
        cb[c] = (uint32_t)v;
    }
    return (size_t)((bit + 7u) >> 3);
}
// Lane descriptors of one stream from its table.  `tab` points at the stream body (table, then the chunk payloads) of `len`
// bytes which starts at byte `off` of the uploaded file; lanes[0 .. plan.nlanes) are filled, *max_bytes = the longest lane's
// byte window (a dual chunk's lanes both see the whole chunk).  Returns 0, or a message.
static inline const char *rc_parse_table(const uint8_t *tab, int64_t off, int64_t len, const RcPlan &plan, int64_t n, int version, RcChunk *lanes, uint32_t *max_bytes)
{
    uint32_t mb = 0;
    if (plan.nlanes == 1 && plan.llog == 0 && !plan.dual && version == 0) {
        lanes[0] = RcChunk{0, 1, (uint32_t)n, 0, (uint32_t)off, (uint32_t)len};
        *max_bytes = (uint32_t)len;
        return nullptr;
    }
    int64_t t = 0;                                   // table cursor
    std::vector<uint32_t> cb(plan.nchunks);
    if (version >= 3) {
        t = (int64_t)rc_table_get(tab, (size_t)std::max<int64_t>(len, 0), cb.data(), plan.nchunks);
        if (!t) return "chunk table: malformed";
    } else {
        for (uint32_t c = 0; c < plan.nchunks; ++c) {
            if (t + 2 > len) return "stream shorter than its chunk table";
            cb[c] = tab[t] | tab[t + 1] << 8;
            t += 2;
        }
    }
    int64_t p = off + t;
    const int64_t end = off + len;
    for (uint32_t c = 0; c < plan.nchunks; ++c) {
        if (p + (int64_t)cb[c] > end) return "chunk overruns its stream";
        if (plan.dual) {
            const uint32_t l0 = 2 * c, l1 = 2 * c + 1;
            lanes[l0] = RcChunk{l0, plan.nlanes, (uint32_t)plan.lane_syms(n, l0), (uint32_t)((int64_t)l0 << plan.llog), (uint32_t)p, cb[c]};
            if (l1 < plan.nlanes)   // backwards from the chunk's last byte
                lanes[l1] = RcChunk{l1, plan.nlanes, (uint32_t)plan.lane_syms(n, l1), (uint32_t)((int64_t)l1 << plan.llog), (uint32_t)(p + cb[c]) - 1u, cb[c] | RC_BACKWARDS};
        } else {
            lanes[c] = RcChunk{c, plan.nlanes, (uint32_t)plan.lane_syms(n, c), (uint32_t)((int64_t)c << plan.llog), (uint32_t)p, cb[c]};
        }
        mb = cb[c] > mb ? cb[c] : mb;
        p += cb[c];
    }
    if (p != end) return "stream has trailing bytes";
    *max_bytes = mb;
    return nullptr;
}

// The staged decoders keep every lane's byte window (+ the dwords the reader runs ahead) in LDS; the 16-ary kernel runs four
// coder states per wave.  A version-3 chunk must fit (both lanes see the whole chunk): ~16 KiB for 16-ary, ~64 KiB for the
// other streams.  The encoder refuses to write a chunk its decoder could not read (possible only at chunk_log2 >= 13 with a
// model that spends > 8 bits per 16-ary symbol).
constexpr int RC_RING_DEPTH = 48;   // rows fetched ahead of the coder: three register sets of 16 (rangecoder.hip)
constexpr uint32_t RC_LDS_CAP = 64u * 1024u;    // dynamic LDS of a decode workgroup (one wave)
static inline uint64_t rc_window_dwords(uint32_t max_bytes) { return ((uint64_t)max_bytes + 3u) / 4u + 3u; }
static inline uint32_t rc_ring_bytes(int) { return 0u; }   // (the row ring lives in registers: rangecoder.hip)
static inline bool rc_window_fits(int lp, uint32_t max_bytes) { return rc_window_dwords(max_bytes) * 4u * (lp == 17 ? 4u : 1u) <= RC_LDS_CAP - rc_ring_bytes(lp); }

static inline uint32_t rc_scratch_stride(uint32_t max_syms) { return (2u * max_syms + 32u + 15u) & ~15u; }

}  // namespace gpcc
