// hostcoder.hip -- torchac's coder on HOST arrays (plain C++; no kernel in this file).
//
// torchac.encode_int16_normalized_cdf / decode_int16_normalized_cdf (requirements.txt:6; torchac 0.9.3's C++ extension) code a
// WHOLE tensor as one range-coder stream on one CPU thread; TC-GS and CAT-3DGS call it that way for every attribute
// (TC-GS/utils/encodings.py:84-176: the CDF table is built with torch.distributions on the GPU, moved to the CPU, coded there).
// A single stream is one dependent chain: on the MI355X it decodes on ONE lane at ~2 Msymbols/s (round 3's torchac shim: 4.8 /
// 2.2 Msymbols/s encode / decode, slower than any CPU), while a host core runs the same chain at > 10 Msymbols/s.  So the
// drop-in for torchac runs the coder here, on the host, and leaves the GPU what it is good at (building the integer CDF rows:
// gauspcc_amd/torchac.py); callers that can choose their format use the chunked device coders (gsac_encode*, gpcc_rc_*).
//
// The loops are the lane loops of rangecoder.hip (carry-less 32-bit coder, 16-bit counts: arithmetic_kernel.cu:94-163,
// 237-356) with the same shortcuts -- renormalisation by count-leading-zeros instead of bit by bit, a
// division-free symbol search on the scaled bounds -- and produce / consume torchac's bytes exactly (tests: bytes == the
// oracle's restatement of the reference loop).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gauspcc.h"
#include "errors.hpp"
#include "hostcoder.hpp"

namespace {

struct BitSink {
    uint8_t *out; int64_t cap, len;
    uint64_t acc; int n;   // n valid bits at the top of acc
    inline void put(uint32_t bits, int k)   // k in [1, 32]
    {
        acc |= (uint64_t)bits << (64 - n - k);
        n += k;
        while (n >= 8) { if (len < cap) out[len] = (uint8_t)(acc >> 56); ++len; acc <<= 8; n -= 8; }
    }
    inline void put_run(uint32_t bit, uint64_t k)
    {
        const uint32_t pat = bit ? 0xFFFFFFFFu : 0u;
        while (k >= 32) { put(pat, 32); k -= 32; }
        if (k) put(pat >> (32 - k), (int)k);
    }
};

inline int clz32h(uint32_t v) { return v ? __builtin_clz(v) : 32; }

// rows of the table as the coder wants them: either the caller's int16 rows, or -- torchac's own calling convention, a float
// table on the host (encode_float_cdf / decode_float_cdf on CPU tensors) -- integerised on the fly, one row per symbol, as
// torchac's _convert_to_int_and_normalize does it: rint(cdf * (2^16 - (Lp - 1))) + j, kept modulo 2^16 (kit/op.py:67-79 is the
// same statement).  fp32 multiply, round-half-even: the values torch computes for `cdf.mul(f).round()`.
struct RowsU16 {
    const uint16_t *cdf; int lp;
    inline const uint16_t *row(int64_t i) { return cdf + i * lp; }
};
struct RowsF32 {
    const float *cdf; int lp; uint16_t *tmp; float scale;
    inline const uint16_t *row(int64_t i)
    {
        const float *r = cdf + i * lp;
        for (int j = 0; j < lp; ++j) tmp[j] = (uint16_t)((int32_t)__builtin_rintf(r[j] * scale) + j);
        return tmp;
    }
};

// BOUNDS: bounds(i, &c_low, &c_high) -> the coded symbol's interval [c_low, c_high) in 2^-16 units (false: a bad symbol, message set)
template <typename BOUNDS>
int host_encode_core(BOUNDS bounds, int64_t n, uint8_t *out, int64_t cap, int64_t *nbytes_out)
{
    BitSink w = {out, cap, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t c_low, c_high;
        if (!bounds(i, &c_low, &c_high)) return GPCC_ERR_ARG;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        // E1 / E2: the n1 leading bits on which low and high agree leave at once (the first with the pending run behind it)
        const int n1 = clz32h(low ^ high);
        if (n1) {
            const uint32_t bits = n1 == 32 ? low : low >> (32 - n1);
            const uint32_t b = bits >> (n1 - 1);
            w.put(b, 1);
            if (pending) { w.put_run(b ^ 1u, pending); pending = 0; }
            if (n1 > 1) w.put(bits & ((1u << (n1 - 1)) - 1u), n1 - 1);
            low = n1 == 32 ? 0u : low << n1;
            high = n1 == 32 ? 0xFFFFFFFFu : (high << n1) | ((1u << n1) - 1u);
        }
        // E3: low = 01.., high = 10.. -- drop the second bit n2 times
        int n2 = clz32h(~(low << 1));
        const int h2 = clz32h(high << 1);
        n2 = n2 < h2 ? n2 : h2;
        n2 = n2 < 31 ? n2 : 31;
        if (n2) {
            pending += (uint64_t)n2;
            low = (low << n2) & 0x7FFFFFFFu;
            high = (high << n2) | 0x80000000u | ((1u << n2) - 1u);
        }
    }
    pending += 1;
    const uint32_t b = low < 0x40000000u ? 0u : 1u;
    w.put(b, 1);
    w.put_run(b ^ 1u, pending);
    if (w.n) w.put(0u, 8 - w.n);
    *nbytes_out = w.len;
    if (w.len > cap) return gpcc::fail(GPCC_ERR_ARG, "output buffer of %lld bytes, the stream takes %lld", (long long)cap, (long long)w.len);
    return GPCC_OK;
}

template <typename ROWS>
int host_encode(const int16_t *sym, ROWS rows, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out)
{
    const int top = lp - 2;
    return host_encode_core([&](int64_t i, uint32_t *c_low, uint32_t *c_high) -> bool {
        const int s = sym[i];
        if (s < 0 || s > top) { (void)gpcc::fail(GPCC_ERR_ARG, "symbol %d at %lld outside [0, %d]", s, (long long)i, top); return false; }
        const uint16_t *row = rows.row(i);
        *c_low = row[s]; *c_high = s == top ? 0x10000u : row[s + 1];
        return true;
    }, n, out, cap, nbytes_out);
}

template <typename ROWS>
int host_decode(ROWS rows, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out)
{
    // bit reservoir: `value` holds the 32 bits at the read position; bits past the end read as zero (arithmetic_kernel.cu:244-262)
    uint64_t res = 0; int nres = 0; int64_t ptr = 0;
    auto take = [&](int k) -> uint32_t {   // k in [0, 32]
        if (k == 0) return 0u;
        while (nres < k) { const uint64_t byte = ptr < nbytes ? bytes[ptr] : 0u; ++ptr; res |= byte << (56 - nres); nres += 8; }
        const uint32_t v = (uint32_t)(res >> (64 - k));
        res <<= k; nres -= k;
        return v;
    };
    uint32_t low = 0, high = 0xFFFFFFFFu, value = take(32);
    const int top = lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const auto row = rows.row(i);
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        const uint32_t x = value - low;
        // the reference picks the largest s with row[s] <= count, count = ((x + 1) 2^16 - 1) / span; row[s] <= count  <=>
        // (span row[s]) >> 16 <= x, exactly (rangecoder.hip header): search on the scaled bounds, no division
        int lo_i = 0, hi_i = top + 1;
        while (lo_i + 1 < hi_i) {
            const int m = (lo_i + hi_i) >> 1;
            if ((uint32_t)((span * (uint64_t)row[m]) >> 16) <= x) lo_i = m; else hi_i = m;
        }
        const int s = lo_i;
        sym_out[i] = (int16_t)s;
        const uint32_t c_low = row[s], c_high = s == top ? 0x10000u : row[s + 1];
        high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        const int n1 = clz32h(low ^ high);
        if (n1) {
            low = n1 == 32 ? 0u : low << n1;
            high = n1 == 32 ? 0xFFFFFFFFu : (high << n1) | ((1u << n1) - 1u);
            value = n1 == 32 ? take(32) : (value << n1) | take(n1);
        }
        int n2 = clz32h(~(low << 1));
        const int h2 = clz32h(high << 1);
        n2 = n2 < h2 ? n2 : h2;
        n2 = n2 < 31 ? n2 : 31;
        if (n2) {
            low = (low << n2) & 0x7FFFFFFFu;
            high = (high << n2) | 0x80000000u | ((1u << n2) - 1u);
            // every underflow step maps value -> 2 (value - 2^30) + next bit: n2 of them keep the top bit and shift the rest
            value = ((value << n2) | take(n2)) ^ 0x80000000u;
        }
    }
    return GPCC_OK;
}

}  // namespace

extern "C" int gsac_host_encode_u16(const int16_t *sym, const uint16_t *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out)
{
    if (!sym || !cdf || !out || !nbytes_out || n < 0 || lp < 2) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    return host_encode(sym, RowsU16{cdf, lp}, n, lp, out, cap, nbytes_out);
}

extern "C" int gsac_host_decode_u16(const uint16_t *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out)
{
    if (!cdf || (!bytes && nbytes) || !sym_out || n < 0 || nbytes < 0 || lp < 2) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    return host_decode(RowsU16{cdf, lp}, bytes, nbytes, n, lp, sym_out);
}

extern "C" int gsac_host_encode_f32(const int16_t *sym, const float *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out)
{
    if (!sym || !cdf || !out || !nbytes_out || n < 0 || lp < 2 || lp > 65536) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    std::vector<uint16_t> tmp((size_t)lp);
    return host_encode(sym, RowsF32{cdf, lp, tmp.data(), (float)(65536 - (lp - 1))}, n, lp, out, cap, nbytes_out);
}

extern "C" int gsac_host_decode_f32(const float *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out)
{
    if (!cdf || (!bytes && nbytes) || !sym_out || n < 0 || nbytes < 0 || lp < 2 || lp > 65536) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    std::vector<uint16_t> tmp((size_t)lp);
    return host_decode(RowsF32{cdf, lp, tmp.data(), (float)(65536 - (lp - 1))}, bytes, nbytes, n, lp, sym_out);
}

// ---- the reference-layout container of gpcc_encode / gpcc_decode (chunk_log2 = 0: one torchac stream per level and stage) on this coder
namespace gpcc {

// streams[k] = the packed (c_low | (c_high - 1) << 16) words of stream k (what the heads emit), n[k] symbols; out[k] receives its bytes.
// The streams of an encode are independent (teacher-forced): a pool of native threads takes them longest first.
int host_encode_streams(const uint32_t *const *streams, const int64_t *n, int nstreams, std::vector<std::vector<uint8_t>> *out, int threads)
{
    out->assign((size_t)nstreams, {});
    std::vector<int> order((size_t)nstreams);
    for (int k = 0; k < nstreams; ++k) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return n[a] > n[b]; });
    if (threads <= 0) threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    threads = std::max(1, std::min(threads, nstreams));
    std::atomic<int> next{0}, bad{0};
    auto work = [&]() {
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= nstreams) return;
            const int k = order[(size_t)t];
            std::vector<uint8_t> &o = (*out)[(size_t)k];
            o.resize((size_t)(2 * n[k] + 64));          // a symbol costs at most 16 bits (counts >= 1 of 2^16), + flush
            const uint32_t *w = streams[k];
            int64_t nb = 0;
            const int rc = host_encode_core([&](int64_t i, uint32_t *c_low, uint32_t *c_high) -> bool {
                *c_low = w[i] & 0xFFFFu; *c_high = (w[i] >> 16) + 1u;
                return *c_low < *c_high;
            }, n[k], o.data(), (int64_t)o.size(), &nb);
            if (rc != GPCC_OK) { bad.store(1); return; }
            o.resize((size_t)nb);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    return bad.load() ? fail(GPCC_ERR_HIP, "internal: an empty coder interval in a reference-layout stream") : GPCC_OK;
}

// host_decode for the three alphabets of the codec (LP = 3 / 5 / 17, compact rows of RS uint16): the same decoder -- same symbol rule, same
// renormalisation -- with the symbol search unrolled and BRANCH-FREE (the generic binary search mispredicts on every symbol of a well-coded
// stream): the largest s with (span * v[s]) >> 16 <= x is built bit by bit from the top, v[0] = 0 always qualifies.
template <int LP, int RS>
static int host_decode_compact_t(const uint16_t *rows, const uint8_t *bytes, int64_t nbytes, int64_t n, uint8_t *sym_out)
{
    uint64_t res = 0; int nres = 0; int64_t ptr = 0;
    auto take = [&](int k) -> uint32_t {
        if (k == 0) return 0u;
        while (nres < k) { const uint64_t byte = ptr < nbytes ? bytes[ptr] : 0u; ++ptr; res |= byte << (56 - nres); nres += 8; }
        const uint32_t v = (uint32_t)(res >> (64 - k));
        res <<= k; nres -= k;
        return v;
    };
    uint32_t low = 0, high = 0xFFFFFFFFu, value = take(32);
    constexpr int top = LP - 2;
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t *r = rows + i * RS;                 // r[m - 1] = v[m], m = 1 .. top
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
        const uint32_t x = value - low;
        int s = 0;
#pragma GCC unroll 4
        for (int step = (top + 1) >> 1; step >= 1; step >>= 1) {
            const int c = s + step;                         // (top + 1 is a power of two: c <= top)
            s = (uint32_t)((span * (uint64_t)r[c - 1]) >> 16) <= x ? c : s;
        }
        sym_out[i] = (uint8_t)s;
        const uint32_t c_low = s ? r[s - 1] : 0u, c_high = s == top ? 0x10000u : r[s];
        high = (low - 1u) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        const int n1 = clz32h(low ^ high);
        if (n1) {
            low = n1 == 32 ? 0u : low << n1;
            high = n1 == 32 ? 0xFFFFFFFFu : (high << n1) | ((1u << n1) - 1u);
            value = n1 == 32 ? take(32) : (value << n1) | take(n1);
        }
        int n2 = clz32h(~(low << 1));
        const int h2 = clz32h(high << 1);
        n2 = n2 < h2 ? n2 : h2;
        n2 = n2 < 31 ? n2 : 31;
        if (n2) {
            low = (low << n2) & 0x7FFFFFFFu;
            high = (high << n2) | 0x80000000u | ((1u << n2) - 1u);
            value = ((value << n2) | take(n2)) ^ 0x80000000u;
        }
    }
    return GPCC_OK;
}

// one stream of n symbols under compact CDF rows (lp - 1 symbols per row): torchac's decoder, symbols as bytes
int host_decode_compact(const uint16_t *rows, int lp, const uint8_t *bytes, int64_t nbytes, int64_t n, uint8_t *sym_out)
{
    if (lp == 3) return host_decode_compact_t<3, 1>(rows, bytes, nbytes, n, sym_out);
    if (lp == 5) return host_decode_compact_t<5, 4>(rows, bytes, nbytes, n, sym_out);
    if (lp == 17) return host_decode_compact_t<17, 16>(rows, bytes, nbytes, n, sym_out);
    return fail(GPCC_ERR_ARG, "internal: compact rows of %d entries", lp);
}

}  // namespace gpcc

// ---- many small files on native threads (the per-slice `.b` files of the attribute loops: include/gauspcc.h)
extern "C" int gpcc_write_files(const char *const *paths, const uint8_t *const *data, const int64_t *sizes, int n, int threads)
{
    if (n < 0 || (n > 0 && (!paths || !data || !sizes))) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    if (threads <= 0) threads = 8;
    threads = threads < n ? threads : (n > 0 ? n : 1);
    std::atomic<int> next{0}, bad{-1};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            bool ok = paths[i] && sizes[i] >= 0 && (sizes[i] == 0 || data[i]);
            if (ok) {
                FILE *f = fopen(paths[i], "wb");
                ok = f != nullptr;
                if (f) {
                    if (sizes[i] > 0) ok = fwrite(data[i], 1, (size_t)sizes[i], f) == (size_t)sizes[i];
                    ok = (fclose(f) == 0) && ok;
                }
            }
            if (!ok) { int expect = -1; bad.compare_exchange_strong(expect, i); }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load() >= 0) return gpcc::fail(GPCC_ERR_ARG, "cannot write %s", paths[bad.load()] ? paths[bad.load()] : "(null path)");
    return GPCC_OK;
}

// The read side: n whole files into one blob (offsets[i] .. offsets[i + 1] = file i), read by native threads.  The 1 002 / 2 343 slice files of a
// million anchors cost the decoders 14 / 27 ms of interpreter time (open + five reads + frombuffer per file under the GIL).  The blob belongs to the
// calling thread and stays valid until its next gpcc_read_files call.
extern "C" int gpcc_read_files(const char *const *paths, int n, int threads, const uint8_t **blob_out, int64_t *offsets_out)
{
    if (n < 0 || !blob_out || !offsets_out || (n > 0 && !paths)) return gpcc::fail(GPCC_ERR_ARG, "bad argument");
    if (threads <= 0) threads = 8;
    threads = threads < n ? threads : (n > 0 ? n : 1);
    std::vector<std::vector<uint8_t>> files((size_t)n);
    std::atomic<int> next{0}, bad{-1};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            bool ok = paths[i] != nullptr;
            if (ok) {
                FILE *f = fopen(paths[i], "rb");
                ok = f != nullptr;
                if (f) {
                    ok = fseek(f, 0, SEEK_END) == 0;
                    const long sz = ok ? ftell(f) : -1;
                    ok = ok && sz >= 0 && fseek(f, 0, SEEK_SET) == 0;
                    if (ok) {
                        files[(size_t)i].resize((size_t)sz);
                        ok = sz == 0 || fread(files[(size_t)i].data(), 1, (size_t)sz, f) == (size_t)sz;
                    }
                    fclose(f);
                }
            }
            if (!ok) { int expect = -1; bad.compare_exchange_strong(expect, i); }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load() >= 0) return gpcc::fail(GPCC_ERR_ARG, "cannot read %s", paths[bad.load()] ? paths[bad.load()] : "(null path)");
    static thread_local std::vector<uint8_t> blob;
    size_t total = 0;
    for (int i = 0; i < n; ++i) { offsets_out[i] = (int64_t)total; total += files[(size_t)i].size(); }
    offsets_out[n] = (int64_t)total;
    blob.resize(total + 16);
    for (int i = 0; i < n; ++i)
        if (!files[(size_t)i].empty()) memcpy(blob.data() + offsets_out[i], files[(size_t)i].data(), files[(size_t)i].size());
    *blob_out = blob.data();
    return GPCC_OK;
}

