// attributes.hip -- the HAC attribute-side kernels around the geometry path:
//   gsac_calculate_cdf   arithmetic.calculate_cdf      arithmetic.zip!arithmetic/arithmetic_kernel.cu:7-54
//   gsac_encode/_decode  arithmetic.arithmetic_encode / arithmetic_decode   ...:94-232, 265-403
//   gsge_forward         _gridencoder.grid_encode_forward   gridencoder.zip!gridencoder/src/gridencoder.cu:46-361
// Same byte format as the reference's CUDA coder (chunks of `chunk_size` symbols, per-chunk byte
// counts), same integerisation of the float CDF rows (rint(cdf * (65536 - (Lp-1))) + index).
//
// Where the reference runs ONE THREAD per chunk (<<<chunks, 1>>>) this version
//   * encodes with the geometry path's lane-per-chunk coder after a fully parallel pre-pass that
//     turns (cdf row, symbol) into the two integers the coder needs;
//   * decodes with one WAVE per chunk: the 64 lanes integerise and scale a whole CDF row at once,
//     a ballot finds the symbol (no binary search, no division), rows are prefetched 4 symbols ahead.
#include "octree.hpp"
#include "primitives.hpp"
#include "rangecoder.hpp"

using namespace gpcc;

namespace {

constexpr int TB = 256;

// ------------------------------------------------------------------ calculate_cdf
// lower[idx][i] of arithmetic.calculate_cdf (arithmetic_kernel.cu:7-28) -- ONE definition, used by the table kernel and by
// the fused coder below, so a stream coded without the table is byte-identical to one coded with it
__device__ __forceinline__ float gaussian_cdf_entry(float mean, float scale, float q, int min_value, int i)
{
    const float sc = (float)fmax((double)scale, 1e-9);                            // max(scale[idx], 1e-9)
    const float sample = (float)(((double)(min_value + i) - 0.5) * (double)q);    // (min + i - 0.5) * Q
    const float arg = -(sample - mean) / (sc * sqrtf(2.0f));
    return (float)(0.5 * (double)erfcf(arg));
}

__global__ __launch_bounds__(TB) void k_gaussian_cdf(const float *__restrict__ mean, const float *__restrict__ scale, const float *__restrict__ Q, int64_t n,
                                                     int min_value, int lp, float *__restrict__ lower)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * lp) return;
    const int64_t idx = t / lp;
    lower[t] = gaussian_cdf_entry(mean[idx], scale[idx], Q[idx], min_value, (int)(t - idx * lp));
}

// Row sources of the coder: a float table row, a uint16 table row (torchac's *_int16_normalized_cdf), or the Gaussian
// parameters of the element -- the table-free path of encoder_gaussian / decoder_gaussian (the reference writes and
// re-reads n x (max - min + 2) floats per slice; here the two entries an encoded symbol needs, or the <= 64 entries a
// decoding wave compares, are evaluated in registers).
struct GaussTable { const float *mean, *scale, *q; int min_value; };
struct GaussRow { float mean, scale, q; int min_value; };
// HAC++'s mixture (HAC-plus/utils/encodings_cuda.py:205-225, 285-299): lower = clamp(sum_i calculate_cdf(mean_i, scale_i, Q) *
// prob_i, 0, 1), the components added in list order in fp32 (a product, then a sum: no fused multiply-add)
constexpr int MIX_MAX = 4;
struct MixTable { const float *mean[MIX_MAX], *scale[MIX_MAX], *prob[MIX_MAX]; const float *q; int k, min_value; };
struct MixRow { float mean[MIX_MAX], scale[MIX_MAX], prob[MIX_MAX]; float q; int k, min_value; };
// one CDF row for every symbol: the Bernoulli coder of the hash tables and masks (encodings_cuda.py: encoder / decoder build an (n, 3) table whose rows
// are all (0, 1 - p, 1) -- 120 MB written and read back for the ten million mask bits of a million anchors)
struct ConstTable { float c[4]; };
struct ConstRow { float c0, c1, c2, c3; };
template <typename CT> struct RowOf { typedef const CT *type; };
template <> struct RowOf<ConstTable> { typedef ConstRow type; };
template <> struct RowOf<GaussTable> { typedef GaussRow type; };
template <> struct RowOf<MixTable> { typedef MixRow type; };
__device__ __forceinline__ const float *row_of(const float *t, int64_t idx, int lp) { return t + idx * lp; }
__device__ __forceinline__ const uint16_t *row_of(const uint16_t *t, int64_t idx, int lp) { return t + idx * lp; }
__device__ __forceinline__ ConstRow row_of(const ConstTable &t, int64_t, int) { return ConstRow{t.c[0], t.c[1], t.c[2], t.c[3]}; }
__device__ __forceinline__ GaussRow row_of(const GaussTable &t, int64_t idx, int) { return GaussRow{t.mean[idx], t.scale[idx], t.q[idx], t.min_value}; }
__device__ __forceinline__ MixRow row_of(const MixTable &t, int64_t idx, int)
{
    MixRow r;
    r.q = t.q[idx]; r.k = t.k; r.min_value = t.min_value;
#pragma unroll
    for (int i = 0; i < MIX_MAX; ++i)
        if (i < t.k) { r.mean[i] = t.mean[i][idx]; r.scale[i] = t.scale[i][idx]; r.prob[i] = t.prob[i][idx]; }
    return r;
}
__device__ __forceinline__ float mix_cdf_entry(const MixRow &row, int m)
{
    float acc = gaussian_cdf_entry(row.mean[0], row.scale[0], row.q, row.min_value, m) * row.prob[0];
#pragma unroll
    for (int i = 1; i < MIX_MAX; ++i)
        if (i < row.k) acc = acc + gaussian_cdf_entry(row.mean[i], row.scale[i], row.q, row.min_value, m) * row.prob[i];
    return fminf(fmaxf(acc, 0.0f), 1.0f);
}

// the mixture's table (tests, and callers that want the reference's two-step form)
__global__ __launch_bounds__(TB) void k_mixture_cdf(MixTable t, int64_t n, int lp, float *__restrict__ lower)
{
    const int64_t q = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (q >= n * lp) return;
    const int64_t idx = q / lp;
    lower[q] = mix_cdf_entry(row_of(t, idx, lp), (int)(q - idx * lp));
}

// ------------------------------------------------------------------ encode pre-pass
// CDF entry -> integer: float rows are integerised on the fly (arithmetic_kernel.cu:124-125 == kit/op.py:67-79),
// uint16 rows (torchac's *_int16_normalized_cdf) are used as they are
__device__ __forceinline__ uint32_t cdf_int(const float *row, int m, float scale) { return (uint32_t)((int)__builtin_rintf(row[m] * scale) + m); }
__device__ __forceinline__ uint32_t cdf_int(const uint16_t *row, int m, float) { return row[m]; }
__device__ __forceinline__ uint32_t cdf_int(const ConstRow &row, int m, float scale)
{
    const float v = m == 0 ? row.c0 : m == 1 ? row.c1 : m == 2 ? row.c2 : row.c3;
    return (uint32_t)((int)__builtin_rintf(v * scale) + m);
}
__device__ __forceinline__ uint32_t cdf_int(const GaussRow &row, int m, float scale)
{
    return (uint32_t)((int)__builtin_rintf(gaussian_cdf_entry(row.mean, row.scale, row.q, row.min_value, m) * scale) + m);
}
__device__ __forceinline__ uint32_t cdf_int(const MixRow &row, int m, float scale) { return (uint32_t)((int)__builtin_rintf(mix_cdf_entry(row, m) * scale) + m); }

template <typename CT>
__global__ __launch_bounds__(TB) void k_hac_pack(const CT cdf, const int16_t *__restrict__ sym, int64_t n, int lp, int chunk, uint32_t nch,
                                                 uint32_t *__restrict__ lohi)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    const float scale = (float)(65536 - (lp - 1));
    const int s = sym[r];
    const auto row = row_of(cdf, r, lp);
    const uint32_t lo = cdf_int(row, s, scale);
    const uint32_t hi = s == lp - 2 ? 0x10000u : cdf_int(row, s + 1, scale);
    const uint32_t c = (uint32_t)(r / chunk), t = (uint32_t)(r - (int64_t)c * chunk);
    lohi[(size_t)t * nch + c] = (lo & 0xFFFFu) | ((hi - 1u) << 16);
}

// ------------------------------------------------------------------ decode: one wave per chunk
// Bit reader of a wave: the chunk's bytes come in 64 words at a time -- lane j loads big-endian word j of the current block, one coalesced load per
// 2 048 bits -- and the coder draws the next word with v_readlane.  (Until round 5 every refill was a load of its own whose latency, ~1 us from HBM
// and never less than an L1 round trip, sat in the symbol chain every 32 bits.)  Bytes past the chunk's end read as zero and are never loaded.
// The state is wave-uniform and kept in scalar registers (readfirstlane): the range arithmetic of the caller runs on the scalar unit.
struct WaveBits {
    const uint8_t *p;     // first byte of the current block of 64 words
    int32_t rem;          // bytes of the chunk from p on
    uint32_t words;       // per lane: word `lane` of the block
    uint32_t widx;        // next word of the block to hand out (64: block exhausted)
    uint32_t nw;
    uint64_t buf;
    uint32_t n;
    __device__ __forceinline__ void load_block()
    {
        const int lane = threadIdx.x & 63;
        const int32_t left = rem - 4 * lane;          // bytes of the chunk at this lane's word
        uint32_t w = 0u;
        if (left >= 4) {
            __builtin_memcpy(&w, p + 4 * lane, 4);
            w = __builtin_bswap32(w);
        } else if (left > 0) {                        // the chunk's last, partial word: byte by byte (nothing behind the chunk is touched)
            for (int b = 0; b < left; ++b) w |= (uint32_t)p[4 * lane + b] << (24 - 8 * b);
        }
        words = w;
        widx = 0u;
    }
    __device__ __forceinline__ uint32_t next_word()
    {
        if (widx == 64u) { p += 256; rem -= 256; load_block(); }
        widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)widx);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)words, (int)widx);
        widx += 1u;
        return w;
    }
    __device__ __forceinline__ void init(const uint8_t *base, uint32_t nbytes) { p = base; rem = (int32_t)nbytes; buf = 0; n = 0; load_block(); nw = next_word(); }
    __device__ __forceinline__ uint32_t take(uint32_t k)
    {
        if (n <= 32) {
            buf |= (uint64_t)nw << (32 - n);
            n += 32;
            nw = next_word();
        }
        const uint32_t r = (uint32_t)((buf >> 1) >> (63 - k));
        buf <<= k;
        n -= k;
        buf = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(buf >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)buf);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
        return r;
    }
};

__device__ __forceinline__ int clz32(uint32_t x) { return x ? __clz((int)x) : 32; }
// value of lane `l` (wave-uniform index) as a scalar: v_readlane_b32 -- the result lives in an SGPR, so what is computed from it (the coder's range
// update) runs on the scalar unit beside the other waves' VALU work; __shfl with a uniform index is a ds_bpermute whose result is a VGPR
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t uniform64(int64_t v) { return (int64_t)(((uint64_t)uniform((uint32_t)((uint64_t)v >> 32)) << 32) | uniform((uint32_t)v)); }
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l)); }

// Row parameters of 64 consecutive symbols at once (round 5): lane j loads row i0 + j -- three coalesced loads per 64 symbols for a GaussTable instead
// of three broadcast loads per symbol whose latency (~1 us from HBM) sat in the decode chain four symbols at a time -- and a row is handed to the
// wave by lane broadcasts.  Table rows (float / uint16) are pointers: nothing to load.
template <typename CT> struct RowLoader {
    __device__ __forceinline__ static auto load(const CT &t, int64_t idx, int lp) { return row_of(t, idx, lp); }
    template <typename R> __device__ __forceinline__ static auto get(const CT &t, const R &, int64_t idx, int lp, int) { return row_of(t, idx, lp); }
    template <typename R> __device__ __forceinline__ static int centre(const R &, int max_symbol) { return max_symbol / 2; }
    template <typename R> __device__ __forceinline__ static int estimate(const R &, float p, int max_symbol) { return (int)(p * (float)max_symbol); }
};
template <> struct RowLoader<GaussTable> {
    __device__ __forceinline__ static GaussRow load(const GaussTable &t, int64_t idx, int lp) { return row_of(t, idx, lp); }
    __device__ __forceinline__ static GaussRow get(const GaussTable &, const GaussRow &mine, int64_t, int, int u)
    {
        return GaussRow{__shfl(mine.mean, u), __shfl(mine.scale, u), __shfl(mine.q, u), mine.min_value};
    }
    // index of the symbol nearest the mean: where a window of 64 candidates is put when the alphabet is wider than a wave
    __device__ __forceinline__ static int centre(const GaussRow &r, int) { return (int)__builtin_rintf(r.mean / r.q) - r.min_value; }
    // index whose CDF entry is about p: the Gaussian's quantile (an ESTIMATE: the window put around it is checked like any other)
    __device__ __forceinline__ static int estimate(const GaussRow &r, float p, int)
    {
        const float z = normcdfinvf(fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f));
        return (int)__builtin_rintf((r.mean + fmaxf(r.scale, 1e-9f) * z) / r.q) - r.min_value;
    }
};
template <> struct RowLoader<MixTable> {
    __device__ __forceinline__ static MixRow load(const MixTable &t, int64_t idx, int lp) { return row_of(t, idx, lp); }
    __device__ __forceinline__ static MixRow get(const MixTable &, const MixRow &mine, int64_t, int, int u)
    {
        MixRow r;
        r.q = __shfl(mine.q, u); r.k = mine.k; r.min_value = mine.min_value;
#pragma unroll
        for (int i = 0; i < MIX_MAX; ++i)
            if (i < mine.k) { r.mean[i] = __shfl(mine.mean[i], u); r.scale[i] = __shfl(mine.scale[i], u); r.prob[i] = __shfl(mine.prob[i], u); }
        return r;
    }
    __device__ __forceinline__ static int centre(const MixRow &r, int)
    {
        int best = 0;
#pragma unroll
        for (int i = 1; i < MIX_MAX; ++i)
            if (i < r.k && r.prob[i] > r.prob[best]) best = i;
        float m = r.mean[0];
#pragma unroll
        for (int i = 1; i < MIX_MAX; ++i) m = best == i ? r.mean[i] : m;
        return (int)__builtin_rintf(m / r.q) - r.min_value;
    }
    __device__ __forceinline__ static int estimate(const MixRow &r, float p, int)
    {   // the quantile of the heaviest component: a starting point
        int best = 0;
#pragma unroll
        for (int i = 1; i < MIX_MAX; ++i)
            if (i < r.k && r.prob[i] > r.prob[best]) best = i;
        float m = r.mean[0], sc = r.scale[0];
#pragma unroll
        for (int i = 1; i < MIX_MAX; ++i) { m = best == i ? r.mean[i] : m; sc = best == i ? r.scale[i] : sc; }
        const float z = normcdfinvf(fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f));
        return (int)__builtin_rintf((m + fmaxf(sc, 1e-9f) * z) / r.q) - r.min_value;
    }
};

// One wave decodes one chunk: `cn` symbols whose rows are base .. base+cn-1 of `cdf`; out(row, symbol) stores the result.
// A symbol is the highest index m with  (span * cdf_int(m)) >> 16  <=  value - low  (index 0 always qualifies; the integerised row is strictly
// increasing: rint(cdf * scale) + m).  The kernel is bound by VALU issue -- one erfc (~100 instructions with its fp64 steps; k of them for HAC++'s
// mixtures) per candidate PASS of the wave, whatever the number of useful lanes -- and the candidates of a row do not depend on the decoder's state.
// So a pass serves FOUR rows: lane (g, c) evaluates candidate s0[g] + c of row g, a window of 16 around the element's mean (the whole row when the
// alphabet has <= 16 symbols); the serial part per symbol is then a scaled compare, a ballot over the row's 16 lanes and the range update.  Round 4
// spent a full 64-candidate pass per symbol (and, for alphabets wider than a wave, a wave-uniform binary search: seven dependent erfc per symbol).
// A window that does not bracket the value falls back to a 64-wide window of the row's own and then to the binary search: the same index either way.
template <typename CT, typename OUT>
__device__ __forceinline__ void hac_decode_chunk(const CT &cdf, const uint8_t *__restrict__ chunk_bytes, uint32_t chunk_nbytes, int64_t base, int cn, int lp,
                                                 OUT out)
{
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const float scale = (float)(65536 - (lp - 1));
    const int max_symbol = lp - 2;            // indices 0 .. lp-2 are searched
    WaveBits in;
    in.init(chunk_bytes, chunk_nbytes);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = in.take(32);
    typedef RowLoader<CT> RL;
    for (int j0 = 0; j0 < cn; j0 += 64) {
        const auto mine = RL::load(cdf, base + min(j0 + lane, cn - 1), lp);
        const int jn = min(64, cn - j0);
        for (int u0 = 0; u0 < jn; u0 += 4) {
            // one pass: 16 candidates of each of the next four rows (rows do not depend on decoded symbols)
            const int ug = min(u0 + g, jn - 1);
            const auto rowg = RL::get(cdf, mine, base + j0 + ug, lp, ug);
            const int s0g = max_symbol > 15 ? max(0, min(RL::centre(rowg, max_symbol) - 7, max_symbol - 15)) : 0;
            const int mg = s0g + c;
            const uint32_t preg = mg <= max_symbol ? cdf_int(rowg, mg, scale) : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u0 + u >= jn) break;
                const int64_t r = base + j0 + u0 + u;
                const uint64_t span = (uint64_t)high - (uint64_t)low + 1u;
                const uint32_t x = value - low;
                const uint32_t t = (uint32_t)((span * (uint64_t)preg) >> 16);
                const uint32_t bal = (uint32_t)(__ballot(g == u && mg <= max_symbol && (mg == 0 || t <= x)) >> (16 * u)) & 0xFFFFu;
                const int top = 31 - clz32(bal);                   // -1: no candidate qualifies
                int s = (int)lane_value((uint32_t)s0g, 16 * u) + top;
                uint32_t lo, hi;
                if (bal != 0u && (top < 15 || s == max_symbol)) {
                    lo = lane_value(t, 16 * u + top);
                    const uint32_t nxt = lane_value(t, 16 * u + min(top + 1, 15));
                    hi = s == max_symbol ? (uint32_t)span : nxt;
                } else {
                    // the narrow window does not bracket the value: a 64-wide window of this row alone, then the wave-uniform binary search
                    const auto row = RL::get(cdf, mine, r, lp, u0 + u);
                    // (centred on the quantile of value - low within the range: wide rows -- HAC's scaling attribute has sigma / Q ~ 50 -- land in it)
                    const int w0 = max_symbol > 63 ? max(0, min(RL::estimate(row, (float)x / (float)span, max_symbol) - 31, max_symbol - 63)) : 0;
                    const int m = w0 + lane;
                    const uint32_t tw = m <= max_symbol ? (uint32_t)((span * (uint64_t)cdf_int(row, m, scale)) >> 16) : 0u;
                    const uint64_t balw = __ballot(m <= max_symbol && (m == 0 || tw <= x));
                    const int topw = 63 - __clzll((long long)balw);
                    s = w0 + topw;
                    if (balw != 0ull && (topw < 63 || s == max_symbol)) {
                        lo = lane_value(tw, topw);
                        const uint32_t nxt = lane_value(tw, min(topw + 1, 63));
                        hi = s == max_symbol ? (uint32_t)span : nxt;
                    } else {
                        int left = 0, right = max_symbol + 1;
                        while (left + 1 < right) {
                            const int mid = (left + right) / 2;
                            const uint32_t v = uniform(cdf_int(row, mid, scale));     // (every lane computed the same number)
                            if ((uint32_t)((span * (uint64_t)v) >> 16) <= x) left = mid; else right = mid;
                        }
                        s = left;
                        lo = (uint32_t)((span * (uint64_t)uniform(cdf_int(row, s, scale))) >> 16);
                        hi = s == max_symbol ? (uint32_t)span : (uint32_t)((span * (uint64_t)uniform(cdf_int(row, s + 1, scale))) >> 16);
                    }
                }
                if (lane == 0) out(r, s);
                high = (low - 1u) + hi;
                low = low + lo;
                const int n1 = clz32(low ^ high);
                low <<= n1; high = (high << n1) | ((1u << n1) - 1u); value = (value << n1) | in.take((uint32_t)n1);
                const int n2 = min(min(clz32(~(low << 1)), clz32(high << 1)), 31);
                low = (low << n2) & (n2 ? 0x7FFFFFFFu : 0xFFFFFFFFu);
                high = (high << n2) | (n2 ? 0x80000000u : 0u) | ((1u << n2) - 1u);
                value = ((value << n2) ^ (n2 ? 0x80000000u : 0u)) | in.take((uint32_t)n2);
                // the coder's state is wave-uniform; saying so keeps it (and the arithmetic above) in scalar registers
                low = uniform(low); high = uniform(high); value = uniform(value);
            }
        }
    }
}

template <typename CT>
__global__ __launch_bounds__(64) void k_hac_decode(const CT cdf, const uint8_t *__restrict__ bytes, const int32_t *__restrict__ cnt,
                                                   const uint32_t *__restrict__ cnt_cum, int16_t *__restrict__ sym, int64_t n, int lp, int chunk)
{
    const int c = blockIdx.x;
    const int64_t base = (int64_t)c * chunk;
    hac_decode_chunk(cdf, bytes + uniform(cnt_cum[c]), uniform((uint32_t)cnt[c]), base, (int)min((int64_t)chunk, n - base), lp,
                     [&](int64_t r, int s) { sym[r] = (int16_t)s; });
}

// The same for MANY slices at once (every 3000-anchor slice of an attribute has its own symbol range [min, max], hence
// its own alphabet): one wave per chunk of any slice, CDF entries from the element's Gaussian parameters, the decoded
// value (sym + min) * Q written directly (encodings_cuda.py:431-432).
struct SliceChunk { int64_t base; int32_t n, slice; uint32_t byte_off, nbytes; };
// CT = GaussTable (HAC) or MixTable (HAC++'s mixture: the feat groups); its min_value is the slice's
template <typename CT>
__global__ __launch_bounds__(64) void k_hac_decode_slices(CT table, const SliceChunk *__restrict__ chunks, const int32_t *__restrict__ smin,
                                                          const int32_t *__restrict__ slp, const uint8_t *__restrict__ bytes, float *__restrict__ x)
{
    // (the chunk record comes in through a vector load; as scalars, the chunk's loop bounds, the bit reader and the coder's range are wave-uniform
    //  for the compiler too and run on the scalar unit)
    const SliceChunk ch = chunks[blockIdx.x];
    const int slice = (int)uniform((uint32_t)ch.slice);
    const int mn = (int)uniform((uint32_t)smin[slice]);
    table.min_value = mn;
    const float *__restrict__ q = table.q;
    hac_decode_chunk(table, bytes + uniform(ch.byte_off), uniform(ch.nbytes), uniform64(ch.base), (int)uniform((uint32_t)ch.n), (int)uniform((uint32_t)slp[slice]),
                     [&](int64_t r, int s) { x[r] = ((float)s + (float)mn) * q[r]; });
}

// ------------------------------------------------------------------ hash-grid forward
__device__ __forceinline__ uint32_t grid_index(int D, uint32_t F, uint32_t hashmap_size, uint32_t res, const uint32_t *pg)
{
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < D && stride <= hashmap_size; ++d) { index += pg[d] * stride; stride *= res; }
    if (stride > hashmap_size) {  // gridencoder.cu:46-60: xor of coordinate * prime
        const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
        uint32_t h = 0;
        for (int d = 0; d < D; ++d) h ^= pg[d] * primes[d];
        index = h;
    }
    return (index % hashmap_size) * F;
}

template <int D, int F>
__global__ __launch_bounds__(TB) void k_grid_forward(const float *__restrict__ inputs, const float *__restrict__ grid, const int *__restrict__ offsets,
                                                     const int *__restrict__ resolutions, float *__restrict__ outputs, uint32_t N, uint32_t Rb,
                                                     const uint8_t *__restrict__ binary_vxl, const int *__restrict__ min_level_id)
{
    const uint32_t b = blockIdx.x * TB + threadIdx.x;
    if (b >= N) return;
    const uint32_t level = min_level_id ? (uint32_t)min_level_id[b] + blockIdx.y : blockIdx.y;
    grid += (size_t)(uint32_t)offsets[level] * F;
    const float *x = inputs + (size_t)b * D;
    float *out = outputs + ((size_t)blockIdx.y * N + b) * F;
    bool oob = false;
    for (int d = 0; d < D; ++d) oob |= (x[d] < 0.0f || x[d] > 1.0f);
    if (oob) { for (int ch = 0; ch < F; ++ch) out[ch] = 0.0f; return; }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t res = (uint32_t)resolutions[level];
    float pos[D];
    uint32_t pg[D];
    for (int d = 0; d < D; ++d) {
        const float t = x[d] * (float)(res - 2);
        pos[d] = t + 0.5f;                           // (float)((double)t + 0.5): identical rounding
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float w_list[1 << D];
    uint32_t idx_list[1 << D];
    bool use[1 << D];
    float wn = 0.0f;
    for (int c = 0; c < (1 << D); ++c) {
        float w = 1.0f;
        uint32_t pl[D];
        for (int d = 0; d < D; ++d) {
            if ((c & (1 << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = min(pg[d] + 1u, res - 1u); }
        }
        bool zero = false;
        for (int d = 0; d < D; ++d) zero |= (pl[d] == 0u || pl[d] == res - 1u);
        bool m = true;
        if (binary_vxl) {  // gridencoder.cu:262-317: any occupied voxel in the corner's footprint
            m = false;
            const float scale_re = (float)(1.0 / ((double)(float)res - 2.0));
            int g0[D], g1[D];
            for (int d = 0; d < D; ++d) {
                const float pn = (float)(((double)(float)pl[d] - 0.5) * (double)scale_re);
                float a = (pn - scale_re) * (float)Rb;
                a = a < 0.0f ? 0.0f : a; a = a > (float)(Rb - 1) ? (float)(Rb - 1) : a;
                g0[d] = (int)a;
                float bb = (pn + scale_re) * (float)Rb;
                bb = bb < 0.0f ? 0.0f : bb; bb = bb > (float)(Rb - 1) ? (float)(Rb - 1) : bb;
                g1[d] = (int)bb;
            }
            if (D == 2) {
                for (int ia = g0[0]; ia <= g1[0] && !m; ++ia)
                    for (int ib = g0[1]; ib <= g1[1] && !m; ++ib) m = binary_vxl[(size_t)ia * Rb + ib] != 0;
            } else if (D == 3) {
                for (int ia = g0[0]; ia <= g1[0] && !m; ++ia)
                    for (int ib = g0[1]; ib <= g1[1] && !m; ++ib)
                        for (int ic = g0[D - 1]; ic <= g1[D - 1] && !m; ++ic) m = binary_vxl[((size_t)ia * Rb + ib) * Rb + ic] != 0;
            } else {
                for (int ia = g0[0]; ia <= g1[0] && !m; ++ia) m = binary_vxl[ia] != 0;
            }
        }
        w_list[c] = w;
        use[c] = !zero && m;
        idx_list[c] = 0;
        if (use[c]) { idx_list[c] = grid_index(D, F, hashmap_size, res, pl); wn += w; }
    }
    if (wn == 0.0f) wn = (float)((double)wn + 1e-9);
    const float wn_re = (float)(1.0 / (double)wn);
    float r[F];
    for (int ch = 0; ch < F; ++ch) r[ch] = 0.0f;
    for (int c = 0; c < (1 << D); ++c)
        if (use[c]) {
            const float ww = w_list[c] * wn_re;
            for (int ch = 0; ch < F; ++ch) r[ch] = __builtin_fmaf(ww, grid[idx_list[c] + ch], r[ch]);  // nvcc contracts mul+add (fmad) here
        }
    for (int ch = 0; ch < F; ++ch) out[ch] = r[ch];
}

template <int D>
int grid_launch_f(hipStream_t st, int F, dim3 g, const float *in, const float *emb, const int *off, const int *res, float *out, uint32_t N, uint32_t Rb,
                  const uint8_t *bv, const int *ml)
{
    switch (F) {
    case 1: k_grid_forward<D, 1><<<g, TB, 0, st>>>(in, emb, off, res, out, N, Rb, bv, ml); break;
    case 2: k_grid_forward<D, 2><<<g, TB, 0, st>>>(in, emb, off, res, out, N, Rb, bv, ml); break;
    case 4: k_grid_forward<D, 4><<<g, TB, 0, st>>>(in, emb, off, res, out, N, Rb, bv, ml); break;
    case 8: k_grid_forward<D, 8><<<g, TB, 0, st>>>(in, emb, off, res, out, N, Rb, bv, ml); break;
    default: return fail(GPCC_ERR_ARG, "GridEncoding: n_features must be 1, 2, 4 or 8");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace

extern "C" int gsac_calculate_cdf(gpcc_ctx *ctx, const float *mean, const float *scale, const float *Q, int64_t n, int min_value, int max_value,
                                  float *lower, void *stream)
{
    if (!ctx || !mean || !scale || !Q || !lower) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0) return GPCC_OK;
    const int lp = max_value - min_value + 2;
    if (lp < 2) return fail(GPCC_ERR_ARG, "max_value < min_value");
    HIP_TRY(hipSetDevice(ctx->device));
    k_gaussian_cdf<<<(unsigned)cdiv(n * lp, TB), TB, 0, (hipStream_t)stream>>>(mean, scale, Q, n, min_value, lp, lower);
    LAUNCH_CHECK();
    return GPCC_OK;
}

template <typename CT>
static int gsac_encode_impl(gpcc_ctx *ctx, const int16_t *sym, CT cdf, int chunk_size, int64_t n, int lp, const uint8_t **bytes_out,
                           int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream, bool keep_arena = false)
{
    if (!ctx || !sym || !bytes_out || !nbytes_out || !cnt_out || !nchunks_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0 || lp < 2) return fail(GPCC_ERR_ARG, "bad size");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    const uint32_t sstride = rc_scratch_stride((uint32_t)std::min<int64_t>(chunk_size, n));
    if (!keep_arena) {
        GP_TRY(ctx->arena.reserve((size_t)nch * chunk_size * 4 + 2 * (size_t)nch * sstride + (size_t)nch * 64 + ((size_t)4 << 20)));
        ctx->arena.reset();
    }
    std::vector<RcChunk> chunks((size_t)nch);
    for (int c = 0; c < nch; ++c) chunks[(size_t)c] = RcChunk{(uint32_t)c, (uint32_t)nch, (uint32_t)std::min<int64_t>(chunk_size, n - (int64_t)c * chunk_size), 0, 0, 0};
    TAKE(lohi, uint32_t, (int64_t)nch * chunk_size); TAKE(dch, RcChunk, nch); TAKE(dcnt, uint32_t, nch + 1); TAKE(doff, uint32_t, nch + 1);
    TAKE(scratch, uint8_t, (size_t)nch * sstride); TAKE(payload, uint8_t, (size_t)nch * sstride);
    HIP_TRY(hipMemcpyAsync(dch, chunks.data(), sizeof(RcChunk) * (size_t)nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    k_hac_pack<CT><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(cdf, sym, n, lp, chunk_size, (uint32_t)nch, lohi);
    LAUNCH_CHECK();
    GP_TRY(rc_encode_launch(st, lohi, dch, nch, scratch, sstride, dcnt));
    GP_TRY(exclusive_scan_u32(ctx, st, dcnt, doff, nch, doff + nch));
    GP_TRY(rc_compact_launch(st, scratch, sstride, dcnt, doff, nullptr, nch, payload));
    GP_TRY(ctx->hstage.reserve(4 * (size_t)nch + 64));
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(ctx->hstage.p);
    HIP_TRY(hipMemcpyAsync(hcnt, dcnt, 4 * (size_t)nch, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hcnt + nch, doff + nch, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    GP_TRY(device_error_check(ctx));
    const size_t total = hcnt[nch];
    GP_TRY(ctx->hbytes.reserve(total + 16));
    if (total) HIP_TRY(hipMemcpyAsync(ctx->hbytes.p, payload, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *bytes_out = ctx->hbytes.p; *nbytes_out = (int64_t)total;
    *cnt_out = reinterpret_cast<const int32_t *>(hcnt); *nchunks_out = nch;
    return GPCC_OK;
}

extern "C" int gsac_encode(gpcc_ctx *ctx, const int16_t *sym, const float *cdf, int chunk_size, int64_t n, int lp, const uint8_t **bytes_out,
                           int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!cdf) return fail(GPCC_ERR_ARG, "null argument");
    return gsac_encode_impl<const float *>(ctx, sym, cdf, chunk_size, n, lp, bytes_out, nbytes_out, cnt_out, nchunks_out, stream);
}

// the same with ONE row for all symbols (lp <= 4 entries, e.g. (0, 1 - p, 1) for a Bernoulli source): byte-identical to gsac_encode on the table that
// repeats the row n times
extern "C" int gsac_encode_const(gpcc_ctx *ctx, const int16_t *sym, const float *row_host, int chunk_size, int64_t n, int lp, const uint8_t **bytes_out,
                                 int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!row_host || lp < 2 || lp > 4) return fail(GPCC_ERR_ARG, "constant-row coder: 2 <= lp <= 4");
    ConstTable t = {};
    for (int i = 0; i < lp; ++i) t.c[i] = row_host[i];
    return gsac_encode_impl<ConstTable>(ctx, sym, t, chunk_size, n, lp, bytes_out, nbytes_out, cnt_out, nchunks_out, stream);
}

extern "C" int gsac_encode_u16(gpcc_ctx *ctx, const int16_t *sym, const uint16_t *cdf, int chunk_size, int64_t n, int lp, const uint8_t **bytes_out,
                               int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!cdf) return fail(GPCC_ERR_ARG, "null argument");
    return gsac_encode_impl<const uint16_t *>(ctx, sym, cdf, chunk_size, n, lp, bytes_out, nbytes_out, cnt_out, nchunks_out, stream);
}

template <typename CT>
static int gsac_decode_impl(gpcc_ctx *ctx, CT cdf, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, int64_t n, int lp,
                           int16_t *sym_out, void *stream, bool keep_arena = false)
{
    if (!ctx || !bytes || !cnt || !sym_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0 || lp < 2) return fail(GPCC_ERR_ARG, "bad size");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    if (nbytes < 0 || nbytes >= ((int64_t)1 << 32)) return fail(GPCC_ERR_FORMAT, "byte stream must be shorter than 4 GiB");
    std::vector<uint32_t> cum((size_t)nch + 1, 0);
    {
        uint64_t run = 0;   // the counts come from a file: summed in 64 bits, checked chunk by chunk (a wrapped 32-bit sum would pass)
        for (int c = 0; c < nch; ++c) {
            if (cnt[c] < 0) return fail(GPCC_ERR_FORMAT, "negative chunk size");
            run += (uint64_t)cnt[c];
            if (run > (uint64_t)nbytes) return fail(GPCC_ERR_FORMAT, "chunk sizes exceed the byte stream");
            cum[(size_t)c + 1] = (uint32_t)run;
        }
    }
    if (!keep_arena) {
        GP_TRY(ctx->arena.reserve((size_t)nbytes + 8 * (size_t)nch + ((size_t)4 << 20)));
        ctx->arena.reset();
    }
    TAKE(db, uint8_t, nbytes + 16); TAKE(dcnt, int32_t, nch); TAKE(dcum, uint32_t, nch + 1);
    HIP_TRY(hipMemcpyAsync(db, bytes, (size_t)nbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dcnt, cnt, 4 * (size_t)nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dcum, cum.data(), 4 * ((size_t)nch + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    k_hac_decode<CT><<<(unsigned)nch, 64, 0, st>>>(cdf, db, dcnt, dcum, sym_out, n, lp, chunk_size);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
}

extern "C" int gsac_decode(gpcc_ctx *ctx, const float *cdf, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, int64_t n, int lp,
                           int16_t *sym_out, void *stream)
{
    if (!cdf) return fail(GPCC_ERR_ARG, "null argument");
    return gsac_decode_impl<const float *>(ctx, cdf, bytes, nbytes, cnt, chunk_size, n, lp, sym_out, stream);
}

extern "C" int gsac_decode_const(gpcc_ctx *ctx, const float *row_host, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, int64_t n, int lp,
                                 int16_t *sym_out, void *stream)
{
    if (!row_host || lp < 2 || lp > 4) return fail(GPCC_ERR_ARG, "constant-row coder: 2 <= lp <= 4");
    ConstTable t = {};
    for (int i = 0; i < lp; ++i) t.c[i] = row_host[i];
    return gsac_decode_impl<ConstTable>(ctx, t, bytes, nbytes, cnt, chunk_size, n, lp, sym_out, stream);
}

extern "C" int gsac_decode_u16(gpcc_ctx *ctx, const uint16_t *cdf, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, int64_t n, int lp,
                               int16_t *sym_out, void *stream)
{
    if (!cdf) return fail(GPCC_ERR_ARG, "null argument");
    return gsac_decode_impl<const uint16_t *>(ctx, cdf, bytes, nbytes, cnt, chunk_size, n, lp, sym_out, stream);
}

// ------------------------------------------------------------------ mlp_grid (a16): Linear - ReLU - Linear
// HAC's context MLP (scene/gaussian_model.py:258-262: Linear(96, 100) - ReLU - Linear(100, 175)) on the hash-grid
// features of a slice of anchors.  Encoder and decoder must obtain bit-identical means / scales / step sizes from it,
// so the arithmetic is specified, as for the geometry heads: acc = bias; for k ascending: acc = fmaf(x[k], W[c][k], acc).
// 16 rows per 256-thread block: the rows and their hidden activations live in LDS, every thread walks the k chain of
// its outputs; the two weight matrices (38 KB + 70 KB) stay in L1/L2.
namespace {
constexpr int MLP_ROWS = 16;
__global__ __launch_bounds__(TB) void k_mlp2(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                                             const float *__restrict__ w2, const float *__restrict__ b2, int64_t n, int din, int dh, int dout,
                                             float slope, float *__restrict__ y)
{
    extern __shared__ float sm[];
    float *xs = sm, *hs = sm + MLP_ROWS * din;
    const int64_t row0 = (int64_t)blockIdx.x * MLP_ROWS;
    const int rows = (int)min((int64_t)MLP_ROWS, n - row0);
    for (int i = threadIdx.x; i < rows * din; i += TB) xs[i] = x[row0 * din + i];
    __syncthreads();
    for (int i = threadIdx.x; i < rows * dh; i += TB) {
        const int r = i / dh, c = i - r * dh;
        const float *w = w1 + (size_t)c * din, *xr = xs + r * din;
        float acc = b1[c];
        for (int k = 0; k < din; ++k) acc = __builtin_fmaf(xr[k], w[k], acc);
        hs[i] = acc > 0.0f ? acc : (slope != 0.0f ? acc * slope : 0.0f);   // ReLU (slope 0) or LeakyReLU(slope): x > 0 ? x : x * slope
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * dout; i += TB) {
        const int r = i / dout, c = i - r * dout;
        const float *w = w2 + (size_t)c * dh, *hr = hs + r * dh;
        float acc = b2[c];
        for (int k = 0; k < dh; ++k) acc = __builtin_fmaf(hr[k], w[k], acc);
        y[(row0 + r) * dout + c] = acc;
    }
}
}  // namespace

namespace {
// The same two layers on the matrix pipe (round 3; k_mlp2 above stays for layer sizes this kernel does not take and as the
// readable statement of the arithmetic).  v_mfma_f32_16x16x4_f32 with the bias as the initial accumulator is the specified
// chain -- acc = b; for k ascending: acc = fmaf(x[k], W[c][k], acc) -- exactly (the heads of the geometry network rely on the
// same fact): MFMA number kk covers k = 4 kk .. 4 kk + 3, lane group g supplying k = 4 kk + g.  16 rows x 16 outputs per
// accumulator tile: 16 anchors are 7 x 24 + 11 x 25 = 443 MFMAs for HAC's 96-100-175 mlp_grid instead of ~55 k scalar
// fmas per row at one lane each (9.5 ms per million anchors at 5.8 TFLOP/s; the matrix pipes need 0.4 ms).
// One persistent workgroup per CU: both weight matrices in LDS ([c][k] at a pitch of K + 2 floats: the 32 lanes of an LDS
// read group hit 32 different banks), every wave takes whole 16-row tiles: the rows staged in LDS, the hidden layer written
// back over them, no block barrier after the weights have landed.
typedef float f32x4m __attribute__((ext_vector_type(4)));
constexpr int MLPM_WAVES_MAX = 8;   // waves per workgroup: 8 (two per SIMD: one's row loads, stores and drains under the other's MFMA chains) when the class's LDS allows, else 4
// DIN / DH / DOUT are the CLASS of the kernel (register arrays and LDS pitches are compile-time); the layer's own sizes din <= DIN,
// dh <= DH, dout <= DOUT are run-time: weights, biases and input columns beyond them are zeros in LDS, so the chain of an output
// is its own k = 0 .. din - 1 steps followed by fmaf(0, 0, acc) steps, which leave acc unchanged.  HAC's 96-100-175 runs in its
// exact class (no padding); HAC++'s mlp_grid (48-100-195 / 225) and its five channel-context MLPs (150 + 10 c - 40 - 30,
// LeakyReLU) in classes <48, 100, 240> and <192, 40, 32> (HAC-plus/scene/gaussian_model.py:117-168, 370-374).
template <int DIN, int DH, int DOUT>
__global__ __launch_bounds__(64 * MLPM_WAVES_MAX) void k_mlp2_mfma(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                                                              const float *__restrict__ w2, const float *__restrict__ b2, int64_t n, int din, int dh, int dout,
                                                              float slope, float *__restrict__ y, int PX)
{
    static_assert(DIN % 4 == 0 && DH % 4 == 0, "whole MFMA k-steps");
    constexpr int NT1 = (DH + 15) / 16, NT2 = (DOUT + 15) / 16, P1 = DIN + 2, P2 = DH + 2;
    // W1 holds DH rows and W2 DOUT rows, not whole tiles of 16: the B operands of the last tile's padding outputs are read from whatever follows
    // (inside the allocation) -- they only reach accumulator columns that are never stored (hidden units >= DH, outputs >= dout)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *W1s = sm, *W2s = W1s + DH * P1, *B1s = W2s + DOUT * P2, *B2s = B1s + NT1 * 16, *XS = B2s + NT2 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, MLPM_WAVES = nthreads >> 6;
    const int e = lane & 15, g = lane >> 4;
    for (int i0 = tid; i0 < DH * DIN; i0 += 4 * nthreads) {     // (four loads in flight per trip, see the tile loads below)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / DIN, k = i - c * DIN; v[u] = (i < DH * DIN && c < dh && k < din) ? w1[(size_t)c * din + k] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / DIN, k = i - c * DIN; if (i < DH * DIN) W1s[c * P1 + k] = v[u]; }
    }
    for (int i0 = tid; i0 < DOUT * DH; i0 += 4 * nthreads) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / DH, k = i - c * DH; v[u] = (i < DOUT * DH && c < dout && k < dh) ? w2[(size_t)c * dh + k] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / DH, k = i - c * DH; if (i < DOUT * DH) W2s[c * P2 + k] = v[u]; }
    }
    for (int i = tid; i < NT1 * 16; i += nthreads) B1s[i] = i < dh ? b1[i] : 0.0f;
    for (int i = tid; i < NT2 * 16; i += nthreads) B2s[i] = i < dout ? b2[i] : 0.0f;
    __syncthreads();
    float *xs = XS + wave * 16 * PX;
    const int64_t ntiles = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * MLPM_WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * MLPM_WAVES) {
        const int64_t row0 = tile * 16;
        // the tile's rows: coalesced float2 loads (rows past n: the last row again), the wave's own LDS slice
        // (loads in batches that are in flight together: as one run-time loop hipcc 7.2 waited for every element before it requested the next --
        //  12 dependent round trips per tile of the 96-column class)
        if (din == DIN) {
            static_assert((16 * DIN / 2) % 64 == 0, "whole trips");
            constexpr int NLD = 16 * DIN / 2 / 64, NB = NLD % 6 == 0 ? 6 : (NLD % 4 == 0 ? 4 : NLD);
            const float *xt = x + (size_t)row0 * DIN;                        // wave-uniform base, 32-bit offsets
            const int last = (int)min((int64_t)15, n - 1 - row0);
#pragma unroll
            for (int b0 = 0; b0 < NLD; b0 += NB) {
                float2 v[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int i = lane + 64 * (b0 + u), r = i / (DIN / 2), c2 = i - r * (DIN / 2);
                    v[u] = *reinterpret_cast<const float2 *>(xt + min(r, last) * DIN + 2 * c2);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int i = lane + 64 * (b0 + u), r = i / (DIN / 2), c2 = i - r * (DIN / 2);
                    *reinterpret_cast<float2 *>(xs + r * PX + 2 * c2) = v[u];
                }
            }
        } else {   // a narrower layer in this class: column by column, zeros beyond din
            const float *xn = x + (size_t)row0 * din;
            const int lastn = (int)min((int64_t)15, n - 1 - row0);
            static_assert((16 * DIN / 64) % 4 == 0, "whole batches");
            for (int i0 = lane; i0 < 16 * DIN; i0 += 4 * 64) {      // unconditional loads (column clamped), four in flight
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u, r = i / DIN, c = i - r * DIN; v[u] = xn[min(r, lastn) * din + min(c, din - 1)]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u, r = i / DIN, c = i - r * DIN; xs[r * PX + c] = c < din ? v[u] : 0.0f; }
            }
        }
        float a[DIN / 4 > DH / 4 ? DIN / 4 : DH / 4];
#pragma unroll
        for (int kk = 0; kk < DIN / 4; ++kk) a[kk] = xs[e * PX + 4 * kk + g];        // A operand: row e, k = 4 kk + g
        // The weight operands of output tile t + 1 are read from LDS while the MFMAs of tile t run (two register sets, the scheduler held to that order):
        // left to itself the compiler placed every ds_read directly in front of the two MFMAs that use it, with one register pair for all of them --
        // a full LDS round trip (~110 cycles) per 64 cycles of matrix work, which is what "32 % of the fp32 matrix peak" was (round 3's figure).
        f32x4m hid[NT1];
        float wb[2][DIN / 4 > DH / 4 ? DIN / 4 : DH / 4];
        {
            const float *wr = W1s + e * P1 + g;                                    // B operand: output 16 t + e, k = 4 kk + g
#pragma unroll
            for (int kk = 0; kk < DIN / 4; ++kk) wb[0][kk] = wr[4 * kk];
        }
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float bias = B1s[16 * t + e];
            f32x4m acc = {bias, bias, bias, bias};
            if (t + 1 < NT1) {
                const float *wr = W1s + (16 * (t + 1) + e) * P1 + g;
#pragma unroll
                for (int kk = 0; kk < DIN / 4; ++kk) wb[(t + 1) & 1][kk] = wr[4 * kk];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < DIN / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wb[t & 1][kk], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            hid[t] = acc;
        }
        // hidden = relu(...) over the rows' slots: lane (g, e) holds rows 4 g .. 4 g + 3 of output 16 t + e (LDS operations of a
        // wave execute in program order: the A reads above are done)
#pragma unroll
        for (int t = 0; t < NT1; ++t)
            if (16 * t + e < DH) {                 // the padding outputs of the last tile have no slot (and no reader)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float h = hid[t][i]; xs[(4 * g + i) * PX + 16 * t + e] = h > 0.0f ? h : (slope != 0.0f ? h * slope : 0.0f); }
            }
#pragma unroll
        for (int kk = 0; kk < DH / 4; ++kk) a[kk] = xs[e * PX + 4 * kk + g];
        {
            const float *wr = W2s + e * P2 + g;
#pragma unroll
            for (int kk = 0; kk < DH / 4; ++kk) wb[0][kk] = wr[4 * kk];
        }
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            const float bias = B2s[16 * t + e];
            f32x4m acc = {bias, bias, bias, bias};
            if (t + 1 < NT2) {
                const float *wr = W2s + (16 * (t + 1) + e) * P2 + g;
#pragma unroll
                for (int kk = 0; kk < DH / 4; ++kk) wb[(t + 1) & 1][kk] = wr[4 * kk];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < DH / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wb[t & 1][kk], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int c = 16 * t + e;
            if (c < dout) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (row0 + 4 * g + i < n) y[(size_t)(row0 + 4 * g + i) * dout + c] = acc[i];
            }
        }
    }
}
template <int DIN, int DH, int DOUT>
static size_t mlpm_lds_bytes(int waves, int px)
{
    constexpr int NT1 = (DH + 15) / 16, NT2 = (DOUT + 15) / 16, P1 = DIN + 2, P2 = DH + 2;
    return sizeof(float) * ((size_t)DH * P1 + (size_t)DOUT * P2 + NT1 * 16 + NT2 * 16 + (size_t)waves * 16 * px);
}
}  // namespace

template <int DIN, int DH, int DOUT>
static int mlp2_mfma_launch(gpcc_ctx *ctx, const float *x, const float *w1, const float *b1, const float *w2, const float *b2, int64_t n, int din, int dh, int dout,
                            float slope, float *y, hipStream_t st)
{
    static PerDeviceOnce attr;
    // eight waves when they fit (if need be with the rows' LDS pitch without its two padding words: two-way conflicts on the 49 A-operand reads of a
    // tile, nothing on the 443 B-operand reads), else four
    constexpr int PXW = (DIN > DH ? DIN : DH);
    constexpr size_t LDS_MAX = 160 * 1024;
    int waves = 8, px = PXW + 2;
    if (mlpm_lds_bytes<DIN, DH, DOUT>(8, px) > LDS_MAX) px = PXW;
    if (mlpm_lds_bytes<DIN, DH, DOUT>(8, px) > LDS_MAX) { waves = 4; px = PXW + 2; }
    const size_t lds = mlpm_lds_bytes<DIN, DH, DOUT>(waves, px);
    GP_TRY(attr.run(ctx->device, [&]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp2_mfma<DIN, DH, DOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        return GPCC_OK;
    }));
    const unsigned grid = (unsigned)std::min<int64_t>(256, cdiv(cdiv(n, 16), waves));
    k_mlp2_mfma<DIN, DH, DOUT><<<grid, 64 * waves, lds, st>>>(x, w1, b1, w2, b2, n, din, dh, dout, slope, y, px);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// act: 0 = ReLU, 1 = LeakyReLU(slope) between the two layers
extern "C" int gshac_mlp2_act(gpcc_ctx *ctx, const float *x, const float *w1, const float *b1, const float *w2, const float *b2, int64_t n, int din, int dh,
                              int dout, int act, float slope, float *y, void *stream)
{
    if (!ctx || !x || !w1 || !b1 || !w2 || !b2 || !y) return fail(GPCC_ERR_ARG, "null argument");
    if (act != 0 && act != 1) return fail(GPCC_ERR_ARG, "mlp2: activation must be 0 (ReLU) or 1 (LeakyReLU)");
    if (n <= 0) return GPCC_OK;
    if (din <= 0 || dh <= 0 || dout <= 0 || (size_t)MLP_ROWS * (size_t)(din + dh) * 4 > 64 * 1024) return fail(GPCC_ERR_ARG, "mlp2: unsupported layer sizes");
    HIP_TRY(hipSetDevice(ctx->device));
    const float sl = act == 1 ? slope : 0.0f;
    hipStream_t st = (hipStream_t)stream;
    static const bool use_mfma = dev_env_int("GAUSPCC_MLP2_MFMA", 1) != 0;
    if (use_mfma) {
        // the smallest class that holds the layer (HAC's mlp_grid in its exact class)
        if (din == 96 && dh == 100 && dout == 175) return mlp2_mfma_launch<96, 100, 175>(ctx, x, w1, b1, w2, b2, n, din, dh, dout, sl, y, st);
        if (din <= 192 && dh <= 40 && dout <= 32) return mlp2_mfma_launch<192, 40, 32>(ctx, x, w1, b1, w2, b2, n, din, dh, dout, sl, y, st);
        if (din <= 48 && dh <= 100 && dout <= 240) return mlp2_mfma_launch<48, 100, 240>(ctx, x, w1, b1, w2, b2, n, din, dh, dout, sl, y, st);
    }
    k_mlp2<<<(unsigned)cdiv(n, MLP_ROWS), TB, (size_t)MLP_ROWS * (size_t)(din + dh) * 4, st>>>(x, w1, b1, w2, b2, n, din, dh, dout, sl, y);
    LAUNCH_CHECK();
    return GPCC_OK;
}

extern "C" int gshac_mlp2(gpcc_ctx *ctx, const float *x, const float *w1, const float *b1, const float *w2, const float *b2, int64_t n, int din, int dh,
                          int dout, float *y, void *stream)
{
    return gshac_mlp2_act(ctx, x, w1, b1, w2, b2, n, din, dh, dout, 0, 0.0f, y, stream);
}


// ------------------------------------------------------------------ fused Gaussian coder (no CDF table)
namespace {
// x_int = round(x / Q) (torch.round: half to even), its min / max over the slice (encodings_cuda.py:343-345)
__global__ __launch_bounds__(TB) void k_quantise_minmax(const float *__restrict__ x, const float *__restrict__ q, int64_t n, int32_t *__restrict__ xi, int32_t *__restrict__ mm)
{
    int mn = INT32_MAX, mx = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int v = (int)__builtin_rintf(x[i] / q[i]);
        xi[i] = v;
        mn = min(mn, v); mx = max(mx, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mn = min(mn, __shfl_xor(mn, d, 64)); mx = max(mx, __shfl_xor(mx, d, 64)); }
    __shared__ int red[TB / 64][2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = mn; red[threadIdx.x >> 6][1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TB / 64; ++w) { mn = min(mn, red[w][0]); mx = max(mx, red[w][1]); }
        atomicMin(&mm[0], mn); atomicMax(&mm[1], mx);
    }
}
__global__ __launch_bounds__(TB) void k_to_symbols(const int32_t *__restrict__ xi, int64_t n, int min_value, int16_t *__restrict__ sym)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) sym[i] = (int16_t)(xi[i] - min_value);
}
// x = (sym + min) * Q (encodings_cuda.py:431-432)
__global__ __launch_bounds__(TB) void k_from_symbols(const int16_t *__restrict__ sym, const float *__restrict__ q, int64_t n, float min_value, float *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) x[i] = ((float)sym[i] + min_value) * q[i];
}
}  // namespace

extern "C" int gsac_encode_gaussian(gpcc_ctx *ctx, const float *x, const float *mean, const float *scale, const float *Q, int64_t n, int chunk_size,
                                    float *min_out, float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out,
                                    int64_t *nchunks_out, void *stream)
{
    if (!ctx || !x || !mean || !scale || !Q || !min_out || !max_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0) return fail(GPCC_ERR_ARG, "bad size");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    const uint32_t sstride = rc_scratch_stride((uint32_t)std::min<int64_t>(chunk_size, n));
    GP_TRY(ctx->arena.reserve((size_t)n * 8 + (size_t)nch * chunk_size * 4 + 2 * (size_t)nch * sstride + (size_t)nch * 64 + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(xi, int32_t, n); TAKE(sym, int16_t, n); TAKE(mm, int32_t, 2);
    const int32_t init[2] = {INT32_MAX, INT32_MIN};
    HIP_TRY(hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, st));
    k_quantise_minmax<<<(unsigned)std::min<int64_t>(cdiv(n, TB), 512), TB, 0, st>>>(x, Q, n, xi, mm);
    LAUNCH_CHECK();
    int32_t hmm[2];
    HIP_TRY(hipMemcpyAsync(hmm, mm, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int lp = hmm[1] - hmm[0] + 2;
    if (lp > 32767) return fail(GPCC_ERR_RANGE, "quantised values span %d levels (int16 symbols)", lp - 1);
    k_to_symbols<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(xi, n, hmm[0], sym);
    LAUNCH_CHECK();
    *min_out = (float)hmm[0]; *max_out = (float)hmm[1];
    return gsac_encode_impl<GaussTable>(ctx, sym, GaussTable{mean, scale, Q, hmm[0]}, chunk_size, n, lp, bytes_out, nbytes_out, cnt_out, nchunks_out, stream, true);
}

static int mix_table(const float *const *mean, const float *const *scale, const float *const *prob, int k, const float *Q, int min_value, MixTable *t)
{
    if (!mean || !scale || !prob || !Q) return fail(GPCC_ERR_ARG, "null argument");
    if (k < 1 || k > MIX_MAX) return fail(GPCC_ERR_ARG, "a mixture has 1..%d components", MIX_MAX);
    *t = MixTable{};
    for (int i = 0; i < k; ++i) {
        if (!mean[i] || !scale[i] || !prob[i]) return fail(GPCC_ERR_ARG, "null argument");
        t->mean[i] = mean[i]; t->scale[i] = scale[i]; t->prob[i] = prob[i];
    }
    t->q = Q; t->k = k; t->min_value = min_value;
    return GPCC_OK;
}

// HAC++'s encoder_gaussian_mixed without the (n, max - min + 2) table (HAC-plus/utils/encodings_cuda.py:205-247)
extern "C" int gsac_encode_gaussian_mixed(gpcc_ctx *ctx, const float *x, const float *const *mean, const float *const *scale, const float *const *prob, int k,
                                          const float *Q, int64_t n, int chunk_size, float *min_out, float *max_out, const uint8_t **bytes_out,
                                          int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!ctx || !x || !min_out || !max_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0) return fail(GPCC_ERR_ARG, "bad size");
    MixTable t;
    GP_TRY(mix_table(mean, scale, prob, k, Q, 0, &t));
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    const uint32_t sstride = rc_scratch_stride((uint32_t)std::min<int64_t>(chunk_size, n));
    GP_TRY(ctx->arena.reserve((size_t)n * 8 + (size_t)nch * chunk_size * 4 + 2 * (size_t)nch * sstride + (size_t)nch * 64 + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(xi, int32_t, n); TAKE(sym, int16_t, n); TAKE(mm, int32_t, 2);
    const int32_t init[2] = {INT32_MAX, INT32_MIN};
    HIP_TRY(hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, st));
    k_quantise_minmax<<<(unsigned)std::min<int64_t>(cdiv(n, TB), 512), TB, 0, st>>>(x, Q, n, xi, mm);
    LAUNCH_CHECK();
    int32_t hmm[2];
    HIP_TRY(hipMemcpyAsync(hmm, mm, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int lp = hmm[1] - hmm[0] + 2;
    // (the reference clamps the symbol to 32767, :227-228, and would then need a table of more than 32767 columns per element)
    if (lp > 32767) return fail(GPCC_ERR_RANGE, "quantised values span %d levels (int16 symbols)", lp - 1);
    k_to_symbols<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(xi, n, hmm[0], sym);
    LAUNCH_CHECK();
    *min_out = (float)hmm[0]; *max_out = (float)hmm[1];
    t.min_value = hmm[0];
    return gsac_encode_impl<MixTable>(ctx, sym, t, chunk_size, n, lp, bytes_out, nbytes_out, cnt_out, nchunks_out, stream, true);
}

// ... and decoder_gaussian_mixed (:271-317)
extern "C" int gsac_decode_gaussian_mixed(gpcc_ctx *ctx, const float *const *mean, const float *const *scale, const float *const *prob, int k, const float *Q,
                                          int64_t n, float min_value, float max_value, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size,
                                          float *x_out, void *stream)
{
    if (!ctx || !x_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0) return fail(GPCC_ERR_ARG, "bad size");
    if (!(min_value >= -1.0e9f && min_value <= 1.0e9f) || !(max_value >= -1.0e9f && max_value <= 1.0e9f))
        return fail(GPCC_ERR_FORMAT, "bad symbol range [%g, %g]", (double)min_value, (double)max_value);
    const int mn = (int)min_value, mx = (int)max_value;
    const int64_t lp64 = (int64_t)mx - mn + 2;
    if (lp64 < 2 || lp64 > 32767) return fail(GPCC_ERR_FORMAT, "bad symbol range [%d, %d]", mn, mx);
    MixTable t;
    GP_TRY(mix_table(mean, scale, prob, k, Q, mn, &t));
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    GP_TRY(ctx->arena.reserve((size_t)n * 2 + (size_t)nbytes + 8 * (size_t)nch + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(sym, int16_t, n);
    GP_TRY(gsac_decode_impl<MixTable>(ctx, t, bytes, nbytes, cnt, chunk_size, n, (int)lp64, sym, stream, true));
    k_from_symbols<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(sym, Q, n, min_value, x_out);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
}

// the mixture's CDF table itself: lower (n, max - min + 2), as HAC++ builds it before arithmetic_encode (:210-225)
extern "C" int gsac_calculate_cdf_mixed(gpcc_ctx *ctx, const float *const *mean, const float *const *scale, const float *const *prob, int k, const float *Q,
                                        int64_t n, int min_value, int max_value, float *lower, void *stream)
{
    if (!ctx || !lower) return fail(GPCC_ERR_ARG, "null argument");
    const int64_t lp = (int64_t)max_value - min_value + 2;
    if (n <= 0 || lp < 2 || lp > 32767) return fail(GPCC_ERR_ARG, "bad size");
    MixTable t;
    GP_TRY(mix_table(mean, scale, prob, k, Q, min_value, &t));
    HIP_TRY(hipSetDevice(ctx->device));
    k_mixture_cdf<<<(unsigned)cdiv(n * lp, TB), TB, 0, (hipStream_t)stream>>>(t, n, (int)lp, lower);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GPCC_OK;
}

extern "C" int gsac_decode_gaussian(gpcc_ctx *ctx, const float *mean, const float *scale, const float *Q, int64_t n, float min_value, float max_value,
                                    const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, float *x_out, void *stream)
{
    if (!ctx || !mean || !scale || !Q || !x_out) return fail(GPCC_ERR_ARG, "null argument");
    if (n <= 0 || chunk_size <= 0) return fail(GPCC_ERR_ARG, "bad size");
    // min / max are floats read from a `.b` file: NaN, infinities and values that do not fit an int are format errors
    if (!(min_value >= -1.0e9f && min_value <= 1.0e9f) || !(max_value >= -1.0e9f && max_value <= 1.0e9f))
        return fail(GPCC_ERR_FORMAT, "bad symbol range [%g, %g]", (double)min_value, (double)max_value);
    const int mn = (int)min_value, mx = (int)max_value;
    const int64_t lp64 = (int64_t)mx - mn + 2;
    if (lp64 < 2 || lp64 > 32767) return fail(GPCC_ERR_FORMAT, "bad symbol range [%d, %d]", mn, mx);
    const int lp = (int)lp64;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv(n, chunk_size);
    GP_TRY(ctx->arena.reserve((size_t)n * 2 + (size_t)nbytes + 8 * (size_t)nch + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(sym, int16_t, n);
    GP_TRY(gsac_decode_impl<GaussTable>(ctx, GaussTable{mean, scale, Q, mn}, bytes, nbytes, cnt, chunk_size, n, lp, sym, stream, true));
    k_from_symbols<<<(unsigned)cdiv(n, TB), TB, 0, st>>>(sym, Q, n, min_value, x_out);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
}

// ------------------------------------------------------------------ all slices of an attribute in one call
namespace {
constexpr int SLICE_PARTS = 8;   // blocks per slice in the min / max pass

__device__ __forceinline__ int slice_of(const int64_t *__restrict__ start, int nslices, int64_t r)
{
    int lo = 0, hi = nslices;           // start[lo] <= r < start[hi]
    while (lo + 1 < hi) { const int m = (lo + hi) >> 1; if (start[m] <= r) lo = m; else hi = m; }
    return lo;
}

// x_int = round(x / Q) and the min / max of every slice: block (part, slice)
__global__ __launch_bounds__(TB) void k_quantise_minmax_slices(const float *__restrict__ x, const float *__restrict__ q, const int64_t *__restrict__ start,
                                                               int32_t *__restrict__ xi, int32_t *__restrict__ mm)
{
    const int s = blockIdx.y;
    const int64_t a = start[s], b = start[s + 1], len = b - a;
    const int64_t lo = a + len * blockIdx.x / SLICE_PARTS, hi = a + len * (blockIdx.x + 1) / SLICE_PARTS;
    int mn = INT32_MAX, mx = INT32_MIN;
    for (int64_t i = lo + threadIdx.x; i < hi; i += TB) {
        const int v = (int)__builtin_rintf(x[i] / q[i]);
        xi[i] = v;
        mn = min(mn, v); mx = max(mx, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mn = min(mn, __shfl_xor(mn, d, 64)); mx = max(mx, __shfl_xor(mx, d, 64)); }
    __shared__ int red[TB / 64][2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = mn; red[threadIdx.x >> 6][1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TB / 64; ++w) { mn = min(mn, red[w][0]); mx = max(mx, red[w][1]); }
        if (hi > lo) { atomicMin(&mm[2 * s], mn); atomicMax(&mm[2 * s + 1], mx); }
    }
}

// (symbol, Gaussian parameters) -> the coder's two integers, in the chunk-interleaved layout of the element's slice
template <typename CT>
__global__ __launch_bounds__(TB) void k_hac_pack_slices(CT table, const int32_t *__restrict__ xi, int64_t n, const int64_t *__restrict__ start, int nslices,
                                                        const int32_t *__restrict__ mm, const int64_t *__restrict__ lohi_base,
                                                        const int32_t *__restrict__ slice_nch, int chunk, uint32_t *__restrict__ lohi)
{
    const int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= n) return;
    const int sl = slice_of(start, nslices, r);
    const int mn = mm[2 * sl], lp = mm[2 * sl + 1] - mn + 2;
    const float sc = (float)(65536 - (lp - 1));
    const int s = xi[r] - mn;
    table.min_value = mn;
    const auto row = row_of(table, r, lp);
    const uint32_t lo = cdf_int(row, s, sc);
    const uint32_t hi = s == lp - 2 ? 0x10000u : cdf_int(row, s + 1, sc);
    const int64_t e = r - start[sl];
    const int64_t c = e / chunk, t = e - c * chunk;
    lohi[lohi_base[sl] + t * slice_nch[sl] + c] = (lo & 0xFFFFu) | ((hi - 1u) << 16);
}
}  // namespace

template <typename CT>
static int encode_slices_impl(gpcc_ctx *ctx, const float *x, CT table, const float *Q, const int64_t *slice_start,
                              int nslices, int chunk_size, float *min_out, float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out,
                              const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!ctx || !x || !Q || !slice_start || !min_out || !max_out || !bytes_out || !nbytes_out || !cnt_out || !nchunks_out)
        return fail(GPCC_ERR_ARG, "null argument");
    if (nslices <= 0 || chunk_size <= 0) return fail(GPCC_ERR_ARG, "bad size");
    for (int s = 0; s < nslices; ++s)
        if (slice_start[s + 1] <= slice_start[s]) return fail(GPCC_ERR_ARG, "slice %d is empty", s);
    if (slice_start[0] != 0) return fail(GPCC_ERR_ARG, "slices must start at element 0");
    const int64_t n = slice_start[nslices];
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // chunk descriptors: slice by slice, the chunks of a slice interleaved among themselves
    std::vector<RcChunk> chunks;
    std::vector<int64_t> lbase((size_t)nslices);
    std::vector<int32_t> snch((size_t)nslices);
    int64_t slots = 0;
    uint32_t max_syms = 1;
    for (int s = 0; s < nslices; ++s) {
        const int64_t len = slice_start[s + 1] - slice_start[s];
        const int nch = (int)cdiv(len, chunk_size);
        lbase[(size_t)s] = slots; snch[(size_t)s] = nch;
        for (int c = 0; c < nch; ++c) {
            const uint32_t cn = (uint32_t)std::min<int64_t>(chunk_size, len - (int64_t)c * chunk_size);
            chunks.push_back(RcChunk{(uint32_t)(slots + c), (uint32_t)nch, cn, 0, 0, 0});
            max_syms = std::max(max_syms, cn);
        }
        slots += (int64_t)nch * chunk_size;
    }
    if (slots >= ((int64_t)1 << 32)) return fail(GPCC_ERR_ARG, "too many symbols in one call");
    const int nch = (int)chunks.size();
    const uint32_t sstride = rc_scratch_stride(max_syms);
    GP_TRY(ctx->arena.reserve((size_t)n * 4 + (size_t)slots * 4 + 2 * (size_t)nch * sstride + (size_t)nch * 64 + (size_t)nslices * 32 + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(xi, int32_t, n); TAKE(mm, int32_t, 2 * nslices); TAKE(dstart, int64_t, nslices + 1); TAKE(dlbase, int64_t, nslices); TAKE(dsnch, int32_t, nslices);
    TAKE(lohi, uint32_t, slots); TAKE(dch, RcChunk, nch); TAKE(dcnt, uint32_t, nch + 1); TAKE(doff, uint32_t, nch + 1);
    TAKE(scratch, uint8_t, (size_t)nch * sstride); TAKE(payload, uint8_t, (size_t)nch * sstride);
    std::vector<int32_t> init((size_t)2 * nslices);
    for (int s = 0; s < nslices; ++s) { init[(size_t)2 * s] = INT32_MAX; init[(size_t)2 * s + 1] = INT32_MIN; }
    HIP_TRY(hipMemcpyAsync(mm, init.data(), 8 * (size_t)nslices, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dstart, slice_start, 8 * ((size_t)nslices + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dlbase, lbase.data(), 8 * (size_t)nslices, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dsnch, snch.data(), 4 * (size_t)nslices, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dch, chunks.data(), sizeof(RcChunk) * (size_t)nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors above go out of scope before the stream drains otherwise
    k_quantise_minmax_slices<<<dim3(SLICE_PARTS, (unsigned)nslices), TB, 0, st>>>(x, Q, dstart, xi, mm);
    LAUNCH_CHECK();
    k_hac_pack_slices<CT><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(table, xi, n, dstart, nslices, mm, dlbase, dsnch, chunk_size, lohi);
    LAUNCH_CHECK();
    GP_TRY(rc_encode_launch(st, lohi, dch, nch, scratch, sstride, dcnt));
    GP_TRY(exclusive_scan_u32(ctx, st, dcnt, doff, nch, doff + nch));
    GP_TRY(rc_compact_launch(st, scratch, sstride, dcnt, doff, nullptr, nch, payload));
    GP_TRY(ctx->hstage.reserve(4 * (size_t)nch + 8 * (size_t)nslices + 64));
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(ctx->hstage.p);
    int32_t *hmm = reinterpret_cast<int32_t *>(hcnt + nch + 1);
    HIP_TRY(hipMemcpyAsync(hcnt, dcnt, 4 * (size_t)nch, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hcnt + nch, doff + nch, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hmm, mm, 8 * (size_t)nslices, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    GP_TRY(device_error_check(ctx));
    for (int s = 0; s < nslices; ++s) {
        if (hmm[2 * s + 1] - hmm[2 * s] + 2 > 32767) return fail(GPCC_ERR_RANGE, "slice %d: quantised values span %d levels (int16 symbols)", s, hmm[2 * s + 1] - hmm[2 * s] + 1);
        min_out[s] = (float)hmm[2 * s]; max_out[s] = (float)hmm[2 * s + 1];
    }
    const size_t total = hcnt[nch];
    GP_TRY(ctx->hbytes.reserve(total + 16));
    if (total) HIP_TRY(hipMemcpyAsync(ctx->hbytes.p, payload, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *bytes_out = ctx->hbytes.p; *nbytes_out = (int64_t)total;
    *cnt_out = reinterpret_cast<const int32_t *>(hcnt); *nchunks_out = nch;
    return GPCC_OK;
}

extern "C" int gsac_encode_gaussian_slices(gpcc_ctx *ctx, const float *x, const float *mean, const float *scale, const float *Q, const int64_t *slice_start,
                                           int nslices, int chunk_size, float *min_out, float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out,
                                           const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    if (!mean || !scale) return fail(GPCC_ERR_ARG, "null argument");
    return encode_slices_impl(ctx, x, GaussTable{mean, scale, Q, 0}, Q, slice_start, nslices, chunk_size, min_out, max_out, bytes_out, nbytes_out, cnt_out, nchunks_out, stream);
}

static int mix_table(const float *const *mean, const float *const *scale, const float *const *prob, int k, const float *Q, int min_value, MixTable *t);

// HAC++: the slices of ONE channel group of `feat` under the two-component mixture (HAC-plus/scene/gaussian_model.py:1306-1321)
extern "C" int gsac_encode_gaussian_mixed_slices(gpcc_ctx *ctx, const float *x, const float *const *mean, const float *const *scale, const float *const *prob, int k,
                                                 const float *Q, const int64_t *slice_start, int nslices, int chunk_size, float *min_out, float *max_out,
                                                 const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream)
{
    MixTable t;
    GP_TRY(mix_table(mean, scale, prob, k, Q, 0, &t));
    return encode_slices_impl(ctx, x, t, Q, slice_start, nslices, chunk_size, min_out, max_out, bytes_out, nbytes_out, cnt_out, nchunks_out, stream);
}

template <typename CT>
static int decode_slices_impl(gpcc_ctx *ctx, CT table, const float *Q, const int64_t *slice_start, int nslices,
                              const float *min_value, const float *max_value, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt,
                              int chunk_size, float *x_out, void *stream)
{
    if (!ctx || !Q || !slice_start || !min_value || !max_value || !bytes || !cnt || !x_out) return fail(GPCC_ERR_ARG, "null argument");
    if (nslices <= 0 || chunk_size <= 0 || slice_start[0] != 0) return fail(GPCC_ERR_ARG, "bad size");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<SliceChunk> chunks;
    std::vector<int32_t> smin((size_t)nslices), slp((size_t)nslices);
    uint64_t off = 0;
    for (int s = 0; s < nslices; ++s) {
        const int64_t len = slice_start[s + 1] - slice_start[s];
        if (len <= 0) return fail(GPCC_ERR_ARG, "slice %d is empty", s);
        if (!(min_value[s] >= -1.0e9f && min_value[s] <= 1.0e9f) || !(max_value[s] >= -1.0e9f && max_value[s] <= 1.0e9f))
            return fail(GPCC_ERR_FORMAT, "slice %d: bad symbol range", s);
        const int mn = (int)min_value[s];
        const int64_t lp64 = (int64_t)(int)max_value[s] - mn + 2;
        if (lp64 < 2 || lp64 > 32767) return fail(GPCC_ERR_FORMAT, "slice %d: bad symbol range", s);
        const int lp = (int)lp64;
        smin[(size_t)s] = mn; slp[(size_t)s] = lp;
        const int nch = (int)cdiv(len, chunk_size);
        for (int c = 0; c < nch; ++c) {
            const int32_t cb = cnt[chunks.size()];
            if (cb < 0) return fail(GPCC_ERR_FORMAT, "negative chunk size");
            chunks.push_back(SliceChunk{slice_start[s] + (int64_t)c * chunk_size, (int32_t)std::min<int64_t>(chunk_size, len - (int64_t)c * chunk_size), s,
                                        (uint32_t)off, (uint32_t)cb});
            off += (uint32_t)cb;
            if ((int64_t)off > nbytes || off >= ((uint64_t)1 << 32)) return fail(GPCC_ERR_FORMAT, "chunk sizes exceed the byte stream");
        }
    }
    const size_t nch = chunks.size();
    GP_TRY(ctx->arena.reserve((size_t)nbytes + sizeof(SliceChunk) * nch + 8 * (size_t)nslices + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(db, uint8_t, nbytes + 16); TAKE(dch, SliceChunk, nch); TAKE(dmin, int32_t, nslices); TAKE(dlp, int32_t, nslices);
    HIP_TRY(hipMemcpyAsync(db, bytes, (size_t)nbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dch, chunks.data(), sizeof(SliceChunk) * nch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dmin, smin.data(), 4 * (size_t)nslices, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dlp, slp.data(), 4 * (size_t)nslices, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    k_hac_decode_slices<CT><<<(unsigned)nch, 64, 0, st>>>(table, dch, dmin, dlp, db, x_out);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
}

extern "C" int gsac_decode_gaussian_slices(gpcc_ctx *ctx, const float *mean, const float *scale, const float *Q, const int64_t *slice_start, int nslices,
                                           const float *min_value, const float *max_value, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt,
                                           int chunk_size, float *x_out, void *stream)
{
    if (!mean || !scale) return fail(GPCC_ERR_ARG, "null argument");
    return decode_slices_impl(ctx, GaussTable{mean, scale, Q, 0}, Q, slice_start, nslices, min_value, max_value, bytes, nbytes, cnt, chunk_size, x_out, stream);
}

extern "C" int gsac_decode_gaussian_mixed_slices(gpcc_ctx *ctx, const float *const *mean, const float *const *scale, const float *const *prob, int k, const float *Q,
                                                 const int64_t *slice_start, int nslices, const float *min_value, const float *max_value, const uint8_t *bytes,
                                                 int64_t nbytes, const int32_t *cnt, int chunk_size, float *x_out, void *stream)
{
    MixTable t;
    GP_TRY(mix_table(mean, scale, prob, k, Q, 0, &t));
    return decode_slices_impl(ctx, t, Q, slice_start, nslices, min_value, max_value, bytes, nbytes, cnt, chunk_size, x_out, stream);
}

extern "C" int gsge_forward(gpcc_ctx *ctx, const float *inputs, const float *embeddings, const int32_t *offsets, const int32_t *resolutions,
                            float *outputs, int64_t N, int num_dim, int n_features, int n_levels, int Rb, const uint8_t *binary_vxl,
                            const int32_t *min_level_id, void *stream)
{
    if (!ctx || !inputs || !embeddings || !offsets || !resolutions || !outputs) return fail(GPCC_ERR_ARG, "null argument");
    if (N <= 0 || n_levels <= 0) return GPCC_OK;
    if (N >= ((int64_t)1 << 31)) return fail(GPCC_ERR_ARG, "too many points");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    dim3 g((unsigned)cdiv(N, TB), (unsigned)n_levels);
    switch (num_dim) {
    case 1: return grid_launch_f<1>(st, n_features, g, inputs, embeddings, offsets, resolutions, outputs, (uint32_t)N, (uint32_t)Rb, binary_vxl, min_level_id);
    case 2: return grid_launch_f<2>(st, n_features, g, inputs, embeddings, offsets, resolutions, outputs, (uint32_t)N, (uint32_t)Rb, binary_vxl, min_level_id);
    case 3: return grid_launch_f<3>(st, n_features, g, inputs, embeddings, offsets, resolutions, outputs, (uint32_t)N, (uint32_t)Rb, binary_vxl, min_level_id);
    default: return fail(GPCC_ERR_ARG, "GridEncoding: num_dim must be 1, 2 or 3");
    }
}
