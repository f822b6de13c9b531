// fuzz_host.cpp -- mutation fuzzer for everything on the HOST side that reads untrusted bytes, built with AddressSanitizer +
// UBSan (-fno-sanitize-recover) by tools/asan_host.sh and run by tests/test_host_fuzz_cpu.py.  No GPU, no HIP: the product's
// parsers are plain C++ headers (csrc/container.hpp, csrc/rc_format.hpp) and one plain C++ source (csrc/hostcoder.hip), and the
// oracle is plain C -- the SAME code gpcc_decode / gauspcc_amd.torchac / the tests' checker run.
//
// Readers being hardened: the container reader of HAC/utils/pcc_utils.py:271-276 and GausPcgc/kit/op.py:40-48 (here:
// container_parse, rc_parse_table / rc_table_get, LEB128 varints, the v0-v4 stream splitter), torchac's decode loop
// (gsac_host_decode_{u16,f32}), gpcc_write_files / gpcc_read_files, and the oracle's orc_decode / orc_stream_decode / orc_chunk_table_get.
//
//   fuzz_host [--parse N] [--decode N] [--coder N] [--seed S]
// Seeds are made in-process: the oracle encodes a few tiny seeded clouds with seeded random weights in every container layout
// (reference layout, versions 3 and 4, two chunk sizes).  Exit code 0 and the line "fuzz_host: ok ..." = no sanitizer report,
// no crash, every unmutated seed still parses and decodes to its cloud.
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../gauspcc_amd/csrc/container.hpp"

namespace gpcc {
thread_local char g_err[512] = "";
thread_local long long g_launches = 0;
}

extern "C" {
// oracle/gpcc_oracle.c
int64_t orc_encode(const float *const *tensors, int C, int k, const int32_t *xyz, int64_t n, int chunk_log2, uint16_t posq_f16, uint8_t *out, int64_t cap);
int64_t orc_decode(const float *const *tensors, int C, int k, const uint8_t *in, int64_t nbytes, int32_t *xyz_out, int64_t cap_pts, uint16_t *posq_f16);
int orc_set_container_version(int v);
int orc_set_threads(int n);
int orc_stream_decode(const uint16_t *cdf, int Lp, const uint8_t *in, int64_t nbytes, int64_t n, int chunk_log2, int version, uint8_t *sym);
int64_t orc_stream_encode(const uint16_t *cdf, int Lp, const uint8_t *sym, int64_t n, int chunk_log2, int version, uint8_t *out, int64_t cap);
int64_t orc_chunk_table_get(const uint8_t *in, int64_t nbytes, int64_t nch, uint32_t *counts);
// csrc/hostcoder.hip
int gsac_host_encode_u16(const int16_t *sym, const uint16_t *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out);
int gsac_host_decode_u16(const uint16_t *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out);
int gsac_host_encode_f32(const int16_t *sym, const float *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out);
int gsac_host_decode_f32(const float *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out);
int gpcc_write_files(const char *const *paths, const uint8_t *const *data, const int64_t *sizes, int n, int threads);
int gpcc_read_files(const char *const *paths, int n, int threads, const uint8_t **blob_out, int64_t *offsets_out);
}
#include "../gauspcc_amd/csrc/hostcoder.hpp"   // the reference-layout container's host coder (round 6): gpcc::host_encode_streams / host_decode_compact


using namespace gpcc;

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd()
{
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t rndn(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0u; }
static inline float rndf() { return (float)((rnd() >> 40) * (1.0 / 16777216.0)); }

struct Model {
    std::vector<std::vector<float>> t;
    std::vector<const float *> p;
    int C = 32, k = 3;
};

static Model make_model(int k)
{
    Model m; m.k = k;
    const int C = 32, K = k * k * k, M[4] = {2, 2, 4, 16};
    auto add = [&](size_t n, float amp) { std::vector<float> v(n); for (auto &x : v) x = (rndf() * 2.0f - 1.0f) * amp; m.t.push_back(std::move(v)); };
    add(256 * C, 1.0f);
    for (int i = 0; i < 18; ++i) add((size_t)K * C * C, 4.0f / sqrtf((float)(C * K)));
    add(8 * C, 1.0f);
    for (int i = 0; i < 4; ++i) add((size_t)C * C, 0.17f);
    for (int i = 0; i < 4; ++i) add((size_t)C, 0.17f);
    for (int i = 0; i < 4; ++i) add((size_t)M[i] * C, 0.17f);
    for (int i = 0; i < 4; ++i) add((size_t)M[i], 0.17f);
    for (int i = 0; i < 3; ++i) add((size_t)(i == 0 ? 2 : i == 1 ? 4 : 16) * C, 1.0f);
    for (auto &v : m.t) m.p.push_back(v.data());
    return m;
}

static std::vector<int32_t> make_cloud(int n, int extent, int32_t shift)
{
    std::vector<uint64_t> keys;
    std::vector<int32_t> xyz;
    while ((int)keys.size() < n) {
        const int32_t c[3] = {(int32_t)rndn((uint32_t)extent), (int32_t)rndn((uint32_t)extent), (int32_t)rndn((uint32_t)extent / 4 + 1)};
        const uint64_t key = ((uint64_t)c[2] << 42) | ((uint64_t)c[1] << 21) | (uint64_t)c[0];
        if (std::find(keys.begin(), keys.end(), key) != keys.end()) continue;
        keys.push_back(key);
        for (int a = 0; a < 3; ++a) xyz.push_back(c[a] + shift);
    }
    return xyz;
}

struct Seed { std::vector<uint8_t> bytes; std::vector<int32_t> cloud; int k; };

static void mutate(std::vector<uint8_t> &b, const std::vector<Seed> &seeds)
{
    const int kind = (int)rndn(10);
    const bool chunked = b.size() > 8 && b[0] == 0xFF && b[1] == 0xFF;
    const size_t hdr = chunked ? std::min<size_t>(b.size(), 12 + 4 * (size_t)b[6]) : 2;
    auto put32 = [&](size_t at, uint32_t v) { if (at + 4 <= b.size()) { b[at] = (uint8_t)v; b[at + 1] = (uint8_t)(v >> 8); b[at + 2] = (uint8_t)(v >> 16); b[at + 3] = (uint8_t)(v >> 24); } };
    auto get32 = [&](size_t at) -> uint32_t { return at + 4 <= b.size() ? ct_get32(b.data() + at) : 0u; };
    switch (kind) {
    case 0: { const int flips = 1 + (int)rndn(8); for (int i = 0; i < flips && !b.empty(); ++i) b[rndn((uint32_t)b.size())] ^= (uint8_t)(1u << rndn(8)); break; }   // bit flips anywhere
    case 1: { if (!b.empty()) b.resize(rndn((uint32_t)b.size())); break; }                                                                         // truncation
    case 2: { const int flips = 1 + (int)rndn(3); for (int i = 0; i < flips && hdr; ++i) b[rndn((uint32_t)hdr)] ^= (uint8_t)(1u << rndn(8)); break; }   // header bits
    case 3: {   // a lying level count / point count
        if (chunked) { const size_t at = 8 + 4 * (size_t)rndn((uint32_t)b[6] + 1u); const uint32_t v = get32(at); const uint32_t alt[6] = {0u, 1u, v + 1u, v - 1u, v * 8u, 0xFFFFFFFFu}; put32(at, alt[rndn(6)]); }
        else if (b.size() > 6) put32(2, rndn(70));
        break;
    }
    case 4: {   // splice: the tail of another seed behind this one's head
        const Seed &o = seeds[rndn((uint32_t)seeds.size())];
        if (!b.empty() && !o.bytes.empty()) { const size_t cut = rndn((uint32_t)b.size()), from = rndn((uint32_t)o.bytes.size()); b.resize(cut); b.insert(b.end(), o.bytes.begin() + (ptrdiff_t)from, o.bytes.end()); }
        break;
    }
    case 5: { const int nset = 1 + (int)rndn(16); for (int i = 0; i < nset && !b.empty(); ++i) b[rndn((uint32_t)b.size())] = (uint8_t)rnd(); break; }   // random bytes
    case 6: { if (chunked && b.size() > 8) { b[2] = (uint8_t)rndn(7); if (rndn(2)) b[3] = (uint8_t)rndn(20); if (rndn(4) == 0) b[6] = (uint8_t)rndn(30); } else if (b.size() > 2) b[0] = 0xFF, b[1] = 0xFF; break; }   // version / chunk_log2 / L
    case 7: {   // a stream length that lies (the first few length fields behind the base level)
        ContainerHdr h;
        if (container_parse(b.data(), (int64_t)b.size(), &h) == GPCC_OK && h.nstreams) {
            const int si = (int)rndn((uint32_t)h.nstreams);
            const size_t at = (size_t)h.s_off[(size_t)si] - 4;
            const uint32_t v = get32(at); const uint32_t alt[5] = {0u, v + 1u, v - 1u, v * 2u, 0x7FFFFFFFu};
            put32(at, alt[rndn(5)]);
        }
        break;
    }
    case 8: {   // garbage inside one stream's chunk table (its first bytes)
        ContainerHdr h;
        if (container_parse(b.data(), (int64_t)b.size(), &h) == GPCC_OK && h.nstreams) {
            const int si = (int)rndn((uint32_t)h.nstreams);
            const int64_t len = std::min<int64_t>(h.s_len[(size_t)si], 12);
            for (int64_t i = 0; i < len; ++i) if (rndn(2)) b[(size_t)(h.s_off[(size_t)si] + i)] = (uint8_t)rnd();
        }
        break;
    }
    default: {   // insert / delete a run of bytes
        if (b.empty()) break;
        const size_t at = rndn((uint32_t)b.size()), len = 1 + rndn(9);
        if (rndn(2)) b.insert(b.begin() + (ptrdiff_t)at, len, (uint8_t)rnd());
        else b.erase(b.begin() + (ptrdiff_t)at, b.begin() + (ptrdiff_t)std::min(b.size(), at + len));
        break;
    }
    }
}

// what gpcc_decode does with the bytes before anything touches the device (codec.hip: decode_entry, decode_body)
static int host_parse(const uint8_t *in, int64_t n, int64_t *lanes_out)
{
    int64_t nodes = 0, nmax = 0, npts = 0;
    const int pre = container_precheck(in, n, &nodes, &nmax, &npts);
    if (pre < 0) return pre;
    ContainerHdr h;
    GP_TRY(container_parse(in, n, &h));
    if (!h.chunked) return GPCC_OK;   // (the reference layout: the level sizes come from the decoded occupancy, one lane per stream)
    std::vector<RcChunk> lanes;
    int64_t total = 0;
    for (int g = 0; g + 1 < h.L; ++g) {
        const int64_t nc = h.lvl_n[g + 1];
        const RcPlan pl = rc_plan(nc, h.chunk_log2, h.version);
        lanes.assign((size_t)4 * pl.nlanes, RcChunk{0, 0, 0, 0, 0, 0});
        uint32_t wb[4] = {0, 0, 0, 0};
        GP_TRY(container_level_tables(in, h, g, nc, lanes.data(), wb));
        // what the device decoder relies on: every lane's bytes inside the file, symbols inside the level
        for (const RcChunk &c : lanes) {
            const uint32_t nb = c.nbytes & ~RC_BACKWARDS;
            const int64_t lo = (c.nbytes & RC_BACKWARDS) ? (int64_t)c.byte_off - (int64_t)nb + 1 : (int64_t)c.byte_off;
            if (lo < 0 || lo + (int64_t)nb > n) { fprintf(stderr, "fuzz_host: lane bytes [%lld, +%u) outside a file of %lld bytes\n", (long long)lo, nb, (long long)n); abort(); }
            if ((int64_t)c.out + (int64_t)c.n > nc) { fprintf(stderr, "fuzz_host: lane symbols beyond the level\n"); abort(); }
        }
        total += (int64_t)lanes.size();
    }
    if (lanes_out) *lanes_out = total;
    return GPCC_OK;
}

int main(int argc, char **argv)
{
    long n_parse = 120000, n_decode = 3000, n_coder = 10000;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--parse")) n_parse = atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--decode")) n_decode = atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--coder")) n_coder = atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) g_rng = strtoull(argv[i + 1], nullptr, 0);
    }
    orc_set_threads(1);
    Model m3 = make_model(3);
    // ---- seeds: tiny clouds (a base level and one to three coded levels) in every layout
    std::vector<Seed> seeds;
    const int sizes[5] = {70, 150, 400, 1200, 40};
    for (int s = 0; s < 5; ++s)
        for (int layout = 0; layout < 5; ++layout) {
            Seed sd; sd.k = 3;
            sd.cloud = make_cloud(sizes[s], s == 4 ? 4096 : 64 << (s / 2), s == 3 ? -1000000 : (s == 1 ? 1500000 : 0));
            const int chunk_log2 = layout == 0 ? 0 : (layout & 1) ? 6 : 11;
            orc_set_container_version(layout <= 2 ? 4 : 3);
            sd.bytes.resize(1 << 20);
            const int64_t nb = orc_encode(m3.p.data(), 32, 3, sd.cloud.data(), sizes[s], chunk_log2, 0x3C00, sd.bytes.data(), (int64_t)sd.bytes.size());
            if (nb <= 0) { fprintf(stderr, "fuzz_host: the oracle cannot encode seed %d/%d\n", s, layout); return 2; }
            sd.bytes.resize((size_t)nb);
            seeds.push_back(std::move(sd));
        }
    orc_set_container_version(4);
    // every seed parses, and decodes to its cloud (as a set: the decoder's order is the raster order of the last level)
    for (const Seed &sd : seeds) {
        int64_t lanes = 0;
        if (host_parse(sd.bytes.data(), (int64_t)sd.bytes.size(), &lanes) != GPCC_OK) { fprintf(stderr, "fuzz_host: a valid seed does not parse: %s\n", g_err); return 2; }
        std::vector<int32_t> out(sd.cloud.size());
        uint16_t pq = 0;
        const int64_t np = orc_decode(m3.p.data(), 32, 3, sd.bytes.data(), (int64_t)sd.bytes.size(), out.data(), (int64_t)out.size() / 3, &pq);
        if (np * 3 != (int64_t)sd.cloud.size()) { fprintf(stderr, "fuzz_host: a valid seed decodes to %lld points\n", (long long)np); return 2; }
        auto key = [](const int32_t *p) { return ((uint64_t)(uint32_t)p[2] << 42) ^ ((uint64_t)(uint32_t)p[1] << 21) ^ (uint64_t)(uint32_t)p[0]; };
        std::vector<uint64_t> a, b;
        for (size_t i = 0; i < out.size(); i += 3) { a.push_back(key(&out[i])); b.push_back(key(&sd.cloud[i])); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        if (a != b) { fprintf(stderr, "fuzz_host: a valid seed decodes to other points\n"); return 2; }
    }
    // ---- 1. the product's host parsers
    long parsed_ok = 0, parsed_err = 0;
    for (long it = 0; it < n_parse; ++it) {
        std::vector<uint8_t> b = seeds[rndn((uint32_t)seeds.size())].bytes;
        const int rounds = 1 + (int)rndn(3);
        for (int r = 0; r < rounds; ++r) mutate(b, seeds);
        // the buffer is exactly as long as the file: any read past the end is a heap overflow the sanitizer sees
        std::vector<uint8_t> exact(b.begin(), b.end());
        if (exact.empty()) exact.reserve(1);
        const int rc = host_parse(exact.data(), (int64_t)exact.size(), nullptr);
        if (rc == GPCC_OK) ++parsed_ok; else ++parsed_err;
    }
    // ---- 2. the oracle's decoder (the checker of every parity test reads the same untrusted layouts)
    long dec_ok = 0, dec_err = 0;
    {
        std::vector<int32_t> out(3 * 20000);
        for (long it = 0; it < n_decode; ++it) {
            const uint32_t pick = rndn(15);
            const Seed &sd = seeds[pick < 10 ? pick : pick + 10];   // the three smallest clouds (70, 150 and 40 points) in every layout: a decode is milliseconds
            std::vector<uint8_t> b = sd.bytes;
            const int rounds = 1 + (int)rndn(2);
            for (int r = 0; r < rounds; ++r) mutate(b, seeds);
            std::vector<uint8_t> exact(b.begin(), b.end());
            if (exact.empty()) exact.reserve(1);
            // a header that announces an absurd level is refused by the product before anything is sized from it; the oracle
            // re-derives sizes from the occupancy, so a corrupt occupancy can at most expand 8-fold per level: bounded by cap_pts
            uint16_t pq = 0;
            const int64_t cap = std::min<int64_t>((int64_t)out.size() / 3, 4 * (int64_t)sd.cloud.size() / 3 + 64);   // (a level beyond the capacity is refused before the network runs on it)
            const int64_t np = orc_decode(m3.p.data(), 32, 3, exact.data(), (int64_t)exact.size(), out.data(), cap, &pq);
            if (np >= 0) ++dec_ok; else ++dec_err;
        }
    }
    // ---- 3. stand-alone stream / table / torchac coders on random (cdf, bytes) with wrong n / lp
    long coder_calls = 0;
    for (long it = 0; it < n_coder; ++it) {
        const int lps[6] = {3, 5, 17, 2, 33, 257};
        const int lp = lps[rndn(6)];
        const int64_t n = 1 + rndn(it % 50 == 0 ? 5000u : 300u);
        std::vector<uint16_t> cdf((size_t)n * lp);
        std::vector<float> cdff((size_t)n * lp);
        const int style = (int)rndn(4);
        for (int64_t i = 0; i < n; ++i) {
            // style 0: valid increasing rows; 1: random u16 garbage; 2: saturated rows; 3: valid floats incl. 0 / 1 edges
            uint32_t acc = 0;
            for (int j = 0; j < lp; ++j) {
                uint32_t v;
                if (style == 1) v = (uint32_t)rnd() & 0xFFFFu;
                else if (style == 2) v = j == 0 ? 0u : (j == lp - 1 ? 0u : (rndn(2) ? 65535u : 1u));
                else { acc += 1 + rndn((uint32_t)std::max(1, (65536 - (lp - j)) / lp - 1)); v = j == 0 ? 0u : (j == lp - 1 ? 0u : std::min(acc, 65535u)); }
                cdf[(size_t)i * lp + j] = (uint16_t)v;
                cdff[(size_t)i * lp + j] = style == 1 ? rndf() * 1.5f - 0.25f : (j == 0 ? 0.0f : (j == lp - 1 ? 1.0f : (float)std::min(acc, 65535u) / 65536.0f));
            }
        }
        const size_t nb = rndn(it % 7 == 0 ? 4u : 2000u);
        std::vector<uint8_t> bytes(nb);
        for (auto &x : bytes) x = (uint8_t)rnd();
        if (bytes.empty()) bytes.reserve(1);
        std::vector<int16_t> sym((size_t)n);
        std::vector<uint8_t> sym8((size_t)n + 4);
        // torchac's loop on host arrays: right sizes, then a stream that ends early
        (void)gsac_host_decode_u16(cdf.data(), bytes.data(), (int64_t)nb, n, lp, sym.data());
        (void)gsac_host_decode_f32(cdff.data(), bytes.data(), (int64_t)nb, n, lp, sym.data());
        (void)gsac_host_decode_u16(cdf.data(), bytes.data(), (int64_t)(nb / 2), n, lp, sym.data());
        // encode what was decoded (symbols are in range by construction), then decode that: a round trip on valid rows
        if (style == 0 || style == 3) {
            std::vector<int16_t> s2((size_t)n);
            for (auto &x : s2) x = (int16_t)rndn((uint32_t)lp - 1u);
            std::vector<uint8_t> enc((size_t)n * 4 + 64);
            int64_t enb = 0;
            if (gsac_host_encode_u16(s2.data(), cdf.data(), n, lp, enc.data(), (int64_t)enc.size(), &enb) == GPCC_OK && style == 0) {
                std::vector<uint8_t> ex(enc.begin(), enc.begin() + (ptrdiff_t)enb);
                if (ex.empty()) ex.reserve(1);
                std::vector<int16_t> back((size_t)n);
                if (gsac_host_decode_u16(cdf.data(), ex.data(), enb, n, lp, back.data()) != GPCC_OK || back != s2) { fprintf(stderr, "fuzz_host: host coder round trip failed (lp %d, n %lld)\n", lp, (long long)n); return 3; }
            }
            int64_t enb2 = 0;
            (void)gsac_host_encode_f32(s2.data(), cdff.data(), n, lp, enc.data(), (int64_t)enc.size(), &enb2);
            // a too-small output buffer is an error, not an overflow
            (void)gsac_host_encode_u16(s2.data(), cdf.data(), n, lp, enc.data(), (int64_t)rndn(8), &enb2);
        }
        // the reference-layout container's host coder (csrc/hostcoder.hpp): compact rows (interior values only) against garbage bytes, and a round
        // trip -- the packed coder words of random symbols under valid rows through host_encode_streams, back through host_decode_compact
        if (lp == 3 || lp == 5 || lp == 17) {
            const int rs = lp == 3 ? 1 : lp == 5 ? 4 : 16;
            std::vector<uint16_t> comp((size_t)n * rs, 0);
            for (int64_t i = 0; i < n; ++i) for (int j = 1; j <= lp - 2; ++j) comp[(size_t)i * rs + (j - 1)] = cdf[(size_t)i * lp + j];
            (void)gpcc::host_decode_compact(comp.data(), lp, bytes.data(), (int64_t)nb, n, sym8.data());
            (void)gpcc::host_decode_compact(comp.data(), lp, bytes.data(), (int64_t)(nb / 3), n, sym8.data());
            if (style == 0) {
                std::vector<uint32_t> words((size_t)n);
                std::vector<uint8_t> want((size_t)n);
                for (int64_t i = 0; i < n; ++i) {
                    const int sy = (int)rndn((uint32_t)lp - 1u);
                    want[(size_t)i] = (uint8_t)sy;
                    const uint32_t lo = cdf[(size_t)i * lp + sy], hi = sy == lp - 2 ? 0x10000u : cdf[(size_t)i * lp + sy + 1];
                    words[(size_t)i] = lo | ((hi - 1u) << 16);
                }
                const uint32_t *sp[2] = {words.data(), words.data()};
                const int64_t sn[2] = {n, n / 2};
                std::vector<std::vector<uint8_t>> outs;
                if (gpcc::host_encode_streams(sp, sn, 2, &outs, 1 + (int)rndn(3)) != GPCC_OK) { fprintf(stderr, "fuzz_host: host_encode_streams failed on valid rows\n"); return 3; }
                std::vector<uint8_t> back((size_t)n);
                std::vector<uint8_t> ex = outs[0];
                if (ex.empty()) ex.reserve(1);
                if (gpcc::host_decode_compact(comp.data(), lp, ex.data(), (int64_t)outs[0].size(), n, back.data()) != GPCC_OK || back != want) {
                    fprintf(stderr, "fuzz_host: reference-layout host coder round trip failed (lp %d, n %lld)\n", lp, (long long)n); return 3;
                }
            }
        }
        // the oracle's stream splitter (versions 0-4) and chunk-table reader on the same garbage
        if (lp == 3 || lp == 5 || lp == 17) {
            const int version = (int)rndn(5);
            (void)orc_stream_decode(cdf.data(), lp, bytes.data(), (int64_t)nb, n, version == 0 ? 0 : 6 + (int)rndn(9), version, sym8.data());
        }
        {
            const int64_t nch = 1 + rndn(64);
            std::vector<uint32_t> counts((size_t)nch);
            (void)orc_chunk_table_get(bytes.data(), (int64_t)nb, nch, counts.data());
            std::vector<uint32_t> cb((size_t)nch);
            (void)rc_table_get(bytes.data(), nb, cb.data(), (uint32_t)nch);
            uint32_t v = 0;
            (void)rc_varint_get(bytes.data(), nb, &v);
        }
        ++coder_calls;
    }
    // ---- 4. the native file writer: good paths, a directory that does not exist, an empty file, null entries
    {
        char dir[] = "/tmp/fuzz_host_XXXXXX";
        if (mkdtemp(dir)) {
            std::string a = std::string(dir) + "/a.b", b = std::string(dir) + "/b.b", c = std::string(dir) + "/nope/c.b";
            const uint8_t payload[5] = {1, 2, 3, 4, 5};
            const char *paths[3] = {a.c_str(), b.c_str(), c.c_str()};
            const uint8_t *data[3] = {payload, nullptr, payload};
            const int64_t sz[3] = {5, 0, 5};
            if (gpcc_write_files(paths, data, sz, 2, 4) != GPCC_OK) { fprintf(stderr, "fuzz_host: gpcc_write_files failed on good paths\n"); return 3; }
            if (gpcc_write_files(paths, data, sz, 3, 4) == GPCC_OK) { fprintf(stderr, "fuzz_host: gpcc_write_files wrote into a missing directory\n"); return 3; }
            const char *np[1] = {nullptr};
            (void)gpcc_write_files(np, data, sz, 1, 1);
            (void)gpcc_write_files(nullptr, nullptr, nullptr, 0, 0);
            // ... and the reader: the two files back (one empty), a missing file, null entries
            const uint8_t *blob = nullptr;
            int64_t offs[4] = {0, 0, 0, 0};
            if (gpcc_read_files(paths, 2, 4, &blob, offs) != GPCC_OK || offs[1] != 5 || offs[2] != 5 || memcmp(blob, payload, 5) != 0) { fprintf(stderr, "fuzz_host: gpcc_read_files failed on good paths\n"); return 3; }
            if (gpcc_read_files(paths, 3, 4, &blob, offs) == GPCC_OK) { fprintf(stderr, "fuzz_host: gpcc_read_files read a missing file\n"); return 3; }
            (void)gpcc_read_files(np, 1, 1, &blob, offs);
            (void)gpcc_read_files(nullptr, 0, 0, &blob, offs);
            (void)gpcc_read_files(paths, 2, 2, nullptr, offs);
            remove(a.c_str()); remove(b.c_str()); rmdir(dir);
        }
    }
    printf("fuzz_host: ok  seeds %zu | parsers: %ld mutants (%ld parsed, %ld refused) | oracle decode: %ld mutants (%ld clouds, %ld refused) | coders: %ld rounds\n",
           seeds.size(), n_parse, parsed_ok, parsed_err, n_decode, dec_ok, dec_err, coder_calls);
    return 0;
}
