// container.hpp -- reading a .bin container's header and stream directory (reference reader: HAC/utils/pcc_utils.py:271-276,
// GausPcgc/kit/op.py:40-48; this library's chunked layouts: HISTORY.md section 5).  Everything here runs on UNTRUSTED bytes and
// is plain C++ (no HIP): gpcc_decode / gpcc_decode_batch call it, and tools/asan_host.sh builds the same code with
// AddressSanitizer + UBSan under a mutation fuzzer (tests/test_host_fuzz_cpu.py).
#pragma once
#include <vector>

#include "errors.hpp"
#include "rc_format.hpp"

namespace gpcc {

static inline uint32_t ct_get32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

struct ContainerHdr {
    bool chunked = false;            // FF FF | version ...: this library's layouts (versions 1-4); else the reference layout
    int version = 0, chunk_log2 = 0, L = -1;
    uint16_t posq = 0;               // bits of np.float16(posQ)
    int64_t lvl_n[MAXLV] = {0};      // chunked: nodes of every stored level, base first ([0] = bn in either layout)
    int64_t npts = -1;               // chunked: point count
    int64_t bn = 0;                  // base nodes
    const uint8_t *bxyz = nullptr;   // bn x 3 int32 (little endian), raster order
    const uint8_t *bocc = nullptr;   // bn occupancy bytes
    int nstreams = 0;
    std::vector<int64_t> s_off, s_len;   // body of every stream inside the file
};

// Cheap consistency checks of a chunked header BEFORE anything is sized from it: a base level below 64 nodes, at most 8 children
// per node, no more symbols than the file can carry (a lane of up to 2^14 symbols costs a table byte and a payload byte at
// least) -- a corrupt header of a few hundred bytes must not reserve gigabytes.  0 = not a chunked header (nothing checked).
static inline int container_precheck(const uint8_t *bytes, int64_t nbytes, int64_t *nodes_out, int64_t *nmax_out, int64_t *npts_out)
{
    if (!(nbytes >= 8 && bytes[0] == 0xFF && bytes[1] == 0xFF && bytes[6] >= 1 && bytes[6] <= 21 && nbytes >= 12 + 4 * (int64_t)bytes[6])) return 0;
    const int L = bytes[6];
    int64_t nodes = 0, nmax = 0, prev = 0;
    for (int d = 0; d < L; ++d) {
        const int64_t v = ct_get32(bytes + 8 + 4 * d);
        if (v <= 0 || (d == 0 ? v >= 64 : v > 8 * prev)) return fail(GPCC_ERR_FORMAT, "bad node count at level %d", d);
        nodes += v; nmax = nmax > v ? nmax : v; prev = v;
    }
    const int64_t npts = ct_get32(bytes + 8 + 4 * L);
    if (npts < 1 || npts > 8 * prev) return fail(GPCC_ERR_FORMAT, "header: %lld points under %lld finest nodes", (long long)npts, (long long)prev);
    if (nodes > (nbytes << 13)) return fail(GPCC_ERR_FORMAT, "header: %lld nodes cannot come from %lld bytes", (long long)nodes, (long long)nbytes);
    if (nodes_out) *nodes_out = nodes;
    if (nmax_out) *nmax_out = nmax;
    if (npts_out) *npts_out = npts;
    return 1;
}

static inline int container_parse(const uint8_t *in, int64_t nbytes, ContainerHdr *h)
{
    int64_t pos = 0;
#define CT_NEED(b) do { if ((int64_t)(b) < 0 || pos + (int64_t)(b) > nbytes) return fail(GPCC_ERR_FORMAT, "truncated bitstream (need %lld bytes at %lld of %lld)", (long long)(b), (long long)pos, (long long)nbytes); } while (0)
    CT_NEED(2);
    h->chunked = in[0] == 0xFF && in[1] == 0xFF;
    if (h->chunked) {
        CT_NEED(8);
        h->version = in[2];
        if (h->version < 1 || h->version > 4) return fail(GPCC_ERR_FORMAT, "unknown container version %d", h->version);
        h->chunk_log2 = in[3];
        if (h->chunk_log2 < 6 || h->chunk_log2 > 14) return fail(GPCC_ERR_FORMAT, "bad chunk_log2 %d", h->chunk_log2);
        h->posq = (uint16_t)(in[4] | in[5] << 8);
        h->L = in[6]; pos = 8;
        if (h->L < 1 || h->L > 21) return fail(GPCC_ERR_FORMAT, "bad level count %d", h->L);
        CT_NEED(4 * h->L + 4);
        for (int d = 0; d < h->L; ++d) { h->lvl_n[d] = ct_get32(in + pos); pos += 4; }
        h->npts = ct_get32(in + pos); pos += 4;
        if (h->npts < 1 || h->npts > 8 * h->lvl_n[h->L - 1]) return fail(GPCC_ERR_FORMAT, "header: %lld points under %lld finest nodes", (long long)h->npts, (long long)h->lvl_n[h->L - 1]);
    } else {
        h->version = 0; h->chunk_log2 = 0;
        h->posq = (uint16_t)(in[0] | in[1] << 8); pos = 2;
    }
    CT_NEED(4);
    h->bn = (int32_t)ct_get32(in + pos); pos += 4;
    if (h->bn <= 0 || h->bn >= 64) return fail(GPCC_ERR_FORMAT, "bad base length %lld", (long long)h->bn);
    CT_NEED(13 * h->bn + 2);
    h->bxyz = in + pos; pos += 12 * h->bn;
    h->bocc = in + pos; pos += h->bn;
    h->nstreams = in[pos] | in[pos + 1] << 8; pos += 2;
    if (h->nstreams % 4) return fail(GPCC_ERR_FORMAT, "stream count %d is not a multiple of 4", h->nstreams);
    if (h->chunked) {
        if (h->nstreams != 4 * (h->L - 1) || h->lvl_n[0] != h->bn) return fail(GPCC_ERR_FORMAT, "header/stream count mismatch");
        for (int g = 0; g + 1 < h->L; ++g)
            if (h->lvl_n[g + 1] <= 0 || h->lvl_n[g + 1] > 8 * h->lvl_n[g]) return fail(GPCC_ERR_FORMAT, "bad node count at level %d", g + 1);
    } else {
        h->L = h->nstreams / 4 + 1;
        if (h->L > 21) return fail(GPCC_ERR_FORMAT, "too many levels");
        h->lvl_n[0] = h->bn;
    }
    h->s_off.assign((size_t)h->nstreams, 0); h->s_len.assign((size_t)h->nstreams, 0);
    for (int si = 0; si < h->nstreams; ++si) {
        CT_NEED(4);
        const int64_t len = ct_get32(in + pos); pos += 4;
        CT_NEED(len);
        h->s_off[(size_t)si] = pos; h->s_len[(size_t)si] = len; pos += len;
    }
#undef CT_NEED
    return GPCC_OK;
}

// the four lane tables of coded level g + 1 (nc nodes): lanes[s * nlanes + l], win_bytes[s] = the longest byte window of stage s
static inline int container_level_tables(const uint8_t *in, const ContainerHdr &h, int g, int64_t nc, RcChunk *lanes, uint32_t win_bytes[4])
{
    const RcPlan pl = rc_plan(nc, h.chunk_log2, h.version);
    for (int s = 0; s < 4; ++s) {
        const int si = 4 * g + s;
        if (si >= h.nstreams) return fail(GPCC_ERR_FORMAT, "level %d has no stream %d", g + 1, s);
        const char *err = rc_parse_table(in + h.s_off[(size_t)si], h.s_off[(size_t)si], h.s_len[(size_t)si], pl, nc, h.version, lanes + (size_t)s * pl.nlanes, &win_bytes[s]);
        if (err) return fail(GPCC_ERR_FORMAT, "stream %d: %s", si, err);
    }
    return GPCC_OK;
}

}  // namespace gpcc
