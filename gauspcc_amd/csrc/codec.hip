// codec.hip -- gpcc_encode / gpcc_decode: the device pipeline behind compress_point_cloud /
// decompress_point_cloud (HAC/utils/pcc_utils.py:24-217, 230-400).
//
// Encode: octree build (Morton order) -> tile lists of every level (tiles.hip) -> prior trunk (5 convs on the
// parent levels), child features, target trunk (5 convs on the coded levels), the four teacher-forced
// stages batched into two 4-job conv launches, four head launches that emit one packed
// (c_low, c_high) word per symbol in raster order -> ONE range-coder launch over every chunk of
// every stream -> scan + compaction -> one D2H copy -> container assembly on the host.
// Host syncs: bbox, level sizes, final byte count (3 per encode; the reference does 8 per level).
//
// Decode: the same network kernels, but stage s+1 needs the symbols of stage s, so each level runs
// 4 x (stage input, 2 convs, head -> CDF rows, chunk-parallel range decode).
#include <algorithm>
#include <chrono>

#include <functional>
#include "hostcoder.hpp"
#include "codec_shared.hpp"
#include "container.hpp"
#include "rangecoder_dev.hpp"

using namespace gpcc;

namespace {

// internal return value of encode_body: a chunk came out larger than the staged decoder's LDS window (rangecoder.hpp:
// rc_window_fits) -- the caller retries with chunk_log2 - 1 (the header records the value used)
constexpr int ENC_RETRY_SMALLER_CHUNKS = -1000;
// internal return value of decode_body: a persistent small-level launch timed out (fused.hip) -- the caller decodes again on the
// launch-per-layer path (ctx->fused_off is set)
constexpr int DEC_RETRY_UNFUSED = -1001;

// lohi slot of stage 0 and stage stride of every row of C: where the heads put a node's coder input (needs the raster ranks)
__global__ __launch_bounds__(256) void k_set_pos(SetLevels S, int64_t nC, uint32_t *__restrict__ pos_out, uint32_t *__restrict__ slots_out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nC) return;
    int d = 1;
    for (int q = 2; q < S.L; ++q) d = i >= (int64_t)S.cbase[q] ? q : d;
    pos_out[i] = S.lohi_base[d] + rc_interleaved(S.m2r[d][i - S.cbase[d]], S.clog[d], S.nch[d]);
    slots_out[i] = S.slots[d];
}

int encode_body(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *xyz, int64_t n, int chunk_log2, uint16_t posq,
                const uint8_t **bytes_out, int64_t *nbytes_out, gpcc_stats *stats, hipStream_t st)
{
    ctx->arena.reset();
    HostTrace ht;
    const bool want_bits = stats && (stats->flags & GPCC_STATS_IDEAL_BITS);   // the reference's bpp estimator (a14): on request only
    GP_TRY(ctx->side_init());
    hipStream_t sd = ctx->side;
    struct SideGuard { hipStream_t s; ~SideGuard() { (void)hipStreamSynchronize(s); } } side_guard{sd};   // error returns leave nothing in flight
    Tree T;
    {
        StageTimer tm(ctx, st, ST_OCTREE, 0.0);
        GP_TRY(tree_build(ctx, st, xyz, n, &T));
        tm.add_bytes(tree_alg_bytes(T));
    }
    ht.mark("enc tree built");
    const int L = T.L;
    if (L == 1) GP_TRY(tree_ranks(ctx, st, &T));   // a cloud that is its own base level: nothing to overlap with
    int64_t coded = 0, nmax = 0;
    for (int d = 0; d < L; ++d) { nmax = std::max(nmax, T.lv[d].n); if (d) coded += T.lv[d].n; }
    if (coded >= ((int64_t)1 << 30)) return fail(GPCC_ERR_ARG, "too many octree nodes");
    // stream-major packed symbols: stream (d, s), d = 1..L-1, occupies slots(d) words (the chunk-interleaved
    // layout pads the last chunk): offset 4 * sum_{d' < d} slots(d') + s * slots(d)
    // (container version 3: rangecoder.hpp -- a stream is cut into LANES of 2^llog symbols, one coder state each; two lanes,
    // one coded forwards and one backwards, share a byte-counted chunk)
    const int CONTAINER_VERSION = ctx->container_version;   // 4: the carry-propagating coder in the lanes; 3: torchac's (gpcc_ctx_set_container_version)
    auto plan = [&](int64_t nc) -> RcPlan { return rc_plan(nc, chunk_log2, CONTAINER_VERSION); };
    auto clog = [&](int64_t nc) -> int { return plan(nc).llog; };                       // lane size log2
    auto slots = [&](int64_t nc) -> int64_t { return chunk_log2 ? (int64_t)plan(nc).nlanes << clog(nc) : nc; };
    int64_t lohi_words = 0;
    for (int d = 1; d < L; ++d) lohi_words += 4 * slots(T.lv[d].n);
    if (lohi_words >= ((int64_t)1 << 32)) return fail(GPCC_ERR_ARG, "too many octree nodes");
    TAKE(lohi, uint32_t, std::max<int64_t>(lohi_words, 1));
    constexpr int NCOUNTERS = MAXLV + 16;   // pairs per level, then 16 accumulators of the ideal code length
    TAKE(pairs_dev, unsigned long long, NCOUNTERS);
    HIP_TRY(hipMemsetAsync(pairs_dev, 0, sizeof(unsigned long long) * NCOUNTERS, st));
    // Encoding is teacher-forced, so every level is independent of the others: all parent levels are
    // concatenated into one "prior set" P (levels 0..L-2) and all coded levels into one "target set" C
    // (levels 1..L-1).  Each of the 18 network layers is then ONE launch over a set instead of one per level.
    int64_t nP = 0;
    for (int d = 0; d + 1 < L; ++d) nP += T.lv[d].n;
    const int64_t nC = coded;
    if (L > 1) {
        int64_t pb[MAXLV] = {0}, cbase[MAXLV] = {0};
        for (int d = 1; d < L; ++d) { pb[d] = pb[d - 1] + T.lv[d - 1].n; }
        for (int d = 2; d < L; ++d) { cbase[d] = cbase[d - 1] + T.lv[d - 1].n; }
        TAKE(occP, uint8_t, nP); TAKE(occC, uint8_t, nC); TAKE(rkeyC, uint64_t, nC);
        TAKE(parentC, uint32_t, nC); TAKE(posC, uint32_t, nC); TAKE(slotsC, uint32_t, nC);
        SetLevels S = {};
        {
            S.L = L;
            int64_t lohi_base = 0;
            for (int d = 0; d < L; ++d) {
                const Level *lv = &T.lv[d];
                S.n[d] = (uint32_t)lv->n; S.pb[d] = (uint32_t)pb[d]; S.cbase[d] = (uint32_t)cbase[d];
                S.occ[d] = lv->occ; S.rkey[d] = lv->rkey; S.parent[d] = lv->parent; S.m2r[d] = lv->m2r;
                if (d) {
                    S.lohi_base[d] = (uint32_t)lohi_base; S.slots[d] = (uint32_t)slots(lv->n); S.clog[d] = clog(lv->n);
                    S.nch[d] = plan(lv->n).nlanes;
                    lohi_base += 4 * slots(lv->n);
                }
            }
            StageTimer tm(ctx, st, ST_ELEM, (double)nP * 2 + (double)nC * (2 + 16 + 8));
            k_set_rows<<<(unsigned)cdiv(std::max(nP, nC), 256), 256, 0, st>>>(S, nP, nC, occP, occC, rkeyC, parentC);
            LAUNCH_CHECK();
        }
        HIP_TRY(hipEventRecord(ctx->ev_main, st));   // the tree is complete on st
        // Second stream: the raster ranks of every level and what depends on them (the coder slots of the rows of C).
        // Their first reader is the head of stage 0, eighteen convolutions away; temporaries come from the top of the
        // arena (Arena::flip), which nothing else in an encode uses.  The ~190 small launches cost the host 0.4 ms to queue:
        // that happens behind the first trunk's launches, when the device has milliseconds of work in hand.
        const std::function<int()> queue_ranks = [&]() -> int {
            HIP_TRY(hipStreamWaitEvent(sd, ctx->ev_main, 0));
            ctx->arena.flip = true;
            int rc = GPCC_OK;
            {
                StageTimer tm(ctx, sd, ST_OCTREE, 0.0);
                rc = tree_ranks(ctx, sd, &T);
            }
            ctx->arena.flip = false;
            GP_TRY(rc);
            {
                StageTimer tm(ctx, sd, ST_ELEM, (double)nC * (4 + 8));
                k_set_pos<<<(unsigned)cdiv(nC, 256), 256, 0, sd>>>(S, nC, posC, slotsC);
                LAUNCH_CHECK();
            }
            HIP_TRY(hipEventRecord(ctx->ev_side, sd));
            return GPCC_OK;
        };
        ht.mark("enc meta queued");
        // Tile lists of every level in one pool, built top-down from the cell maps (tiles.hip; one stream sync for the pool
        // size).  A level's list is the same in the prior and in the target set -- tile entries are row indices inside the
        // level -- so the two sets are two views of the pool: levels 0..L-2 and 1..L-1.
        ConvTiles tilesP, tilesC;
        {
            const int NPc = cell_map_entries(m->k);
            TileLevel tl[MAXLV];
            const int32_t *cell_prev = nullptr;
            for (int d = 0; d < L; ++d) {
                int32_t *own = nullptr;
                if (d + 1 < L) { TAKE(cm, int32_t, (int64_t)NPc * T.lv[d].n); own = cm; }
                tl[d] = TileLevel{&T.lv[d], d ? &T.lv[d - 1] : nullptr, cell_prev, own};
                cell_prev = own;
            }
            const int R = conv_pick_rows(nC, m->k), H = conv_pick_height(nC, R);
            TilePool pool;
            StageTimer tm(ctx, st, ST_TILES, 0.0);
            GP_TRY(tiles_build(ctx, st, tl, L, m->k, R, H, &pool, pairs_dev));
            GP_TRY(tiles_view(ctx, st, pool, 0, L - 1, pb, &tilesP));
            GP_TRY(tiles_view(ctx, st, pool, 1, L, cbase + 1, &tilesC));
            tm.add_bytes(pool.alg_bytes);
        }
        ht.mark("enc tiles built");
        TAKE(pF, float, nP * m->C); TAKE(pA, float, nP * m->C); TAKE(pB, float, nP * m->C);
        { StageTimer tm(ctx, st, ST_ELEM, (double)nP * 129); GP_TRY(embed_occ(st, m->prior_emb, occP, nP, pF, m->C)); }
        GP_TRY(run_trunk(ctx, 0, st, m, 0, Trunk{pF, pA, pB}, tilesP, nP));           // -> pA
        GP_TRY(queue_ranks());   // the device now has milliseconds of convolutions queued: the host time of these launches is free
        TAKE(cX, float, nC * m->C); TAKE(cA, float, nC * m->C); TAKE(cB, float, nC * m->C);
        { StageTimer tm(ctx, st, ST_ELEM, (double)nC * (128 + 12 + 128)); GP_TRY(child_features(st, pA, parentC, rkeyC, m->temb, nC, cX, m->C)); }
        GP_TRY(run_trunk(ctx, 1, st, m, 5, Trunk{cX, cA, cB}, tilesC, nC));           // -> cA  (X of pcc_utils.py:109)
        // stages: cX, cB are free now; inputs u[s], mid v[s], outputs y[s]
        TAKE(u1, float, nC * m->C); TAKE(u2, float, nC * m->C); TAKE(u3, float, nC * m->C);
        TAKE(v1, float, nC * m->C); TAKE(v2, float, nC * m->C);
        float *u[4] = {cA, u1, u2, u3};
        float *v[4] = {cX, cB, v1, v2};
        {
            const float *const embs[3] = {m->semb[0], m->semb[1], m->semb[2]};
            float *const outs[3] = {u1, u2, u3};
            StageTimer tm(ctx, st, ST_ELEM, (double)nC * (128 + 1 + 3 * 128));
            GP_TRY(stage_inputs_gt(st, cA, embs, occC, nC, outs, m->C));
        }
        ConvBatch cb = {}; cb.C = m->C;
        for (int s = 0; s < 4; ++s) cb.job[s] = ConvJob{u[s], m->conv[10 + 2 * s], nullptr, v[s]};
        GP_TRY(sparse_conv(ctx, 1, st, cb, 4, tilesC, nC, 1));
        TAKE(y0, float, nC * m->C);
        float *y[4] = {y0, u1, u2, u3};
        for (int s = 0; s < 4; ++s) cb.job[s] = ConvJob{v[s], m->conv[10 + 2 * s + 1], nullptr, y[s]};
        GP_TRY(sparse_conv(ctx, 1, st, cb, 4, tilesC, nC, 0));
        GP_TRY(dbg_mark(ctx, st, 1, pA, (size_t)nP * 128)); GP_TRY(dbg_mark(ctx, st, 2, cA, (size_t)nC * 128));
        for (int s = 0; s < 4; ++s) GP_TRY(dbg_mark(ctx, st, 3 + s, y[s], (size_t)nC * 128));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_side, 0));   // ranks -> posC / slotsC
        GP_TRY(dbg_mark(ctx, st, 7, posC, (size_t)nC * 4)); GP_TRY(dbg_mark(ctx, st, 8, slotsC, (size_t)nC * 4));
        ctx->arena.release_top_low();                       // what is enqueued on st from here on runs behind the rank pass: its temporaries are free
        StageTimer tm_heads(ctx, st, ST_HEADS, (double)nC * 4 * (128 + 1 + 8 + 4));
        for (int s = 0; s < 4; ++s) {
            HeadArgs ha = {}; ha.C = m->C;
            ha.x = y[s]; ha.n = nC; ha.stage_m = STAGE_M[s];
            ha.w1 = m->hw1[s]; ha.b1 = m->hb1[s]; ha.w2 = m->hw2[s]; ha.b2 = m->hb2[s]; ha.frag = m->hfrag[s];
            ha.occ = occC; ha.stage = s; ha.lohi = lohi; ha.mode = 0; ha.pos = posC; ha.slots = slotsC;
            ha.bits = want_bits ? reinterpret_cast<double *>(pairs_dev + MAXLV) : nullptr;   // the 16 slots behind the pair counters
            GP_TRY(head_cdf(st, ha));
        }
    }
    // ---- range coder over every lane of every stream
    const int nstreams = 4 * (L - 1);
    // The reference layout (chunk_log2 = 0: ONE torchac stream per level and stage): the CODER runs on the host (hostcoder.hpp says why: a stream
    // is one dependent chain, torchac is a CPU coder, a host core runs the chain several times faster than a GPU lane) -- the heads' packed coder
    // words come down once, the streams (independent in an encode) are coded on a pool of native threads, the container is put together here.
    // GAUSPCC_V0_DEVICE_CODER=1 (developer knob): the one-lane-per-stream device path below, kept as the cross-check.
    static const bool v0_device = dev_env_int("GAUSPCC_V0_DEVICE_CODER", 0) != 0;
    if (!chunk_log2 && !v0_device) {
        const Level *base = &T.lv[0];
        TAKE(base_xyz, int32_t, 3 * base->n);
        TAKE(base_occ, uint8_t, base->n);
        GP_TRY(level_to_raster(ctx, st, base, T.bias, base_xyz, base_occ));
        const size_t off_w = 0, off_pairs = ((size_t)4 * (size_t)std::max<int64_t>(lohi_words, 1) + 63) & ~(size_t)63;
        const size_t off_bx = off_pairs + 8 * (size_t)NCOUNTERS, off_bo = off_bx + 12 * (size_t)base->n;
        GP_TRY(ctx->hcoder.reserve(off_bo + (size_t)base->n + 64));
        uint8_t *hc = ctx->hcoder.p;
        {
            StageTimer tm(ctx, st, ST_CODER, (double)coded * 4 * 4);
            if (lohi_words) HIP_TRY(hipMemcpyAsync(hc + off_w, lohi, 4 * (size_t)lohi_words, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipMemcpyAsync(hc + off_pairs, pairs_dev, 8 * NCOUNTERS, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hc + off_bx, base_xyz, 12 * (size_t)base->n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hc + off_bo, base_occ, (size_t)base->n, hipMemcpyDeviceToHost, st));
        ht.mark("enc all queued");
        HIP_TRY(hipStreamSynchronize(st));
        ht.mark("enc network done (sync)");
        std::vector<const uint32_t *> sp((size_t)std::max(nstreams, 1));
        std::vector<int64_t> sn((size_t)std::max(nstreams, 1));
        {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(hc + off_w);
            int64_t pre = 0; int si = 0;
            for (int d = 1; d < L; ++d) {
                const int64_t nc = T.lv[d].n;
                for (int s = 0; s < 4; ++s, ++si) { sp[(size_t)si] = w + pre + (int64_t)s * nc; sn[(size_t)si] = nc; }
                pre += 4 * nc;
            }
        }
        std::vector<std::vector<uint8_t>> sb;
        GP_TRY(host_encode_streams(sp.data(), sn.data(), nstreams, &sb, 0));
        ht.mark("enc streams coded (host)");
        size_t fsize = 2 + 4 + 13 * (size_t)base->n + 2;
        for (int si = 0; si < nstreams; ++si) fsize += 4 + sb[(size_t)si].size();
        GP_TRY(ctx->hbytes.reserve(fsize + 16));
        uint8_t *out = ctx->hbytes.p;
        size_t pos = 0;
        out[0] = (uint8_t)posq; out[1] = (uint8_t)(posq >> 8); pos = 2;
        put32(out + pos, (uint32_t)base->n); pos += 4;
        memcpy(out + pos, hc + off_bx, 12 * (size_t)base->n); pos += 12 * (size_t)base->n;
        memcpy(out + pos, hc + off_bo, (size_t)base->n); pos += (size_t)base->n;
        out[pos] = (uint8_t)nstreams; out[pos + 1] = (uint8_t)(nstreams >> 8); pos += 2;
        for (int si = 0; si < nstreams; ++si) {
            const std::vector<uint8_t> &b = sb[(size_t)si];
            put32(out + pos, (uint32_t)b.size()); pos += 4;
            if (!b.empty()) memcpy(out + pos, b.data(), b.size());
            pos += b.size();
        }
        if (pos != fsize) return fail(GPCC_ERR_HIP, "internal: container size mismatch (%zu vs %zu)", pos, fsize);
        if (ctx->prof.on && ctx->prof.stages) ctx->prof.stage_bytes[ST_CODER] += 3.0 * (double)(fsize - (2 + 4 + 13 * (size_t)base->n + 2 + 4 * (size_t)nstreams));
        const unsigned long long *hp = reinterpret_cast<const unsigned long long *>(hc + off_pairs);
        unsigned long long set_pairs[2] = {0, 0};
        for (int d = 0; d < L; ++d) { if (d + 1 < L) set_pairs[0] += hp[d]; if (d) set_pairs[1] += hp[d]; }
        if (ctx->prof.on) GP_TRY(prof_collect(ctx, set_pairs, 2));
        *bytes_out = out; *nbytes_out = (int64_t)pos;
        if (stats) {
            memset(stats, 0, sizeof *stats);
            stats->num_points = n; stats->num_bytes = (int64_t)pos; stats->num_levels = L; stats->coded_nodes = coded;
            for (int d = 0; d < L; ++d) stats->level_nodes[d] = T.lv[d].n;
            stats->conv_pairs = (int64_t)set_pairs[0] * 5 + (int64_t)set_pairs[1] * 13;
            const double *hb = reinterpret_cast<const double *>(hp + MAXLV);
            for (int i = 0; i < 16; ++i) stats->ideal_bits += hb[i];
        }
        return GPCC_OK;
    }
    std::vector<RcChunk> chunks;         // one descriptor per lane
    std::vector<uint32_t> gaps;          // reference layout: container bytes in front of a lane's payload that are not payload (the stream lengths);
                                         // chunked: the stream of every lane (the gaps depend on the byte counts: rc_layout_launch)
    std::vector<uint32_t> stream_first(nstreams + 1, 0);
    uint32_t max_syms = 1;
    size_t table_bound = 0;              // most bytes the chunk tables can take
    {
        int64_t pre = 0; int si = 0;
        uint32_t gap = 0;
        for (int d = 1; d < L; ++d) {
            const int64_t nc = T.lv[d].n;
            const RcPlan pl = plan(nc);
            for (int s = 0; s < 4; ++s, ++si) {
                stream_first[si] = (uint32_t)chunks.size();
                const int64_t base = pre + (int64_t)s * slots(nc);
                gap += 4u;
                table_bound += 6 * (size_t)pl.nchunks + 8;   // escape code: 48 bits a chunk; first count + k
                if (pl.dual && ((base & 1) || pl.llog < 4)) return fail(GPCC_ERR_HIP, "internal: stream %d starts on an odd slot or has lanes below 16 symbols", si);   // k_rc_compact tells a chunk's backwards lane by the parity of RcChunk::first; k_rc_decode_lds stores 16 symbols at a time
                for (uint32_t c = 0; c < pl.nlanes; ++c) {
                    const int64_t cn = pl.lane_syms(nc, c);
                    gaps.push_back(chunk_log2 ? (uint32_t)si : gap);
                    chunks.push_back(RcChunk{(uint32_t)(base + c), pl.nlanes, (uint32_t)cn, 0, 0, 0});
                    max_syms = std::max<uint32_t>(max_syms, (uint32_t)cn);
                }
            }
            pre += 4 * slots(nc);
        }
        stream_first[nstreams] = (uint32_t)chunks.size();
    }
    const int nchunks = (int)chunks.size();
    const Level *base = &T.lv[0];
    TAKE(base_xyz, int32_t, 3 * base->n);
    TAKE(base_occ, uint8_t, base->n);
    GP_TRY(level_to_raster(ctx, st, base, T.bias, base_xyz, base_occ));
    // staging layout (pinned): [lane descs | cnt | pairs | base xyz | base occ | gaps or lane streams | first lane of every stream]
    const size_t off_desc = 0, off_cnt = off_desc + sizeof(RcChunk) * (size_t)std::max(nchunks, 1);
    const size_t off_pairs = off_cnt + 4 * (size_t)std::max(nchunks, 1) + 8, off_bx = off_pairs + 8 * NCOUNTERS, off_bo = off_bx + 12 * (size_t)base->n;
    const size_t off_gap = (off_bo + (size_t)base->n + 63) & ~(size_t)63;
    const size_t off_sf = off_gap + 4 * (size_t)std::max(nchunks, 1);
    GP_TRY(ctx->hstage.reserve(off_sf + 4 * (size_t)(nstreams + 1) + 64));
    const uint32_t gap_bound = 4u * (uint32_t)nstreams + (chunk_log2 ? (uint32_t)table_bound : 0u);
    uint8_t *hs = ctx->hstage.p;
    uint32_t total_payload = 0;
    uint8_t *payload_dev = nullptr;
    bool direct = false;
    const size_t pos0_hdr = (chunk_log2 ? 8 + 4 * (size_t)L + 4 : 2) + 4 + 13 * (size_t)base->n + 2;   // where the streams start
    if (nchunks) {
        memcpy(hs + off_desc, chunks.data(), sizeof(RcChunk) * (size_t)nchunks);
        memcpy(hs + off_gap, gaps.data(), 4 * (size_t)nchunks);
        memcpy(hs + off_sf, stream_first.data(), 4 * (size_t)(nstreams + 1));
        const uint32_t stride = rc_scratch_stride(max_syms);
        TAKE(dchunks, RcChunk, nchunks);
        TAKE(dgap, uint32_t, nchunks);
        HIP_TRY(hipMemcpyAsync(dgap, hs + off_gap, 4 * (size_t)nchunks, hipMemcpyHostToDevice, st));
        TAKE(dcnt, uint32_t, nchunks + 1);
        TAKE(doff, uint32_t, nchunks + 1);
        TAKE(scratch, uint8_t, (size_t)nchunks * stride);
        HIP_TRY(hipMemcpyAsync(dchunks, hs + off_desc, sizeof(RcChunk) * (size_t)nchunks, hipMemcpyHostToDevice, st));
        StageTimer tm(ctx, st, ST_CODER, (double)coded * 4 * 4);   // + 3 x the payload (written, compacted), added when it is known
        GP_TRY(rc_encode_launch(st, lohi, dchunks, nchunks, scratch, stride, dcnt, chunk_log2 ? rc_coder_of_version(CONTAINER_VERSION) : RC_CODER_CARRYLESS));
        GP_TRY(dbg_mark(ctx, st, 10, dcnt, (size_t)nchunks * 4));
        GP_TRY(exclusive_scan_u32(ctx, st, dcnt, doff, nchunks, doff + nchunks));
        HIP_TRY(hipMemcpyAsync(hs + off_cnt, dcnt, 4 * (size_t)nchunks, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hs + off_cnt + 4 * (size_t)nchunks, doff + nchunks, 4, hipMemcpyDeviceToHost, st));
        // worst case payload = all scratch; compact into a buffer of that size -- every lane at its final distance from the
        // first stream's length field -- and copy back only the used part, in one piece
        TAKE(payload, uint8_t, (size_t)nchunks * stride + gap_bound + 32);
        if (chunk_log2) {
            // chunked containers: the device itself moves the compacted payload into the pinned output buffer, at its final
            // place behind the header (known now: it depends on L and the base level only), so the one sync below ends the
            // call.  The buffer is sized for the worst case (every lane at its scratch stride).  The reference layout keeps
            // the copy-after-sync: one lane per stream would make that bound the size of the whole symbol array.
            // The varint tables make the distance between payloads depend on the byte counts: a one-workgroup kernel turns the
            // counts into the gap in front of every lane (the host later fills the gaps with the tables it rebuilds from the counts).
            TAKE(dsf, uint32_t, nstreams + 1);
            TAKE(dlst, uint32_t, nchunks);
            TAKE(dgap_total, uint32_t, 1);
            HIP_TRY(hipMemcpyAsync(dsf, hs + off_sf, 4 * (size_t)(nstreams + 1), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dlst, hs + off_gap, 4 * (size_t)nchunks, hipMemcpyHostToDevice, st));
            GP_TRY(rc_layout_launch(st, dcnt, dsf, nstreams, dlst, nchunks, true, dgap, dgap_total));
            const size_t worst = pos0_hdr + (size_t)nchunks * stride + gap_bound + 64;
            GP_TRY(ctx->hbytes.reserve(worst));
            const uint32_t mis = (uint32_t)(pos0_hdr & 15);   // same 16-byte phase on both sides: whole-word copies
            GP_TRY(rc_compact_launch(st, scratch, stride, dcnt, doff, dgap, nchunks, payload + mis, dchunks));
            GP_TRY(rc_to_host_launch(st, payload, doff + nchunks, mis, dgap_total, ctx->hbytes.p + (pos0_hdr - mis)));
            direct = true;
        } else {
            GP_TRY(rc_compact_launch(st, scratch, stride, dcnt, doff, dgap, nchunks, payload));
        }
        payload_dev = payload;
    }
    HIP_TRY(hipMemcpyAsync(hs + off_pairs, pairs_dev, 8 * NCOUNTERS, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + off_bx, base_xyz, 12 * (size_t)base->n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + off_bo, base_occ, (size_t)base->n, hipMemcpyDeviceToHost, st));
    ht.mark("enc all queued");
    HIP_TRY(hipStreamSynchronize(st));
    ht.mark("enc coded (sync)");
    const uint32_t *hcnt = reinterpret_cast<const uint32_t *>(hs + off_cnt);
    if (nchunks) total_payload = hcnt[nchunks];
    if (ctx->prof.on && ctx->prof.stages) ctx->prof.stage_bytes[ST_CODER] += 3.0 * total_payload;
    // pairs of the two sets (the conv launches are tagged 0 = prior set, 1 = target set)
    unsigned long long set_pairs[2] = {0, 0};
    {
        const unsigned long long *hp = reinterpret_cast<const unsigned long long *>(hs + off_pairs);
        for (int d = 0; d < L; ++d) { if (d + 1 < L) set_pairs[0] += hp[d]; if (d) set_pairs[1] += hp[d]; }
    }
    if (ctx->prof.on) GP_TRY(prof_collect(ctx, set_pairs, 2));
    // ---- container
    // per stream: table bytes and payload bytes, from the lane byte counts
    std::vector<size_t> s_tab((size_t)nstreams, 0), s_pay((size_t)nstreams, 0);
    size_t tables = 0;
    for (int si = 0; si < nstreams; ++si) {
        const int c0 = (int)stream_first[si], c1 = (int)stream_first[si + 1];
        for (int c = c0; c < c1; ++c) s_pay[(size_t)si] += hcnt[c];
        if (chunk_log2) {   // never write a chunk the staged decoder could not hold (rangecoder.hpp: rc_window_fits)
            const int stage_lp[4] = {STAGE_M[0] + 1, STAGE_M[1] + 1, STAGE_M[2] + 1, STAGE_M[3] + 1};
            uint32_t mb = 0;
            for (int c = c0; c < c1; c += 2) mb = std::max(mb, hcnt[c] + (c + 1 < c1 ? hcnt[c + 1] : 0u));
            // (GAUSPCC_TEST_CHUNK_BYTES: a test's stand-in for the window, so that the re-encode with smaller chunks is seen working
            // on an ordinary cloud: tests/test_gpu_parity.py)
            static const uint32_t test_cap = (uint32_t)env_int("GAUSPCC_TEST_CHUNK_BYTES", 0);
            if (!rc_window_fits(stage_lp[si & 3], mb) || (test_cap && mb > test_cap && chunk_log2 > 7)) {
                // (possible only at chunk_log2 >= 13 with a model that spends > 8 bits per 16-ary symbol: gpcc_encode codes the cloud again with smaller chunks)
                (void)fail(GPCC_ERR_ARG, "a chunk of stream %d takes %u bytes, more than the decoder's window holds at chunk_log2 = %d: use a smaller chunk_log2", si, mb, chunk_log2);
                return ENC_RETRY_SMALLER_CHUNKS;
            }
        }
        if (chunk_log2)
            s_tab[(size_t)si] = rc_table_size([&](uint32_t c) { const int l = c0 + 2 * (int)c; return hcnt[l] + (l + 1 < c1 ? hcnt[l + 1] : 0u); }, (uint32_t)((c1 - c0 + 1) / 2));
        tables += s_tab[(size_t)si];
    }
    size_t fsize = pos0_hdr + 4 * (size_t)nstreams + total_payload + tables;
    if (direct) { if (fsize + 16 > ctx->hbytes.cap) return fail(GPCC_ERR_HIP, "internal: payload beyond its bound"); }
    else GP_TRY(ctx->hbytes.reserve(fsize + 16));
    uint8_t *out = ctx->hbytes.p;
    size_t pos = 0;
    if (chunk_log2) {
        out[0] = 0xFF; out[1] = 0xFF; out[2] = (uint8_t)CONTAINER_VERSION; out[3] = (uint8_t)chunk_log2; out[4] = (uint8_t)posq; out[5] = (uint8_t)(posq >> 8); out[6] = (uint8_t)L; out[7] = 0;
        pos = 8;
        for (int d = 0; d < L; ++d) { put32(out + pos, (uint32_t)T.lv[d].n); pos += 4; }
        put32(out + pos, (uint32_t)n); pos += 4;
    } else {
        out[0] = (uint8_t)posq; out[1] = (uint8_t)(posq >> 8); pos = 2;
    }
    put32(out + pos, (uint32_t)base->n); pos += 4;
    memcpy(out + pos, hs + off_bx, 12 * (size_t)base->n); pos += 12 * (size_t)base->n;
    memcpy(out + pos, hs + off_bo, (size_t)base->n); pos += (size_t)base->n;
    out[pos] = (uint8_t)nstreams; out[pos + 1] = (uint8_t)(nstreams >> 8); pos += 2;
    // the payload comes straight from the device into its final place in one copy; the stream lengths and chunk tables
    // are then written into the gaps it left
    {
        const size_t pos0 = pos, body = (size_t)total_payload + 4 * (size_t)nstreams + tables;
        if (pos0 != pos0_hdr) return fail(GPCC_ERR_HIP, "internal: header size mismatch");
        if (body && !direct) { HIP_TRY(hipMemcpyAsync(out + pos0, payload_dev, body, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
        size_t p = pos0;
        for (int si = 0; si < nstreams; ++si) {
            const int c0 = (int)stream_first[si], c1 = (int)stream_first[si + 1];
            put32(out + p, (uint32_t)(s_tab[(size_t)si] + s_pay[(size_t)si])); p += 4;
            if (chunk_log2)
                p += rc_table_put(out + p, [&](uint32_t c) { const int l = c0 + 2 * (int)c; return hcnt[l] + (l + 1 < c1 ? hcnt[l + 1] : 0u); }, (uint32_t)((c1 - c0 + 1) / 2));
            p += s_pay[(size_t)si];
        }
        pos = p;
    }
    ht.mark("enc payload d2h");
    if (pos != fsize) return fail(GPCC_ERR_HIP, "internal: container size mismatch (%zu vs %zu)", pos, fsize);
    *bytes_out = out; *nbytes_out = (int64_t)pos;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->num_points = n; stats->num_bytes = (int64_t)pos; stats->num_levels = L; stats->coded_nodes = coded;
        const unsigned long long *hp = reinterpret_cast<const unsigned long long *>(hs + off_pairs);
        for (int d = 0; d < L; ++d) stats->level_nodes[d] = T.lv[d].n;
        stats->conv_pairs = (int64_t)set_pairs[0] * 5 + (int64_t)set_pairs[1] * 13;   // prior set: 5 convs, target set: 5 + 8
        const double *hb = reinterpret_cast<const double *>(hp + MAXLV);
        for (int i = 0; i < 16; ++i) stats->ideal_bits += hb[i];
    }
    return GPCC_OK;
}

// The words a decode hands back at its one sync -- what every level's occupancy expanded to (cstart[n] of its parent level, left there by
// the expansion's own scan), the leaves' count, the pair counters of the profile, the sticky timeout word of the persistent launches -- gathered
// by ONE single-wave launch straight into pinned memory.  Until round 6 each of them was a runtime blit of its own (per level a 4-byte
// device-to-device copy and a 4-byte device-to-host copy: ~30 __amd_rocclr_copyBuffer launches per decode).
struct DecTail { const uint32_t *tot[MAXLV + 1]; int ntot; const unsigned long long *pairs; const uint32_t *tmo; };
__global__ __launch_bounds__(64) void k_dec_tail(DecTail t, uint32_t *__restrict__ h_tot, unsigned long long *__restrict__ h_pairs, uint32_t *__restrict__ h_tmo)
{
    const int l = (int)threadIdx.x;
    if (l < t.ntot) h_tot[l] = *t.tot[l];
    if (l < MAXLV) h_pairs[l] = t.pairs[l];
    if (l == 0) *h_tmo = t.tmo ? *t.tmo : 0u;
}

// out_user / out_cap: optional caller-owned device buffer for the points (capacity in points); else a context-owned one
int decode_body(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *in, int64_t nbytes, const int32_t **xyz_out, int64_t *n_out,
                uint16_t *posq_out, gpcc_stats *stats, hipStream_t st, int32_t *out_user, int64_t out_cap)
{
    ctx->arena.reset();
    HostTrace ht;

    // header and stream directory (container.hpp: plain C++, the part of a decode that the sanitizer build fuzzes)
    ContainerHdr hdr;
    GP_TRY(container_parse(in, nbytes, &hdr));
    const bool v1 = hdr.chunked;
    const int chunk_log2 = hdr.chunk_log2, version = hdr.version;
    const int L = hdr.L;
    int64_t lvl_n[MAXLV] = {0};
    for (int d = 0; d < MAXLV; ++d) lvl_n[d] = hdr.lvl_n[d];
    const int64_t npts_hdr = hdr.npts, bn = hdr.bn;
    const uint8_t *bxyz = hdr.bxyz, *bocc = hdr.bocc;
    const std::vector<int64_t> &s_len = hdr.s_len;
    *posq_out = hdr.posq;
    // ---- base level -> Morton order on the host (< 64 nodes)
    // internal frame (octree.hpp: Tree::bias): 2^20 when the whole cloud lies inside (-2^20, 2^20), else the base level's own
    // minimum per axis -- in leaf units a multiple of 2^L, which is all the tree needs
    int64_t bias_base[3] = {CB >> L, CB >> L, CB >> L}, bias_leaf[3] = {CB, CB, CB};
    {
        int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        for (int64_t i = 0; i < bn; ++i)
            for (int a = 0; a < 3; ++a) { const int64_t c = (int32_t)get32(bxyz + 12 * i + 4 * a); lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
        bool inside = true;
        for (int a = 0; a < 3; ++a) inside = inside && lo[a] + (int64_t)(CB >> L) >= 0 && hi[a] + (int64_t)(CB >> L) < ((int64_t)1 << (21 - L));
        if (!inside)
            for (int a = 0; a < 3; ++a) { bias_base[a] = -lo[a]; bias_leaf[a] = -lo[a] * ((int64_t)1 << L); }
    }
    struct BN { uint64_t mk, rk; uint8_t occ; };
    std::vector<BN> bnodes((size_t)bn);
    uint32_t mn[3] = {~0u, ~0u, ~0u}, mx[3] = {0, 0, 0};
    for (int64_t i = 0; i < bn; ++i) {
        uint32_t b[3];
        for (int a = 0; a < 3; ++a) {
            const int64_t c = (int32_t)get32(bxyz + 12 * i + 4 * a) + bias_base[a];
            if (c < 0 || c >= ((int64_t)1 << (21 - L))) return fail(GPCC_ERR_FORMAT, "base coordinate out of range");
            b[a] = (uint32_t)c; mn[a] = std::min(mn[a], b[a]); mx[a] = std::max(mx[a], b[a]);
        }
        bnodes[(size_t)i] = BN{morton3(b[0], b[1], b[2]), rkey3(b[0], b[1], b[2]), bocc[i]};
        if (!bocc[i]) return fail(GPCC_ERR_FORMAT, "empty base occupancy");
    }
    std::sort(bnodes.begin(), bnodes.end(), [](const BN &a, const BN &b) { return a.mk < b.mk; });
    for (int64_t i = 1; i < bn; ++i) if (bnodes[(size_t)i].mk == bnodes[(size_t)i - 1].mk) return fail(GPCC_ERR_FORMAT, "duplicate base node");
    int hb = 1;
    for (int a = 0; a < 3; ++a) { int b = 0; uint32_t v = mn[a] ^ mx[a]; while (v) { ++b; v >>= 1; } hb = std::max(hb, b); }

    // ---- device state
    auto alloc_level = [&](Level *lv, int64_t n, int lvl) -> int {
        lv->n = n; lv->lvl = lvl;
        TAKE(rkey, uint64_t, n); TAKE(occ, uint8_t, n); TAKE(cstart, uint32_t, n + 1); TAKE(parent, uint32_t, n); TAKE(m2r, uint32_t, n); TAKE(r2m, uint32_t, n);
        lv->rkey = rkey; lv->occ = occ; lv->cstart = cstart; lv->parent = parent; lv->m2r = m2r; lv->r2m = r2m;
        // the arrays are carved back to back: level_expand_rank zeroes them with ONE memset over this recorded span
        lv->span0 = reinterpret_cast<char *>(rkey); lv->span_bytes = (size_t)(reinterpret_cast<char *>(r2m + n) - reinterpret_cast<char *>(rkey));
        return GPCC_OK;
    };
    // the container goes up on a stream of its own: the first reader is the range decoder of the first coded level, behind a
    // parent trunk, the structure of that level and its own trunk (the copy was the first ~0.2 ms of every decode on st)
    TAKE(dbytes, uint8_t, nbytes + 16);
    GP_TRY(ctx->side_init());
    hipStream_t sd = ctx->side;
    struct SideGuard { hipStream_t s, x; ~SideGuard() { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(x); } } side_guard{sd, ctx->xfer};   // error returns leave nothing in flight
    HIP_TRY(hipMemcpyAsync(dbytes, in, (size_t)nbytes, hipMemcpyHostToDevice, ctx->xfer));
    HIP_TRY(hipEventRecord(ctx->ev_bytes, ctx->xfer));
    // lane descriptors of every level are staged in pinned memory that is written once (no reuse, so no sync before
    // a level's table may be overwritten): chunked containers know all level sizes from the header
    size_t desc_total = 0;
    for (int g = 0; g + 1 < L; ++g) desc_total += 4 * (size_t)rc_plan(v1 ? lvl_n[g + 1] : 0, chunk_log2, version).nlanes;
    const size_t desc_off = (4096 + 16 * (size_t)bn + 63) & ~(size_t)63;
    GP_TRY(ctx->hstage.reserve(desc_off + sizeof(RcChunk) * desc_total + 64));
    RcChunk *hdesc = reinterpret_cast<RcChunk *>(ctx->hstage.p + desc_off);
    size_t desc_used = 0;
    uint32_t win_bytes[MAXLV][4] = {};   // longest byte window of a lane, per (level, stage): decides how the decoder reads it
    // the four lane tables of coded level g + 1 (nc nodes), parsed from the container into the pinned staging area
    auto level_chunks = [&](int g, int64_t nc, size_t *at) -> int {
        const RcPlan pl = rc_plan(nc, chunk_log2, version);
        if (desc_used + (size_t)4 * pl.nlanes > desc_total) return fail(GPCC_ERR_FORMAT, "level %d: chunk tables exceed the header's level sizes", g + 1);
        RcChunk *lanes = hdesc + desc_used;
        *at = desc_used;
        desc_used += (size_t)4 * pl.nlanes;
        GP_TRY(container_level_tables(in, hdr, g, nc, lanes, win_bytes[g]));
        return GPCC_OK;
    };
    // chunked containers know every level's size from the header: all tables go up once, behind the container.  Parsing
    // ~10^4 chunk sizes is ~0.1 ms of host time: it happens once the first level's device work is queued (upload_tables).
    RcChunk *dchunks_all = nullptr;
    size_t desc_at[MAXLV] = {0};
    if (v1 && L > 1) {
        for (int g = 0; g + 1 < L; ++g)
            if (lvl_n[g + 1] <= 0 || lvl_n[g + 1] > 8 * lvl_n[g]) return fail(GPCC_ERR_FORMAT, "bad node count at level %d", g + 1);
        TAKE(dca, RcChunk, std::max<size_t>(desc_total, 1));
        dchunks_all = dca;
    }
    // Two batches (round 4: the host parsed ALL tables between the first level's launches and the second level's, ~0.15 ms with the
    // device idle): the tables of the first TAB_EARLY coded levels -- small levels, a tenth of the chunks -- go up in front of the first
    // coded level; the rest is parsed once that level's chain is queued and goes up behind its own event, which level TAB_EARLY waits for.
    constexpr int TAB_EARLY = 6;
    int tab_parsed = 0;
    auto upload_tables = [&](int g_end, hipEvent_t ev) -> int {
        const size_t from = desc_used;
        for (; tab_parsed < g_end && tab_parsed + 1 < L; ++tab_parsed) GP_TRY(level_chunks(tab_parsed, lvl_n[tab_parsed + 1], &desc_at[tab_parsed]));
        if (desc_used > from) HIP_TRY(hipMemcpyAsync(dchunks_all + from, hdesc + from, sizeof(RcChunk) * (desc_used - from), hipMemcpyHostToDevice, ctx->xfer));
        HIP_TRY(hipEventRecord(ev, ctx->xfer));   // (ev_bytes: replaces the record behind the container alone)
        return GPCC_OK;
    };
    Level cur;
    GP_TRY(alloc_level(&cur, bn, L));
    {
        uint64_t *hr = reinterpret_cast<uint64_t *>(ctx->hstage.p + 1024);
        uint8_t *ho = ctx->hstage.p + 1024 + 8 * (size_t)bn;
        for (int64_t i = 0; i < bn; ++i) { hr[i] = bnodes[(size_t)i].rk; ho[i] = bnodes[(size_t)i].occ; }
        HIP_TRY(hipMemcpyAsync(cur.rkey, hr, 8 * (size_t)bn, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(cur.occ, ho, (size_t)bn, hipMemcpyHostToDevice, st));
        ht.mark("dec parse+h2d queued");   // no sync: nothing else in this call touches bytes 1024 .. 4095 of the staging area
    }
    GP_TRY(level_raster_rank(ctx, st, &cur, hb));
    TAKE(dtotal, uint32_t, 4);
    uint32_t *htotal = reinterpret_cast<uint32_t *>(ctx->hstage.p);
    // a level's tile list is built from its parent's cell map (tiles.hip) and leaves its own behind for the level below;
    // sizes are only known level by level in v0, so everything is carved per level from the arena
    const int NPc = cell_map_entries(m->k);
    const int64_t zero_base[1] = {0};
    TAKE(pairs_dev, unsigned long long, MAXLV);
    HIP_TRY(hipMemsetAsync(pairs_dev, 0, sizeof(unsigned long long) * MAXLV, st));
    ConvTiles tilesP;
    int32_t *cellP = nullptr;
    {
        TAKE(cm, int32_t, (int64_t)NPc * bn);
        cellP = cm;
        const TileLevel tl = {&cur, nullptr, nullptr, cellP};
        const int R = conv_pick_rows(bn, m->k);
        TilePool pool;
        GP_TRY(tiles_build(ctx, st, &tl, 1, m->k, R, conv_pick_height(bn, R), &pool, pairs_dev));
        GP_TRY(tiles_view(ctx, st, pool, 0, 1, zero_base, &tilesP));
    }
    int64_t coded = 0;
    DecTail tail = {};
    // Small levels (fused.hpp): a chunked container's levels of at most FUSE_MAX_NODES nodes get a pair plan instead of a tile list,
    // their chain runs as one persistent launch (plus one for the finished level's prior trunk).  planP: the plan of `cur`.
    const bool fuse_ctx = fused_enabled() && !ctx->fused_off && v1 && version >= 1 && m->C == 32;   // (the persistent kernels are the 32-channel MFMA path)
    const int fmode = fused_mode();
    PairPlan planP;
    int64_t planP_np = 0;          // nodes of the level above planP's (the grid policy's density hint)
    bool any_fused = false;
    // Two streams.  `st` carries the network of a level (18 convolutions, heads, range decoder).  The octree work that only
    // needs the parent level's occupancy -- expansion into the child level, raster ranks, the child's tile list (and cell
    // map) -- runs on the context's side stream beside the parent trunk's five convolutions and fills the idle tails
    // of their launches.  ev_main: the parent level is complete on st; ev_side: the child's structure is ready.
    HIP_TRY(hipEventRecord(ctx->ev_main, st));
    // Arena discipline: a level's child arrays, neighbour map, tile list and chunk table come from the bottom (kept: they are
    // the next level's parent data), its feature buffers from the top (rewound at the end of the level).
    for (int g = 0; g + 1 < L; ++g) {
        const size_t top_mk = ctx->arena.top_mark();
        const int64_t np = cur.n;
        // ---- st: parent trunk
        TAKE_TOP(pF, float, np * m->C); TAKE_TOP(pA, float, np * m->C); TAKE_TOP(pB, float, np * m->C);
        float *Pp = nullptr;
        if (planP.valid()) { TAKE_TOP(pp, float, planP.pcap * 32); Pp = pp; }
        if (planP.valid() && fmode == 1) {
            // (profiling: a persistent launch counts as the convolutions it contains -- 5 here, 13 for a level's chain -- over its whole
            // time, heads / coder phases and barriers included: the conv roofline figure stays conservative)
            ConvRec rec = {0, 0, g, 1, 0, 0, (long long)np, 0, 5, 1};
            if (ctx->prof.on) GP_TRY(prof_event(ctx, st, &rec.e0));
            GP_TRY(fused_parent_trunk(ctx, st, m, planP, planP_np, cur.occ, pF, pA, pB, Pp));
            if (ctx->prof.on) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
            any_fused = true;
        } else {
            { StageTimer tm(ctx, st, ST_ELEM, (double)np * 129); GP_TRY(embed_occ(st, m->prior_emb, cur.occ, np, pF, m->C)); }
            GP_TRY(dbg_mark(ctx, st, g * 100 + 1, pF, (size_t)np * 128));
            GP_TRY(run_trunk(ctx, g, st, m, 0, Trunk{pF, pA, pB}, tilesP, np, planP.valid() ? &planP : nullptr, Pp));
        }
        GP_TRY(dbg_mark(ctx, st, g * 100 + 2, pA, (size_t)np * 128));
        // ---- side: the child level's structure
        HIP_TRY(hipStreamWaitEvent(sd, ctx->ev_main, 0));
        Level chi;
        int64_t nc = lvl_n[g + 1];
        if (!v1) {
            GP_TRY(level_expand(ctx, sd, &cur, nullptr, dtotal));
            HIP_TRY(hipMemcpyAsync(htotal, dtotal, 4, hipMemcpyDeviceToHost, sd));
            HIP_TRY(hipStreamSynchronize(sd));
            nc = htotal[0];
            lvl_n[g + 1] = nc;
        }
        if (nc <= 0 || nc > 8 * cur.n) return fail(GPCC_ERR_FORMAT, "bad node count at level %d", g + 1);
        GP_TRY(alloc_level(&chi, nc, L - g - 1));
        {
            const int hbl = std::min(21, std::max(1, hb + g + 1));
            StageTimer tm(ctx, sd, ST_OCTREE, (double)np * 13 + (double)nc * 12 + (double)cdiv(3 * hbl, 8) * (double)nc * 24 + (double)nc * 8);
            // chunked containers: the count the occupancy expanded to stays where the expansion's scan left it (cur.cstart[np]); every level's
            // word is compared with the header at the final sync (k_dec_tail)
            GP_TRY(level_expand_rank(ctx, sd, &cur, &chi, v1 ? nullptr : dtotal, hb + g + 1));
            tail.tot[1 + g] = cur.cstart + np;
        }
        int32_t *cellC = nullptr;
        if (g + 2 < L) { TAKE(cm, int32_t, (int64_t)NPc * nc); cellC = cm; }      // the last level has no level below it
        // lane descriptors of this level's four streams (chunked containers: all levels parsed and uploaded once, behind the container)
        if (v1 && g == 0) GP_TRY(upload_tables(TAB_EARLY, ctx->ev_bytes));
        const RcPlan pl = rc_plan(nc, chunk_log2, version);
        ConvTiles tilesC;
        PairPlan planC;
        const bool child_plan = fuse_ctx && fused_level_ok(nc, m->k) && pl.dual == (version >= 3) && fused_windows_fit(nc, cur.n, pl.nlanes, win_bytes[g]);
        if (child_plan) {
            StageTimer tm(ctx, sd, ST_TILES, 0.0);
            GP_TRY(pairplan_build(ctx, sd, &cur, cellP, &chi, cellC, m->k, &planC, pairs_dev + g + 1));
        } else {
            const TileLevel tl = {&chi, &cur, cellP, cellC};
            const int R = conv_pick_rows(nc, m->k);
            TilePool pool;
            StageTimer tm(ctx, sd, ST_TILES, 0.0);
            GP_TRY(tiles_build(ctx, sd, &tl, 1, m->k, R, conv_pick_height(nc, R), &pool, pairs_dev + g + 1));
            GP_TRY(tiles_view(ctx, sd, pool, 0, 1, zero_base, &tilesC));
            tm.add_bytes(pool.alg_bytes);
        }
        GP_TRY(dbg_mark(ctx, sd, g * 100 + 3, chi.rkey, (size_t)nc * 8)); GP_TRY(dbg_mark(ctx, sd, g * 100 + 4, chi.parent, (size_t)nc * 4));
        GP_TRY(dbg_mark(ctx, sd, g * 100 + 5, chi.m2r, (size_t)nc * 4));
        if (!child_plan) { GP_TRY(dbg_mark(ctx, sd, g * 100 + 6, tilesC.first, (size_t)(tilesC.nblk + 1) * 4)); GP_TRY(dbg_mark(ctx, sd, g * 100 + 7, tilesC.order, (size_t)tilesC.nblk * 4)); }
        HIP_TRY(hipEventRecord(ctx->ev_side, sd));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_side, 0));
        const int64_t S = chunk_log2 ? (int64_t)1 << pl.llog : nc;   // symbols of a lane
        const int nch = (int)pl.nlanes;
        const int clog = pl.llog;
        const RcChunk *dchunks = nullptr;
        if (v1) {
            dchunks = dchunks_all + desc_at[g];
        } else {
            size_t at = 0;
            GP_TRY(level_chunks(g, nc, &at));
            TAKE(dch, RcChunk, 4 * nch);
            HIP_TRY(hipMemcpyAsync(dch, hdesc + at, sizeof(RcChunk) * 4 * (size_t)nch, hipMemcpyHostToDevice, st));   // pinned, write-once: no sync
            dchunks = dch;
        }
        // ---- st: child trunk and the four stages
        TAKE_TOP(cX, float, nc * m->C); TAKE_TOP(cA, float, nc * m->C); TAKE_TOP(cB, float, nc * m->C); TAKE_TOP(cU, float, nc * m->C);
        float *Pc = nullptr;
        if (child_plan) { TAKE_TOP(pc, float, planC.pcap * 32); Pc = pc; }
        TAKE_TOP(cdf, uint16_t, rc_rows_capacity(nch, S) * 16);  // interleaved rows + the decoder's look-ahead
        uint8_t *sym[4];
        for (int s = 0; s < 4; ++s) { TAKE_TOP(sy, uint8_t, nc + 4); sym[s] = sy; }   // + the last group of four of the last lane
        if (child_plan && fmode == 1) {
            // the level's whole chain in one persistent launch (fused.hip)
            if (g == 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_bytes, 0));
            if (g == TAB_EARLY) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_tables, 0));
            FusedChild fa = {};
            fa.pA = pA; fa.np = np; fa.parent = chi.parent; fa.rkey = chi.rkey; fa.m2r = chi.m2r; fa.bytes = dbytes; fa.chunks = dchunks; fa.nlanes = (uint32_t)nch; fa.llog = clog;
            for (int s = 0; s < 4; ++s) { fa.win_bytes[s] = win_bytes[g][s]; fa.sym[s] = sym[s]; }
            fa.cX = cX; fa.cA = cA; fa.cB = cB; fa.cU = cU; fa.P = Pc; fa.cdf = cdf; fa.occ = chi.occ; fa.coder = rc_coder_of_version(version);
            ConvRec rec = {0, 0, g + 1, 1, 0, 0, (long long)nc, 0, 13, 1};
            if (ctx->prof.on) GP_TRY(prof_event(ctx, st, &rec.e0));
            GP_TRY(fused_child_level(ctx, st, m, planC, fa));
            if (ctx->prof.on) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
            any_fused = true;
        } else {
        { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (128 + 12 + 128)); GP_TRY(child_features(st, pA, chi.parent, chi.rkey, m->temb, nc, cX, m->C)); }
        GP_TRY(dbg_mark(ctx, st, g * 100 + 8, cX, (size_t)nc * 128));
        GP_TRY(run_trunk(ctx, g + 1, st, m, 5, Trunk{cX, cA, cB}, tilesC, nc, child_plan ? &planC : nullptr, Pc));  // -> cA
        GP_TRY(dbg_mark(ctx, st, g * 100 + 9, cA, (size_t)nc * 128));
        for (int s = 0; s < 4; ++s) {
            const float *xin = cA;
            if (s) { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (128 + 4 + s + 128)); GP_TRY(stage_input_dec(st, cA, m->semb[s - 1], sym, chi.m2r, s, nc, cU, m->C)); xin = cU; }
            ConvBatch cb = {}; cb.C = m->C;
            if (child_plan) {
                GP_TRY(plan_conv(st, planC, ConvJob{xin, m->conv[10 + 2 * s], nullptr, cX}, Pc, 1));
                GP_TRY(plan_conv(st, planC, ConvJob{cX, m->conv[10 + 2 * s + 1], nullptr, cB}, Pc, 0));
            } else {
            GP_TRY(conv_chain_begin(ctx, st));
            cb.job[0] = ConvJob{xin, m->conv[10 + 2 * s], nullptr, cX};
            GP_TRY(sparse_conv(ctx, g + 1, st, cb, 1, tilesC, nc, 1));
            cb.job[0] = ConvJob{cX, m->conv[10 + 2 * s + 1], nullptr, cB};
            GP_TRY(sparse_conv(ctx, g + 1, st, cb, 1, tilesC, nc, 0));
            GP_TRY(conv_chain_end(ctx, st));
            }
            GP_TRY(dbg_mark(ctx, st, g * 100 + 10 + 5 * s, xin, (size_t)nc * 128)); GP_TRY(dbg_mark(ctx, st, g * 100 + 11 + 5 * s, cX, (size_t)nc * 128));
            GP_TRY(dbg_mark(ctx, st, g * 100 + 12 + 5 * s, cB, (size_t)nc * 128));
            HeadArgs ha = {}; ha.C = m->C;
            ha.x = cB; ha.n = nc; ha.stage_m = STAGE_M[s];
            ha.w1 = m->hw1[s]; ha.b1 = m->hb1[s]; ha.w2 = m->hw2[s]; ha.b2 = m->hb2[s]; ha.frag = m->hfrag[s];
            ha.m2r = chi.m2r; ha.cdf = cdf; ha.mode = 1; ha.chunk_log2 = clog; ha.nch = (uint32_t)nch;
            const int row_bytes = STAGE_M[s] == 2 ? 2 : STAGE_M[s] == 4 ? 8 : 32;     // compact CDF row
            const size_t cdf_bytes = (size_t)rc_rows_capacity(nch, S) * 16 * 2;
            if (ctx->dbg_on) HIP_TRY(hipMemsetAsync(cdf, 0, cdf_bytes, st));   // developer trace: rows the head does not write read as zeros
            { StageTimer tm(ctx, st, ST_HEADS, (double)nc * (128 + 4 + row_bytes)); GP_TRY(head_cdf(st, ha)); }
            GP_TRY(dbg_mark(ctx, st, g * 100 + 13 + 5 * s, cdf, cdf_bytes));
            if (g == 0 && s == 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_bytes, 0));
            if (g == TAB_EARLY && s == 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_tables, 0));
            static const bool v0_device = dev_env_int("GAUSPCC_V0_DEVICE_CODER", 0) != 0;
            if (!v1 && !v0_device) {
                // the reference layout: the stream is ONE chain of nc symbols -- decoded on the host by the library's torchac coder (hostcoder.hpp):
                // the stage's compact CDF rows come down, the symbols go up; the next stage's input waits for them on the stream as it always did
                const int lp = STAGE_M[s] + 1, rs = rc_row_stride(lp);
                const size_t rows_b = (size_t)nc * (size_t)rs * 2, off_sym = (rows_b + 63) & ~(size_t)63;
                {   // (the reference layout's header carries no level sizes: the block grows with the levels -- geometrically, so that a first decode re-pins it a few times, not once per level)
                    const size_t need = off_sym + (size_t)nc + 64;
                    if (need > ctx->hcoder.cap) GP_TRY(ctx->hcoder.reserve(std::max(need, 4 * ctx->hcoder.cap)));
                }
                StageTimer tm(ctx, st, ST_CODER, (double)nc * (row_bytes + 1) + (double)s_len[4 * g + s]);
                HIP_TRY(hipMemcpyAsync(ctx->hcoder.p, cdf, rows_b, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                GP_TRY(host_decode_compact(reinterpret_cast<const uint16_t *>(ctx->hcoder.p), lp, in + hdr.s_off[(size_t)(4 * g + s)], s_len[(size_t)(4 * g + s)], nc,
                                           ctx->hcoder.p + off_sym));
                HIP_TRY(hipMemcpyAsync(sym[s], ctx->hcoder.p + off_sym, (size_t)nc, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));   // (the pinned block is reused by the next stage)
            } else {
                StageTimer tm(ctx, st, ST_CODER, (double)nc * (row_bytes + 1) + (double)s_len[4 * g + s]);
                GP_TRY(rc_decode_launch(st, cdf, STAGE_M[s] + 1, dbytes, dchunks + (size_t)s * nch, nch, win_bytes[g][s], pl.dual, sym[s], rc_coder_of_version(version)));
            }
            GP_TRY(dbg_mark(ctx, st, g * 100 + 14 + 5 * s, sym[s], (size_t)nc));
        }
        { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (4 + 4 + 1)); GP_TRY(assemble_occ(st, sym, chi.m2r, nc, chi.occ)); }
        }
        GP_TRY(dbg_mark(ctx, st, g * 100 + 30, chi.occ, (size_t)nc));
        HIP_TRY(hipEventRecord(ctx->ev_main, st));
        if (v1 && g == 0) GP_TRY(upload_tables(L, ctx->ev_tables));   // the other levels' tables: parsed while the device runs the first coded level
        ctx->arena.top_rewind(top_mk);
        coded += nc;
        cur = chi; cellP = cellC; tilesP = tilesC; planP = child_plan ? planC : PairPlan(); planP_np = np;
        ht.mark("dec level queued", g + 1, nc);
    }
    // ---- leaves
    GP_TRY(level_expand(ctx, st, &cur, nullptr, v1 ? nullptr : dtotal));
    if (!v1) HIP_TRY(hipMemcpyAsync(htotal, dtotal, 4, hipMemcpyDeviceToHost, st));
    tail.tot[0] = cur.cstart + cur.n;
    // chunked containers carry the point count: the leaves are queued behind the last level without a sync and every count
    // of the header is compared with what the decoded occupancy expanded to at the one sync below (the expansions are
    // bounded by the header's sizes, so a wrong header produced garbage, not out-of-bounds accesses)
    int64_t npts = npts_hdr;
    if (!v1) {
        HIP_TRY(hipStreamSynchronize(st));
        ht.mark("dec levels done (sync)");
        npts = htotal[0];
    }
    int32_t *xyz = out_user;
    if (out_user) {
        if (npts > out_cap) return fail(GPCC_ERR_ARG, "decoded %lld points, the output buffer holds %lld", (long long)npts, (long long)out_cap);
    } else {
        TAKE(own, int32_t, 3 * std::max<int64_t>(npts, 1));
        xyz = own;
    }
    GP_TRY(leaves_reference_order(ctx, st, &cur, bias_leaf, xyz, npts));
    // (pinned -- bytes 512 .. 703 of the staging area, behind htotal's words -- not the stack: an error return between this copy and the
    // sync must not leave a transfer pending into a dead frame)
    unsigned long long *hpairs = reinterpret_cast<unsigned long long *>(ctx->hstage.p + 512);
    static_assert(512 + sizeof(unsigned long long) * MAXLV <= 1024, "hpairs sits in front of the base level's staging bytes");
    htotal[40] = 0;
    if (v1) {
        tail.ntot = L; tail.pairs = pairs_dev; tail.tmo = any_fused ? fused_timeout_word(ctx) : nullptr;
        k_dec_tail<<<1, 64, 0, st>>>(tail, htotal, hpairs, htotal + 40);
        LAUNCH_CHECK();
    } else {
        HIP_TRY(hipMemcpyAsync(hpairs, pairs_dev, sizeof(unsigned long long) * MAXLV, hipMemcpyDeviceToHost, st));
        if (any_fused && fused_timeout_word(ctx)) HIP_TRY(hipMemcpyAsync(htotal + 40, fused_timeout_word(ctx), 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    ht.mark("dec leaves done (sync)");
    if (htotal[40]) {
        // a persistent launch gave up waiting for its workgroups (fused.hip: bounded spins): nothing it produced is trusted
        ctx->fused_off = true;
        GP_TRY(fused_reset(ctx, st));
        return DEC_RETRY_UNFUSED;
    }
    if (v1) {
        for (int g = 0; g + 1 < L; ++g)
            if (htotal[1 + g] != (uint32_t)lvl_n[g + 1])
                return fail(GPCC_ERR_FORMAT, "level %d: header says %lld nodes, occupancy expands to %u", g + 1, (long long)lvl_n[g + 1], htotal[1 + g]);
        if ((int64_t)htotal[0] != npts_hdr) return fail(GPCC_ERR_FORMAT, "decoded %lld points, header says %lld", (long long)htotal[0], (long long)npts_hdr);
    }
    if (ctx->prof.on) GP_TRY(prof_collect(ctx, hpairs, L));
    ht.mark("dec prof collect");
    *xyz_out = xyz; *n_out = npts;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->num_points = npts; stats->num_bytes = nbytes; stats->num_levels = L; stats->coded_nodes = coded;
        int64_t cp = 0;
        for (int d = 0; d < L; ++d) {
            stats->level_nodes[d] = lvl_n[d];
            cp += (int64_t)hpairs[d] * ((d + 1 < L ? 5 : 0) + (d > 0 ? 13 : 0));
        }
        stats->conv_pairs = cp;
    }
    return GPCC_OK;
}

}  // namespace

// developer / test knob: scales the first workspace estimate of both calls, so that the grow-and-retry path (a level of the
// tree, the tile pool, the rank pass on the second stream or a feature buffer running out of arena) can be driven on purpose
static size_t arena_scaled(size_t want)
{
    static const double scale = [] { const char *e = getenv("GAUSPCC_ARENA_SCALE"); const double v = e ? atof(e) : 1.0; return v > 0.0 ? v : 1.0; }();
    return scale == 1.0 ? want : std::max<size_t>((size_t)((double)want * scale), (size_t)1 << 20);
}

extern "C" int gpcc_encode(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *xyz_dev, int64_t n, int chunk_log2, uint16_t posq_f16,
                           const uint8_t **bytes_out, int64_t *nbytes_out, gpcc_stats *stats, void *stream)
{
    if (!ctx || !m || !xyz_dev || !bytes_out || !nbytes_out) return fail(GPCC_ERR_ARG, "null argument");
    if (chunk_log2 != 0 && (chunk_log2 < 6 || chunk_log2 > 14)) return fail(GPCC_ERR_ARG, "chunk_log2 must be 0 or 6..14");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    size_t want = arena_scaled(arena_estimate(n, m->K));
    int rc = GPCC_OK;
    for (int attempt = 0; attempt < 6; ++attempt) {
        GP_TRY(ctx->arena.reserve(want));
        rc = encode_body(ctx, m, xyz_dev, n, chunk_log2, posq_f16, bytes_out, nbytes_out, stats, st);
        if (rc == ENC_RETRY_SMALLER_CHUNKS) {
            // an oversize chunk (16-ary streams at chunk_log2 >= 13 under a high-entropy model): the staged decoder has no path
            // for it, so the limit is part of the format -- code again with half the chunk size instead of failing the call
            HIP_TRY(hipStreamSynchronize(st));
            if (chunk_log2 <= 6) { rc = GPCC_ERR_ARG; break; }
            chunk_log2 -= 1; attempt -= 1;
            continue;
        }
        if (rc != GPCC_ERR_NOMEM) break;
        HIP_TRY(hipStreamSynchronize(st));
        want *= 2;  // deep / very sparse trees: more nodes per point than the estimate
    }
    if (rc == GPCC_OK && stats) stats->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const int de = device_error_check(ctx)) rc = de;   // a device-side fault outranks whatever the garbage it left behind was taken for
    return rc;
}

static int decode_entry(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes, const int32_t **xyz_dev_out,
                        int64_t *n_out, uint16_t *posq_f16_out, gpcc_stats *stats, void *stream, int32_t *out_user, int64_t out_cap)
{
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    // v1 headers carry the node counts; otherwise start from a size-based guess and grow on demand
    size_t want = arena_estimate(std::max<int64_t>(nbytes, 1 << 16), m->K);
    // the cheap consistency checks BEFORE the header sizes the workspace (container.hpp: container_precheck)
    int64_t nodes = 0, nmax = 0, npts = 0;
    const int pre = container_precheck(bytes, nbytes, &nodes, &nmax, &npts);
    if (pre < 0) return pre;
    if (pre > 0) {
        const int L = bytes[6];
        want = (size_t)nmax * 2600 + (size_t)nodes * (size_t)(4 * 125 + m->K * 81 / 16 + 96) + (size_t)npts * 32 + (size_t)nbytes + ((size_t)48 << 20);
        if (fused_enabled()) {   // small levels (fused.hpp): the product buffer n K + 1 rows, the plan and its build scratch
            int64_t nf = 0;
            for (int d = 0; d < L; ++d) { const int64_t v = get32(bytes + 8 + 4 * d); if (fused_level_ok(v, m->k)) nf = std::max(nf, v); }
            want += (size_t)nf * (size_t)m->K * (128 + 10 + 9) + ((size_t)4 << 20);
        }
    }
    want = arena_scaled(want);
    ctx->fused_note_decode();   // (re-arms the persistent small-level path some decodes after a timeout)
    int rc = GPCC_OK;
    const int32_t *xyz = nullptr;
    for (int attempt = 0; attempt < 6; ++attempt) {
        GP_TRY(ctx->arena.reserve(want));
        rc = decode_body(ctx, m, bytes, nbytes, &xyz, n_out, posq_f16_out, stats, st, out_user, out_cap);
        if (rc == DEC_RETRY_UNFUSED) {
            static const bool loud = getenv("GAUSPCC_FUSED_QUIET") == nullptr;
            if (loud) fprintf(stderr, "[gauspcc] a persistent small-level launch timed out on device %d; the launch-per-layer path serves this context's next %d decodes\n", ctx->device, ctx->fused_rearm_after);
            attempt -= 1;
            continue;
        }
        if (rc != GPCC_ERR_NOMEM) break;
        HIP_TRY(hipStreamSynchronize(st));
        want *= 2;
    }
    if (rc == GPCC_OK && xyz_dev_out) *xyz_dev_out = xyz;
    if (rc == GPCC_OK && stats) stats->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const int de = device_error_check(ctx)) rc = de;
    return rc;
}

extern "C" int gpcc_decode(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes, const int32_t **xyz_dev_out,
                           int64_t *n_out, uint16_t *posq_f16_out, gpcc_stats *stats, void *stream)
{
    if (!ctx || !m || !bytes || !xyz_dev_out || !n_out || !posq_f16_out) return fail(GPCC_ERR_ARG, "null argument");
    return decode_entry(ctx, m, bytes, nbytes, xyz_dev_out, n_out, posq_f16_out, stats, stream, nullptr, 0);
}

extern "C" int gpcc_decode_to(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes, int32_t *xyz_dev, int64_t capacity_points,
                              int64_t *n_out, uint16_t *posq_f16_out, gpcc_stats *stats, void *stream)
{
    if (!ctx || !m || !bytes || !xyz_dev || capacity_points < 0 || !n_out || !posq_f16_out) return fail(GPCC_ERR_ARG, "null argument");
    return decode_entry(ctx, m, bytes, nbytes, nullptr, n_out, posq_f16_out, stats, stream, xyz_dev, capacity_points);
}
