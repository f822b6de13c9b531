// codec_batch.hip -- gpcc_encode_batch / gpcc_decode_batch: K scenes through ONE chain of launches.
//
// Reference: the batch column of the codec's coordinates (HAC/utils/pcc_utils.py:73: coords = [b, x, y, z]; kit/op.py:17-30
// sorts by batch last) and the file loop of the stand-alone CLI (GausPcgc/compress_ue_4stage_conv.py:72-75).  A scene is a
// serial chain of ~60 dependent launches per octree level whatever its size (codec.hip), so a 100 k-point scene runs at a
// third and a 10 k-point one at a tenth of the 1 M-point rate.  Here the K scenes are one virtual octree (forest.hpp): every
// convolution, head, coder and scan launch of a depth covers all K scenes, each scene still gets its own container, and every
// container is byte-identical to the scene's solo encode (tests/test_gpu_batch.py) because no node ever sees another scene's
// nodes and the float chains do not depend on how rows are packed into tiles.
//
// Scenes that cannot share a tree (an extent of 2^20 or more, more coordinate budget than 21 bits hold, the reference
// container layout, mixed container versions) are coded one by one by the same entry points.
#include <algorithm>
#include <chrono>
#include <functional>

#include "codec_shared.hpp"
#include "container.hpp"
#include "forest.hpp"
#include "rangecoder_dev.hpp"

using namespace gpcc;

// codec.hip
extern "C" int gpcc_encode(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *xyz_dev, int64_t n, int chunk_log2, uint16_t posq_f16, const uint8_t **bytes_out, int64_t *nbytes_out,
                           gpcc_stats *stats, void *stream);
extern "C" int gpcc_decode_to(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes, int32_t *xyz_dev, int64_t capacity_points, int64_t *n_out,
                              uint16_t *posq_f16_out, gpcc_stats *stats, void *stream);

namespace {

// internal: code the scenes one by one instead (not an error)
constexpr int BATCH_SOLO = -2001;
// internal: a persistent small-level launch timed out (fused.hip) -- decode again on the launch-per-layer path (ctx->fused_off is set)
constexpr int BATCH_RETRY_UNFUSED = -2002;

struct FPosArgs {
    int L;
    uint32_t cbase[MAXLV];
    const uint32_t *m2r[MAXLV];
    const ForestSeg *seg[MAXLV];
    int nseg[MAXLV];
};

// lohi slot of stage 0 and stage stride of every row of the target set: the scene's stream inside its level (codec.hip: k_set_pos)
__global__ __launch_bounds__(256) void k_fset_pos(FPosArgs S, int64_t nC, uint32_t *__restrict__ pos_out, uint32_t *__restrict__ slots_out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nC) return;
    int d = 1;
    for (int q = 2; q < S.L; ++q) d = i >= (int64_t)S.cbase[q] ? q : d;
    const uint32_t r = S.m2r[d][i - S.cbase[d]];
    const ForestSeg *seg = S.seg[d];
    int lo = 0, hi = S.nseg[d];
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg[mid].row0 <= r) lo = mid; else hi = mid; }
    const ForestSeg s = seg[lo];
    pos_out[i] = s.base + rc_interleaved(r - s.row0, s.llog, s.nlanes);
    slots_out[i] = s.slots;
}

// pinned staging of an encode: forest_build's tables, then the per-level scene records
size_t enc_seg_offset(int K) { return (forest_build_pinned(K) + 63) & ~(size_t)63; }
size_t enc_seg_bytes(int K) { return (size_t)MAXLV * (K + 1) * sizeof(ForestSeg); }

int encode_batch_body(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *const *xyz, const int64_t *npts, int K, int chunk_log2, const uint16_t *posq,
                      const uint8_t **bytes_out, int64_t *offsets_out, gpcc_stats *stats, hipStream_t st)
{
    ctx->arena.reset();
    HostTrace ht;
    GP_TRY(ctx->side_init());
    hipStream_t sd = ctx->side;
    struct SideGuard { hipStream_t s; ~SideGuard() { (void)hipStreamSynchronize(s); } } side_guard{sd};
    GP_TRY(ctx->hbatch.reserve(enc_seg_offset(K) + enc_seg_bytes(K) + 64));
    Forest F;
    int bad = -1;
    {
        StageTimer tm(ctx, st, ST_OCTREE, 0.0);
        GP_TRY(forest_build(ctx, st, xyz, npts, K, m->k, &F, &bad));
        tm.add_bytes(tree_alg_bytes(F.T));
    }
    ht.mark("benc forest built");
    Tree &T = F.T;
    const int L = F.L;
    std::vector<int> internal_of((size_t)K);
    for (int qi = 0; qi < K; ++qi) internal_of[(size_t)F.sc[(size_t)qi].user] = qi;
    if (L == 1) GP_TRY(forest_ranks(ctx, st, &F));
    int64_t coded = 0;
    for (int d = 1; d < L; ++d) coded += T.lv[d].n;
    if (coded >= ((int64_t)1 << 30)) return BATCH_SOLO;
    const int CONTAINER_VERSION = ctx->container_version;
    auto plan = [&](int64_t nc) -> RcPlan { return rc_plan(nc, chunk_log2, CONTAINER_VERSION); };
    auto slots = [&](int64_t nc) -> int64_t { const RcPlan p = plan(nc); return (int64_t)p.nlanes << p.llog; };
    // packed symbols: stream (d, scene, s) occupies slots(n of the scene's level d) words; the streams of a (d, scene) are consecutive
    std::vector<ForestSeg> seg[MAXLV];
    int64_t lohi_words = 0;
    for (int d = 0; d < L; ++d) {
        seg[d].resize((size_t)F.Kd[d] + 1);
        for (int qi = 0; qi < F.Kd[d]; ++qi) {
            ForestSeg &s = seg[d][(size_t)qi];
            s = ForestSeg{};
            s.row0 = F.row0[d][(size_t)qi];
            if (d) {
                const int64_t nc = F.sc[(size_t)qi].n[d];
                const RcPlan p = plan(nc);
                s.nlanes = p.nlanes; s.llog = p.llog; s.slots = (uint32_t)slots(nc); s.base = (uint32_t)lohi_words;
                lohi_words += 4 * slots(nc);
            }
        }
        ForestSeg &e = seg[d][(size_t)F.Kd[d]];
        e = ForestSeg{}; e.row0 = (uint32_t)T.lv[d].n;
    }
    if (lohi_words >= ((int64_t)1 << 32)) return BATCH_SOLO;
    GP_TRY(forest_upload_segs(ctx, st, &F, seg, ctx->hbatch.p + enc_seg_offset(K), enc_seg_bytes(K)));
    TAKE(lohi, uint32_t, std::max<int64_t>(lohi_words, 1));
    constexpr int NCOUNTERS = MAXLV + 16;
    TAKE(pairs_dev, unsigned long long, NCOUNTERS);
    HIP_TRY(hipMemsetAsync(pairs_dev, 0, sizeof(unsigned long long) * NCOUNTERS, st));
    int64_t nP = 0;
    for (int d = 0; d + 1 < L; ++d) nP += T.lv[d].n;
    const int64_t nC = coded;
    if (L > 1) {
        int64_t pb[MAXLV] = {0}, cbase[MAXLV] = {0};
        for (int d = 1; d < L; ++d) pb[d] = pb[d - 1] + T.lv[d - 1].n;
        for (int d = 2; d < L; ++d) cbase[d] = cbase[d - 1] + T.lv[d - 1].n;
        TAKE(occP, uint8_t, nP); TAKE(occC, uint8_t, nC); TAKE(rkeyC, uint64_t, nC);
        TAKE(parentC, uint32_t, nC); TAKE(posC, uint32_t, nC); TAKE(slotsC, uint32_t, nC);
        SetLevels S = {};
        FPosArgs PA = {};
        {
            S.L = L; PA.L = L;
            for (int d = 0; d < L; ++d) {
                const Level *lv = &T.lv[d];
                S.n[d] = (uint32_t)lv->n; S.pb[d] = (uint32_t)pb[d]; S.cbase[d] = (uint32_t)cbase[d];
                S.occ[d] = lv->occ; S.rkey[d] = lv->rkey; S.parent[d] = lv->parent; S.m2r[d] = lv->m2r;
                PA.cbase[d] = (uint32_t)cbase[d]; PA.m2r[d] = lv->m2r; PA.seg[d] = F.seg_dev[d]; PA.nseg[d] = F.Kd[d];
            }
            StageTimer tm(ctx, st, ST_ELEM, (double)nP * 2 + (double)nC * (2 + 16 + 8));
            k_set_rows<<<(unsigned)cdiv(std::max(nP, nC), 256), 256, 0, st>>>(S, nP, nC, occP, occC, rkeyC, parentC);
            LAUNCH_CHECK();
        }
        HIP_TRY(hipEventRecord(ctx->ev_main, st));
        // second stream: the raster ranks of every merged level and the coder slots that depend on them (codec.hip: queue_ranks)
        const std::function<int()> queue_ranks = [&]() -> int {
            HIP_TRY(hipStreamWaitEvent(sd, ctx->ev_main, 0));
            ctx->arena.flip = true;
            int rc = GPCC_OK;
            {
                StageTimer tm(ctx, sd, ST_OCTREE, 0.0);
                rc = forest_ranks(ctx, sd, &F);
            }
            ctx->arena.flip = false;
            GP_TRY(rc);
            {
                StageTimer tm(ctx, sd, ST_ELEM, (double)nC * (4 + 8));
                k_fset_pos<<<(unsigned)cdiv(nC, 256), 256, 0, sd>>>(PA, nC, posC, slotsC);
                LAUNCH_CHECK();
            }
            HIP_TRY(hipEventRecord(ctx->ev_side, sd));
            return GPCC_OK;
        };
        // tile lists of every merged level in one pool; the base level hangs under the forest's root level, so it is an
        // ordinary level with a parent (tiles.hip)
        ConvTiles tilesP, tilesC;
        {
            const int NPc = cell_map_entries(m->k);
            TileLevel tl[MAXLV];
            const int32_t *cell_prev = F.cell_root;
            for (int d = 0; d < L; ++d) {
                int32_t *own = nullptr;
                if (d + 1 < L) { TAKE(cm, int32_t, (int64_t)NPc * T.lv[d].n); own = cm; }
                tl[d] = TileLevel{&T.lv[d], d ? &T.lv[d - 1] : &F.root, cell_prev, own};
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
        ht.mark("benc tiles built");
        TAKE(pF, float, nP * m->C); TAKE(pA, float, nP * m->C); TAKE(pB, float, nP * m->C);
        { StageTimer tm(ctx, st, ST_ELEM, (double)nP * 129); GP_TRY(embed_occ(st, m->prior_emb, occP, nP, pF, m->C)); }
        GP_TRY(run_trunk(ctx, 0, st, m, 0, Trunk{pF, pA, pB}, tilesP, nP));
        GP_TRY(queue_ranks());
        TAKE(cX, float, nC * m->C); TAKE(cA, float, nC * m->C); TAKE(cB, float, nC * m->C);
        { StageTimer tm(ctx, st, ST_ELEM, (double)nC * (128 + 12 + 128)); GP_TRY(child_features(st, pA, parentC, rkeyC, m->temb, nC, cX, m->C)); }
        GP_TRY(run_trunk(ctx, 1, st, m, 5, Trunk{cX, cA, cB}, tilesC, nC));
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
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_side, 0));
        ctx->arena.release_top_low();
        StageTimer tm_heads(ctx, st, ST_HEADS, (double)nC * 4 * (128 + 1 + 8 + 4));
        for (int s = 0; s < 4; ++s) {
            HeadArgs ha = {}; ha.C = m->C;
            ha.x = y[s]; ha.n = nC; ha.stage_m = STAGE_M[s];
            ha.w1 = m->hw1[s]; ha.b1 = m->hb1[s]; ha.w2 = m->hw2[s]; ha.b2 = m->hb2[s]; ha.frag = m->hfrag[s];
            ha.occ = occC; ha.stage = s; ha.lohi = lohi; ha.mode = 0; ha.pos = posC; ha.slots = slotsC;
            ha.bits = nullptr;
            GP_TRY(head_cdf(st, ha));
        }
    }
    // ---- range coder over every lane of every stream; streams in container order: scene (the caller's order), level, stage
    int nstreams = 0;
    for (int u = 0; u < K; ++u) nstreams += 4 * (F.sc[(size_t)internal_of[(size_t)u]].L - 1);
    std::vector<RcChunk> chunks;
    std::vector<uint32_t> lane_stream;
    std::vector<uint32_t> stream_first((size_t)nstreams + 1, 0), stream_extra((size_t)std::max(nstreams, 1), 0);
    std::vector<size_t> hdr_bytes((size_t)K), scene_stream0((size_t)K + 1, 0);
    uint32_t max_syms = 1;
    size_t table_bound = 0, hdr_total = 0;
    {
        int si = 0;
        uint32_t pending = 0;   // header bytes of the scenes in front of the next stream
        for (int u = 0; u < K; ++u) {
            const int qi = internal_of[(size_t)u];
            const ForestScene &sc = F.sc[(size_t)qi];
            hdr_bytes[(size_t)u] = 8 + 4 * (size_t)sc.L + 4 + 4 + 13 * (size_t)sc.n[0] + 2;
            hdr_total += hdr_bytes[(size_t)u];
            pending += (uint32_t)hdr_bytes[(size_t)u];
            scene_stream0[(size_t)u] = (size_t)si;
            for (int d = 1; d < sc.L; ++d) {
                const int64_t nc = sc.n[d];
                const RcPlan pl = plan(nc);
                const ForestSeg &sg = seg[d][(size_t)qi];
                for (int s = 0; s < 4; ++s, ++si) {
                    stream_first[(size_t)si] = (uint32_t)chunks.size();
                    stream_extra[(size_t)si] = pending; pending = 0;
                    const int64_t base = (int64_t)sg.base + (int64_t)s * sg.slots;
                    table_bound += 6 * (size_t)pl.nchunks + 8;
                    if (pl.dual && ((base & 1) || pl.llog < 4)) return fail(GPCC_ERR_HIP, "internal: stream %d starts on an odd slot or has lanes below 16 symbols", si);
                    for (uint32_t c = 0; c < pl.nlanes; ++c) {
                        const int64_t cn = pl.lane_syms(nc, c);
                        lane_stream.push_back((uint32_t)si);
                        chunks.push_back(RcChunk{(uint32_t)(base + c), pl.nlanes, (uint32_t)cn, 0, 0, 0});
                        max_syms = std::max<uint32_t>(max_syms, (uint32_t)cn);
                    }
                }
            }
        }
        scene_stream0[(size_t)K] = (size_t)si;
        stream_first[(size_t)nstreams] = (uint32_t)chunks.size();
    }
    const int nchunks = (int)chunks.size();
    const Level *base = &T.lv[0];
    TAKE(base_xyz, int32_t, 3 * base->n);
    TAKE(base_occ, uint8_t, base->n);
    const int64_t zero_bias[3] = {0, 0, 0};
    GP_TRY(level_to_raster(ctx, st, base, zero_bias, base_xyz, base_occ));
    // staging (pinned): [lane descs | cnt + total | pairs | base xyz | base occ | lane streams | first lane of every stream | extra of every stream]
    const size_t off_desc = 0, off_cnt = off_desc + sizeof(RcChunk) * (size_t)std::max(nchunks, 1);
    const size_t off_pairs = off_cnt + 4 * (size_t)std::max(nchunks, 1) + 8, off_bx = off_pairs + 8 * NCOUNTERS, off_bo = off_bx + 12 * (size_t)base->n;
    const size_t off_ls = (off_bo + (size_t)base->n + 63) & ~(size_t)63;
    const size_t off_sf = off_ls + 4 * (size_t)std::max(nchunks, 1);
    const size_t off_ex = off_sf + 4 * (size_t)(nstreams + 1);
    GP_TRY(ctx->hstage.reserve(off_ex + 4 * (size_t)std::max(nstreams, 1) + 64));
    uint8_t *hs = ctx->hstage.p;
    const uint32_t stride = rc_scratch_stride(max_syms);
    const size_t gap_bound = hdr_total + 4 * (size_t)nstreams + table_bound;
    const size_t worst = (size_t)nchunks * stride + gap_bound + 64;
    GP_TRY(ctx->hbytes.reserve(worst + 16));
    if (nchunks) {
        memcpy(hs + off_desc, chunks.data(), sizeof(RcChunk) * (size_t)nchunks);
        memcpy(hs + off_ls, lane_stream.data(), 4 * (size_t)nchunks);
        memcpy(hs + off_sf, stream_first.data(), 4 * (size_t)(nstreams + 1));
        memcpy(hs + off_ex, stream_extra.data(), 4 * (size_t)nstreams);
        TAKE(dchunks, RcChunk, nchunks);
        TAKE(dgap, uint32_t, nchunks);
        TAKE(dcnt, uint32_t, nchunks + 1);
        TAKE(doff, uint32_t, nchunks + 1);
        TAKE(scratch, uint8_t, (size_t)nchunks * stride);
        HIP_TRY(hipMemcpyAsync(dchunks, hs + off_desc, sizeof(RcChunk) * (size_t)nchunks, hipMemcpyHostToDevice, st));
        StageTimer tm(ctx, st, ST_CODER, (double)coded * 4 * 4);
        GP_TRY(rc_encode_launch(st, lohi, dchunks, nchunks, scratch, stride, dcnt, rc_coder_of_version(CONTAINER_VERSION)));
        GP_TRY(exclusive_scan_u32(ctx, st, dcnt, doff, nchunks, doff + nchunks));
        HIP_TRY(hipMemcpyAsync(hs + off_cnt, dcnt, 4 * (size_t)nchunks, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hs + off_cnt + 4 * (size_t)nchunks, doff + nchunks, 4, hipMemcpyDeviceToHost, st));
        TAKE(payload, uint8_t, worst + 32);
        TAKE(dsf, uint32_t, nstreams + 1);
        TAKE(dlst, uint32_t, nchunks);
        TAKE(dex, uint32_t, nstreams);
        TAKE(dssize, uint32_t, nstreams);
        TAKE(dgap_total, uint32_t, 1);
        HIP_TRY(hipMemcpyAsync(dsf, hs + off_sf, 4 * (size_t)(nstreams + 1), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dlst, hs + off_ls, 4 * (size_t)nchunks, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dex, hs + off_ex, 4 * (size_t)nstreams, hipMemcpyHostToDevice, st));
        // the blob of all containers leaves the device in one piece: the payload buffer IS the blob from its first byte, with holes
        // where the host writes headers, stream lengths and chunk tables after the sync
        GP_TRY(rc_layout_many_launch(st, dcnt, dsf, nstreams, dlst, nchunks, true, dex, dssize, dgap, dgap_total));
        GP_TRY(rc_compact_launch(st, scratch, stride, dcnt, doff, dgap, nchunks, payload, dchunks));
        GP_TRY(rc_to_host_launch(st, payload, doff + nchunks, 0, dgap_total, ctx->hbytes.p));
    }
    HIP_TRY(hipMemcpyAsync(hs + off_pairs, pairs_dev, 8 * NCOUNTERS, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + off_bx, base_xyz, 12 * (size_t)base->n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + off_bo, base_occ, (size_t)base->n, hipMemcpyDeviceToHost, st));
    ht.mark("benc all queued");
    HIP_TRY(hipStreamSynchronize(st));
    ht.mark("benc coded (sync)");
    const uint32_t *hcnt = reinterpret_cast<const uint32_t *>(hs + off_cnt);
    unsigned long long set_pairs[2] = {0, 0};
    {
        const unsigned long long *hp = reinterpret_cast<const unsigned long long *>(hs + off_pairs);
        for (int d = 0; d < L; ++d) { if (d + 1 < L) set_pairs[0] += hp[d]; if (d) set_pairs[1] += hp[d]; }
    }
    if (ctx->prof.on) GP_TRY(prof_collect(ctx, set_pairs, 2));
    // ---- containers: header, then per stream its length, chunk table and (already in place) payload
    uint8_t *out = ctx->hbytes.p;
    size_t pos = 0;
    const int32_t *hbx = reinterpret_cast<const int32_t *>(hs + off_bx);
    const uint8_t *hbo = hs + off_bo;
    for (int u = 0; u < K; ++u) {
        const int qi = internal_of[(size_t)u];
        const ForestScene &sc = F.sc[(size_t)qi];
        offsets_out[u] = (int64_t)pos;
        if (pos + hdr_bytes[(size_t)u] > ctx->hbytes.cap) return fail(GPCC_ERR_HIP, "internal: container beyond its bound");
        uint8_t *o = out + pos;
        o[0] = 0xFF; o[1] = 0xFF; o[2] = (uint8_t)CONTAINER_VERSION; o[3] = (uint8_t)chunk_log2; o[4] = (uint8_t)posq[u]; o[5] = (uint8_t)(posq[u] >> 8); o[6] = (uint8_t)sc.L; o[7] = 0;
        size_t p = 8;
        for (int d = 0; d < sc.L; ++d) { put32(o + p, (uint32_t)sc.n[d]); p += 4; }
        put32(o + p, (uint32_t)sc.npts); p += 4;
        put32(o + p, (uint32_t)sc.n[0]); p += 4;
        const uint32_t r0 = F.row0[0][(size_t)qi];
        for (int64_t i = 0; i < sc.n[0]; ++i)
            for (int a = 0; a < 3; ++a) { put32(o + p, (uint32_t)((int64_t)hbx[3 * (r0 + i) + a] - (sc.bias[a] >> sc.L))); p += 4; }
        memcpy(o + p, hbo + r0, (size_t)sc.n[0]); p += (size_t)sc.n[0];
        const int ns = 4 * (sc.L - 1);
        o[p] = (uint8_t)ns; o[p + 1] = (uint8_t)(ns >> 8); p += 2;
        if (p != hdr_bytes[(size_t)u]) return fail(GPCC_ERR_HIP, "internal: header size mismatch");
        pos += p;
        for (size_t si = scene_stream0[(size_t)u]; si < scene_stream0[(size_t)u + 1]; ++si) {
            const int c0 = (int)stream_first[si], c1 = (int)stream_first[si + 1];
            size_t pay = 0;
            uint32_t mb = 0;
            for (int c = c0; c < c1; ++c) pay += hcnt[c];
            for (int c = c0; c < c1; c += 2) mb = std::max(mb, hcnt[c] + (c + 1 < c1 ? hcnt[c + 1] : 0u));
            const int stage_lp = STAGE_M[si & 3] + 1;
            if (!rc_window_fits(stage_lp, mb)) return BATCH_SOLO;   // an oversize chunk: the solo path codes that scene again with smaller chunks
            auto cb = [&](uint32_t c) { const int l = c0 + 2 * (int)c; return hcnt[l] + (l + 1 < c1 ? hcnt[l + 1] : 0u); };
            const uint32_t nch = (uint32_t)((c1 - c0 + 1) / 2);
            const size_t tab = rc_table_size(cb, nch);
            if (pos + 4 + tab + pay > ctx->hbytes.cap) return fail(GPCC_ERR_HIP, "internal: container beyond its bound");
            put32(out + pos, (uint32_t)(tab + pay)); pos += 4;
            pos += rc_table_put(out + pos, cb, nch);
            pos += pay;
        }
        if (stats) {
            gpcc_stats *s = &stats[u];
            memset(s, 0, sizeof *s);
            s->num_points = sc.npts; s->num_bytes = (int64_t)pos - offsets_out[u]; s->num_levels = sc.L;
            for (int d = 0; d < sc.L; ++d) { s->level_nodes[d] = sc.n[d]; if (d) s->coded_nodes += sc.n[d]; }
        }
    }
    offsets_out[K] = (int64_t)pos;
    if (nchunks) {
        const size_t expect = (size_t)hcnt[nchunks] + hdr_total + 4 * (size_t)nstreams;   // + tables: checked through the running position
        if (pos < expect) return fail(GPCC_ERR_HIP, "internal: blob size mismatch (%zu below %zu)", pos, expect);
    }
    if (stats) stats[0].conv_pairs = (int64_t)set_pairs[0] * 5 + (int64_t)set_pairs[1] * 13;   // of the whole batch (a scene's share is not kept)
    *bytes_out = out;
    return GPCC_OK;
}

// ---------------------------------------------------------------------------------------------------------------- decode
// header of a chunked container (container.hpp: the parser gpcc_decode runs); BATCH_SOLO for the reference layout
int parse_chunked(const uint8_t *in, int64_t nbytes, int scene, ContainerHdr *h)
{
    const int rc = container_parse(in, nbytes, h);
    if (rc != GPCC_OK) { char msg[400]; snprintf(msg, sizeof msg, "%s", g_err); return fail(rc, "scene %d: %s", scene, msg); }
    if (!h->chunked) return BATCH_SOLO;
    int64_t nodes = 0;
    for (int d = 0; d < h->L; ++d) nodes += h->lvl_n[d];
    if (nodes > (nbytes << 13)) return fail(GPCC_ERR_FORMAT, "scene %d: header: %lld nodes cannot come from %lld bytes", scene, (long long)nodes, (long long)nbytes);
    return GPCC_OK;
}


// The words a batched decode hands back at its one sync, gathered by ONE single-wave launch straight into pinned memory (codec.hip: k_dec_tail is
// the one-scene form): per level what the merged occupancy expanded to (cstart[rows with children] of the parent level), the moved-boundary flag
// of forest_check_bounds, the leaves of every scene, the pair counters, the sticky timeout word.  (Until round 6: two 4-byte blits per level.)
struct BDecTail { const uint32_t *tot[MAXLV]; int ntot; const uint32_t *bounds_flag, *leaf_cnt; int K; const unsigned long long *pairs; const uint32_t *tmo; };
__global__ __launch_bounds__(64) void k_bdec_tail(BDecTail t, uint32_t *__restrict__ h_leaf, uint32_t *__restrict__ h_lvl, unsigned long long *__restrict__ h_pairs,
                                                  uint32_t *__restrict__ h_tmo)
{
    const int l = (int)threadIdx.x;
    for (int i = l; i < t.K; i += 64) h_leaf[i] = t.leaf_cnt[i];
    if (l < MAXLV) {
        h_lvl[l] = l == MAXLV - 1 ? *t.bounds_flag : l < t.ntot ? *t.tot[l] : 0u;
        h_pairs[l] = t.pairs[l];
    }
    if (l == 0) *h_tmo = t.tmo ? *t.tmo : 0u;
}

int decode_batch_body(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *const *in, const int64_t *nbytes, int K, int32_t *const *xyz_out, const int64_t *cap_out, int64_t *n_out,
                      uint16_t *posq_out, gpcc_stats *stats, hipStream_t st)
{
    ctx->arena.reset();
    HostTrace ht;
    // ---- headers, internal order (deepest scenes first)
    std::vector<ContainerHdr> H((size_t)K);
    int64_t blob = 0;
    for (int u = 0; u < K; ++u) {
        GP_TRY(parse_chunked(in[u], nbytes[u], u, &H[(size_t)u]));
        if (H[(size_t)u].version != H[0].version) return BATCH_SOLO;
        if (cap_out[u] < H[(size_t)u].npts) return fail(GPCC_ERR_ARG, "scene %d: %lld points, the output buffer holds %lld", u, (long long)H[(size_t)u].npts, (long long)cap_out[u]);
        if (H[(size_t)u].L > 20) return BATCH_SOLO;
        blob += (nbytes[u] + 15) & ~(int64_t)15;
    }
    if (blob >= (int64_t)1 << 32) return BATCH_SOLO;
    const int version = H[0].version;
    std::vector<int> order((size_t)K);
    for (int u = 0; u < K; ++u) order[(size_t)u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return H[(size_t)a].L > H[(size_t)b].L; });
    Forest F;
    F.K = K; F.sc.resize((size_t)K);
    const int L = H[(size_t)order[0]].L;
    F.L = L;
    // ---- base levels: every scene in a frame of its own (its base minimum at 0), z slabs one behind the other
    struct BN { uint64_t mk, rk; uint8_t occ; };
    std::vector<BN> bnodes;
    std::vector<uint32_t> bparent;
    std::vector<uint64_t> root_rk;
    std::vector<uint8_t> root_occ;
    std::vector<uint32_t> root_cs;
    F.root0.assign((size_t)K + 1, 0u);
    int64_t zcur = 0, xymax = 0;
    for (int qi = 0; qi < K; ++qi) {
        const int u = order[(size_t)qi];
        const ContainerHdr &h = H[(size_t)u];
        ForestScene &sc = F.sc[(size_t)qi];
        sc.user = u; sc.L = h.L; sc.npts = h.npts;
        for (int d = 0; d < h.L; ++d) sc.n[d] = h.lvl_n[d];
        int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        for (int64_t i = 0; i < h.bn; ++i)
            for (int a = 0; a < 3; ++a) { const int64_t c = (int32_t)get32(h.bxyz + 12 * i + 4 * a); lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
        int64_t tz = 0;
        if (!forest_place(h.L, 0, hi[2] - lo[2], &zcur, &tz)) return BATCH_SOLO;
        for (int a = 0; a < 2; ++a) {
            if (((hi[a] - lo[a] + 1) << (h.L - 1)) > ((int64_t)1 << 21)) return BATCH_SOLO;
            xymax = std::max(xymax, hi[a] - lo[a]);
        }
        const int64_t t3[3] = {-lo[0], -lo[1], -lo[2] + tz};
        for (int a = 0; a < 3; ++a) sc.bias[a] = t3[a] * ((int64_t)1 << h.L);
        const size_t b0 = bnodes.size();
        for (int64_t i = 0; i < h.bn; ++i) {
            uint32_t b[3];
            for (int a = 0; a < 3; ++a) b[a] = (uint32_t)((int64_t)(int32_t)get32(h.bxyz + 12 * i + 4 * a) + t3[a]);
            if (!h.bocc[i]) return fail(GPCC_ERR_FORMAT, "scene %d: empty base occupancy", u);
            bnodes.push_back(BN{morton3(b[0], b[1], b[2]), rkey3(b[0], b[1], b[2]), h.bocc[i]});
        }
        std::sort(bnodes.begin() + (ptrdiff_t)b0, bnodes.end(), [](const BN &a, const BN &b) { return a.mk < b.mk; });
        for (size_t i = b0 + 1; i < bnodes.size(); ++i) if (bnodes[i].mk == bnodes[i - 1].mk) return fail(GPCC_ERR_FORMAT, "scene %d: duplicate base node", u);
        // root level: the would-be parents of the base nodes (siblings are neighbours in Morton order)
        F.root0[(size_t)qi] = (uint32_t)root_rk.size();
        for (size_t i = b0; i < bnodes.size(); ++i) {
            const uint64_t pk = bnodes[i].mk >> 3;
            if (i == b0 || (bnodes[i - 1].mk >> 3) != pk) {
                root_rk.push_back(rkey3(compact1by2(pk), compact1by2(pk >> 1), compact1by2(pk >> 2)));
                root_occ.push_back(0);
                root_cs.push_back((uint32_t)i);
            }
            root_occ.back() |= (uint8_t)(1u << (bnodes[i].mk & 7));
            bparent.push_back((uint32_t)root_rk.size() - 1u);
        }
        sc.nroot = (int64_t)root_rk.size() - F.root0[(size_t)qi];
    }
    F.root0[(size_t)K] = (uint32_t)root_rk.size();
    root_cs.push_back((uint32_t)bnodes.size());
    F.hb0 = 1;
    { int64_t v = std::max(zcur, xymax + 1); while (((int64_t)1 << F.hb0) < v + 1) ++F.hb0; }
    for (int d = 0; d <= L + 1; ++d) F.Kd[d] = 0;
    for (int qi = 0; qi < K; ++qi) for (int d = 0; d < F.sc[(size_t)qi].L; ++d) F.Kd[d] = qi + 1;
    int64_t lvl_n[MAXLV] = {0};
    int64_t total_pts = 0;
    for (int d = 0; d < L; ++d) {
        F.row0[d].assign((size_t)F.Kd[d] + 1, 0u);
        int64_t acc = 0;
        for (int qi = 0; qi < F.Kd[d]; ++qi) { F.row0[d][(size_t)qi] = (uint32_t)acc; acc += F.sc[(size_t)qi].n[d]; }
        if (acc >= (int64_t)1 << 31) return BATCH_SOLO;
        F.row0[d][(size_t)F.Kd[d]] = (uint32_t)acc;
        lvl_n[d] = acc;
    }
    for (int qi = 0; qi < K; ++qi) total_pts += F.sc[(size_t)qi].npts;
    F.npts = total_pts;
    const int64_t bn = lvl_n[0], nroot = (int64_t)root_rk.size();

    // ---- coder geometry of every (level, scene): lanes, CDF row slots, symbol slots
    // sym slots: a scene's symbols start on a multiple of 16 and leave 4 bytes of slack behind them (the staged decoders store
    // 16 / 4 symbols at a time: rangecoder_dev.hpp), so a level's symbol arrays are indexed by a padded rank
    std::vector<ForestSeg> seg[MAXLV];
    uint32_t nch_tot[MAXLV] = {0}, smax[MAXLV] = {0};
    int64_t symlen[MAXLV] = {0};
    std::vector<int> clog2_of((size_t)K);
    for (int qi = 0; qi < K; ++qi) clog2_of[(size_t)qi] = H[(size_t)order[(size_t)qi]].chunk_log2;
    for (int d = 0; d < L; ++d) {
        seg[d].resize((size_t)F.Kd[d] + 1);
        uint32_t lane0 = 0;
        int64_t sym0 = 0;
        for (int qi = 0; qi < F.Kd[d]; ++qi) {
            ForestSeg &s = seg[d][(size_t)qi];
            s = ForestSeg{};
            s.row0 = F.row0[d][(size_t)qi];
            if (d) {
                const int64_t nc = F.sc[(size_t)qi].n[d];
                const RcPlan p = rc_plan(nc, clog2_of[(size_t)qi], version);
                if (p.dual != (version >= 3) || p.llog < 4) return BATCH_SOLO;
                s.lane0 = lane0; s.nlanes = p.nlanes; s.llog = p.llog; s.base = (uint32_t)sym0;
                lane0 += p.nlanes;
                smax[d] = std::max(smax[d], 1u << p.llog);
                sym0 = (sym0 + nc + 4 + 15) & ~(int64_t)15;
            }
        }
        ForestSeg &e = seg[d][(size_t)F.Kd[d]];
        e = ForestSeg{}; e.row0 = (uint32_t)lvl_n[d];
        nch_tot[d] = lane0; symlen[d] = sym0 + 16;
    }
    size_t desc_total = 0;
    for (int d = 1; d < L; ++d) desc_total += 4 * (size_t)nch_tot[d];
    // pinned staging, reserved once: scene records | leaf scene records | base + root arrays | lane descriptors | counts
    const size_t pin_seg = 0, pin_seg_b = (size_t)MAXLV * (K + 1) * sizeof(ForestSeg);
    const size_t pin_leaf = pin_seg + pin_seg_b, pin_leaf_b = (size_t)K * sizeof(ForestLeafScene);
    const size_t pin_base = (pin_leaf + pin_leaf_b + 63) & ~(size_t)63, pin_base_b = (size_t)bn * (8 + 1 + 4) + (size_t)nroot * (8 + 1) + (size_t)(nroot + 1) * 4 + (size_t)(K + 1) * 4 + 64;
    const size_t pin_desc = (pin_base + pin_base_b + 63) & ~(size_t)63, pin_desc_b = sizeof(RcChunk) * std::max<size_t>(desc_total, 1);
    const size_t pin_cnt = (pin_desc + pin_desc_b + 63) & ~(size_t)63, pin_cnt_b = 4 * (size_t)(K + MAXLV * (K + 1) + MAXLV + 64) + 8 * (size_t)MAXLV + 8;   // + the pair counts (8-byte aligned behind the words)
    GP_TRY(ctx->hbatch.reserve(pin_cnt + pin_cnt_b + 64));
    uint8_t *pin = ctx->hbatch.p;
    GP_TRY(ctx->side_init());
    hipStream_t sd = ctx->side;
    struct SideGuard { hipStream_t s, x; ~SideGuard() { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(x); } } side_guard{sd, ctx->xfer};
    // ---- the containers go up on a stream of their own, one behind the other (16-byte aligned)
    TAKE(dbytes, uint8_t, blob + 16);
    std::vector<int64_t> byte0((size_t)K);
    {
        int64_t at = 0;
        for (int u = 0; u < K; ++u) {
            byte0[(size_t)u] = at;
            HIP_TRY(hipMemcpyAsync(dbytes + at, in[u], (size_t)nbytes[u], hipMemcpyHostToDevice, ctx->xfer));
            at += (nbytes[u] + 15) & ~(int64_t)15;
        }
    }
    // ---- lane tables of every level: [stage][lanes of the level, scene after scene].  Parsed in two batches like the solo decoder's
    // (codec.hip: upload_tables): the tables of the first TAB_EARLY coded levels -- small levels, few lanes -- go up behind the containers;
    // the rest is parsed once the first level's device work is queued (0.4 ms of host time at 32 scenes that the device no longer waits for)
    RcChunk *hdesc = reinterpret_cast<RcChunk *>(pin + pin_desc);
    size_t desc_at[MAXLV + 1] = {0};
    uint32_t win_bytes[MAXLV][4] = {};
    for (int d = 1; d < L; ++d) desc_at[d + 1] = desc_at[d] + 4 * (size_t)nch_tot[d];
    TAKE(dchunks_all, RcChunk, std::max<size_t>(desc_total, 1));
    constexpr int TAB_EARLY = 5;
    auto parse_levels = [&](int d0, int d1, hipEvent_t ev) -> int {
        std::vector<RcChunk> tmp;
        for (int d = d0; d < d1; ++d) {
            const size_t at = desc_at[d];
            for (int qi = 0; qi < F.Kd[d]; ++qi) {
                const int u = order[(size_t)qi];
                const ContainerHdr &h = H[(size_t)u];
                const int64_t nc = F.sc[(size_t)qi].n[d];
                const RcPlan pl = rc_plan(nc, h.chunk_log2, version);
                const ForestSeg &sg = seg[d][(size_t)qi];
                tmp.resize(pl.nlanes);
                for (int s = 0; s < 4; ++s) {
                    const int si = 4 * (d - 1) + s;
                    uint32_t wb = 0;
                    const char *err = rc_parse_table(in[u] + h.s_off[(size_t)si], h.s_off[(size_t)si], h.s_len[(size_t)si], pl, nc, version, tmp.data(), &wb);
                    if (err) return fail(GPCC_ERR_FORMAT, "scene %d stream %d: %s", u, si, err);
                    win_bytes[d][s] = std::max(win_bytes[d][s], wb);
                    RcChunk *dst = hdesc + at + (size_t)s * nch_tot[d] + sg.lane0;
                    for (uint32_t l = 0; l < pl.nlanes; ++l) {
                        RcChunk c = tmp[l];
                        c.first = sg.lane0 + l; c.stride = nch_tot[d];
                        c.out = sg.base + (l << pl.llog);
                        c.byte_off += (uint32_t)byte0[(size_t)u];
                        dst[l] = c;
                    }
                }
            }
        }
        if (d1 > d0 && desc_at[d1] > desc_at[d0])
            HIP_TRY(hipMemcpyAsync(dchunks_all + desc_at[d0], hdesc + desc_at[d0], sizeof(RcChunk) * (desc_at[d1] - desc_at[d0]), hipMemcpyHostToDevice, ctx->xfer));
        HIP_TRY(hipEventRecord(ev, ctx->xfer));
        return GPCC_OK;
    };
    const int d_early = std::min(L, 1 + TAB_EARLY);
    GP_TRY(forest_upload_segs(ctx, st, &F, seg, pin + pin_seg, pin_seg_b));
    // ---- base and root levels
    auto alloc_level = [&](Level *lv, int64_t n, int lvl) -> int {
        lv->n = n; lv->lvl = lvl;
        TAKE(rkey, uint64_t, n); TAKE(occ, uint8_t, n); TAKE(cstart, uint32_t, n + 1); TAKE(parent, uint32_t, n); TAKE(m2r, uint32_t, n); TAKE(r2m, uint32_t, n);
        lv->rkey = rkey; lv->occ = occ; lv->cstart = cstart; lv->parent = parent; lv->m2r = m2r; lv->r2m = r2m;
        lv->span0 = reinterpret_cast<char *>(rkey); lv->span_bytes = (size_t)(reinterpret_cast<char *>(r2m + n) - reinterpret_cast<char *>(rkey));
        return GPCC_OK;
    };
    Tree &T = F.T;
    T.L = L; T.npts = total_pts; T.hb = std::min(21, F.hb0 + L);
    Level &cur0 = T.lv[0];
    GP_TRY(alloc_level(&cur0, bn, L));
    GP_TRY(alloc_level(&F.root, nroot, L + 1));
    TAKE(droot0, uint32_t, K + 1);
    {
        uint8_t *p = pin + pin_base;
        uint64_t *h_rk = reinterpret_cast<uint64_t *>(p); p += 8 * (size_t)bn;
        uint64_t *h_rrk = reinterpret_cast<uint64_t *>(p); p += 8 * (size_t)nroot;
        uint32_t *h_par = reinterpret_cast<uint32_t *>(p); p += 4 * (size_t)bn;
        uint32_t *h_rcs = reinterpret_cast<uint32_t *>(p); p += 4 * (size_t)(nroot + 1);
        uint32_t *h_r0 = reinterpret_cast<uint32_t *>(p); p += 4 * (size_t)(K + 1);
        uint8_t *h_occ = p; p += (size_t)bn;
        uint8_t *h_rocc = p;
        for (int64_t i = 0; i < bn; ++i) { h_rk[i] = bnodes[(size_t)i].rk; h_occ[i] = bnodes[(size_t)i].occ; h_par[i] = bparent[(size_t)i]; }
        for (int64_t i = 0; i < nroot; ++i) { h_rrk[i] = root_rk[(size_t)i]; h_rocc[i] = root_occ[(size_t)i]; }
        for (int64_t i = 0; i <= nroot; ++i) h_rcs[i] = root_cs[(size_t)i];
        for (int qi = 0; qi <= K; ++qi) h_r0[qi] = F.root0[(size_t)qi];
        HIP_TRY(hipMemcpyAsync(cur0.rkey, h_rk, 8 * (size_t)bn, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(cur0.occ, h_occ, (size_t)bn, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(cur0.parent, h_par, 4 * (size_t)bn, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(F.root.rkey, h_rrk, 8 * (size_t)nroot, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(F.root.occ, h_rocc, (size_t)nroot, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(F.root.cstart, h_rcs, 4 * (size_t)(nroot + 1), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(droot0, h_r0, 4 * (size_t)(K + 1), hipMemcpyHostToDevice, st));
    }
    const int NPc = cell_map_entries(m->k);
    TAKE(cell_root, int32_t, (int64_t)NPc * nroot);
    F.cell_root = cell_root;
    GP_TRY(forest_root_cells(ctx, st, &F.root, droot0, K, m->k, cell_root));
    GP_TRY(level_raster_rank(ctx, st, &cur0, F.hb0));
    ht.mark("bdec parse+h2d queued");
    TAKE(dtotal, uint32_t, 4);
    TAKE(dlevel_tot, uint32_t, MAXLV);   // nodes every level expanded to; the last word: a scene boundary that moved (forest_check_bounds)
    HIP_TRY(hipMemsetAsync(dlevel_tot, 0, 4 * (size_t)MAXLV, st));
    const int64_t zero_base[1] = {0};
    TAKE(pairs_dev, unsigned long long, MAXLV);
    HIP_TRY(hipMemsetAsync(pairs_dev, 0, sizeof(unsigned long long) * MAXLV, st));
    ConvTiles tilesP;
    int32_t *cellP = nullptr;
    {
        TAKE(cm, int32_t, (int64_t)NPc * bn);
        cellP = cm;
        const TileLevel tl = {&cur0, &F.root, cell_root, cellP};
        const int R = conv_pick_rows(bn, m->k);
        TilePool pool;
        GP_TRY(tiles_build(ctx, st, &tl, 1, m->k, R, conv_pick_height(bn, R), &pool, pairs_dev));
        GP_TRY(tiles_view(ctx, st, pool, 0, 1, zero_base, &tilesP));
    }
    // small levels (fused.hpp): a merged level of at most FUSE_MAX_NODES nodes gets a pair plan, its chain is one persistent launch
    const bool fuse_ctx = fused_enabled() && !ctx->fused_off && m->C == 32;
    const int fmode = fused_mode();
    PairPlan planP;
    int64_t planP_np = 0;
    bool any_fused = false;
    BDecTail tail = {};
    HIP_TRY(hipEventRecord(ctx->ev_main, st));
    // the early lane tables: parsed HERE, with the base level's launches (root cells, raster ranks, tile list) already queued -- until round 6 the
    // host parsed them before the first launch, with the device idle (their first reader is the host itself: win_bytes decides the first
    // coded level's class a few lines down; on the device the first range-decoder phase, behind ev_bytes = containers AND early tables)
    GP_TRY(parse_levels(1, d_early, ctx->ev_bytes));
    for (int g = 0; g + 1 < L; ++g) {
        const size_t top_mk = ctx->arena.top_mark();
        Level &cur = T.lv[g];
        const int64_t np = cur.n;
        const int64_t np_in = F.inner(g);   // rows with children in level g + 1: the scenes that go on
        // ---- st: parent trunk (only the rows that have children need it: a prefix -- but a block of the tile list may straddle
        // the boundary, so the trunk runs on the whole level; the rows of finished scenes are a level's tail and cost their share)
        TAKE_TOP(pF, float, np * m->C); TAKE_TOP(pA, float, np * m->C); TAKE_TOP(pB, float, np * m->C);
        float *Pp = nullptr;
        if (planP.valid()) { TAKE_TOP(pp, float, planP.pcap * 32); Pp = pp; }
        if (planP.valid() && fmode == 1) {
            ConvRec rec = {0, 0, g, 1, 0, 0, (long long)np, 0, 5, 1};
            if (ctx->prof.on) GP_TRY(prof_event(ctx, st, &rec.e0));
            GP_TRY(fused_parent_trunk(ctx, st, m, planP, planP_np, cur.occ, pF, pA, pB, Pp));
            if (ctx->prof.on) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
            any_fused = true;
        } else {
            { StageTimer tm(ctx, st, ST_ELEM, (double)np * 129); GP_TRY(embed_occ(st, m->prior_emb, cur.occ, np, pF, m->C)); }
            GP_TRY(run_trunk(ctx, g, st, m, 0, Trunk{pF, pA, pB}, tilesP, np, planP.valid() ? &planP : nullptr, Pp));
        }
        // ---- side: the child level's structure
        HIP_TRY(hipStreamWaitEvent(sd, ctx->ev_main, 0));
        Level &chi = T.lv[g + 1];
        const int64_t nc = lvl_n[g + 1];
        GP_TRY(alloc_level(&chi, nc, L - g - 1));
        {
            Level pv = cur; pv.n = np_in;
            StageTimer tm(ctx, sd, ST_OCTREE, (double)np * 13 + (double)nc * 12 + (double)nc * 8);
            GP_TRY(level_expand_rank(ctx, sd, &pv, &chi, nullptr, F.hb0 + g + 1));
            tail.tot[g] = cur.cstart + np_in;   // what the occupancy expanded to: left there by the expansion's scan, gathered at the one sync (k_bdec_tail)
            GP_TRY(forest_check_bounds(sd, cur.cstart, F.seg_dev[g], F.seg_dev[g + 1], F.Kd[g + 1], dlevel_tot + MAXLV - 1));
        }
        int32_t *cellC = nullptr;
        if (g + 2 < L) { TAKE(cm, int32_t, (int64_t)NPc * nc); cellC = cm; }
        ConvTiles tilesC;
        PairPlan planC;
        const bool child_plan = fuse_ctx && fused_level_ok(nc, m->k) && fused_windows_fit(nc, cur.n, nch_tot[g + 1], win_bytes[g + 1]);
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
        // CDF row slot and symbol slot of every node (scene records of level g + 1)
        TAKE(cpos, uint32_t, nc);
        TAKE(spos, uint32_t, nc);
        GP_TRY(forest_cdf_pos(sd, F.seg_dev[g + 1], F.Kd[g + 1], chi.m2r, nc, nch_tot[g + 1], cpos, spos));
        HIP_TRY(hipEventRecord(ctx->ev_side, sd));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_side, 0));
        const int64_t S = smax[g + 1];
        const int nch = (int)nch_tot[g + 1];
        const RcChunk *dchunks = dchunks_all + desc_at[g + 1];
        // ---- st: child trunk and the four stages
        TAKE_TOP(cX, float, nc * m->C); TAKE_TOP(cA, float, nc * m->C); TAKE_TOP(cB, float, nc * m->C); TAKE_TOP(cU, float, nc * m->C);
        float *Pc = nullptr;
        if (child_plan) { TAKE_TOP(pc, float, planC.pcap * 32); Pc = pc; }
        TAKE_TOP(cdf, uint16_t, rc_rows_capacity(nch, S) * 16);
        uint8_t *sym[4];
        for (int s = 0; s < 4; ++s) { TAKE_TOP(sy, uint8_t, symlen[g + 1]); sym[s] = sy; }
        if (g + 1 == d_early && d_early < L) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_tables, 0));   // the first level whose tables went up in the second batch
        if (child_plan && fmode == 1) {
            // the level's whole chain in one persistent launch (fused.hip)
            if (g == 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_bytes, 0));
            FusedChild fa = {};
            fa.pA = pA; fa.np = np; fa.parent = chi.parent; fa.rkey = chi.rkey; fa.m2r = chi.m2r; fa.bytes = dbytes; fa.chunks = dchunks; fa.nlanes = (uint32_t)nch; fa.llog = 4;
            fa.cpos = cpos; fa.spos = spos;
            for (int s = 0; s < 4; ++s) { fa.win_bytes[s] = win_bytes[g + 1][s]; fa.sym[s] = sym[s]; }
            fa.cX = cX; fa.cA = cA; fa.cB = cB; fa.cU = cU; fa.P = Pc; fa.cdf = cdf; fa.occ = chi.occ; fa.coder = rc_coder_of_version(version);
            ConvRec rec = {0, 0, g + 1, 1, 0, 0, (long long)nc, 0, 13, 1};
            if (ctx->prof.on) GP_TRY(prof_event(ctx, st, &rec.e0));
            GP_TRY(fused_child_level(ctx, st, m, planC, fa));
            if (ctx->prof.on) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
            any_fused = true;
        } else {
        { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (128 + 12 + 128)); GP_TRY(child_features(st, pA, chi.parent, chi.rkey, m->temb, nc, cX, m->C)); }
        GP_TRY(run_trunk(ctx, g + 1, st, m, 5, Trunk{cX, cA, cB}, tilesC, nc, child_plan ? &planC : nullptr, Pc));
        for (int s = 0; s < 4; ++s) {
            const float *xin = cA;
            if (s) { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (128 + 4 + s + 128)); GP_TRY(stage_input_dec(st, cA, m->semb[s - 1], sym, spos, s, nc, cU, m->C)); xin = cU; }
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
            HeadArgs ha = {}; ha.C = m->C;
            ha.x = cB; ha.n = nc; ha.stage_m = STAGE_M[s];
            ha.w1 = m->hw1[s]; ha.b1 = m->hb1[s]; ha.w2 = m->hw2[s]; ha.b2 = m->hb2[s]; ha.frag = m->hfrag[s];
            ha.m2r = chi.m2r; ha.cdf = cdf; ha.mode = 1; ha.pos = cpos;
            const int row_bytes = STAGE_M[s] == 2 ? 2 : STAGE_M[s] == 4 ? 8 : 32;
            { StageTimer tm(ctx, st, ST_HEADS, (double)nc * (128 + 4 + row_bytes)); GP_TRY(head_cdf(st, ha)); }
            if (g == 0 && s == 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_bytes, 0));
            {
                StageTimer tm(ctx, st, ST_CODER, (double)nc * (row_bytes + 1));
                GP_TRY(rc_decode_launch(st, cdf, STAGE_M[s] + 1, dbytes, dchunks + (size_t)s * nch, nch, win_bytes[g + 1][s], version >= 3, sym[s], rc_coder_of_version(version)));
            }
        }
        { StageTimer tm(ctx, st, ST_ELEM, (double)nc * (4 + 4 + 1)); GP_TRY(assemble_occ(st, sym, spos, nc, chi.occ)); }
        }
        HIP_TRY(hipEventRecord(ctx->ev_main, st));
        if (g == 0 && d_early < L) GP_TRY(parse_levels(d_early, L, ctx->ev_tables));   // the other levels' tables: parsed while the device runs the first coded level
        ctx->arena.top_rewind(top_mk);
        cellP = cellC; tilesP = tilesC; planP = child_plan ? planC : PairPlan(); planP_np = np;
        ht.mark("bdec level queued", g + 1, nc);
    }
    // ---- leaves: every level hands out the points of the scenes that end there
    ForestLeafScene *hleaf = reinterpret_cast<ForestLeafScene *>(pin + pin_leaf);
    TAKE(dleaf, ForestLeafScene, K);
    TAKE(dleaf_cnt, uint32_t, K);
    HIP_TRY(hipMemsetAsync(dleaf_cnt, 0, 4 * (size_t)K, st));
    for (int qi = 0; qi < K; ++qi) {
        const ForestScene &sc = F.sc[(size_t)qi];
        ForestLeafScene &ls = hleaf[qi];
        ls.rank0 = F.row0[sc.L - 1][(size_t)qi];
        ls.xyz = xyz_out[sc.user]; ls.cap = std::min<int64_t>(cap_out[sc.user], sc.npts);
        for (int a = 0; a < 3; ++a) ls.bias[a] = sc.bias[a];
    }
    HIP_TRY(hipMemcpyAsync(dleaf, hleaf, sizeof(ForestLeafScene) * (size_t)K, hipMemcpyHostToDevice, st));
    for (int d = 0; d < L; ++d) {
        const int q0 = d + 1 < L ? F.Kd[d + 1] : 0, q1 = F.Kd[d];   // the scenes whose last level is d
        if (q1 <= q0) continue;
        GP_TRY(forest_leaves(ctx, st, &T.lv[d], (int64_t)F.row0[d][(size_t)q0], dleaf + q0, q1 - q0, dleaf_cnt + q0));
    }
    // ---- one sync: every count of every header against what the decoded occupancy expanded to
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(pin + pin_cnt);
    uint32_t *h_leaf = hcnt, *h_lvl = hcnt + K;

    // (pinned, not the stack: an error return between this copy and the sync must not leave a transfer pending into a dead frame)
    unsigned long long *hpairs = reinterpret_cast<unsigned long long *>(pin + ((pin_cnt + 4 * (size_t)(K + MAXLV * (K + 1) + MAXLV + 64) + 7) & ~(size_t)7));
    uint32_t *h_tmo = hcnt + K + MAXLV;
    *h_tmo = 0;
    tail.ntot = L - 1; tail.bounds_flag = dlevel_tot + MAXLV - 1; tail.leaf_cnt = dleaf_cnt; tail.K = K; tail.pairs = pairs_dev;
    tail.tmo = any_fused ? fused_timeout_word(ctx) : nullptr;
    k_bdec_tail<<<1, 64, 0, st>>>(tail, h_leaf, h_lvl, hpairs, h_tmo);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    ht.mark("bdec leaves done (sync)");
    if (*h_tmo) {
        // a persistent launch gave up waiting for its workgroups (fused.hip: bounded spins): nothing it produced is trusted
        ctx->fused_off = true;
        GP_TRY(fused_reset(ctx, st));
        return BATCH_RETRY_UNFUSED;
    }
    for (int g = 0; g + 1 < L; ++g)
        if (h_lvl[g] != (uint32_t)lvl_n[g + 1]) return fail(GPCC_ERR_FORMAT, "level %d: the headers say %lld nodes, the occupancy expands to %u", g + 1, (long long)lvl_n[g + 1], h_lvl[g]);
    if (h_lvl[MAXLV - 1]) return fail(GPCC_ERR_FORMAT, "a scene's occupancy expands to a node count its header does not state");
    for (int qi = 0; qi < K; ++qi)
        if ((int64_t)h_leaf[qi] != F.sc[(size_t)qi].npts) return fail(GPCC_ERR_FORMAT, "scene %d: decoded %u points, header says %lld", F.sc[(size_t)qi].user, h_leaf[qi], (long long)F.sc[(size_t)qi].npts);
    if (ctx->prof.on) GP_TRY(prof_collect(ctx, hpairs, L));
    for (int u = 0; u < K; ++u) {
        const ContainerHdr &h = H[(size_t)u];
        n_out[u] = h.npts; posq_out[u] = h.posq;
        if (stats) {
            gpcc_stats *s = &stats[u];
            memset(s, 0, sizeof *s);
            s->num_points = h.npts; s->num_bytes = nbytes[u]; s->num_levels = h.L;
            for (int d = 0; d < h.L; ++d) { s->level_nodes[d] = h.lvl_n[d]; if (d) s->coded_nodes += h.lvl_n[d]; }
        }
    }
    if (stats) {
        int64_t cp = 0;
        for (int d = 0; d < L; ++d) cp += (int64_t)hpairs[d] * ((d + 1 < L ? 5 : 0) + (d > 0 ? 13 : 0));
        stats[0].conv_pairs = cp;
    }
    return GPCC_OK;
}

size_t arena_scaled_b(size_t want)
{
    static const double scale = [] { const char *e = getenv("GAUSPCC_ARENA_SCALE"); const double v = e ? atof(e) : 1.0; return v > 0.0 ? v : 1.0; }();
    return scale == 1.0 ? want : std::max<size_t>((size_t)((double)want * scale), (size_t)1 << 20);
}

}  // namespace

extern "C" int gpcc_encode_batch(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *const *xyz_dev, const int64_t *n, int nscenes, int chunk_log2, const uint16_t *posq_f16,
                                 const uint8_t **bytes_out, int64_t *offsets_out, gpcc_stats *stats, int *batched_out, void *stream)
{
    if (!ctx || !m || !xyz_dev || !n || !posq_f16 || !bytes_out || !offsets_out || nscenes < 1) return fail(GPCC_ERR_ARG, "null argument");
    if (chunk_log2 != 0 && (chunk_log2 < 6 || chunk_log2 > 14)) return fail(GPCC_ERR_ARG, "chunk_log2 must be 0 or 6..14");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = BATCH_SOLO;
    if (batched_out) *batched_out = 0;
    static const bool never = env_int("GAUSPCC_BATCH", 1) == 0;   // cross-check knob: every scene on the solo path
    if (chunk_log2 != 0 && nscenes <= FOREST_MAX_SCENES && !never) {
        int64_t total = 0;
        for (int q = 0; q < nscenes; ++q) total += n[q] > 0 ? n[q] : 0;
        size_t want = arena_scaled_b(arena_estimate(total, m->K) + (size_t)total * 40 + (size_t)nscenes * 65536);
        for (int attempt = 0; attempt < 6; ++attempt) {
            GP_TRY(ctx->arena.reserve(want));
            rc = encode_batch_body(ctx, m, xyz_dev, n, nscenes, chunk_log2, posq_f16, bytes_out, offsets_out, stats, st);
            if (rc != GPCC_ERR_NOMEM) break;
            HIP_TRY(hipStreamSynchronize(st));
            want *= 2;
        }
        if (rc == GPCC_OK && batched_out) *batched_out = 1;
        if (rc == FOREST_UNFIT) rc = BATCH_SOLO;
    }
    if (rc == BATCH_SOLO) {
        // one by one; the containers are collected in a buffer of the library's (the context's output buffer is reused by every call)
        HIP_TRY(hipStreamSynchronize(st));
        static thread_local std::vector<uint8_t> blob;
        blob.clear();
        for (int q = 0; q < nscenes; ++q) {
            const uint8_t *b = nullptr; int64_t nb = 0;
            gpcc_stats s1; memset(&s1, 0, sizeof s1);
            const int r1 = gpcc_encode(ctx, m, xyz_dev[q], n[q], chunk_log2, posq_f16[q], &b, &nb, &s1, stream);
            if (r1 != GPCC_OK) {
                char msg[400]; snprintf(msg, sizeof msg, "%s", g_err);
                return fail(r1, "scene %d: %s", q, msg);
            }
            offsets_out[q] = (int64_t)blob.size();
            blob.insert(blob.end(), b, b + nb);
            if (stats) stats[q] = s1;
        }
        offsets_out[nscenes] = (int64_t)blob.size();
        GP_TRY(ctx->hbytes.reserve(blob.size() + 16));
        memcpy(ctx->hbytes.p, blob.data(), blob.size());
        *bytes_out = ctx->hbytes.p;
        rc = GPCC_OK;
    }
    if (rc == GPCC_OK && stats) stats[0].device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const int de = device_error_check(ctx)) rc = de;
    return rc;
}

extern "C" int gpcc_decode_batch(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *const *bytes, const int64_t *nbytes, int nscenes, int32_t *const *xyz_dev,
                                 const int64_t *capacity_points, int64_t *n_out, uint16_t *posq_f16_out, gpcc_stats *stats, int *batched_out, void *stream)
{
    if (!ctx || !m || !bytes || !nbytes || !xyz_dev || !capacity_points || !n_out || !posq_f16_out || nscenes < 1) return fail(GPCC_ERR_ARG, "null argument");
    for (int q = 0; q < nscenes; ++q) if (!bytes[q] || !xyz_dev[q] || capacity_points[q] < 0 || nbytes[q] < 0) return fail(GPCC_ERR_ARG, "scene %d: null argument", q);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = BATCH_SOLO;
    if (batched_out) *batched_out = 0;
    static const bool never = env_int("GAUSPCC_BATCH", 1) == 0;
    if (nscenes <= FOREST_MAX_SCENES && !never) {
        // workspace from the headers (verified inside decode_batch_body before anything is sized from them: parse_chunked)
        int64_t nodes = 0, nmax_sum[MAXLV] = {0}, total_bytes = 0, npts = 0;
        bool ok = true;
        for (int q = 0; q < nscenes && ok; ++q) {
            const uint8_t *b = bytes[q];
            total_bytes += nbytes[q];
            if (!(nbytes[q] >= 8 && b[0] == 0xFF && b[1] == 0xFF && b[6] >= 1 && b[6] <= 21 && nbytes[q] >= 12 + 4 * (int64_t)b[6])) { ok = false; break; }
            const int L = b[6];
            int64_t prev = 0, mine = 0;
            for (int d = 0; d < L; ++d) {
                const int64_t v = get32(b + 8 + 4 * d);
                if (v <= 0 || (d == 0 ? v >= 64 : v > 8 * prev)) { ok = false; break; }
                mine += v; nmax_sum[d] += v; prev = v;
            }
            if (!ok) break;
            const int64_t np = get32(b + 8 + 4 * L);
            if (np < 1 || np > 8 * prev || mine > (nbytes[q] << 13)) { ok = false; break; }
            nodes += mine; npts += np;
        }
        if (ok) {
            ctx->fused_note_decode();
            int64_t nmax = 0;
            for (int d = 0; d < MAXLV; ++d) nmax = std::max(nmax, nmax_sum[d]);
            size_t want = arena_scaled_b((size_t)nmax * 2700 + (size_t)nodes * (size_t)(4 * 125 + m->K * 81 / 16 + 96) + (size_t)npts * 32 + (size_t)total_bytes + (size_t)nscenes * 65536 + ((size_t)48 << 20));
            if (fused_enabled()) {   // small levels (fused.hpp): the product buffer n K + 1 rows, the plan and its build scratch
                int64_t nf = 0;
                for (int d = 0; d < MAXLV; ++d) if (nmax_sum[d] > 0 && fused_level_ok(nmax_sum[d], m->k)) nf = std::max(nf, nmax_sum[d]);
                want += (size_t)nf * (size_t)m->K * (128 + 10 + 9) + ((size_t)4 << 20);
            }
            for (int attempt = 0; attempt < 6; ++attempt) {
                GP_TRY(ctx->arena.reserve(want));
                rc = decode_batch_body(ctx, m, bytes, nbytes, nscenes, xyz_dev, capacity_points, n_out, posq_f16_out, stats, st);
                if (rc == BATCH_RETRY_UNFUSED) {
                    static const bool loud = getenv("GAUSPCC_FUSED_QUIET") == nullptr;
                    if (loud) fprintf(stderr, "[gauspcc] a persistent small-level launch timed out on device %d; the launch-per-layer path serves this context's next %d decodes\n", ctx->device, ctx->fused_rearm_after);
                    attempt -= 1;
                    continue;
                }
                if (rc != GPCC_ERR_NOMEM) break;
                HIP_TRY(hipStreamSynchronize(st));
                want *= 2;
            }
            if (rc == GPCC_OK && batched_out) *batched_out = 1;
        }
        // (a header that fails the cheap checks: the solo path reports it with the scene's own message)
    }
    if (rc == BATCH_SOLO) {
        HIP_TRY(hipStreamSynchronize(st));
        for (int q = 0; q < nscenes; ++q) {
            gpcc_stats s1; memset(&s1, 0, sizeof s1);
            const int r1 = gpcc_decode_to(ctx, m, bytes[q], nbytes[q], xyz_dev[q], capacity_points[q], &n_out[q], &posq_f16_out[q], &s1, stream);
            if (r1 != GPCC_OK) {
                char msg[400]; snprintf(msg, sizeof msg, "%s", g_err);
                return fail(r1, "scene %d: %s", q, msg);
            }
            if (stats) stats[q] = s1;
        }
        rc = GPCC_OK;
    }
    if (rc == GPCC_OK && stats) stats[0].device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const int de = device_error_check(ctx)) rc = de;
    return rc;
}
