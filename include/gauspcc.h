/*
 * gauspcc.h -- C ABI of libgauspcc.so, the MI355X (gfx950) implementation of the
 * GausPcc hot path.  Plain pointers and sizes only; every device pointer is a HIP
 * device address on the context's device, every `stream` is a hipStream_t passed
 * as void* (NULL = the legacy default stream).  gpcc_encode / gpcc_decode also use a second stream the
 * context owns (event-fenced against `stream`); a context serves one call at a time.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  All functions return 0 on success or a negative
 * gpcc_status; gpcc_last_error() gives the thread-local message.  The Python shim
 * (gauspcc_amd/) raises on non-zero, mirroring the reference's Python exceptions.
 */
#ifndef GAUSPCC_H
#define GAUSPCC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPCC_API __attribute__((visibility("default")))

typedef enum {
    GPCC_OK = 0,
    GPCC_ERR_HIP = -1,        /* a HIP runtime call failed */
    GPCC_ERR_ARG = -2,        /* bad argument */
    GPCC_ERR_RANGE = -3,      /* the cloud leaves (-2^20, 2^20) AND its extent is 2^20 or more (any int32 position is fine) */
    GPCC_ERR_DUPLICATE = -4,  /* duplicate point (the reference silently corrupts occupancy here) */
    GPCC_ERR_FORMAT = -5,     /* malformed / truncated bitstream */
    GPCC_ERR_NOMEM = -6
} gpcc_status;

typedef enum { GPCC_F32 = 0, GPCC_F64 = 1, GPCC_I32 = 2, GPCC_I64 = 3 } gpcc_dtype;

typedef struct gpcc_ctx gpcc_ctx;     /* per-device context: workspace arena, staging buffers */
typedef struct gpcc_model gpcc_model; /* GausPcgc weights resident in HBM, MFMA-friendly layout */

GPCC_API const char *gpcc_last_error(void);
GPCC_API int gpcc_version(void);

GPCC_API int gpcc_ctx_create(int device, gpcc_ctx **out);
GPCC_API void gpcc_ctx_destroy(gpcc_ctx *ctx);
/* Chunked containers (chunk_log2 != 0) are this library's own layout (the versions' history: HISTORY.md section 5): version 4 (default) runs a
 * carry-propagating range coder in the lanes of a stream -- same 16-bit CDF rows and rate as torchac's coder, a third of the
 * decoder's dependent chain --, version 3 torchac's coder (arithmetic_kernel.cu:94-163) itself.  Sets what gpcc_encode and
 * gpcc_rc_encode / gpcc_rc_decode use; gpcc_decode reads versions 0 (the reference layout, pcc_utils.py:198-203) to 4. */
GPCC_API int gpcc_ctx_set_container_version(gpcc_ctx *ctx, int version);
/* bytes the context holds now: device (workspace arena, product buffer of the small levels) and pinned host staging.  The
 * reference's CLIs print torch.cuda.max_memory_allocated() (compress_ue_4stage_conv.py:283); this library allocates its
 * workspace itself, so its CLIs add this figure.  Either pointer may be null. */
GPCC_API int gpcc_ctx_bytes(const gpcc_ctx *ctx, int64_t *device_bytes, int64_t *pinned_bytes);

/* developer counter: kernel launches the calling thread has enqueued through this library since the last reset (reset != 0
 * zeroes it); memcpy / memset nodes are not counted.  bench.py reports kernels per decode with it. */
GPCC_API long long gpcc_debug_launches(int reset);

/* Device-side faults that must not abort the process (today: a device-wide scan whose tiles made no progress for ~10 s) raise a sticky
 * word in the context instead of trapping.  Every entry point that synchronises checks it and returns GPCC_ERR_HIP once (the word and
 * the scan state are reset: the next call starts clean); callers of the stage-level entry points that return before their work has run
 * call this after synchronising the stream.  GPCC_OK when nothing was raised. */
GPCC_API int gpcc_device_error_check(gpcc_ctx *ctx);

/* ---- a2  calculate_morton_order            src/gs_compress/HAC/utils/pcc_utils.py:12-22
 * perm_out[N] (int64, device): argsort of x + y*M + z*M^2 after the per-axis min shift
 * (M = max over all axes + 1), i.e. the (z,y,x) raster order; stable for equal keys. */
GPCC_API int gpcc_raster_order(gpcc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n,
                      int64_t *perm_out_dev, void *stream);

/* ---- a1  voxelise (caller side)     src/ai_pcc/GausPcgc/compress_ue_4stage_conv.py:89-94: `xyz / 0.001 + 131072` (when the
 * data is not pre-quantised) then `torch.round(xyz / posQ).int()`; HAC/scene/gaussian_model.py:1107: `round(anchor / voxel)`.
 * out[i] = int32(rint(((x[i] / d1) + add) / d2)) over the 3n coordinates, evaluated in the array's OWN dtype (GPCC_F32 or
 * GPCC_F64) one correctly rounded operation at a time, as the reference's numpy / torch-CPU expressions are; flags bit 0
 * enables the `/ d1 + add` step, bit 1 the `/ d2` step.  xyz_dev (n,3) and out_dev (n,3) int32 are device pointers. */
#define GPCC_VOX_SCALE 1
#define GPCC_VOX_POSQ 2
GPCC_API int gpcc_voxelise(gpcc_ctx *ctx, const void *xyz_dev, int dtype, int64_t n, double d1, double add, double d2, int flags,
                  int32_t *out_dev, void *stream);

/* ---- Network(channels, kernel_size).load_state_dict     pcc_utils.py:65-67, 266-268
 * tensors: GPCC_T_COUNT host pointers to contiguous float32 arrays, upstream layouts
 * (conv kernels (k^3, Cin, Cout); Linear (out, in)); order = gpcc_tensor_id. */
typedef enum {
    GPCC_T_PRIOR_EMB = 0,  /* prior_embedding.weight (256, C) */
    GPCC_T_CONV0 = 1,      /* 18 conv kernels: prior_resnet.{0,2.conv0,2.conv1,3.conv0,3.conv1},
                              target_resnet.{same five}, spatial_conv_s{0..3}.{0,2} */
    GPCC_T_TEMB = 19,      /* target_embedding.target_res_embedding.weight (8, C) */
    GPCC_T_HW1 = 20,       /* pred_head_s{0..3}.0.weight (C, C) */
    GPCC_T_HB1 = 24,       /* pred_head_s{0..3}.0.bias   (C)    */
    GPCC_T_HW2 = 28,       /* pred_head_s{0..3}.2.weight ({2,2,4,16}, C) */
    GPCC_T_HB2 = 32,       /* pred_head_s{0..3}.2.bias   ({2,2,4,16})    */
    GPCC_T_SEMB = 36,      /* pred_head_s{1,2,3}_emb.weight ({2,4,16}, C) */
    GPCC_T_COUNT = 39
} gpcc_tensor_id;

GPCC_API int gpcc_model_create(gpcc_ctx *ctx, int channels, int kernel_size, const float *const *tensors,
                      gpcc_model **out);
GPCC_API void gpcc_model_destroy(gpcc_model *m);

#define GPCC_STATS_IDEAL_BITS 1
typedef struct {
    int64_t num_points;
    int64_t num_bytes;
    int32_t num_levels;        /* stored levels L (base + coded) */
    int32_t flags;             /* IN: GPCC_STATS_IDEAL_BITS asks gpcc_encode for ideal_bits (costs ~4 % of an encode); OUT: 0 */
    int64_t level_nodes[24];   /* nodes per stored level, base first */
    int64_t coded_nodes;       /* sum of nodes over coded levels */
    int64_t conv_pairs;        /* sum over all 18*levels convs of (output node, present neighbour) pairs */
    double device_ms;          /* wall time between the syncs that bracket the call */
    double ideal_bits;         /* encode only, when requested through flags: sum over coded symbols of clamp(-log2(p_gt + 1e-10), 0, 50), the reference's
                                  bpp estimator before the division by N (network_ue_4stage_conv.py:100-182, a14) */
} gpcc_stats;

/* ---- a12  compress_point_cloud (the timed span :78-189 + container :192-203)
 * xyz_dev: (N,3) int32 device, duplicate-free, any order.  chunk_log2 = 0 writes the
 * reference container layout (one torchac range-coder stream per level and stage: a stream is ONE
 * dependent chain, so for this layout the coder -- not the network -- runs on the host, on the library's own
 * torchac coder, csrc/hostcoder.hpp); 6..14 writes the chunked container whose streams are cut into
 * chunks of at most 2^chunk_log2 symbols that decode in parallel.  A chunk must fit the staged decoder's LDS
 * window (64 KiB; 16 KiB for the 16-ary streams): should one come out larger -- possible only at chunk_log2 >= 13 with
 * a model that spends more than 8 bits per 16-ary symbol -- the cloud is coded again with chunk_log2 - 1 (the header
 * records the value used; readers never see the request).  posq_f16 = bits of np.float16(posQ).
 * On success *bytes_out points at a context-owned host buffer valid until the next call. */
GPCC_API int gpcc_encode(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *xyz_dev, int64_t n,
                int chunk_log2, uint16_t posq_f16, const uint8_t **bytes_out, int64_t *nbytes_out,
                gpcc_stats *stats, void *stream);

/* ---- a13  decompress_point_cloud (the timed span :279-385)
 * bytes: the whole .bin file (host).  On success *xyz_dev_out points at a context-owned
 * (N,3) int32 DEVICE buffer (integer leaf coordinates, before the posQ multiply) in the
 * reference's output order (FCG of the raster-sorted last level, pcc_utils.py:375),
 * valid until the next call on this context. */
GPCC_API int gpcc_decode(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes,
                const int32_t **xyz_dev_out, int64_t *n_out, uint16_t *posq_f16_out,
                gpcc_stats *stats, void *stream);

/* The same into a caller-owned DEVICE buffer of capacity_points x 3 int32 (the chunked containers carry the point count
 * in their header -- u32 at byte 8 + 4 L, L = byte 6 -- so a caller can size it before the call); fails with
 * GPCC_ERR_ARG when the cloud does not fit.  Saves the copy out of the context's buffer. */
GPCC_API int gpcc_decode_to(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *bytes, int64_t nbytes,
                int32_t *xyz_dev, int64_t capacity_points, int64_t *n_out, uint16_t *posq_f16_out,
                gpcc_stats *stats, void *stream);

/* ---- batched a12 / a13: K scenes through ONE chain of launches
 * Reference: the codec's coordinate tensor has a batch column -- coords = [b, x, y, z], src/gs_compress/HAC/utils/pcc_utils.py:73
 * (b pinned to 0), sort_CF orders by batch last (src/ai_pcc/GausPcgc/kit/op.py:17-30) -- and the stand-alone CLI loops over files
 * (src/ai_pcc/GausPcgc/compress_ue_4stage_conv.py:72-75).  Every scene gets its own container, byte-identical to what gpcc_encode
 * writes for it alone; the scenes share every convolution / head / coder / scan launch of an octree depth (csrc/forest.hpp).
 * xyz_dev / n / posq_f16: HOST arrays of nscenes device pointers, point counts and posQ bits.  On success *bytes_out points at a
 * context-owned host buffer holding the containers one after the other, scene i at [offsets_out[i], offsets_out[i + 1])
 * (offsets_out: nscenes + 1 entries, caller-owned).  stats (nullable): nscenes records; conv_pairs and device_ms of the whole
 * batch are in stats[0].  *batched_out (nullable): 1 when the scenes shared one tree, 0 when they were coded one by one (the
 * reference container layout chunk_log2 = 0, a scene whose extent exceeds 2^18 voxels, more than 256 scenes, or more scenes than the
 * 21-bit coordinate frame stacks: same bytes either way). */
GPCC_API int gpcc_encode_batch(gpcc_ctx *ctx, const gpcc_model *m, const int32_t *const *xyz_dev, const int64_t *n, int nscenes,
                int chunk_log2, const uint16_t *posq_f16, const uint8_t **bytes_out, int64_t *offsets_out,
                gpcc_stats *stats, int *batched_out, void *stream);
/* bytes / nbytes: HOST arrays of nscenes container pointers (host memory) and sizes; xyz_dev / capacity_points: nscenes caller-owned
 * DEVICE buffers ((capacity, 3) int32 each; a chunked container's header carries its point count, see gpcc_decode_to).  Every
 * scene's points come out as gpcc_decode gives them for that container alone.  Containers of the reference layout or of mixed
 * versions are decoded one by one (*batched_out = 0). */
GPCC_API int gpcc_decode_batch(gpcc_ctx *ctx, const gpcc_model *m, const uint8_t *const *bytes, const int64_t *nbytes, int nscenes,
                int32_t *const *xyz_dev, const int64_t *capacity_points, int64_t *n_out, uint16_t *posq_f16_out,
                gpcc_stats *stats, int *batched_out, void *stream);

/* Live timing of the dominant kernel (the sparse convolution): while enabled, every launch is
 * bracketed by HIP events on the stream it runs on.  conv_pair_jobs = sum over launches of
 * (output node, present neighbour) pairs x jobs in the launch; algorithmic flops = 2*C*C*conv_pair_jobs. */
typedef struct {
    double conv_ms;
    int64_t conv_launches;
    int64_t conv_pair_jobs;
    /* the decoder's persistent small-level launches (csrc/fused.hpp), kept apart from the k_sparse_conv family above: their time
     * covers whole chains of layers (heads, range-decoder phases and grid barriers included); fused_launches counts the
     * convolutions they contain (13 per level chain, 5 per prior trunk), fused_pair_jobs their (pair, convolution) products */
    double fused_ms;
    int64_t fused_launches;
    int64_t fused_pair_jobs;
} gpcc_profile;
GPCC_API int gpcc_profile_enable(gpcc_ctx *ctx, int on);   /* 0 off, 1 the convolution, 2 also the stages below (these reset the accumulators); 3 = stop recording, keep what was collected */
GPCC_API int gpcc_profile_get(gpcc_ctx *ctx, gpcc_profile *out);

/* The HBM-bound stages of gpcc_encode / gpcc_decode (SURVEY.md 8d "which roofline"), bracketed by HIP events on the stream
 * their kernels run on while gpcc_profile_enable(ctx, 2) is in force: ms = bracketed time, bytes = the ALGORITHMIC traffic of
 * the bracketed work (an ideal layer-by-layer implementation's reads + writes; formulas in HISTORY.md section 4). */
typedef struct {
    char name[48];
    double ms;
    double bytes;
    int64_t brackets;
    double critical_ms;   /* of ms: the part during which no convolution bracket was open on any stream of the call -- the octree / tile-list
                           * work of a decode runs beside the parent trunk's convolutions, the encoder's rank pass beside its first trunk */
} gpcc_stage;
GPCC_API int gpcc_profile_stages(gpcc_ctx *ctx, gpcc_stage *out, int cap, int *n_out);

/* Developer trace (no reference counterpart): while enabled, gpcc_decode records a checksum of every intermediate buffer
 * of every level (features, level structure, symbols) on the stream that produced it; _get returns the (tag, sum) pairs of
 * the decode that just returned (tag = 100 x level + buffer id, codec.hip) and clears the list.  Two runs of one container
 * must give identical lists: the first difference names the stage that misbehaved (tools/inflight_check.py). */
GPCC_API int gpcc_debug_trace_enable(gpcc_ctx *ctx, int on);
GPCC_API int gpcc_debug_trace_get(gpcc_ctx *ctx, int *tags, unsigned long long *sums, int cap);
/* keep a device copy of every marked buffer whose tag % 100 == tag_mod (-1: none); _get copies the last copy of `tag` out and
 * returns its size in bytes (-1: no such copy) */
GPCC_API int gpcc_debug_capture(gpcc_ctx *ctx, int tag_mod);
GPCC_API long long gpcc_debug_capture_get(gpcc_ctx *ctx, int tag, void *host, long long cap);

/* The library's exclusive prefix sum of uint32 values (every level build, rank derivation, sort pass and tile list runs on it: no
 * reference counterpart, torch.cumsum / unique do this job upstream).  in / out device arrays of n values (in == out allowed), in2 /
 * out2 optional: a second independent scan of the same length in the same launch where the size allows; total_dev (nullable)
 * receives the sum of `in`.  Exported for the unit tests of the scan kernels (tests/test_gpu_primitives.py). */
GPCC_API int gpcc_debug_exclusive_scan(gpcc_ctx *ctx, const uint32_t *in_dev, uint32_t *out_dev, const uint32_t *in2_dev, uint32_t *out2_dev, int64_t n,
                                       uint32_t *total_dev, void *stream);

/* Write n host buffers to n files on `threads` native threads (<= 0: 8).  The attribute loops of HAC / HAC++ produce one `.b` file per
 * 3 000-anchor slice and attribute (HAC/scene/gaussian_model.py:1176-1213: 1 002 files per million anchors; HAC++: 2 338) -- in Python that is
 * ~55 us of interpreter time per file under the GIL; here it is open / write / close.  Returns GPCC_OK or GPCC_ERR_ARG with the first
 * path that failed in gpcc_last_error().  No context, no GPU call. */
GPCC_API int gpcc_write_files(const char *const *paths, const uint8_t *const *data, const int64_t *sizes, int n, int threads);

/* The read side: n whole files into one blob, file i at offsets_out[i] .. offsets_out[i + 1] (offsets_out has n + 1 entries), read by `threads`
 * native threads (<= 0: 8).  The decoders of the attribute loops read 1 002 (HAC) / 2 343 (HAC++) slice files per million anchors: 14 / 27 ms of
 * interpreter time in Python.  *blob_out belongs to the calling thread and is valid until its next call.  GPCC_OK or GPCC_ERR_ARG with the first
 * file that could not be read. */
GPCC_API int gpcc_read_files(const char *const *paths, int n, int threads, const uint8_t **blob_out, int64_t *offsets_out);

/* Copy out of a context-owned device buffer (e.g. gpcc_decode's points) into caller memory,
 * ordered on `stream`; returns after the copy has completed. */
GPCC_API int gpcc_memcpy_d2d(gpcc_ctx *ctx, void *dst_dev, const void *src_dev, int64_t nbytes, void *stream);

/* ---- stage-level entry points (what the reference does through torch / torchsparse /
 * torchac calls); used by the parity tests and by callers that want a single stage. */

/* op.sort_CF order of integer coordinates             src/ai_pcc/GausPcgc/kit/op.py:17-30
 * xyz (n,3) int32 device -> perm (n) uint32 device such that xyz[perm] is (z,y,x) sorted. */
GPCC_API int gpcc_sort_zyx(gpcc_ctx *ctx, const int32_t *xyz_dev, int64_t n, uint32_t *perm_dev, void *stream);

/* FOG loop                                            kit/nn.py:38-55, pcc_utils.py:83-89
 * Builds the octree of xyz on the device and copies every stored level to the host in
 * raster order: coords_out[d] (n_d,3) int32, occ_out[d] (n_d) uint8; capacity per level
 * cap_nodes.  levels_out/level_nodes_out as in gpcc_stats. */
GPCC_API int gpcc_build_octree(gpcc_ctx *ctx, const int32_t *xyz_dev, int64_t n, int32_t *levels_out,
                      int64_t *level_nodes_out, int32_t **coords_out_host, uint8_t **occ_out_host,
                      int64_t cap_nodes, void *stream);

/* spnn.Conv3d (stride 1) on raster-sorted coordinates: torchsparse 2.1.0 (not in tree), call
 * sites network_ue_4stage_conv.py:17-62.  in/out (n,C) float32 device in LOGICAL channel
 * order; w = upstream (k^3,C,C) host; residual may be NULL.  Also returns the number of
 * (node, neighbour) pairs.  relu: bit 0 = ReLU; bit 1 (test knob) = run the pair-plan form the decoder uses on its small levels
 * (csrc/fused.hpp; at most 16384 points) instead of the block-tile kernels -- same sums in the same order, same bits. */
GPCC_API int gpcc_conv3d(gpcc_ctx *ctx, const int32_t *xyz_sorted_dev, int64_t n, int channels, int kernel_size,
                const float *in_dev, const float *w_host, const float *res_dev, int relu,
                float *out_dev, int64_t *pairs_out, void *stream);

/* pred_head_s* + cdf + _convert_to_int_and_normalize  network_ue_4stage_conv.py:65-94,
 * pcc_utils.py:146-171, kit/op.py:50-79.  x (n,C) device logical order; prob (n,m) float32,
 * cdf (n,m+1) uint16 device outputs (either may be NULL). */
GPCC_API int gpcc_head_cdf(gpcc_ctx *ctx, const float *x_dev, int64_t n, int channels, int m,
                  const float *w1_host, const float *b1_host, const float *w2_host, const float *b2_host,
                  float *prob_dev, uint16_t *cdf_dev, void *stream);

/* torchac.encode_int16_normalized_cdf / decode_int16_normalized_cdf   pcc_utils.py:174-177,322-366
 * cdf (n,Lp) uint16 device, sym (n) uint8 device.  chunk_log2 as in gpcc_encode
 * (0 = one stream, bytes identical to torchac's).  encode: *bytes_out context-owned host buffer.
 * For chunk_log2 != 0 both calls code the lanes with the coder of the CONTEXT's container version (gpcc_ctx_set_container_version:
 * 4 = the carry-propagating coder, the default; 3 = torchac's): a caller holding version-3 stage streams sets version 3 first. */
GPCC_API int gpcc_rc_encode(gpcc_ctx *ctx, const uint16_t *cdf_dev, int lp, const uint8_t *sym_dev, int64_t n,
                   int chunk_log2, const uint8_t **bytes_out, int64_t *nbytes_out, void *stream);
GPCC_API int gpcc_rc_decode(gpcc_ctx *ctx, const uint16_t *cdf_dev, int lp, const uint8_t *bytes, int64_t nbytes,
                   int64_t n, int chunk_log2, uint8_t *sym_dev, void *stream);

/* ================= HAC attribute-side kernels (a15, a17-a19) ================= */

/* arithmetic.calculate_cdf(mean, scale, Q, min_value, max_value) -> lower (n, max-min+2) float32
 * HAC/submodules/arithmetic.zip!arithmetic/arithmetic.cpp:4-15, arithmetic_kernel.cu:7-54.  All device. */
GPCC_API int gsac_calculate_cdf(gpcc_ctx *ctx, const float *mean_dev, const float *scale_dev, const float *q_dev, int64_t n,
                                int min_value, int max_value, float *lower_dev, void *stream);

/* arithmetic.arithmetic_encode(sym int16 (n), cdf float (n,Lp), chunk_size, N, Lp) -> (bytes uint8, cnt int32[chunks])
 * arithmetic.cpp:18-29, arithmetic_kernel.cu:94-232.  sym / cdf device; outputs are context-owned HOST buffers. */
GPCC_API int gsac_encode(gpcc_ctx *ctx, const int16_t *sym_dev, const float *cdf_dev, int chunk_size, int64_t n, int lp,
                         const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream);

/* arithmetic.arithmetic_decode(cdf, bytes, cnt, chunk_size, N, Lp) -> sym int16 (n)
 * arithmetic.cpp:32-43, arithmetic_kernel.cu:265-403.  cdf device, bytes / cnt host, sym_out device. */
GPCC_API int gsac_decode(gpcc_ctx *ctx, const float *cdf_dev, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size,
                         int64_t n, int lp, int16_t *sym_out_dev, void *stream);

/* The same coder over already-integerised rows: torchac.encode_int16_normalized_cdf / decode_int16_normalized_cdf
 * (requirements.txt:6; call sites src/gs_compress/HAC/utils/pcc_utils.py:174-177, TC-GS/utils/encodings.py:38-174).
 * cdf (n, Lp) uint16 device.  chunk_size = n gives torchac's single stream. */
GPCC_API int gsac_encode_u16(gpcc_ctx *ctx, const int16_t *sym_dev, const uint16_t *cdf_dev, int chunk_size, int64_t n, int lp,
                             const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_u16(gpcc_ctx *ctx, const uint16_t *cdf_dev, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size,
                             int64_t n, int lp, int16_t *sym_out_dev, void *stream);

/* The same coder with ONE row for all n symbols (lp = 2 .. 4 entries, host floats): the Bernoulli coder of HAC's hash tables and masks
 * (HAC/utils/encodings_cuda.py:228-262 builds an (n, 3) table whose rows are all (0, 1 - p, 1) -- 120 MB for the ten million mask bits of a
 * million anchors).  Byte-identical to gsac_encode / gsac_decode on that table. */
GPCC_API int gsac_encode_const(gpcc_ctx *ctx, const int16_t *sym_dev, const float *row_host, int chunk_size, int64_t n, int lp,
                               const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_const(gpcc_ctx *ctx, const float *row_host, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size,
                               int64_t n, int lp, int16_t *sym_out_dev, void *stream);

/* torchac's coder itself, on HOST arrays and one host thread -- the drop-in for torchac.encode_int16_normalized_cdf /
 * decode_int16_normalized_cdf (torchac 0.9.3; call sites TC-GS/utils/encodings.py:84-176, CAT-3DGS/utils/encodings.py:39-175,
 * HAC/utils/pcc_utils.py:174-177): ONE stream for the whole tensor, byte-identical to torchac's.  A single stream is a single
 * dependent chain, which a host core runs several times faster than one GPU lane; gauspcc_amd.torchac builds the integer rows
 * where the caller's tensors live and codes here.  cdf (n, lp) uint16 rows, sym (n) int16, all host memory; out: cap bytes
 * (4 n + 64 always suffice).  No context, no GPU call. */
GPCC_API int gsac_host_encode_u16(const int16_t *sym, const uint16_t *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out);
GPCC_API int gsac_host_decode_u16(const uint16_t *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out);
/* The same for torchac.encode_float_cdf / decode_float_cdf on a HOST float table (torchac's own calling convention: CPU tensors;
 * TC-GS/utils/encodings.py:84-129 moves its table to the CPU first): row i is integerised on the fly as torchac's
 * _convert_to_int_and_normalize does -- rint(cdf * (2^16 - (lp - 1))) + j modulo 2^16, fp32 -- so the table is read once,
 * 4 bytes per entry, and no int16 copy of it is ever built.  Bytes == the _u16 form on the pre-integerised rows.  lp <= 65536. */
GPCC_API int gsac_host_encode_f32(const int16_t *sym, const float *cdf, int64_t n, int lp, uint8_t *out, int64_t cap, int64_t *nbytes_out);
GPCC_API int gsac_host_decode_f32(const float *cdf, const uint8_t *bytes, int64_t nbytes, int64_t n, int lp, int16_t *sym_out);

/* encoder_gaussian / decoder_gaussian in one call each, WITHOUT the (n, max-min+2) float CDF table of
 * arithmetic.calculate_cdf (src/gs_compress/HAC/utils/encodings_cuda.py:336-371, 399-433):
 *   encode: x_int = round(x / Q), min/max over the slice, the two CDF entries of every symbol evaluated on the fly,
 *           the chunked coder -> (min, max, bytes, cnt) = what the reference writes into the `.b` file;
 *   decode: the <= 64 entries a decoding wave compares evaluated on the fly; x = (sym + min) * Q.
 * Byte-identical to gsac_calculate_cdf + gsac_encode (same entry formula, same integerisation).
 * x / mean / scale / Q device (n); bytes / cnt host; min / max are the float values the file stores. */
GPCC_API int gsac_encode_gaussian(gpcc_ctx *ctx, const float *x_dev, const float *mean_dev, const float *scale_dev, const float *q_dev, int64_t n,
                                  int chunk_size, float *min_out, float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out,
                                  const int32_t **cnt_out, int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_gaussian(gpcc_ctx *ctx, const float *mean_dev, const float *scale_dev, const float *q_dev, int64_t n, float min_value,
                                  float max_value, const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, float *x_out_dev,
                                  void *stream);

/* HAC++'s Gaussian-MIXTURE coder (src/gs_compress/HAC-plus/utils/encodings_cuda.py:177-317; callers
 * HAC-plus/scene/gaussian_model.py:1315, 1499): the CDF row of element i is
 *     lower[i][t] = clamp( sum_c calculate_cdf(mean_c[i], scale_c[i], Q[i])[t] * prob_c[i], 0, 1 ),   c = 0 .. k-1 in list order, fp32
 * with the same integerisation, chunking, symbol range and `.b` layout as the single Gaussian.  As above the table is never
 * materialised: gsac_encode_gaussian_mixed == calculate_cdf x k, multiply, add, clamp, arithmetic_encode of :205-247 in one
 * call, gsac_decode_gaussian_mixed == :285-317.  mean / scale / prob: host arrays of k (1..4) device pointers, (n) each.
 * gsac_calculate_cdf_mixed writes the table itself, (n, max - min + 2) fp32 on the device (tests; two-step callers). */
GPCC_API int gsac_encode_gaussian_mixed(gpcc_ctx *ctx, const float *x_dev, const float *const *mean_dev, const float *const *scale_dev,
                                        const float *const *prob_dev, int k, const float *q_dev, int64_t n, int chunk_size, float *min_out,
                                        float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out,
                                        int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_gaussian_mixed(gpcc_ctx *ctx, const float *const *mean_dev, const float *const *scale_dev, const float *const *prob_dev, int k,
                                        const float *q_dev, int64_t n, float min_value, float max_value, const uint8_t *bytes, int64_t nbytes,
                                        const int32_t *cnt, int chunk_size, float *x_out_dev, void *stream);
GPCC_API int gsac_calculate_cdf_mixed(gpcc_ctx *ctx, const float *const *mean_dev, const float *const *scale_dev, const float *const *prob_dev, int k,
                                      const float *q_dev, int64_t n, int min_value, int max_value, float *lower_dev, void *stream);

/* The same for every slice of an attribute in ONE call: conduct_encoding / conduct_decoding code each 3000-anchor
 * slice into its own `.b` file (own min / max, hence own alphabet; HAC/scene/gaussian_model.py:1123-1206, 1262-1311) --
 * 334 slices x 3 attributes per million anchors, each a serial chain of 10000-symbol chunks.  Here all chunks of all
 * slices are coded concurrently.  slice_start (nslices + 1, host): element ranges, slice_start[0] = 0.
 * encode: min_out / max_out (nslices, host); cnt lists the chunks slice by slice (ceil(len / chunk_size) each), bytes
 * are their payloads in that order -- the caller cuts them into the per-slice files.  decode: the reverse. */
GPCC_API int gsac_encode_gaussian_slices(gpcc_ctx *ctx, const float *x_dev, const float *mean_dev, const float *scale_dev, const float *q_dev,
                                         const int64_t *slice_start, int nslices, int chunk_size, float *min_out, float *max_out,
                                         const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out, int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_gaussian_slices(gpcc_ctx *ctx, const float *mean_dev, const float *scale_dev, const float *q_dev, const int64_t *slice_start,
                                         int nslices, const float *min_value, const float *max_value, const uint8_t *bytes, int64_t nbytes,
                                         const int32_t *cnt, int chunk_size, float *x_out_dev, void *stream);
/* The same for HAC++'s K-component mixture (gsac_encode_gaussian_mixed): all 3000-anchor slices of ONE ten-channel group of `feat` in one
 * call -- conduct_encoding / conduct_decoding code feat as five such groups per slice, every group's second component coming from the
 * channel-context MLP on the groups already coded (HAC-plus/scene/gaussian_model.py:1300-1321, 1484-1504). */
GPCC_API int gsac_encode_gaussian_mixed_slices(gpcc_ctx *ctx, const float *x_dev, const float *const *mean_dev, const float *const *scale_dev,
                                               const float *const *prob_dev, int k, const float *q_dev, const int64_t *slice_start, int nslices, int chunk_size,
                                               float *min_out, float *max_out, const uint8_t **bytes_out, int64_t *nbytes_out, const int32_t **cnt_out,
                                               int64_t *nchunks_out, void *stream);
GPCC_API int gsac_decode_gaussian_mixed_slices(gpcc_ctx *ctx, const float *const *mean_dev, const float *const *scale_dev, const float *const *prob_dev, int k,
                                               const float *q_dev, const int64_t *slice_start, int nslices, const float *min_value, const float *max_value,
                                               const uint8_t *bytes, int64_t nbytes, const int32_t *cnt, int chunk_size, float *x_out_dev, void *stream);

/* GaussianModel.mlp_grid = nn.Sequential(Linear(din, dh), ReLU, Linear(dh, dout)) (src/gs_compress/HAC/scene/gaussian_model.py:258-262,
 * called through get_grid_mlp at :1152-1153 and :1281-1282): y = W2 relu(W1 x + b1) + b2 for n rows.
 * w1 (dh, din), w2 (dout, dh) row-major as nn.Linear stores them; all pointers device.  Specified fp32 order
 * (bias, then fmaf over k ascending), so the encoder and the decoder see the same context parameters. */
GPCC_API int gshac_mlp2(gpcc_ctx *ctx, const float *x_dev, const float *w1_dev, const float *b1_dev, const float *w2_dev, const float *b2_dev,
                        int64_t n, int din, int dh, int dout, float *y_dev, void *stream);
/* The same with the activation as an argument: act 0 = ReLU, 1 = LeakyReLU(slope) -- HAC++'s channel-context MLPs
 * (Channel_CTX_fea.MLP_d0..4: Linear(150 + 10 c, 40) - LeakyReLU - Linear(40, 30), HAC-plus/scene/gaussian_model.py:117-168, called through
 * get_deform_mlp.forward(feat, mean_scale, to_dec=c) at :1306 and :1490): context MLPs too -- encoder and decoder must obtain the same
 * bits, hence the same specified fp32 order. */
GPCC_API int gshac_mlp2_act(gpcc_ctx *ctx, const float *x_dev, const float *w1_dev, const float *b1_dev, const float *w2_dev, const float *b2_dev,
                            int64_t n, int din, int dh, int dout, int act, float slope, float *y_dev, void *stream);

/* _gridencoder.grid_encode_forward (inputs (N,D) in [0,1], embeddings (sO,F), offsets (L+1), resolutions (L),
 * outputs (L,N,F), ..., Rb, binary_vxl, min_level_id)   gridencoder.zip!gridencoder/src/gridencoder.h:12-22,
 * gridencoder.cu:100-361 (forward only; dy_dx / backward are training-side and out of scope).  All device. */
GPCC_API int gsge_forward(gpcc_ctx *ctx, const float *inputs_dev, const float *embeddings_dev, const int32_t *offsets_dev,
                          const int32_t *resolutions_dev, float *outputs_dev, int64_t n, int num_dim, int n_features, int n_levels,
                          int rb, const uint8_t *binary_vxl_dev, const int32_t *min_level_id_dev, void *stream);

/* ================= generate_neural_gaussians, inference path =================
 * src/gs_compress/HAC/gaussian_renderer/__init__.py:25-172 after attribute quantisation (:103-114, done by the caller):
 * view direction / distance (:116-118), optional feature bank (:121-132), opacity / colour / covariance MLPs on
 * [feat | view | dist] (:134-152), masking by neural opacity > 0 (:136-141, 160-162) and the assembly of position,
 * scale and rotation (:165-171).  n visible anchors, feat_dim in {32, 50}, K = n_offsets.
 * rows: the visible anchors as n ascending row indices into the model's tensors (the reference gathers `tensor[visible_mask]` five
 * times, :54-58 -- 0.8 GB of copies per million anchors; here the kernels read the rows in place), or NULL: the tensors hold exactly
 * the n anchors of the call.
 * mlp: 16 device pointers = {w1, b1, w2, b2} of mlp_feature_bank (all four NULL when use_feat_bank is off), mlp_opacity,
 * mlp_cov, mlp_color (nn.Linear layouts; gaussian_model.py:229-256).  mask (n, K) in {0, 1}; cam_center (3) device.
 * Outputs have room for n*K Gaussians; *count_out of them are written (anchor-major, the reference's order). */
GPCC_API int gsnn_generate(gpcc_ctx *ctx, int64_t n, const int32_t *rows, int feat_dim, int n_offsets, const float *anchor, const float *feat, const float *offsets,
                           const float *scaling, const float *mask, const float *cam_center, const float *const *mlp, float *xyz_out,
                           float *color_out, float *opacity_out, float *scale_out, float *rot_out, int64_t *count_out, void *stream);

/* ================= Gaussian splat rasteriser, forward only (a20) =================
 * diff_gaussian_rasterization (Scaffold-GS fork; zip missing from the reference tree).  Mirrors the C++ API the
 * reference's viewer calls: CudaRasterizer::Rasterizer::visible_filter / ::forward,
 * TC-GS/SIBR_viewers/src/projects/gaussianviewer/renderer/GaussianView.cpp:535-553, 660-688; Python call
 * sites HAC/gaussian_renderer/__init__.py:199-225, 268-303.  All pointers are device pointers; viewmatrix /
 * projmatrix are the 4x4 row-vector (transposed) matrices of HAC/scene/cameras.py:48-57; colours are
 * precomputed (shs = None in every call site); out_color is (3, H, W).  *num_rendered_out is the reference's count -- the tiles of every
 * splat's 3-sigma bounding square -- although the lists that are sorted and blended hold only the (splat, tile) pairs that can reach alpha >=
 * 1 / 255 somewhere in the tile (csrc/rasterizer.hip: tile_touches; the image is bit-identical either way). */
GPCC_API int gsr_visible_filter(gpcc_ctx *ctx, int P, int W, int H, const float *means3D, const float *scales, float scale_modifier,
                                const float *rotations, const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                float tan_fovx, float tan_fovy, int prefiltered, int *radii, void *stream);
GPCC_API int gsr_forward(gpcc_ctx *ctx, int P, const float *background, int W, int H, const float *means3D, const float *colors_precomp,
                         const float *opacities, const float *scales, float scale_modifier, const float *rotations,
                         const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, float tan_fovx, float tan_fovy,
                         int prefiltered, float *out_color, int *radii, int64_t *num_rendered_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GAUSPCC_H */
