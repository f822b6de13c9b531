// network_any.hip -- the context network for channel counts other than 32 (the reference's signatures take `channels`:
// HAC/utils/pcc_utils.py:28,65 -> Network(channels, kernel_size), GausPcgc/network_ue_4stage_conv.py:12).  Every shipped
// configuration uses 32, and the fast kernels (network.hip: MFMA tiles, 128-byte rows in a physical channel order) are built
// on that width; this file is the SAME arithmetic for any width up to 64 as plain one-thread-per-element kernels in logical
// channel order -- the normative chains of DESIGN.md section 2 (per kernel offset an fma chain over k ascending from zero,
// offsets added in ascending order, then residual, then ReLU; Linear = bias + fma chain), so the bytes equal the oracle's for
// that width.  Correct and complete, not fast: speed is not the point of a width nobody ships.
#include "network_dev.hpp"
#include "octree.hpp"

namespace gpcc {

namespace {

constexpr int TB = 256;
inline unsigned nblk(int64_t n) { return (unsigned)cdiv(n, TB); }

template <int C>
__global__ __launch_bounds__(TB) void k_any_embed(const float *__restrict__ emb, const uint8_t *__restrict__ occ, int64_t n, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * C) return;
    out[t] = emb[(size_t)occ[t / C] * C + (t % C)];
}

template <int C>
__global__ __launch_bounds__(TB) void k_any_child(const float *__restrict__ F, const uint32_t *__restrict__ parent, const uint64_t *__restrict__ rkey, const float *__restrict__ temb,
                                                  int64_t n, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * C) return;
    const int64_t i = t / C;
    const int c = (int)(t % C);
    const uint64_t k = rkey[i];
    const int q = (int)((rk_x(k) & 1) | ((rk_y(k) & 1) << 1) | ((rk_z(k) & 1) << 2));
    out[t] = F[(size_t)parent[i] * C + c] + temb[q * C + c];
}

template <int C>
__global__ __launch_bounds__(TB) void k_any_stage_gt(const float *__restrict__ X, const float *__restrict__ e1, const float *__restrict__ e2, const float *__restrict__ e3,
                                                     const uint8_t *__restrict__ occ, int64_t n, float *__restrict__ o1, float *__restrict__ o2, float *__restrict__ o3)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * C) return;
    const uint32_t o = occ[t / C];
    const int c = (int)(t % C);
    const float x = X[t];
    o1[t] = x + e1[((o >> 7) & 1u) * C + c];    // pcc_utils.py:121,128,136
    o2[t] = x + e2[((o >> 6) & 3u) * C + c];
    o3[t] = x + e3[((o >> 4) & 15u) * C + c];
}

struct AnySym { const uint8_t *s[3]; };
template <int C>
__global__ __launch_bounds__(TB) void k_any_stage_dec(const float *__restrict__ X, const float *__restrict__ emb, AnySym sp, const uint32_t *__restrict__ m2r, int stage, int64_t n,
                                                      float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * C) return;
    const uint32_t r = m2r[t / C];
    uint32_t prev = sp.s[0][r];
    if (stage >= 2) prev = prev * 2 + sp.s[1][r];
    if (stage >= 3) prev = prev * 4 + sp.s[2][r];
    out[t] = X[t] + emb[prev * C + (t % C)];
}

// One workgroup per block of the tile list; the block's rows accumulate in `out` itself, tile by tile (tiles of a block are in
// ascending kernel-offset order and the entries of one tile have distinct output rows: the normative order of additions).
template <int C>
__global__ __launch_bounds__(TB) void k_any_conv(ConvJob J, ConvTiles T, int relu)
{
    const int blk = (int)T.order[blockIdx.x];
    int lvi = 0;
    for (int i = 1; i < T.nlv; ++i) lvi = blk >= (int)T.lv_blk0[i] ? i : lvi;
    const int lrow0 = (blk - (int)T.lv_blk0[lvi]) * T.H;
    const int nrows = min(T.H, (int)T.lv_rows[lvi] - lrow0);
    const size_t row0 = (size_t)T.lv_row0[lvi] + (size_t)lrow0;
    const float *__restrict__ in = J.in + (size_t)T.lv_row0[lvi] * C;    // tile entries are row indices inside the level
    float *out = J.out + row0 * C;
    for (int idx = threadIdx.x; idx < nrows * C; idx += TB) out[idx] = 0.0f;
    __syncthreads();
    const uint32_t t0 = T.first[blk], t1 = T.first[blk + 1];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t oc = T.toc[t];
        const uint32_t o = oc & 0xFFFFu, cnt = oc >> 16;
        const float *__restrict__ W = J.w + (size_t)o * C * C;
        for (int idx = threadIdx.x; idx < (int)cnt * C; idx += TB) {
            const int e = idx / C, c = idx % C;
            const int slot = T.tr[(size_t)t * 16 + e];
            if (slot == 0) continue;
            const float *__restrict__ x = in + (size_t)(uint32_t)T.tj[(size_t)t * 16 + e] * C;
            float p = 0.0f;
#pragma unroll 8
            for (int k = 0; k < C; ++k) p = __builtin_fmaf(x[k], W[k * C + c], p);
            float *a = out + (size_t)(slot - 1) * C + c;
            *a = *a + p;
        }
        __syncthreads();
    }
    const float *res = J.res ? J.res + row0 * C : nullptr;
    for (int idx = threadIdx.x; idx < nrows * C; idx += TB) {
        float v = out[idx];
        if (res) v = v + res[idx];
        if (relu) v = v > 0.0f ? v : 0.0f;
        out[idx] = v;
    }
}

// Linear - ReLU - Linear - softmax - CDF for one node per thread (network.hip: k_head at any width, logical channel order)
template <int C, int M, int MODE>
__global__ __launch_bounds__(TB) void k_any_head(HeadArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= a.n) return;
    float x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = a.x[(size_t)i * C + c];
    float hdn[C];
    for (int c = 0; c < C; ++c) {
        float acc = a.b1[c];
#pragma unroll
        for (int k = 0; k < C; ++k) acc = __builtin_fmaf(x[k], a.w1[c * C + k], acc);
        hdn[c] = acc > 0.0f ? acc : 0.0f;
    }
    float z[M];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float acc = a.b2[j];
#pragma unroll
        for (int k = 0; k < C; ++k) acc = __builtin_fmaf(hdn[k], a.w2[j * C + k], acc);
        z[j] = acc;
        mx = acc > mx ? acc : mx;
    }
    head_tail<M, MODE>(a, i, z, mx, blockIdx.x);
}

template <int C, int MODE>
int any_head_launch(hipStream_t st, const HeadArgs &a)
{
    const unsigned g = nblk(a.n);
    switch (a.stage_m) {
    case 2: k_any_head<C, 2, MODE><<<g, TB, 0, st>>>(a); break;
    case 4: k_any_head<C, 4, MODE><<<g, TB, 0, st>>>(a); break;
    case 16: k_any_head<C, 16, MODE><<<g, TB, 0, st>>>(a); break;
    default: return fail(GPCC_ERR_ARG, "head width must be 2, 4 or 16");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

}  // namespace

#define ANY_C(C, CALL16, CALL64) do { if ((C) == 16) { CALL16; } else if ((C) == 64) { CALL64; } else return fail(GPCC_ERR_ARG, "channels must be 16, 32 or 64 (got %d)", (C)); } while (0)

int any_embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out, int C)
{
    if (n <= 0) return GPCC_OK;
    ANY_C(C, (k_any_embed<16><<<nblk(n * 16), TB, 0, st>>>(emb, occ, n, out)), (k_any_embed<64><<<nblk(n * 64), TB, 0, st>>>(emb, occ, n, out)));
    LAUNCH_CHECK();
    return GPCC_OK;
}

int any_child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out, int C)
{
    if (n <= 0) return GPCC_OK;
    ANY_C(C, (k_any_child<16><<<nblk(n * 16), TB, 0, st>>>(F, parent, rkey_c, temb, n, out)), (k_any_child<64><<<nblk(n * 64), TB, 0, st>>>(F, parent, rkey_c, temb, n, out)));
    LAUNCH_CHECK();
    return GPCC_OK;
}

int any_stage_inputs_gt(hipStream_t st, const float *X, const float *const emb[3], const uint8_t *occ, int64_t n, float *const out[3], int C)
{
    if (n <= 0) return GPCC_OK;
    ANY_C(C, (k_any_stage_gt<16><<<nblk(n * 16), TB, 0, st>>>(X, emb[0], emb[1], emb[2], occ, n, out[0], out[1], out[2])),
          (k_any_stage_gt<64><<<nblk(n * 64), TB, 0, st>>>(X, emb[0], emb[1], emb[2], occ, n, out[0], out[1], out[2])));
    LAUNCH_CHECK();
    return GPCC_OK;
}

int any_stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out, int C)
{
    if (n <= 0) return GPCC_OK;
    const AnySym sp = {{sym_r[0], sym_r[1], sym_r[2]}};
    ANY_C(C, (k_any_stage_dec<16><<<nblk(n * 16), TB, 0, st>>>(X, emb, sp, m2r, stage, n, out)), (k_any_stage_dec<64><<<nblk(n * 64), TB, 0, st>>>(X, emb, sp, m2r, stage, n, out)));
    LAUNCH_CHECK();
    return GPCC_OK;
}

int any_sparse_conv(hipStream_t st, const ConvBatch &jobs, int njobs, const ConvTiles &T, int64_t n, int relu, int C)
{
    if (n <= 0 || T.nblk <= 0) return GPCC_OK;
    for (int j = 0; j < njobs; ++j) {
        ANY_C(C, (k_any_conv<16><<<(unsigned)T.nblk, TB, 0, st>>>(jobs.job[j], T, relu)), (k_any_conv<64><<<(unsigned)T.nblk, TB, 0, st>>>(jobs.job[j], T, relu)));
        LAUNCH_CHECK();
    }
    return GPCC_OK;
}

int any_head_cdf(hipStream_t st, const HeadArgs &a, int C)
{
    if (a.n <= 0) return GPCC_OK;
    if (C == 16) return a.mode == 0 ? any_head_launch<16, 0>(st, a) : a.mode == 1 ? any_head_launch<16, 1>(st, a) : any_head_launch<16, 2>(st, a);
    if (C == 64) return a.mode == 0 ? any_head_launch<64, 0>(st, a) : a.mode == 1 ? any_head_launch<64, 1>(st, a) : any_head_launch<64, 2>(st, a);
    return fail(GPCC_ERR_ARG, "channels must be 16, 32 or 64 (got %d)", C);
}

}  // namespace gpcc
