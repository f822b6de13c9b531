// network.hip -- GausPcgc context network kernels for gfx950.
//
// k_sparse_conv   submanifold 3-D convolution, C = 32, exact fp32 on v_mfma_f32_32x32x2_f32.
//                 One wave owns 32 output nodes x 32 output channels (16 accumulator VGPRs) and
//                 walks the k^3 offsets in ascending order; an offset whose 32 neighbour slots are
//                 all empty is skipped by a wave ballot.  Per offset: 4 x 16-byte gathers of the
//                 neighbour row half (A operand, contiguous thanks to the physical channel order),
//                 4 x 16-byte loads of the pre-swizzled weight fragment (B operand), 16 MFMAs.
//                 The accumulation order per output element is the oracle's: offsets ascending,
//                 k = 0..31, one fma per term (MFMA f32 == fmaf chain), so results are bit-exact.
//                 Bound: fp32 MFMA (2*32*32 flop per (node, neighbour) pair).
// k_head          Linear-ReLU-Linear-softmax-cumsum-integerise, one node per lane, weights through
//                 the scalar cache.  Negligible next to the convolutions.
#include "network.hpp"
#include "octree.hpp"

namespace gpcc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CONV_WAVES = 4;

__global__ __launch_bounds__(64 * CONV_WAVES) void k_sparse_conv(ConvBatch jobs, const int32_t *__restrict__ nbrT, int n, int K, int relu)
{
    const ConvJob J = jobs.job[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * CONV_WAVES + wave) * 32;
    if (row0 >= n) return;
    const int r = lane & 31, h = lane >> 5;
    const int node = row0 + r;
    const bool inb = node < n;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float4 *__restrict__ wf = reinterpret_cast<const float4 *>(J.w) + (size_t)lane * 4;
    for (int o = 0; o < K; ++o) {
        const int j = inb ? nbrT[(size_t)o * n + node] : -1;
        if (__ballot(j >= 0) == 0ull) continue;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (j >= 0) {
            const float4 *__restrict__ p = reinterpret_cast<const float4 *>(J.in + (size_t)j * 32 + 16 * h);
            a0 = p[0]; a1 = p[1]; a2 = p[2]; a3 = p[3];
        }
        const float4 *__restrict__ w = wf + (size_t)o * 256;
        const float4 b0 = w[0], b1 = w[1], b2 = w[2], b3 = w[3];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, b2.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, b2.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, b3.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, b3.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, b3.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, b3.w, acc, 0, 0, 0);
    }
    // D layout: lane holds output channel c = lane & 31 for rows (i&3) + 8*(i>>2) + 4*h
    const int pc = phys_of(r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < n) {
            float v = acc[i];
            if (J.res) v = v + J.res[(size_t)row * 32 + pc];
            if (relu) v = v > 0.0f ? v : 0.0f;
            J.out[(size_t)row * 32 + pc] = v;
        }
    }
}

static int prof_event(gpcc_ctx *ctx, hipStream_t st, int *idx)
{
    Prof &p = ctx->prof;
    if (p.used == (int)p.pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        p.pool.push_back(e);
    }
    *idx = p.used++;
    HIP_TRY(hipEventRecord(p.pool[(size_t)*idx], st));
    return GPCC_OK;
}

int sparse_conv(gpcc_ctx *ctx, int level, hipStream_t st, const ConvBatch &jobs, int njobs, const int32_t *nbrT, int64_t n, int K, int relu)
{
    if (n <= 0) return GPCC_OK;
    if (n >= (int64_t)1 << 31) return fail(GPCC_ERR_ARG, "level too large");
    const bool prof = ctx && ctx->prof.on;
    ConvRec rec = {0, 0, level, njobs};
    if (prof) GP_TRY(prof_event(ctx, st, &rec.e0));
    dim3 grid((unsigned)cdiv(n, 32 * CONV_WAVES), (unsigned)njobs);
    k_sparse_conv<<<grid, 64 * CONV_WAVES, 0, st>>>(jobs, nbrT, (int)n, K, relu);
    LAUNCH_CHECK();
    if (prof) { GP_TRY(prof_event(ctx, st, &rec.e1)); ctx->prof.recs.push_back(rec); }
    return GPCC_OK;
}

int prof_collect(gpcc_ctx *ctx, const unsigned long long *pairs, int nlevels)
{
    Prof &p = ctx->prof;
    for (const ConvRec &r : p.recs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.pool[(size_t)r.e0], p.pool[(size_t)r.e1]));
        p.conv_ms += ms;
        p.conv_launches += 1;
        if (r.level >= 0 && r.level < nlevels) p.conv_pair_jobs += (int64_t)pairs[r.level] * r.njobs;
    }
    p.recs.clear();
    p.used = 0;
    return GPCC_OK;
}

// ------------------------------------------------------------------ row-wise elementwise kernels
constexpr int TB = 256;
static inline unsigned nblk(int64_t n) { return (unsigned)cdiv(n, TB); }

__global__ __launch_bounds__(TB) void k_embed_occ(const float4 *__restrict__ emb, const uint8_t *__restrict__ occ, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    out[t] = emb[(size_t)occ[i] * 8 + (t & 7)];
}
int embed_occ(hipStream_t st, const float *emb, const uint8_t *occ, int64_t n, float *out)
{
    k_embed_occ<<<nblk(n * 8), TB, 0, st>>>((const float4 *)emb, occ, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(TB) void k_child_features(const float4 *__restrict__ F, const uint32_t *__restrict__ parent, const uint64_t *__restrict__ rkey_c,
                                                       const float4 *__restrict__ temb, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const int g = (int)(t & 7);
    const uint64_t k = rkey_c[i];
    const int q = (int)((rk_x(k) & 1) | ((rk_y(k) & 1) << 1) | ((rk_z(k) & 1) << 2));
    out[t] = add4(F[(size_t)parent[i] * 8 + g], temb[q * 8 + g]);
}
int child_features(hipStream_t st, const float *F, const uint32_t *parent, const uint64_t *rkey_c, const float *temb, int64_t n, float *out)
{
    k_child_features<<<nblk(n * 8), TB, 0, st>>>((const float4 *)F, parent, rkey_c, (const float4 *)temb, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_stage_input_gt(const float4 *__restrict__ X, const float4 *__restrict__ emb, const uint8_t *__restrict__ occ,
                                                       int stage, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const uint32_t o = occ[i];
    const uint32_t prev = stage == 1 ? (o >> 7) & 1u : stage == 2 ? (o >> 6) & 3u : (o >> 4) & 15u;  // pcc_utils.py:121,128,136
    out[t] = add4(X[t], emb[prev * 8 + (t & 7)]);
}
int stage_input_gt(hipStream_t st, const float *X, const float *emb, const uint8_t *occ, int stage, int64_t n, float *out)
{
    k_stage_input_gt<<<nblk(n * 8), TB, 0, st>>>((const float4 *)X, (const float4 *)emb, occ, stage, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

struct SymPtrs { const uint8_t *s[4]; };

__global__ __launch_bounds__(TB) void k_stage_input_dec(const float4 *__restrict__ X, const float4 *__restrict__ emb, SymPtrs sp, const uint32_t *__restrict__ m2r,
                                                        int stage, int64_t n, float4 *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    int64_t i = t >> 3;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    uint32_t prev = sp.s[0][r];
    if (stage >= 2) prev = prev * 2 + sp.s[1][r];
    if (stage >= 3) prev = prev * 4 + sp.s[2][r];
    out[t] = add4(X[t], emb[prev * 8 + (t & 7)]);
}
int stage_input_dec(hipStream_t st, const float *X, const float *emb, const uint8_t *const sym_r[3], const uint32_t *m2r, int stage, int64_t n, float *out)
{
    SymPtrs sp = {{sym_r[0], sym_r[1], sym_r[2], nullptr}};
    k_stage_input_dec<<<nblk(n * 8), TB, 0, st>>>((const float4 *)X, (const float4 *)emb, sp, m2r, stage, n, (float4 *)out);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_assemble_occ(SymPtrs sp, const uint32_t *__restrict__ m2r, int64_t n, uint8_t *__restrict__ occ)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = m2r[i];
    occ[i] = (uint8_t)(sp.s[0][r] * 128 + sp.s[1][r] * 64 + sp.s[2][r] * 16 + sp.s[3][r]);
}
int assemble_occ(hipStream_t st, const uint8_t *const sym_r[4], const uint32_t *m2r, int64_t n, uint8_t *occ)
{
    SymPtrs sp = {{sym_r[0], sym_r[1], sym_r[2], sym_r[3]}};
    k_assemble_occ<<<nblk(n), TB, 0, st>>>(sp, m2r, n, occ);
    LAUNCH_CHECK();
    return GPCC_OK;
}

__global__ __launch_bounds__(TB) void k_rows_permute(const float *__restrict__ in, float *__restrict__ out, int64_t n, int to_physical)
{
    int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * 32) return;
    const int64_t i = t >> 5;
    const int c = (int)(t & 31);
    if (to_physical) out[i * 32 + phys_of(c)] = in[t];
    else out[t] = in[i * 32 + phys_of(c)];
}
int rows_permute(hipStream_t st, const float *in, float *out, int64_t n, int to_physical)
{
    k_rows_permute<<<nblk(n * 32), TB, 0, st>>>(in, out, n, to_physical);
    LAUNCH_CHECK();
    return GPCC_OK;
}

// ------------------------------------------------------------------ heads
// Same operation sequence as orc_exp() in oracle/gpcc_oracle.c (bit-exact by construction).
__device__ __forceinline__ float dev_exp(float x)
{
    if (x < -86.0f) return 0.0f;
    const float t = x * 1.44269504088896341f;
    const float nf = __builtin_rintf(t);
    float r = __builtin_fmaf(nf, -0.693145751953125f, x);
    r = __builtin_fmaf(nf, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int bits = __float_as_int(y);
    bits += (int)nf * (1 << 23);
    return __int_as_float(bits);
}

template <int M, int MODE>
__global__ __launch_bounds__(TB) void k_head(HeadArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= a.n) return;
    float x[32];
    {
        const float4 *__restrict__ px = reinterpret_cast<const float4 *>(a.x + (size_t)i * 32);
        float raw[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 v = px[g];
            raw[4 * g] = v.x; raw[4 * g + 1] = v.y; raw[4 * g + 2] = v.z; raw[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) x[c] = MODE == 2 ? raw[c] : raw[phys_of(c)];
    }
    const float *__restrict__ W1 = a.w1;
    const float *__restrict__ B1 = a.b1;
    const float *__restrict__ W2 = a.w2;
    const float *__restrict__ B2 = a.b2;
    float hdn[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float acc = B1[c];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(x[k], W1[c * 32 + k], acc);
        hdn[c] = acc > 0.0f ? acc : 0.0f;
    }
    float z[M], e[M];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float acc = B2[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(hdn[k], W2[j * 32 + k], acc);
        z[j] = acc;
        mx = acc > mx ? acc : mx;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        e[j] = dev_exp(z[j] - mx);
        s = j == 0 ? e[0] : s + e[j];
    }
    const float scale = (float)(65536 - M);
    uint32_t v[M + 1];
    v[0] = 0;
    float c = 0.0f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const float p = e[j] / s;
        if (MODE == 2 && a.prob) a.prob[(size_t)i * M + j] = p;
        c = c + p;
        const float cc = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        v[j + 1] = ((uint32_t)((int)__builtin_rintf(cc * scale) + j + 1)) & 0xFFFFu;
    }
    if (MODE == 0) {
        const uint32_t o = a.occ[i];
        const int sym = a.stage == 0 ? (o >> 7) & 1 : a.stage == 1 ? (o >> 6) & 1 : a.stage == 2 ? (o >> 4) & 3 : o & 15;  // pcc_utils.py:112-115
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j == sym) { lo = v[j]; hi = j == M - 1 ? 0x10000u : v[j + 1]; }
        a.lohi[a.m2r[i]] = lo | ((hi - 1u) << 16);
    } else {
        const size_t row = MODE == 1 ? (size_t)a.m2r[i] : (size_t)i;
        if (a.cdf) {
            uint16_t *dst = a.cdf + row * (M + 1);
#pragma unroll
            for (int j = 0; j <= M; ++j) dst[j] = (uint16_t)v[j];
        }
    }
}

template <int MODE>
static int head_launch(hipStream_t st, const HeadArgs &a)
{
    const unsigned g = nblk(a.n);
    switch (a.stage_m) {
    case 2: k_head<2, MODE><<<g, TB, 0, st>>>(a); break;
    case 4: k_head<4, MODE><<<g, TB, 0, st>>>(a); break;
    case 16: k_head<16, MODE><<<g, TB, 0, st>>>(a); break;
    default: return fail(GPCC_ERR_ARG, "head width must be 2, 4 or 16");
    }
    LAUNCH_CHECK();
    return GPCC_OK;
}

int head_cdf(hipStream_t st, const HeadArgs &a)
{
    if (a.n <= 0) return GPCC_OK;
    if (a.mode == 0) return head_launch<0>(st, a);
    if (a.mode == 1) return head_launch<1>(st, a);
    return head_launch<2>(st, a);
}

}  // namespace gpcc
