// neural_gaussians.hip -- generate_neural_gaussians (inference path) on gfx950: SURVEY.md 8(f) row 2.
//
// Reference: src/gs_compress/HAC/gaussian_renderer/__init__.py:25-172 (the same function in HAC-plus / TC-GS / CAT-3DGS):
// per visible anchor the view direction and distance, the optional feature bank, three two-layer MLPs (opacity / colour /
// covariance) on [feat | view | dist], masking by neural opacity, and the assembly of the surviving Gaussians' position,
// colour, opacity, scale and rotation.  In PyTorch that is ~40 small kernels and an (n K, 22) concatenation that is
// written, masked and split again; here
//   k_anchor_mlps   one lane per anchor: inputs and hidden activations in registers, weights through the scalar cache
//                   (every lane of a wave reads the same weight -> SGPR operands), the K x (1 + 3 + 7) head outputs and
//                   the per-Gaussian keep flag written once;
//   scan            exclusive scan of the flags;
//   k_assemble      one lane per kept Gaussian writes its 14 output floats.
// fp32 throughout; RD evaluation tolerates rounding differences (PSNR within 0.01 dB), so unlike the codec kernels this
// arithmetic is not pinned to an order -- the test compares against the same operations in PyTorch fp32 with a tolerance.
#include "primitives.hpp"

using namespace gpcc;

namespace {

constexpr int TB = 256;

struct Mlp { const float *w1, *b1, *w2, *b2; };   // Linear(din, F) - ReLU - Linear(F, dout); nn.Linear layouts

struct NGArgs {
    const float *anchor, *feat, *offsets, *scaling, *mask;   // (n,3) (n,F) (n,K,3) (n,6) (n,K)
    const int32_t *rows;                                      // anchors of the model this call is about (visible_mask as an index list), or null: rows 0 .. n - 1
    int64_t n;
    int K;
    const float *cam;                // camera centre (3), device memory: read by the kernels, no copy to the host in front of the call
    Mlp bank, opacity, cov, color;   // bank.w1 == nullptr: no feature bank
    float *nopa;                     // (n K)     neural opacity * mask
    float *dense;                    // (n K, 10) colour (3) | scale_rot (7)
    uint32_t *keep;                  // (n K)     neural opacity > 0
};

template <int F, int DIN>
__device__ __forceinline__ void hidden_layer(const float (&x)[DIN], const Mlp &m, float (&h)[F])
{
#pragma unroll 2
    for (int c = 0; c < F; ++c) {
        float a = m.b1[c];
        const float *w = m.w1 + c * DIN;
#pragma unroll
        for (int k = 0; k < DIN; ++k) a = __builtin_fmaf(x[k], w[k], a);
        h[c] = a > 0.0f ? a : 0.0f;
    }
}

template <int F>
__device__ __forceinline__ float out_unit(const float (&h)[F], const Mlp &m, int j)
{
    float a = m.b2[j];
    const float *w = m.w2 + j * F;
#pragma unroll
    for (int k = 0; k < F; ++k) a = __builtin_fmaf(h[k], w[k], a);
    return a;
}

template <int F>
__global__ __launch_bounds__(TB) void k_anchor_mlps(NGArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= a.n) return;
    const int64_t src = a.rows ? a.rows[i] : i;     // row of the model's tensors; outputs are indexed by i
    constexpr int DIN = F + 4;
    float x[DIN];
    // ob_view / ob_dist (:116-118)
    const float vx = a.anchor[3 * src] - a.cam[0], vy = a.anchor[3 * src + 1] - a.cam[1], vz = a.anchor[3 * src + 2] - a.cam[2];
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    x[F] = vx / dist; x[F + 1] = vy / dist; x[F + 2] = vz / dist; x[F + 3] = dist;
#pragma unroll
    for (int k = 0; k < F; ++k) x[k] = a.feat[src * F + k];
    if (a.bank.w1) {   // view-adaptive feature (:121-132): softmax-weighted mix of the feature at strides 4 / 2 / 1
        float hb[F];
        const float cv[4] = {x[F], x[F + 1], x[F + 2], x[F + 3]};
        hidden_layer<F, 4>(cv, a.bank, hb);
        float z0 = out_unit<F>(hb, a.bank, 0), z1 = out_unit<F>(hb, a.bank, 1), z2 = out_unit<F>(hb, a.bank, 2);
        const float zm = fmaxf(z0, fmaxf(z1, z2));
        z0 = expf(z0 - zm); z1 = expf(z1 - zm); z2 = expf(z2 - zm);
        const float zs = z0 + z1 + z2;
        z0 /= zs; z1 /= zs; z2 /= zs;
        float y[F];
#pragma unroll
        for (int k = 0; k < F; ++k) y[k] = x[(k % (F / 4)) * 4] * z0 + x[(k % (F / 2)) * 2] * z1 + x[k] * z2;
#pragma unroll
        for (int k = 0; k < F; ++k) x[k] = y[k];
    }
    float h[F];
    const int K = a.K;
    // opacity: tanh head, times the binary grid mask; a Gaussian survives when the product is positive (:136-141)
    hidden_layer<F, DIN>(x, a.opacity, h);
    for (int j = 0; j < K; ++j) {
        const float o = tanhf(out_unit<F>(h, a.opacity, j)) * a.mask[src * K + j];
        a.nopa[i * K + j] = o;
        a.keep[i * K + j] = o > 0.0f ? 1u : 0u;
    }
    // colour: sigmoid head (:147-148)
    hidden_layer<F, DIN>(x, a.color, h);
    for (int j = 0; j < 3 * K; ++j) {
        const float v = out_unit<F>(h, a.color, j);
        a.dense[(i * K + j / 3) * 10 + j % 3] = 1.0f / (1.0f + expf(-v));
    }
    // covariance: linear head, 7 per Gaussian (:151-152)
    hidden_layer<F, DIN>(x, a.cov, h);
    for (int j = 0; j < 7 * K; ++j) a.dense[(i * K + j / 7) * 10 + 3 + j % 7] = out_unit<F>(h, a.cov, j);
}

struct AsmArgs {
    const float *anchor, *offsets, *scaling, *nopa, *dense;
    const int32_t *rows;
    const uint32_t *keep, *pos;
    int64_t nk;
    int K;
    float *xyz, *color, *opacity, *scale, *rot;
};

__global__ __launch_bounds__(TB) void k_assemble(AsmArgs a)
{
    const int64_t g = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (g >= a.nk || !a.keep[g]) return;
    const int64_t i = a.rows ? (int64_t)a.rows[g / a.K] : g / a.K;
    const int64_t gs = i * a.K + g % a.K;      // the candidate's row in the model's (n, K, 3) offsets
    const uint32_t p = a.pos[g];
    const float *d = a.dense + g * 10, *sc = a.scaling + i * 6;
    a.opacity[p] = a.nopa[g];
    a.color[3 * p] = d[0]; a.color[3 * p + 1] = d[1]; a.color[3 * p + 2] = d[2];
    // scaling = scaling[3:] * sigmoid(scale_rot[:3]); rot = normalize(scale_rot[3:7]) (:166-168)
#pragma unroll
    for (int k = 0; k < 3; ++k) a.scale[3 * p + k] = sc[3 + k] * (1.0f / (1.0f + expf(-d[3 + k])));
    const float q0 = d[6], q1 = d[7], q2 = d[8], q3 = d[9];
    const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);   // F.normalize: x / max(|x|, eps)
    a.rot[4 * p] = q0 / nrm; a.rot[4 * p + 1] = q1 / nrm; a.rot[4 * p + 2] = q2 / nrm; a.rot[4 * p + 3] = q3 / nrm;
    // xyz = anchor + offsets * scaling[:3] (:170-171)
#pragma unroll
    for (int k = 0; k < 3; ++k) a.xyz[3 * p + k] = a.anchor[3 * i + k] + a.offsets[gs * 3 + k] * sc[k];
}


// ---- the same function on the matrix pipe (round 5) ------------------------------------------------------------------
// k_anchor_mlps spends 13.6 k scalar-operand fmas per anchor at one lane each (1.78 ms per million anchors, 15 TFLOP/s) and
// hands the K x 10 head outputs of EVERY candidate Gaussian to the assembly through HBM (400 MB written, 400 MB read) although
// about half of them are dropped.  Here a wave owns 16 anchors: [feat | view | dist] staged in its LDS slice, every layer a
// chain of v_mfma_f32_16x16x4_f32 with the bias as the initial accumulator (lane (g, e) supplies row e / output e at
// k = 4 kk + g and receives rows 4 g .. 4 g + 3 of output e), all weights of the launch in LDS at bank-conflict-free
// pitches.  Two launches around the scan of the keep flags:
//   k_ng_opacity   x -> opacity MLP -> tanh * mask -> nopa, keep                 (the flags the scan needs; 69 MFMAs per tile)
//   k_ng_emit      x again -> colour and covariance MLPs -> the tile's 16 x 10 K outputs in LDS -> the surviving Gaussians
//                  written straight to their final rows (position from the scan): no (n K, 10) intermediate, no assembly pass.
// x is rebuilt rather than kept: 200 MB of feature reads against 430 MB of x written and read back.  n_offsets <= 16 (the
// reference's configurations use 10: HAC/arguments/__init__.py:55); wider models stay on k_anchor_mlps.
typedef float f32x4n __attribute__((ext_vector_type(4)));
constexpr int NG_AUX = 9;     // anchor (3) + scaling (6) per row of a tile, k_ng_emit's LDS copy
#define NG_MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int F> struct NGM {
    static constexpr int DINP = (F + 4 + 3) / 4 * 4;   // [feat | view | dist] padded to whole k-steps: 56 / 36
    static constexpr int DHP = (F + 3) / 4 * 4;        // hidden width padded: 52 / 32
    static constexpr int NT1 = (F + 15) / 16;          // hidden output tiles: 4 / 2
    static constexpr int P1 = DINP + 2, P2 = DHP + 2, PB = 6;   // LDS pitches of W1 / W2 / the bank's W1 (4 inputs)
};

struct MlpLds { const float *w1, *b1, *w2, *b2; };

// floats of one MLP's weights in LDS (first-layer pitch p1, nt2 output tiles)
template <int F> __host__ __device__ constexpr int ng_mlp_floats(int p1, int nt2) { return NGM<F>::NT1 * 16 * p1 + nt2 * 16 * NGM<F>::P2 + NGM<F>::NT1 * 16 + nt2 * 16; }

template <int F>
__device__ __forceinline__ MlpLds ng_stage(float *&sm, const Mlp &m, int din, int dinp, int p1, int dout, int nt2, int tid, int nthreads)
{
    using M = NGM<F>;
    float *W1s = sm, *W2s = W1s + M::NT1 * 16 * p1, *B1s = W2s + nt2 * 16 * M::P2, *B2s = B1s + M::NT1 * 16;
    sm = B2s + nt2 * 16;
    // (four loads in flight per trip: written as a plain loop, hipcc 7.2 waits for every element's load before it issues the next -- one HBM / L2 round trip
    //  per element, ~15 us at the head of every workgroup)
    for (int i0 = tid; i0 < M::NT1 * 16 * dinp; i0 += 4 * nthreads) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / dinp, k = i - c * dinp; v[u] = (i < M::NT1 * 16 * dinp && c < F && k < din) ? m.w1[c * din + k] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / dinp, k = i - c * dinp; if (i < M::NT1 * 16 * dinp) W1s[c * p1 + k] = v[u]; }
    }
    for (int i0 = tid; i0 < nt2 * 16 * M::DHP; i0 += 4 * nthreads) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / M::DHP, k = i - c * M::DHP; v[u] = (i < nt2 * 16 * M::DHP && c < dout && k < F) ? m.w2[c * F + k] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nthreads, c = i / M::DHP, k = i - c * M::DHP; if (i < nt2 * 16 * M::DHP) W2s[c * M::P2 + k] = v[u]; }
    }
    for (int i = tid; i < M::NT1 * 16; i += nthreads) B1s[i] = i < F ? m.b1[i] : 0.0f;
    for (int i = tid; i < nt2 * 16; i += nthreads) B2s[i] = i < dout ? m.b2[i] : 0.0f;
    return MlpLds{W1s, B1s, W2s, B2s};
}

__device__ __forceinline__ void ng_wave_sync()
{   // LDS operations of a wave execute in program order; this only keeps the compiler from moving them across a phase boundary
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The weight operands of output tile t + 1 are read from LDS while the MFMAs of tile t run -- two register sets and the scheduler held to that order
// (__builtin_amdgcn_sched_barrier): left to itself the compiler re-read BOTH operands from LDS directly in front of every pair of MFMAs, one LDS round
// trip per 64 cycles of matrix work (the same finding as in attributes.hip: k_mlp2_mfma).
template <int NK>
__device__ __forceinline__ void ng_load_w(float (&w)[NK], const float *wr)
{
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) w[kk] = wr[4 * kk];
}
template <int NK>
__device__ __forceinline__ f32x4n ng_chain(const float (&a)[NK], const float (&w)[NK], float bias)
{
    f32x4n acc = {bias, bias, bias, bias};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) NG_MF(acc, a[kk], w[kk]);
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

// relu(x W1^T + b1) of the tile: A operands in registers, result into hs[16][P2] (columns F .. DHP - 1 come out as zeros)
template <int F, int NK>
__device__ __forceinline__ void ng_hidden(const float (&a)[NK], const float *W1s, int p1, const float *B1s, float *hs, int e, int g)
{
    using M = NGM<F>;
    f32x4n acc[M::NT1];
    float wb[2][NK];
    ng_load_w<NK>(wb[0], W1s + e * p1 + g);
#pragma unroll
    for (int t = 0; t < M::NT1; ++t) {
        if (t + 1 < M::NT1) ng_load_w<NK>(wb[(t + 1) & 1], W1s + (16 * (t + 1) + e) * p1 + g);
        acc[t] = ng_chain<NK>(a, wb[t & 1], B1s[16 * t + e]);
    }
#pragma unroll
    for (int t = 0; t < M::NT1; ++t)
        if (16 * t + e < M::DHP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) hs[(4 * g + i) * M::P2 + 16 * t + e] = acc[t][i] > 0.0f ? acc[t][i] : 0.0f;
        }
}

template <int F>
__device__ __forceinline__ f32x4n ng_out_tile(const float (&a)[NGM<F>::DHP / 4], const MlpLds &m, int t, int e, int g)
{
    using M = NGM<F>;
    float w[M::DHP / 4];
    ng_load_w<M::DHP / 4>(w, m.w2 + (16 * t + e) * M::P2 + g);
    return ng_chain<M::DHP / 4>(a, w, m.b2[16 * t + e]);
}

// the output tiles 0 .. nt - 1 of a second layer, two per trip so that the register set of a tile is a compile-time choice: fn(t, acc)
template <int F, typename FN>
__device__ __forceinline__ void ng_layer2(const float (&a)[NGM<F>::DHP / 4], const MlpLds &m, int nt, int e, int g, FN fn)
{
    using M = NGM<F>;
    constexpr int NK = M::DHP / 4;
    float w0[NK], w1[NK];
    ng_load_w<NK>(w0, m.w2 + e * M::P2 + g);
    for (int t = 0; t < nt; t += 2) {
        if (t + 1 < nt) ng_load_w<NK>(w1, m.w2 + (16 * (t + 1) + e) * M::P2 + g);
        fn(t, ng_chain<NK>(a, w0, m.b2[16 * t + e]));
        if (t + 1 < nt) {
            if (t + 2 < nt) ng_load_w<NK>(w0, m.w2 + (16 * (t + 2) + e) * M::P2 + g);
            fn(t + 1, ng_chain<NK>(a, w1, m.b2[16 * (t + 1) + e]));
        }
    }
}

// [feat | view | dist | 0] of anchors row0 .. row0 + 15 into xs[16][px] (rows past n: the last anchor again), the feature bank applied
// ROWS (a.rows != nullptr) is a template parameter, not a run-time test: hipcc 7.2 materialised the uniform `a.rows != nullptr` as a lane mask under the
// partial EXEC of the feature-load loop and branched on it again inside the divergent emission loop -- lanes that were inactive at the first place read
// rows[] through the null pointer (found by tests/test_gpu_hac_plus_codec.py on an un-decoded model: a memory fault that depended on which Gaussians survive)
template <int F, bool ROWS>
__device__ __forceinline__ void ng_build_x(const NGArgs &a, const float (&cam)[3], int64_t row0, float *xs, int px, float *hs, bool bank, const MlpLds &bk, int lane, float *aux = nullptr)
{
    using M = NGM<F>;
    const int e = lane & 15, g = lane >> 4;
    // every load of the tile is requested before the first one is used: as a run-time loop (`for (i = lane; i < 16 * F / 2; i += 64)`) the compiler
    // waited for each float2 before asking for the next -- seven dependent round trips per tile, 5 of the 7.8 us a wave spent on a tile of k_ng_opacity
    constexpr int NLD = (16 * (F / 2) + 63) / 64;
    float2 fv[NLD];
    float ax = 0.0f, ay = 0.0f, az = 0.0f, sv[6] = {};
    {
        int64_t rws[NLD];
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = lane + 64 * it, r = i / (F / 2);
            rws[it] = row0 + r < a.n ? row0 + r : a.n - 1;
            if (ROWS) rws[it] = i < 16 * (F / 2) ? (int64_t)a.rows[rws[it]] : 0;
        }
        int64_t arow = row0 + lane < a.n ? row0 + lane : a.n - 1;
        if (ROWS) arow = lane < 16 ? (int64_t)a.rows[arow] : 0;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = lane + 64 * it, r = i / (F / 2), c2 = i - r * (F / 2);
            fv[it] = i < 16 * (F / 2) ? *reinterpret_cast<const float2 *>(a.feat + rws[it] * F + 2 * c2) : make_float2(0.0f, 0.0f);
        }
        if (lane < 16) { ax = a.anchor[3 * arow]; ay = a.anchor[3 * arow + 1]; az = a.anchor[3 * arow + 2]; }
        if (aux && lane < 16) {   // k_ng_emit: the tile's anchors and scalings, kept in LDS for the emission (aux[16][NG_AUX])
#pragma unroll
            for (int k = 0; k < 6; ++k) sv[k] = a.scaling[arow * 6 + k];
        }
    }
    if (aux && lane < 16) {
        float *q = aux + lane * NG_AUX;
        q[0] = ax; q[1] = ay; q[2] = az;
#pragma unroll
        for (int k = 0; k < 6; ++k) q[3 + k] = sv[k];
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int i = lane + 64 * it, r = i / (F / 2), c2 = i - r * (F / 2);
        if (i < 16 * (F / 2)) *reinterpret_cast<float2 *>(xs + r * px + 2 * c2) = fv[it];
    }
    if (lane < 16) {   // ob_view / ob_dist (:116-118)
        const float vx = ax - cam[0], vy = ay - cam[1], vz = az - cam[2];
        const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
        float *x = xs + lane * px + F;
        x[0] = vx / dist; x[1] = vy / dist; x[2] = vz / dist; x[3] = dist;
        for (int k = F + 4; k < M::DINP; ++k) xs[lane * px + k] = 0.0f;
    }
    ng_wave_sync();
    if (!bank) return;
    // view-adaptive feature (:121-132): softmax(bank MLP(view, dist)) mixes the feature at strides 4 / 2 / 1
    {
        const float av[1] = {xs[e * px + F + g]};
        ng_hidden<F, 1>(av, bk.w1, M::PB, bk.b1, hs, e, g);
    }
    ng_wave_sync();
    float a2[M::DHP / 4];
#pragma unroll
    for (int kk = 0; kk < M::DHP / 4; ++kk) a2[kk] = hs[e * M::P2 + 4 * kk + g];
    const f32x4n z = ng_out_tile<F>(a2, bk, 0, e, g);
    ng_wave_sync();
    if (e < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[(4 * g + i) * M::P2 + e] = z[i];
    }
    ng_wave_sync();
    if (lane < 16) {
        float z0 = hs[lane * M::P2], z1 = hs[lane * M::P2 + 1], z2 = hs[lane * M::P2 + 2];
        const float zm = fmaxf(z0, fmaxf(z1, z2));
        z0 = expf(z0 - zm); z1 = expf(z1 - zm); z2 = expf(z2 - zm);
        const float zs = z0 + z1 + z2;
        hs[lane * M::P2] = z0 / zs; hs[lane * M::P2 + 1] = z1 / zs; hs[lane * M::P2 + 2] = z2 / zs;
    }
    ng_wave_sync();
    constexpr int NIT = (16 * F + 63) / 64;
    float y[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        if (idx < 16 * F) {
            const int r = idx / F, k = idx - r * F;
            const float *x = xs + r * px, *w = hs + r * M::P2;
            y[it] = x[(k % (F / 4)) * 4] * w[0] + x[(k % (F / 2)) * 2] * w[1] + x[k] * w[2];
        }
    }
    ng_wave_sync();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        if (idx < 16 * F) { const int r = idx / F, k = idx - r * F; xs[r * px + k] = y[it]; }
    }
    ng_wave_sync();
}

template <int F, bool ROWS>
__global__ __launch_bounds__(512) void k_ng_opacity(NGArgs a)
{
    using M = NGM<F>;
    extern __shared__ __attribute__((aligned(16))) float ngsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6, e = lane & 15, g = lane >> 4;
    float *sm = ngsm;
    const bool bank = a.bank.w1 != nullptr;
    MlpLds bk = {};
    if (bank) bk = ng_stage<F>(sm, a.bank, 4, 4, M::PB, 3, 1, tid, blockDim.x);
    const MlpLds op = ng_stage<F>(sm, a.opacity, F + 4, M::DINP, M::P1, a.K, 1, tid, blockDim.x);
    __syncthreads();
    constexpr int PX = M::P1;
    float *xs = sm + wave * 16 * (PX + M::P2), *hs = xs + 16 * PX;
    const int64_t ntiles = (a.n + 15) / 16;
    const float cam[3] = {a.cam[0], a.cam[1], a.cam[2]};     // read once: inside the tile loop it was a round trip of its own per tile
    for (int64_t tile = (int64_t)blockIdx.x * waves + wave; tile < ntiles; tile += (int64_t)gridDim.x * waves) {
        const int64_t row0 = tile * 16;
        // the grid masks of this lane's four outputs: requested now, used after the MLP (their round trip is off the tile's chain)
        float mk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + 4 * g + i;
            mk[i] = 0.0f;
            if (e < a.K && row < a.n) mk[i] = a.mask[(ROWS ? (int64_t)a.rows[row] : row) * a.K + e];
        }
        ng_build_x<F, ROWS>(a, cam, row0, xs, PX, hs, bank, bk, lane);
        float a1[M::DINP / 4];
#pragma unroll
        for (int kk = 0; kk < M::DINP / 4; ++kk) a1[kk] = xs[e * PX + 4 * kk + g];
        ng_hidden<F, M::DINP / 4>(a1, op.w1, M::P1, op.b1, hs, e, g);
        ng_wave_sync();
        float a2[M::DHP / 4];
#pragma unroll
        for (int kk = 0; kk < M::DHP / 4; ++kk) a2[kk] = hs[e * M::P2 + 4 * kk + g];
        const f32x4n v = ng_out_tile<F>(a2, op, 0, e, g);
        // opacity: tanh head, times the binary grid mask; a Gaussian survives when the product is positive (:136-141)
        if (e < a.K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row0 + 4 * g + i;
                if (row < a.n) {
                    const float o = tanhf(v[i]) * mk[i];
                    a.nopa[row * a.K + e] = o;
                    a.keep[row * a.K + e] = o > 0.0f ? 1u : 0u;
                }
            }
        }
        ng_wave_sync();
    }
}

struct EmitArgs {
    const float *offsets, *scaling;
    const uint32_t *pos;
    int px;                            // pitch of the wave's tile buffer: even, >= DINP + 2 and >= 10 K
    float *xyz, *color, *opacity, *scale, *rot;
};

// NIT = candidates per lane in the emission = ceil(16 K / 64): a template parameter so that their attributes stay in registers
#ifndef NG_EMIT_LOADS_AT
#define NG_EMIT_LOADS_AT 1
#endif
template <int F, bool ROWS, int NIT>
__global__ __launch_bounds__(512) void k_ng_emit(NGArgs a, EmitArgs o)
{
    using M = NGM<F>;
    extern __shared__ __attribute__((aligned(16))) float ngsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6, e = lane & 15, g = lane >> 4;
    const int K = a.K, ntc = (3 * K + 15) / 16, ntv = (7 * K + 15) / 16, PX = o.px;
    float *sm = ngsm;
    const bool bank = a.bank.w1 != nullptr;
    MlpLds bk = {};
    if (bank) bk = ng_stage<F>(sm, a.bank, 4, 4, M::PB, 3, 1, tid, blockDim.x);
    const MlpLds col = ng_stage<F>(sm, a.color, F + 4, M::DINP, M::P1, 3 * K, ntc, tid, blockDim.x);
    const MlpLds cov = ng_stage<F>(sm, a.cov, F + 4, M::DINP, M::P1, 7 * K, ntv, tid, blockDim.x);
    __syncthreads();
    float *xs = sm + wave * 16 * (PX + M::P2 + NG_AUX), *hs = xs + 16 * PX, *aux = hs + 16 * M::P2;
    const int64_t ntiles = (a.n + 15) / 16;
    const float cam[3] = {a.cam[0], a.cam[1], a.cam[2]};     // read once: inside the tile loop it was a round trip of its own per tile
    for (int64_t tile = (int64_t)blockIdx.x * waves + wave; tile < ntiles; tile += (int64_t)gridDim.x * waves) {
        const int64_t row0 = tile * 16;
        ng_build_x<F, ROWS>(a, cam, row0, xs, PX, hs, bank, bk, lane, aux);
        // the emission's inputs (flag, position, attributes of the lane's NIT candidates): requested at NG_EMIT_LOADS_AT -- 0: where they are used, behind the
        // covariance MLP; 1: in front of it; 2: in front of both MLPs -- so that their round trip runs beside the matrix work
        const uint32_t *__restrict__ keepp = a.keep, *__restrict__ posp = o.pos;
        const float *__restrict__ nopap = a.nopa, *__restrict__ offp = o.offsets;
        bool kp[NIT];
        int64_t gis[NIT], srcs[NIT];
        int rr[NIT], jj[NIT];
        uint32_t kf[NIT], pp[NIT];
        float no[NIT];
        float of[NIT][3];
        auto emission_loads = [&]() {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = lane + 64 * it;
                rr[it] = idx / K; jj[it] = idx - rr[it] * K;
                gis[it] = (row0 + rr[it]) * K + jj[it];
                kp[it] = idx < 16 * K && row0 + rr[it] < a.n;
            }
            // ONE batch: flag, position and offset of every candidate are requested together, the dropped candidates' too (anchor and scaling of the tile's
            // rows sit in LDS since ng_build_x) -- the flag -> position -> attributes chain was three dependent round trips per tile
#pragma unroll
            for (int it = 0; it < NIT; ++it) srcs[it] = ROWS ? (kp[it] ? (int64_t)a.rows[row0 + rr[it]] : 0) : (kp[it] ? row0 + rr[it] : 0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int64_t gi = kp[it] ? gis[it] : 0;
                kf[it] = keepp[gi]; pp[it] = posp[gi]; no[it] = nopap[gi];
#pragma unroll
                for (int k = 0; k < 3; ++k) of[it][k] = offp[(srcs[it] * K + (kp[it] ? jj[it] : 0)) * 3 + k];
            }
        };
        if (NG_EMIT_LOADS_AT == 2) emission_loads();
        float a1[M::DINP / 4];
#pragma unroll
        for (int kk = 0; kk < M::DINP / 4; ++kk) a1[kk] = xs[e * PX + 4 * kk + g];
        float a2[M::DHP / 4];
        // colour: sigmoid head (:147-148) -> columns 0 .. 3 K - 1 of the tile buffer (x is in registers now)
        ng_hidden<F, M::DINP / 4>(a1, col.w1, M::P1, col.b1, hs, e, g);
        ng_wave_sync();
#pragma unroll
        for (int kk = 0; kk < M::DHP / 4; ++kk) a2[kk] = hs[e * M::P2 + 4 * kk + g];
        ng_layer2<F>(a2, col, ntc, e, g, [&](int t, const f32x4n &v) {
            const int c = 16 * t + e;
            if (c < 3 * K) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xs[(4 * g + i) * PX + c] = 1.0f / (1.0f + expf(-v[i]));
            }
        });
        ng_wave_sync();
        if (NG_EMIT_LOADS_AT == 1) emission_loads();
        // covariance: linear head, 7 per Gaussian (:151-152) -> columns 3 K .. 10 K - 1
        ng_hidden<F, M::DINP / 4>(a1, cov.w1, M::P1, cov.b1, hs, e, g);
        ng_wave_sync();
#pragma unroll
        for (int kk = 0; kk < M::DHP / 4; ++kk) a2[kk] = hs[e * M::P2 + 4 * kk + g];
        ng_layer2<F>(a2, cov, ntv, e, g, [&](int t, const f32x4n &v) {
            const int c = 16 * t + e;
            if (c < 7 * K) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xs[(4 * g + i) * PX + 3 * K + c] = v[i];
            }
        });
        ng_wave_sync();
        // the surviving Gaussians of the tile, each to its final row (:160-171); the emission is bound by its loads' round trips, not by its stores
        // (HISTORY.md section 4)
        {
            float *__restrict__ oop = o.opacity, *__restrict__ ocol = o.color, *__restrict__ osc = o.scale, *__restrict__ orot = o.rot, *__restrict__ oxyz = o.xyz;
            if (NG_EMIT_LOADS_AT == 0) emission_loads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) kp[it] = kp[it] && kf[it] != 0u;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (kp[it]) {
                    const uint32_t p = pp[it];
                    const float *c3 = xs + rr[it] * PX + 3 * jj[it], *d = xs + rr[it] * PX + 3 * K + 7 * jj[it], *ar = aux + rr[it] * NG_AUX;
                    oop[p] = no[it];
                    ocol[3 * p] = c3[0]; ocol[3 * p + 1] = c3[1]; ocol[3 * p + 2] = c3[2];
#pragma unroll
                    for (int k = 0; k < 3; ++k) osc[3 * p + k] = ar[6 + k] * (1.0f / (1.0f + expf(-d[k])));
                    const float q0 = d[3], q1 = d[4], q2 = d[5], q3 = d[6];
                    const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
                    orot[4 * p] = q0 / nrm; orot[4 * p + 1] = q1 / nrm; orot[4 * p + 2] = q2 / nrm; orot[4 * p + 3] = q3 / nrm;
#pragma unroll
                    for (int k = 0; k < 3; ++k) oxyz[3 * p + k] = ar[k] + of[it][k] * ar[3 + k];
                }
        }
        ng_wave_sync();
    }
}

constexpr size_t NG_LDS_MAX = 160 * 1024;

template <int F> static size_t ng_lds_opacity(bool bank, int waves)
{
    using M = NGM<F>;
    return 4 * ((size_t)(bank ? ng_mlp_floats<F>(M::PB, 1) : 0) + ng_mlp_floats<F>(M::P1, 1) + (size_t)waves * 16 * (M::P1 + M::P2));
}
template <int F> static int ng_emit_pitch(int K) { const int px = std::max(NGM<F>::P1, 10 * K); return px + (px & 1); }
template <int F> static size_t ng_lds_emit(bool bank, int K, int waves)
{
    using M = NGM<F>;
    return 4 * ((size_t)(bank ? ng_mlp_floats<F>(M::PB, 1) : 0) + ng_mlp_floats<F>(M::P1, (3 * K + 15) / 16) + ng_mlp_floats<F>(M::P1, (7 * K + 15) / 16) +
                (size_t)waves * 16 * (ng_emit_pitch<F>(K) + M::P2 + NG_AUX));
}

// number of waves per workgroup for the two launches, 0 when the model does not fit the matrix-pipe path
template <int F> static int ng_mfma_waves(bool bank, int K)
{
    if (K > 16) return 0;
    for (int waves = 8; waves >= 4; waves -= 4)
        if (ng_lds_emit<F>(bank, K, waves) <= NG_LDS_MAX && ng_lds_opacity<F>(bank, waves) <= NG_LDS_MAX) return waves;
    return 0;
}

template <int F, bool ROWS>
static int ng_mfma_opacity(gpcc_ctx *ctx, const NGArgs &a, int waves, hipStream_t st)
{
    const size_t lds = ng_lds_opacity<F>(a.bank.w1 != nullptr, waves);
    static PerDeviceOnce attr;
    GP_TRY(attr.run(ctx->device, [&]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ng_opacity<F, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NG_LDS_MAX));
        return GPCC_OK;
    }));
    const unsigned grid = (unsigned)std::min<int64_t>(256, cdiv(cdiv(a.n, 16), waves));
    k_ng_opacity<F, ROWS><<<grid, 64 * waves, lds, st>>>(a);
    LAUNCH_CHECK();
    return GPCC_OK;
}

template <int F, bool ROWS, int NIT>
static int ng_mfma_emit_n(gpcc_ctx *ctx, const NGArgs &a, EmitArgs o, int waves, hipStream_t st)
{
    const size_t lds = ng_lds_emit<F>(a.bank.w1 != nullptr, a.K, waves);
    o.px = ng_emit_pitch<F>(a.K);
    static PerDeviceOnce attr;
    GP_TRY(attr.run(ctx->device, [&]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ng_emit<F, ROWS, NIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NG_LDS_MAX));
        return GPCC_OK;
    }));
    const unsigned grid = (unsigned)std::min<int64_t>(256, cdiv(cdiv(a.n, 16), waves));
    k_ng_emit<F, ROWS, NIT><<<grid, 64 * waves, lds, st>>>(a, o);
    LAUNCH_CHECK();
    return GPCC_OK;
}

template <int F, bool ROWS>
static int ng_mfma_emit(gpcc_ctx *ctx, const NGArgs &a, const EmitArgs &o, int waves, hipStream_t st)
{
    switch ((16 * a.K + 63) / 64) {
    case 1: return ng_mfma_emit_n<F, ROWS, 1>(ctx, a, o, waves, st);
    case 2: return ng_mfma_emit_n<F, ROWS, 2>(ctx, a, o, waves, st);
    case 3: return ng_mfma_emit_n<F, ROWS, 3>(ctx, a, o, waves, st);
    default: return ng_mfma_emit_n<F, ROWS, 4>(ctx, a, o, waves, st);
    }
}

}  // namespace

extern "C" int gsnn_generate(gpcc_ctx *ctx, int64_t n, const int32_t *rows, int feat_dim, int n_offsets, const float *anchor, const float *feat, const float *offsets,
                             const float *scaling, const float *mask, const float *cam_center, const float *const *mlp /* 16 pointers */,
                             float *xyz_out, float *color_out, float *opacity_out, float *scale_out, float *rot_out, int64_t *count_out, void *stream)
{
    if (!ctx || !anchor || !feat || !offsets || !scaling || !mask || !cam_center || !mlp || !xyz_out || !color_out || !opacity_out || !scale_out || !rot_out || !count_out)
        return fail(GPCC_ERR_ARG, "null argument");
    *count_out = 0;
    if (n <= 0) return GPCC_OK;
    if (feat_dim != 32 && feat_dim != 50) return fail(GPCC_ERR_ARG, "generate_neural_gaussians: feat_dim must be 32 or 50 (got %d)", feat_dim);
    if (n_offsets < 1 || n_offsets > 64) return fail(GPCC_ERR_ARG, "generate_neural_gaussians: bad n_offsets %d", n_offsets);
    if (n * n_offsets >= ((int64_t)1 << 31)) return fail(GPCC_ERR_ARG, "too many Gaussians");
    for (int q = 4; q < 16; ++q)
        if (!mlp[q]) return fail(GPCC_ERR_ARG, "null MLP tensor %d", q);
    if (mlp[0] && feat_dim % 4) return fail(GPCC_ERR_ARG, "the feature bank needs feat_dim divisible by 4");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nk = n * n_offsets;
    static const bool use_mfma = dev_env_int("GAUSPCC_NG_MFMA", 1) != 0;
    const int waves = !use_mfma ? 0 : feat_dim == 32 ? ng_mfma_waves<32>(mlp[0] != nullptr, n_offsets) : ng_mfma_waves<50>(mlp[0] != nullptr, n_offsets);
    GP_TRY(ctx->arena.reserve((size_t)nk * (4 + (waves ? 0 : 40) + 4 + 4) + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(nopa, float, nk); TAKE(keep, uint32_t, nk); TAKE(pos, uint32_t, nk + 1);
    NGArgs a = {};
    a.anchor = anchor; a.feat = feat; a.offsets = offsets; a.scaling = scaling; a.mask = mask; a.rows = rows; a.n = n; a.K = n_offsets;
    a.cam = cam_center;
    a.bank = Mlp{mlp[0], mlp[1], mlp[2], mlp[3]};
    a.opacity = Mlp{mlp[4], mlp[5], mlp[6], mlp[7]};
    a.cov = Mlp{mlp[8], mlp[9], mlp[10], mlp[11]};
    a.color = Mlp{mlp[12], mlp[13], mlp[14], mlp[15]};
    a.nopa = nopa; a.keep = keep;
    // the survivor count lands in pinned memory: a copy into the stack variable is staged by the runtime and holds the host until it is done -- the emission
    // was enqueued ~18 us after the scan ended
    GP_TRY(ctx->hstage.reserve(64));
    volatile uint32_t *htotal = reinterpret_cast<volatile uint32_t *>(ctx->hstage.p);
    *htotal = 0;
    if (waves) {   // matrix pipe: flags -> scan -> colour / covariance and the surviving rows in one launch
        if (rows) { if (feat_dim == 32) GP_TRY((ng_mfma_opacity<32, true>(ctx, a, waves, st))); else GP_TRY((ng_mfma_opacity<50, true>(ctx, a, waves, st))); }
        else { if (feat_dim == 32) GP_TRY((ng_mfma_opacity<32, false>(ctx, a, waves, st))); else GP_TRY((ng_mfma_opacity<50, false>(ctx, a, waves, st))); }
        GP_TRY(exclusive_scan_u32(ctx, st, keep, pos, nk, pos + nk));
        HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(htotal), pos + nk, 4, hipMemcpyDeviceToHost, st));
        EmitArgs o = {offsets, scaling, pos, 0, xyz_out, color_out, opacity_out, scale_out, rot_out};
        if (rows) { if (feat_dim == 32) GP_TRY((ng_mfma_emit<32, true>(ctx, a, o, waves, st))); else GP_TRY((ng_mfma_emit<50, true>(ctx, a, o, waves, st))); }
        else { if (feat_dim == 32) GP_TRY((ng_mfma_emit<32, false>(ctx, a, o, waves, st))); else GP_TRY((ng_mfma_emit<50, false>(ctx, a, o, waves, st))); }
    } else {
        TAKE(dense, float, nk * 10);
        a.dense = dense;
        if (feat_dim == 32) k_anchor_mlps<32><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(a);
        else k_anchor_mlps<50><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(a);
        LAUNCH_CHECK();
        GP_TRY(exclusive_scan_u32(ctx, st, keep, pos, nk, pos + nk));
        HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(htotal), pos + nk, 4, hipMemcpyDeviceToHost, st));
        AsmArgs b = {anchor, offsets, scaling, nopa, dense, rows, keep, pos, nk, n_offsets, xyz_out, color_out, opacity_out, scale_out, rot_out};
        k_assemble<<<(unsigned)cdiv(nk, TB), TB, 0, st>>>(b);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipStreamSynchronize(st));
    GP_TRY(device_error_check(ctx));
    *count_out = (int64_t)*htotal;
    return GPCC_OK;
}
