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
    int64_t n;
    int K;
    float cam[3];
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
    constexpr int DIN = F + 4;
    float x[DIN];
    // ob_view / ob_dist (:116-118)
    const float vx = a.anchor[3 * i] - a.cam[0], vy = a.anchor[3 * i + 1] - a.cam[1], vz = a.anchor[3 * i + 2] - a.cam[2];
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    x[F] = vx / dist; x[F + 1] = vy / dist; x[F + 2] = vz / dist; x[F + 3] = dist;
#pragma unroll
    for (int k = 0; k < F; ++k) x[k] = a.feat[i * F + k];
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
        const float o = tanhf(out_unit<F>(h, a.opacity, j)) * a.mask[i * K + j];
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
    const uint32_t *keep, *pos;
    int64_t nk;
    int K;
    float *xyz, *color, *opacity, *scale, *rot;
};

__global__ __launch_bounds__(TB) void k_assemble(AsmArgs a)
{
    const int64_t g = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (g >= a.nk || !a.keep[g]) return;
    const int64_t i = g / a.K;
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
    for (int k = 0; k < 3; ++k) a.xyz[3 * p + k] = a.anchor[3 * i + k] + a.offsets[g * 3 + k] * sc[k];
}

}  // namespace

extern "C" int gsnn_generate(gpcc_ctx *ctx, int64_t n, int feat_dim, int n_offsets, const float *anchor, const float *feat, const float *offsets,
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
    GP_TRY(ctx->arena.reserve((size_t)nk * (4 + 40 + 4 + 4) + ((size_t)4 << 20)));
    ctx->arena.reset();
    TAKE(nopa, float, nk); TAKE(dense, float, nk * 10); TAKE(keep, uint32_t, nk); TAKE(pos, uint32_t, nk + 1);
    NGArgs a = {};
    a.anchor = anchor; a.feat = feat; a.offsets = offsets; a.scaling = scaling; a.mask = mask; a.n = n; a.K = n_offsets;
    float cam[3];
    HIP_TRY(hipMemcpyAsync(cam, cam_center, 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    a.cam[0] = cam[0]; a.cam[1] = cam[1]; a.cam[2] = cam[2];
    a.bank = Mlp{mlp[0], mlp[1], mlp[2], mlp[3]};
    a.opacity = Mlp{mlp[4], mlp[5], mlp[6], mlp[7]};
    a.cov = Mlp{mlp[8], mlp[9], mlp[10], mlp[11]};
    a.color = Mlp{mlp[12], mlp[13], mlp[14], mlp[15]};
    a.nopa = nopa; a.dense = dense; a.keep = keep;
    if (feat_dim == 32) k_anchor_mlps<32><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(a);
    else k_anchor_mlps<50><<<(unsigned)cdiv(n, TB), TB, 0, st>>>(a);
    LAUNCH_CHECK();
    GP_TRY(exclusive_scan_u32(ctx, st, keep, pos, nk, pos + nk));
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, pos + nk, 4, hipMemcpyDeviceToHost, st));
    AsmArgs b = {anchor, offsets, scaling, nopa, dense, keep, pos, nk, n_offsets, xyz_out, color_out, opacity_out, scale_out, rot_out};
    k_assemble<<<(unsigned)cdiv(nk, TB), TB, 0, st>>>(b);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    *count_out = (int64_t)total;
    return GPCC_OK;
}
