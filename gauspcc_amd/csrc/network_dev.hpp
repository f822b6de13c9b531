// network_dev.hpp -- device functions of the context network shared by network.hip and fused.hip: the head's softmax / CDF
// tail and its matrix-pipe body as a function of ONE wave (64 nodes), so that the stand-alone head kernel and the fused
// small-level kernel run the same arithmetic.
#pragma once
#include "network.hpp"
#include "rangecoder.hpp"

namespace gpcc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
constexpr int HEAD_LDS_FLOATS = 512 + 1024;   // per wave: hidden tile + logits of 64 nodes

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

// softmax -> cdf -> integerise -> the mode's output, for node i with logits z (max mx); shared by both head kernels
template <int M, int MODE>
__device__ __forceinline__ void head_tail(const HeadArgs &a, int64_t i, const float (&z)[M], float mx, unsigned bidx)
{
    float e[M];
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
    float pr[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const float p = e[j] / s;
        pr[j] = p;
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
        const size_t slot = a.pos ? (size_t)a.pos[i] + (size_t)a.stage * a.slots[i] : (size_t)rc_interleaved(a.m2r[i], a.chunk_log2, a.nch);
        a.lohi[slot] = lo | ((hi - 1u) << 16);
        if (a.bits) {  // ideal code length of the ground-truth symbol (the training loss of the reference, a14)
            float pg = pr[0];
#pragma unroll
            for (int j = 1; j < M; ++j) pg = j == sym ? pr[j] : pg;
            double b = -log2((double)pg + 1e-10);
            b = b < 0.0 ? 0.0 : (b > 50.0 ? 50.0 : b);
            if ((int64_t)(bidx + 1) * 256 <= a.n) {   // every lane of the block is live: reduce over the wave first
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d);
                if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(a.bits + (bidx & 15), b);
            } else {
                unsafeAtomicAdd(a.bits + (bidx & 15), b);
            }
        }
    } else if (MODE == 1) {
        constexpr int RS = M == 2 ? 1 : M == 4 ? 4 : 16;
        // (a.pos: a merged level of several scenes -- forest.hpp -- brings every node's row slot; else the one stream's interleave)
        uint16_t *dst = a.cdf + (a.pos ? (size_t)a.pos[i] : (size_t)rc_interleaved(a.m2r[i], a.chunk_log2, a.nch)) * RS;
        if (M == 2) dst[0] = (uint16_t)v[1];
        else if (M == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(v[1] | (v[2] << 16), v[3]);
        else {
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = v[2 * k + 1] | ((2 * k + 2 < M ? v[2 * k + 2] : 0u) << 16);
            reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            reinterpret_cast<uint4 *>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    } else {
        if (a.cdf) {
            uint16_t *dst = a.cdf + (size_t)i * (M + 1);
#pragma unroll
            for (int j = 0; j <= M; ++j) dst[j] = (uint16_t)v[j];
        }
    }
}

// Encode / decode heads on the matrix pipe.  The scalar kernel above spends 90 % of its cycles waiting (weights stream
// through the scalar cache, 1800 mostly dependent instructions per wave); here a wave takes 64 nodes as four 16-row
// tiles: hidden = relu(b1 + x W1^T) is 16 MFMAs per tile (bias as the initial accumulator, k ascending -- the chain the
// oracle runs), the 16 x 32 hidden tile turns from the accumulator layout into the A-operand layout through wave-
// private LDS, logits = b2 + hidden W2^T is 8 more, and the logits go through LDS to one lane per node for the
// sequential softmax / CDF tail.  Weights live in 24 VGPRs for the whole wave.
// one wave: nodes nb .. nb + 63 (loads clamped to a.n - 1, outputs only for nodes below nlimit <= a.n); hbuf = HEAD_LDS_FLOATS
// floats of LDS private to the wave.  bidx = the value the stand-alone kernel's blockIdx.x has (spreads the a14 accumulators).
template <int M, int MODE>
__device__ __forceinline__ void head_wave(const HeadArgs &a, int64_t nb, int64_t nlimit, int lane, float *hbuf, unsigned bidx)
{
    float *zbuf = hbuf + 512;
    const int e = lane & 15, g = lane >> 4;
    if (nb >= nlimit) return;
    const float4 *__restrict__ fr = reinterpret_cast<const float4 *>(a.frag) + lane;
    // [half][q][lane][4]: k-step kk = 4 q + r of output half `half`
    const float4 w10a = fr[0], w10b = fr[64], w11a = fr[128], w11b = fr[192], w2a = fr[256], w2b = fr[320];
    const float b1lo = a.frag[1536 + e], b1hi = a.frag[1536 + 16 + e], b2e = a.frag[1568 + e];
#define MF(c, x, w) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w, c, 0, 0, 0)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        const int64_t row = min(nb + 16 * t4 + e, a.n - 1);
        const float *px = a.x + row * 32 + 4 * g;       // physical channel order: logical k = 4 kk + g sits at 4 g + kk / 16 + 4 g + (kk - 4)
        const float4 a0 = ld4(px), a1 = ld4(px + 16);
        f32x4 c0 = {b1lo, b1lo, b1lo, b1lo}, c1 = {b1hi, b1hi, b1hi, b1hi};
        MF(c0, a0.x, w10a.x); MF(c1, a0.x, w11a.x);
        MF(c0, a0.y, w10a.y); MF(c1, a0.y, w11a.y);
        MF(c0, a0.z, w10a.z); MF(c1, a0.z, w11a.z);
        MF(c0, a0.w, w10a.w); MF(c1, a0.w, w11a.w);
        MF(c0, a1.x, w10b.x); MF(c1, a1.x, w11b.x);
        MF(c0, a1.y, w10b.y); MF(c1, a1.y, w11b.y);
        MF(c0, a1.z, w10b.z); MF(c1, a1.z, w11b.z);
        MF(c0, a1.w, w10b.w); MF(c1, a1.w, w11b.w);
        // relu; hidden[row][c] -> hbuf[row][c & 3][c >> 2] so that lane (e, g) finds its eight k-steps contiguous
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h0 = c0[i] > 0.0f ? c0[i] : 0.0f, h1 = c1[i] > 0.0f ? c1[i] : 0.0f;
            hbuf[(4 * g + i) * 32 + (e & 3) * 8 + (e >> 2)] = h0;            // c = e
            hbuf[(4 * g + i) * 32 + (e & 3) * 8 + 4 + (e >> 2)] = h1;        // c = 16 + e
        }
        const float4 h0 = *reinterpret_cast<const float4 *>(hbuf + e * 32 + g * 8), h1 = *reinterpret_cast<const float4 *>(hbuf + e * 32 + g * 8 + 4);
        f32x4 z = {b2e, b2e, b2e, b2e};
        MF(z, h0.x, w2a.x); MF(z, h0.y, w2a.y); MF(z, h0.z, w2a.z); MF(z, h0.w, w2a.w);
        MF(z, h1.x, w2b.x); MF(z, h1.y, w2b.y); MF(z, h1.z, w2b.z); MF(z, h1.w, w2b.w);
#pragma unroll
        for (int i = 0; i < 4; ++i) zbuf[(16 * t4 + 4 * g + i) * 16 + e] = z[i];
    }
#undef MF
    const int64_t i = nb + lane;
    if (i >= nlimit) return;
    float z[M];
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < M; ++j) { z[j] = zbuf[lane * 16 + j]; mx = z[j] > mx ? z[j] : mx; }
    head_tail<M, MODE>(a, i, z, mx, bidx);
}


}  // namespace gpcc
