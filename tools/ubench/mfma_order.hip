// Is an fp32 MFMA an exact k-ordered chain of fused multiply-adds from its accumulator?  The normative conv arithmetic
//  needs that: p = fmaf(x[k], w[k], p) for k ascending.  Checked here for the form the conv loop uses
// (16x16x4) and for the one the "32x32x2 pair step" of HISTORY.md section 9 would use: every output element of a random
// product is compared, bit for bit, with the ascending chain, the descending chain, and "sum of exact products, then + c".
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave; A[m][k], B[k][n], C[m][n] in plain row-major arrays; the kernels move them through the documented lane layouts
__global__ void k16(const float *A, const float *B, const float *C, float *D)   // M = N = 16, K = 4
{
    const int l = threadIdx.x, m = l & 15, k = l >> 4;          // A: lane holds A[m = l % 16][k = l / 16]; B: B[k = l / 16][n = l % 16]
    f32x4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[(4 * (l >> 4) + i) * 16 + (l & 15)];   // D: lane holds rows 4 (l / 16) + i, column l % 16
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 4 + k], B[k * 16 + (l & 15)], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}
__global__ void k32(const float *A, const float *B, const float *C, float *D)   // M = N = 32, K = 2
{
    const int l = threadIdx.x, m = l & 31, k = l >> 5;          // A[m = l % 32][k = l / 32]; B[k = l / 32][n = l % 32]
    f32x16 c;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) c[4 * j + i] = C[(8 * j + 4 * (l >> 5) + i) * 32 + (l & 31)];   // rows 8 j + 4 (l / 32) + i, column l % 32
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * 2 + k], B[k * 32 + (l & 31)], c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) D[(8 * j + 4 * (l >> 5) + i) * 32 + (l & 31)] = c[4 * j + i];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

template <int M, int K, typename F> static void run(const char *name, F launch)
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> mant(1.0f, 2.0f);
    std::uniform_int_distribution<int> ex(-12, 12), sg(0, 1);
    auto rnd = [&] { return (sg(rng) ? -1.f : 1.f) * std::ldexp(mant(rng), ex(rng)); };
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, M * K * 4); hipMalloc(&dB, K * M * 4); hipMalloc(&dC, M * M * 4); hipMalloc(&dD, M * M * 4);
    long asc = 0, desc = 0, sumfirst = 0, total = 0, layout_bad = 0;
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<float> A(M * K), B(K * M), C(M * M), D(M * M);
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        for (auto &v : C) v = trial % 4 == 0 ? 0.f : rnd();
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        launch(dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < M; ++n) {
                float a = C[m * M + n], d = C[m * M + n];
                double s = 0.0;
                for (int k = 0; k < K; ++k) a = fmaf(A[m * K + k], B[k * M + n], a);
                for (int k = K - 1; k >= 0; --k) d = fmaf(A[m * K + k], B[k * M + n], d);
                for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)B[k * M + n];
                const float sf = (float)(s + (double)C[m * M + n]);
                const uint32_t got = bits(D[m * M + n]);
                ++total; asc += got == bits(a); desc += got == bits(d); sumfirst += got == bits(sf);
                if (std::fabs(D[m * M + n] - a) > 1e-3f * (std::fabs(a) + 1e-6f)) ++layout_bad;
            }
    }
    printf("%s: %ld elements; bit-identical to the ascending fma chain %ld, descending chain %ld, exact-sum-then-round %ld; far off (layout) %ld\n", name, total, asc, desc,
           sumfirst, layout_bad);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
}

int main()
{
    run<16, 4>("v_mfma_f32_16x16x4_f32", [](float *a, float *b, float *c, float *d) { k16<<<1, 64>>>(a, b, c, d); });
    run<32, 2>("v_mfma_f32_32x32x2_f32", [](float *a, float *b, float *c, float *d) { k32<<<1, 64>>>(a, b, c, d); });
    return 0;
}
