// Microbenchmark: throughput of v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 independent accumulator chains
// (round 4: the 32x32x2 pair step of tools/gen_conv_loop3.py has ONE chain of 16 per step, the 16x16x4 one four chains of 8).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k(float *out, int iters, float a0, float b0)
{
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c[k % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k % CHAINS], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[4096] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void k32(float *out, int iters, float a0, float b0)
{
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c[k % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k % CHAINS], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[4096] = t1 - t0;
}

template <int CHAINS>
void run32(int waves_per_simd, float *d)
{
    const int iters = 10000;
    dim3 grid(256), block(256 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<CHAINS><<<grid, block>>>(d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    k32<CHAINS><<<grid, block>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (long long *)d + 4096, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    printf("32x32x2 chains %d waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD, %.1f ticks per MFMA per wave, %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms,
           ms * 1e6 / mfma_per_simd, (double)cyc / (iters * 16), 2.0 * 32 * 32 * 2 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

template <int CHAINS>
void run(int waves_per_simd, float *d)
{
    const int iters = 20000;
    dim3 grid(256), block(256 * waves_per_simd);  // one block per CU, 4 or 8 waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<grid, block>>>(d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    k<CHAINS><<<grid, block>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (long long *)d + 4096, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    printf("chains %d waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD, wave-0 clock64 %lld -> %.1f ticks per MFMA per wave, %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms,
           ms * 1e6 / mfma_per_simd, cyc, (double)cyc / (iters * 16), 2.0 * 16 * 16 * 4 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *d; hipMalloc(&d, 1 << 22);
    for (int w = 1; w <= 2; ++w) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    for (int w = 1; w <= 2; ++w) { run32<1>(w, d); run32<2>(w, d); run32<4>(w, d); }
    return 0;
}
