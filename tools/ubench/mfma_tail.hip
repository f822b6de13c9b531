// Gate for VERDICT (round 5) item 1: would the part-filled tail tiles of k_sparse_conv be cheaper packed 4 rows at a time on
// v_mfma_f32_4x4x1_16B_f32 (sixteen independent (4 x 1)(1 x 4) blocks per instruction, every block with its OWN weight operand)?
//
//   part 1  semantics: operand / result lane layout of the 16-block form, and whether 32 chained K = 1 instructions from a zero
//           accumulator are the ascending fmaf chain of the normative arithmetic, bit for bit.
//   part 2  rates, one wave per SIMD on every CU (the conv kernel's occupancy), all in "cycles per 64 row-slots" so that the rows are
//           comparable: a tail batch = 16 groups of 4 rows = 256 x 4x4x1 MFMAs (2 048 MFMA cycles); four 16-row tiles = 64 x 16x16x4
//           MFMAs (2 048 MFMA cycles).
//             a. 4x4x1 stream, operands in registers (the instruction's own rate)
//             b. 4x4x1 stream, per-lane weights streamed: lane (block b, channel i) reads W2[offset_b][chunk][q][i][0..3] -- 64 b128
//                loads per batch (the layout VERDICT asks for, interleaved so that a block's four lanes read 64 contiguous bytes per
//                load) + the 64 gathered rows (8 b128 per lane), products added to LDS sums (8 reads + 8 writes of 16 bytes per lane)
//             c. 16x16x4 stream, operands in registers
//             d. 16x16x4 tile stream as the product kernel feeds it: per tile 4 coalesced b128 weight loads (1 KiB each), 2 b128
//                gathers, 16 MFMAs from zero, 2 + 2 LDS accesses of 16 bytes (the one-tile loop; the pair loop halves the weight loads)
//           The gate (VERDICT): b's flop rate >= 85 % of d's.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_tail.hip -o tools/ubench/mfma_tail && tools/ubench/mfma_tail
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------ part 1: semantics
// One wave.  A[b][i] (16 blocks x 4), B[b][j]; K instructions chained; D[b][i][j].  Assumed layout (checked): lane 4 b + i supplies
// A_b[i], lane 4 b + j supplies B_b[j], result register i of lane 4 b + j holds D_b[i][j].
__global__ void k_sem(const float *A, const float *B, float *D, int K)   // A, B: [K][64]; D: [64 lanes][4]
{
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + l], B[k * 64 + l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int semantics()
{
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> mant(1.0f, 2.0f);
    std::uniform_int_distribution<int> ex(-12, 12), sg(0, 1);
    auto rnd = [&] { return (sg(rng) ? -1.f : 1.f) * std::ldexp(mant(rng), ex(rng)); };
    const int K = 32;
    float *dA, *dB, *dD;
    CHECK(hipMalloc(&dA, K * 64 * 4)); CHECK(hipMalloc(&dB, K * 64 * 4)); CHECK(hipMalloc(&dD, 256 * 4));
    long total = 0, asc = 0, desc = 0, far = 0;
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<float> A(K * 64), B(K * 64), D(256);
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        k_sem<<<1, 64>>>(dA, dB, dD, K);
        CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < 16; ++b)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float a = 0.f, d = 0.f;
                    for (int k = 0; k < K; ++k) a = fmaf(A[k * 64 + 4 * b + i], B[k * 64 + 4 * b + j], a);
                    for (int k = K - 1; k >= 0; --k) d = fmaf(A[k * 64 + 4 * b + i], B[k * 64 + 4 * b + j], d);
                    const float got = D[(4 * b + j) * 4 + i];
                    ++total; asc += bits(got) == bits(a); desc += bits(got) == bits(d);
                    if (std::fabs(got - a) > 1e-3f * (std::fabs(a) + 1e-6f)) ++far;
                }
    }
    printf("part 1  v_mfma_f32_4x4x1_16B_f32, 32 chained instructions from zero: %ld elements; bit-identical to the ascending fmaf chain %ld, to the descending chain %ld; "
           "far off (wrong lane layout) %ld\n", total, asc, desc, far);
    hipFree(dA); hipFree(dB); hipFree(dD);
    return 0;
}

// ------------------------------------------------------------------ part 2: rates
constexpr int NOFF = 125;                 // kernel offsets (k = 5)
constexpr int ROWP = 36;                  // LDS row pitch in floats (144 bytes, the product kernel's)
constexpr int NROWS = 255;

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// a. / b.  MODE 0: operands in registers; MODE 1: streamed weights + gathered rows + LDS sums
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_tail(const float *__restrict__ W2, const float *__restrict__ X, int nrows_x, float *out, int nbatch, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 2, i = lane & 3;
    float *acc = lds + wave * ((NROWS + 1) * ROWP);
    const uint32_t wid = blockIdx.x * 4 + wave;
    for (int t = lane; t < (NROWS + 1) * ROWP; t += 64) acc[t] = 0.f;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    if (MODE == 0) {
        float a = 1.f + lane, bb = 2.f + lane;
        for (int bt = 0; bt < nbatch; ++bt) {
            f32x4 c[8];
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) c[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) c[ch] = __builtin_amdgcn_mfma_f32_4x4x1f32(a + (float)(ch + 8 * (k & 3)), bb + (float)bt, c[ch], 0, 0, 0);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) keep += c[ch];
        }
    } else {
        // the wave's block of rows sits at base; a group's offset and its four rows change per batch (hash of batch, block)
        const uint32_t base = (uint32_t)(((uint64_t)wid * 255u) % (uint32_t)(nrows_x - 4096));
        auto wptr = [&](int bt) { return W2 + (size_t)(hash32(wid * 7919u + (uint32_t)bt * 16u + (uint32_t)b) % NOFF) * 1024 + i * 4; };
        auto xptr = [&](int bt) { return X + (size_t)(base + hash32(wid * 104729u + (uint32_t)bt * 64u + (uint32_t)lane) % 4096u) * 32; };
        auto slot = [&](int bt) { return 1u + hash32(wid * 31u + (uint32_t)bt * 64u + (uint32_t)lane + 12345u) % (uint32_t)NROWS; };
        // weights: a ring of 4 channel chunks (8 b128 each) three chunks ahead; rows of batch bt + 1 loaded during batch bt
        float4 wr[4][8], x[8], xn[8];
        const float *wp = wptr(0), *wpn = wptr(1);
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = ld4(xptr(0) + 4 * q);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) wr[c][q] = ld4(wp + c * 128 + q * 16);
        for (int bt = 0; bt < nbatch; ++bt) {
            const uint32_t sl = slot(bt) * ROWP;
            const float *xpn = xptr(bt + 1);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                // prefetch chunk ch + 3 (of this batch or the next)
                {
                    const int pc = ch + 3;
                    const float *p = pc < 8 ? wp + pc * 128 : wpn + (pc - 8) * 128;
#pragma unroll
                    for (int q = 0; q < 8; ++q) wr[pc & 3][q] = ld4(p + q * 16);
                }
                if (ch == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) xn[q] = ld4(xpn + 4 * q);
                }
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 w = wr[ch & 3][q], xv = x[q];
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, xv.x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, xv.y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, xv.z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, xv.w, c, 0, 0, 0);
                }
                // lane (b, j = i) holds channels 4 ch .. 4 ch + 3 of its row: one 16-byte read-add-write of the row's sums
                float4 *s = reinterpret_cast<float4 *>(acc + sl + 4 * ch);
                float4 v = *s;
                v.x += c[0]; v.y += c[1]; v.z += c[2]; v.w += c[3];
                *s = v;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = xn[q];
            wp = wpn; wpn = wptr(bt + 2);
        }
    }
    const long long t1 = clock64();
    float s = keep[0] + keep[1] + keep[2] + keep[3];
    for (int t = lane; t < (NROWS + 1) * ROWP; t += 64) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// c. / d.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_full(const float *__restrict__ W, const float *__restrict__ X, int nrows_x, float *out, int ntile, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 15, g = lane >> 4;
    float *acc = lds + wave * ((NROWS + 1) * ROWP);
    const uint32_t wid = blockIdx.x * 4 + wave;
    for (int t = lane; t < (NROWS + 1) * ROWP; t += 64) acc[t] = 0.f;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
    if (MODE == 0) {
        float a = 1.f + lane, bb = 2.f + lane;
        for (int t = 0; t < ntile; ++t) {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { MF(c0, a + (float)(k & 3), bb + (float)t); MF(c1, a + (float)(4 + (k & 3)), bb + (float)t); }
            keep += c0 + c1;
        }
    } else {
        const uint32_t base = (uint32_t)(((uint64_t)wid * 255u) % (uint32_t)(nrows_x - 4096));
        struct AB { float4 a0, a1, b00, b01, b10, b11; };
        auto load = [&](int t) {
            const float *p = X + (size_t)(base + hash32(wid * 104729u + (uint32_t)t * 16u + (uint32_t)e) % 4096u) * 32 + 4 * g;
            const float *w = W + (size_t)(hash32(wid * 7919u + (uint32_t)t) % NOFF) * 1024 + lane * 4;
            AB r;
            r.a0 = ld4(p); r.a1 = ld4(p + 16);
            r.b00 = ld4(w); r.b01 = ld4(w + 256); r.b10 = ld4(w + 512); r.b11 = ld4(w + 768);
            return r;
        };
        constexpr int RING = 4;
        AB ring[RING];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) ring[s] = load(s);
        for (int t = 0; t < ntile; t += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                ring[(s + RING - 1) % RING] = load(t + s + RING - 1);
                const AB &v = ring[s];
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
                MF(c0, v.b00.x, v.a0.x); MF(c1, v.b10.x, v.a0.x);
                MF(c0, v.b00.y, v.a0.y); MF(c1, v.b10.y, v.a0.y);
                MF(c0, v.b00.z, v.a0.z); MF(c1, v.b10.z, v.a0.z);
                MF(c0, v.b00.w, v.a0.w); MF(c1, v.b10.w, v.a0.w);
                MF(c0, v.b01.x, v.a1.x); MF(c1, v.b11.x, v.a1.x);
                MF(c0, v.b01.y, v.a1.y); MF(c1, v.b11.y, v.a1.y);
                MF(c0, v.b01.z, v.a1.z); MF(c1, v.b11.z, v.a1.z);
                MF(c0, v.b01.w, v.a1.w); MF(c1, v.b11.w, v.a1.w);
                // transposed product: lane (g, e) holds four consecutive channels of row e per accumulator
                const uint32_t sl = (1u + hash32(wid * 31u + (uint32_t)(t + s) * 16u + (uint32_t)e + 12345u) % (uint32_t)NROWS) * ROWP;
                float4 *s0 = reinterpret_cast<float4 *>(acc + sl + 4 * g), *s1 = reinterpret_cast<float4 *>(acc + sl + 16 + 4 * g);
                float4 v0 = *s0, v1 = *s1;
                v0.x += c0[0]; v0.y += c0[1]; v0.z += c0[2]; v0.w += c0[3];
                v1.x += c1[0]; v1.y += c1[1]; v1.z += c1[2]; v1.w += c1[3];
                *s0 = v0; *s1 = v1;
            }
        }
    }
#undef MF
    const long long t1 = clock64();
    float s = keep[0] + keep[1] + keep[2] + keep[3];
    for (int t = lane; t < (NROWS + 1) * ROWP; t += 64) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    if (semantics()) return 1;
    const int nrows_x = 1 << 20;                      // 128 MiB of feature rows (a level of the 1 M-point cloud)
    float *W, *X, *out; long long *cyc;
    CHECK(hipMalloc(&W, NOFF * 1024 * 4)); CHECK(hipMalloc(&X, (size_t)nrows_x * 128)); CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMemset(W, 0, NOFF * 1024 * 4)); CHECK(hipMemset(X, 0, (size_t)nrows_x * 128));
    const size_t lds = 4 * (NROWS + 1) * ROWP * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_full<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_full<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char *name, double units64, float ms) {   // units64 = batches of 64 row-slots per wave
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = units64 * 64.0 * 2048.0 * 1024.0;        // 1 024 waves
        printf("  %-62s %8.3f ms  %7.0f cycles per 64 row-slots (2 048 = the matrix pipe alone)  %6.1f TFLOP/s of issued rows\n", name, ms, (double)c / units64, flop / (ms * 1e-3) / 1e12);
        return flop / (ms * 1e-3) / 1e12;
    };
    printf("part 2  one wave per SIMD, 256 CUs; a tail batch (256 x 4x4x1) and four 16-row tiles (64 x 16x16x4) are both 64 row-slots\n");
    float ms;
    const int nb = 2000, nt = 8000;
    k_tail<0><<<256, 256, lds>>>(W, X, nrows_x, out, 50, cyc);
    hipEventRecord(e0); k_tail<0><<<256, 256, lds>>>(W, X, nrows_x, out, nb, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    report("a. 4x4x1_16B stream, operands in registers", nb, ms);
    k_tail<1><<<256, 256, lds>>>(W, X, nrows_x, out, 50, cyc);
    hipEventRecord(e0); k_tail<1><<<256, 256, lds>>>(W, X, nrows_x, out, nb, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double tb = report("b. 4x4x1_16B, per-lane weights streamed + gathers + LDS sums", nb, ms);
    k_full<0><<<256, 256, lds>>>(W, X, nrows_x, out, 200, cyc);
    hipEventRecord(e0); k_full<0><<<256, 256, lds>>>(W, X, nrows_x, out, nt, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    report("c. 16x16x4 stream, operands in registers", nt / 4.0, ms);
    k_full<1><<<256, 256, lds>>>(W, X, nrows_x, out, 200, cyc);
    hipEventRecord(e0); k_full<1><<<256, 256, lds>>>(W, X, nrows_x, out, nt, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double td = report("d. 16x16x4 tiles: coalesced weights + gathers + LDS sums", nt / 4.0, ms);
    CHECK(hipDeviceSynchronize());
    printf("gate: b / d = %.2f (VERDICT's bar: >= 0.85)\n", tb / td);
    return 0;
}
