// Microbenchmark: how many other instructions fit in the shadow of back-to-back v_mfma_f32_16x16x4_f32 issued by ONE wave?
// Per MFMA the loop issues P instructions of one kind (VALU / LDS read / LDS write / global load hitting one L1 line),
// 16 MFMAs per iteration on two accumulator chains, 1 or 2 waves per SIMD.  Prints cycles per MFMA (32 = pipe-bound).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int KIND, int P>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters)
{
    __shared__ float lds[512 * 8];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    const unsigned laddr = (unsigned)(threadIdx.x * 16) & 8191u;   // conflict-free 16-byte slots
    const float *gp = in + (threadIdx.x & 63) * 4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            ".rept 8\n"
            "v_mfma_f32_16x16x4_f32 v[16:19], v8, v9, v[16:19]\n"
            ".rept " STR(%[p]) "\n"
            ".if %[kind] == 0\n v_lshl_add_u32 v30, v31, 2, v32\n .endif\n"
            ".if %[kind] == 1\n ds_read_b128 v[40:43], %[la]\n .endif\n"
            ".if %[kind] == 2\n ds_write_b128 %[la], v[44:47]\n .endif\n"
            ".if %[kind] == 3\n global_load_dwordx4 v[40:43], %[ga], off\n .endif\n"
            ".if %[kind] == 4\n ds_read2_b32 v[40:41], %[la] offset1:16\n .endif\n"
            ".if %[kind] == 5\n s_add_u32 s40, s41, 3\n .endif\n"
            ".endr\n"
            "v_mfma_f32_16x16x4_f32 v[20:23], v8, v9, v[20:23]\n"
            ".rept " STR(%[p]) "\n"
            ".if %[kind] == 0\n v_lshl_add_u32 v33, v34, 2, v35\n .endif\n"
            ".if %[kind] == 1\n ds_read_b128 v[48:51], %[la]\n .endif\n"
            ".if %[kind] == 2\n ds_write_b128 %[la], v[44:47]\n .endif\n"
            ".if %[kind] == 3\n global_load_dwordx4 v[48:51], %[ga], off\n .endif\n"
            ".if %[kind] == 4\n ds_read2_b32 v[48:49], %[la] offset1:16\n .endif\n"
            ".if %[kind] == 5\n s_add_u32 s42, s43, 3\n .endif\n"
            ".endr\n"
            ".endr\n"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            :
            : [la] "v"(laddr), [ga] "v"(gp), [p] "n"(P), [kind] "n"(KIND)
            : "v8", "v9", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
              "v48", "v49", "v50", "v51", "s40", "s41", "s42", "s43", "scc", "memory");
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[0] = t1 - t0;
}

template <int KIND, int P>
void run(const char *name, float *d, const float *in)
{
    for (int w = 1; w <= 2; ++w) {
        const int iters = 2000;
        k<KIND, P><<<256, 256 * w>>>(d, in, 10);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND, P><<<256, 256 * w>>>(d, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
        printf("%-10s P=%d waves/SIMD %d: %6.1f cycles per MFMA per wave (wave 0), %6.1f ns per MFMA per SIMD\n", name, P, w, (double)cyc / (iters * 16.0), ms * 1e6 / (iters * 16.0 * w));
    }
}

int main()
{
    float *d, *in;
    hipMalloc(&d, 1 << 20); hipMalloc(&in, 1 << 20);
    hipMemset(in, 0, 1 << 20);
    run<0, 0>("none", d, in);
    run<0, 2>("valu", d, in); run<0, 4>("valu", d, in); run<0, 6>("valu", d, in); run<0, 7>("valu", d, in); run<0, 8>("valu", d, in);
    run<5, 4>("salu", d, in); run<5, 8>("salu", d, in);
    run<1, 1>("ds_rd128", d, in); run<1, 2>("ds_rd128", d, in);
    run<4, 1>("ds_rd2", d, in); run<4, 2>("ds_rd2", d, in);
    run<2, 1>("ds_wr128", d, in); run<2, 2>("ds_wr128", d, in);
    run<3, 1>("gload x4", d, in); run<3, 2>("gload x4", d, in);
    return 0;
}
