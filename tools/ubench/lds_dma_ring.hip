// A DEPTH-deep ring of LDS-DMA loads (global_load_lds_ushort) consumed one slot per step behind a hand-written
// s_waitcnt vmcnt(DEPTH - 2): does every step read the row it expects?  (The range decoder's row ring, in isolation.)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_ring.hip -o /tmp/ring && /tmp/ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int DEPTH = 16;
__device__ __forceinline__ void dma_u16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int STORES>
__global__ __launch_bounds__(64) void ring(const uint16_t *src, int stride, int steps, uint32_t *out, uint32_t *junk)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const char *p = reinterpret_cast<const char *>(src + (size_t)blockIdx.x * 64 + lane);
    const size_t step = (size_t)stride * 2;
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_groupstaticsize() + 1024u));
    const uint32_t *r = lds + 256;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { dma_u16(p, base + d * 256u); p += step; }
    vm_wait<DEPTH - 1>();
    uint32_t next = r[lane];
    uint32_t bad = 0, acc = 0;
    for (int i0 = 0; i0 < steps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int i = i0 + d;
            const uint32_t v = next;
            vm_wait<DEPTH - 2>();
            next = r[((d + 1) % DEPTH) * 64 + lane];
            dma_u16(p, base + d * 256u); p += step;
            const uint32_t want = (uint32_t)(uint16_t)(((size_t)blockIdx.x * 64 + lane + (size_t)i * stride) * 2654435761u >> 7);
            if (v != want && bad == 0) bad = (uint32_t)i + 1;
            acc = acc * 31u + v;
            if (STORES && (d & 3) == 3) junk[((size_t)blockIdx.x * 64 + lane) * 4 + (d >> 2)] = acc;
        }
    }
    out[blockIdx.x * 64 + lane] = bad;
}
int main()
{
    const int blocks = 256, stride = blocks * 64, steps = 1024;
    const size_t n = (size_t)stride * (steps + 2 * DEPTH);
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (uint16_t)(i * 2654435761u >> 7);
    uint16_t *d; uint32_t *o, *j;
    hipMalloc(&d, n * 2); hipMalloc(&o, blocks * 64 * 4); hipMalloc(&j, blocks * 64 * 16);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode) ring<1><<<blocks, 64, 1024 + DEPTH * 256>>>(d, stride, steps, o, j);
        else ring<0><<<blocks, 64, 1024 + DEPTH * 256>>>(d, stride, steps, o, j);
        std::vector<uint32_t> r(blocks * 64);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int nbad = 0; uint32_t first = 0;
        for (auto v : r) if (v) { ++nbad; if (!first) first = v; }
        printf("stores in the loop: %d  lanes with a wrong row: %d of %zu (first wrong step %u)\n", mode, nbad, r.size(), first ? first - 1 : 0);
    }
    return 0;
}
