// Probe of LDS-DMA semantics on gfx950: where does lane l's datum of a global_load_lds_{ushort,dword} land?
// hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_probe.hip -o tools/ubench/lds_dma_probe && tools/ubench/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint16_t *src, uint32_t *out, int mode)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint16_t *p = src + lane * 3;   // odd stride: unaligned dwords
    const uint32_t dst = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_groupstaticsize() + 256u));
    unsigned keep;
    if (mode == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main()
{
    uint16_t h[256]; for (int i = 0; i < 256; ++i) h[i] = (uint16_t)(0x1000 + i);
    uint16_t *d; uint32_t *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 2048); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64, 2048>>>(d, o, mode);
        uint32_t r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
        printf("mode %d (%s): lds dwords 60..76 (dst = byte 256 = dword 64):\n", mode, mode ? "dword" : "ushort");
        for (int i = 60; i < 76; ++i) printf(" [%d]=%08x", i, r[i]);
        printf("\n last touched dword: ");
        int last = -1; for (int i = 0; i < 512; ++i) if (r[i] != 0xDEADBEEFu) last = i;
        printf("%d\n", last);
    }
    return 0;
}
