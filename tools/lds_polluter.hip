// (self-check: pollute_probe() launches a kernel that only READS its 64 KiB -- 0 % pattern words on a fresh device, 100 % while the
//  polluter runs and after it has stopped: LDS is not cleared between kernels.  Load the library AFTER torch has initialised HIP.)
// Developer tool: keeps every CU's LDS (and a few KiB of VGPR-staged stores) full of a recognisable non-zero pattern by
// launching a fill kernel over and over on its own stream, so that a kernel of the library that reads LDS it has not
// written -- harmless on an idle device, whose LDS reads as zeros or as the kernel's own leftovers -- computes a wrong
// result under the parity tests.  (Found this way in round 3: 16-bit LDS-DMA loads leave the upper half of the dword alone.)
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_polluter.hip -o tools/liblds_polluter.so
//   GAUSPCC_TEST_POLLUTE=tools/liblds_polluter.so python -m pytest tests -m gpu ...   (tests/conftest.py starts it)
#include <hip/hip_runtime.h>
#include <atomic>
#include <thread>

__global__ __launch_bounds__(256) void k_pollute(unsigned pattern, unsigned *sink)
{
    extern __shared__ unsigned lds[];
    const unsigned words = 64u * 1024u / 4u;
    for (unsigned i = threadIdx.x; i < words; i += 256u) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 97u) % words] != pattern) sink[0] = 1;   // keeps the stores alive
}

// how much of a fresh workgroup's LDS holds the pattern right now: a kernel that only READS its 64 KiB (the tool's self-check)
__global__ __launch_bounds__(256) void k_probe(unsigned pattern, unsigned *hits)
{
    extern __shared__ unsigned lds[];
    unsigned n = 0;
    for (unsigned i = threadIdx.x; i < 64u * 1024u / 4u; i += 256u) n += lds[i] == pattern;
    atomicAdd(hits, n);
}
extern "C" double pollute_probe(unsigned pattern, int workgroups)
{
    unsigned *hits = nullptr, h = 0;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc(&hits, 4) != hipSuccess) return -1.0;
    (void)hipMemsetAsync(hits, 0, 4, s);
    k_probe<<<workgroups, 256, 64 * 1024, s>>>(pattern, hits);
    (void)hipMemcpyAsync(&h, hits, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(hits); (void)hipStreamDestroy(s);
    return (double)h / ((double)workgroups * 16384.0);
}

static std::atomic<bool> g_run{false};
static std::thread g_thread;

extern "C" int pollute_start(int device, unsigned pattern)
{
    if (g_run.exchange(true)) return 0;
    g_thread = std::thread([=] {
        hipSetDevice(device);
        hipStream_t s;
        hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        unsigned *sink = nullptr;
        hipMalloc(&sink, 4);
        while (g_run.load()) {
            for (int i = 0; i < 8; ++i) k_pollute<<<1024, 256, 64 * 1024, s>>>(pattern, sink);
            hipStreamSynchronize(s);
        }
        hipFree(sink);
        hipStreamDestroy(s);
    });
    return 0;
}
extern "C" int pollute_stop()
{
    if (!g_run.exchange(false)) return 0;
    g_thread.join();
    return 0;
}
