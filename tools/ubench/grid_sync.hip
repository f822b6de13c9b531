// What does it cost to keep a chain of dependent layers inside ONE launch on this part, against a chain of launches?
// (task 1: measure before fusing the decoder's per-level launch chains.)
//
//  part 1  a chain of K dependent small kernels on one stream: eager (host enqueues, one sync at the end) against hipGraph
//          replay of the same chain -- us per launch.  Kernel = 160 workgroups x 256 threads, one 16-byte load + store per
//          thread (the size of a 5 k-node level's elementwise kernel); `work` repeats the body to stretch the kernel.
//  part 2  ONE persistent launch of P phases with a grid barrier behind each: every workgroup rewrites its 64 rows of a
//          (G x 64, 32) u32 array from pseudo-random rows of the previous phase's array (other workgroups' rows, re-read every
//          second phase: the L1-warm re-read that exposes a missing acquire), the result is compared word for word with the
//          host's.  Barrier forms:
//            flat      one monotonic counter, every workgroup: release fence, arrive, poll, acquire fence
//            xcd       per-XCC counters; the last arriver of an XCC does the release fence and arrives at the top counter,
//                      the last of those raises one generation word per XCC; every workgroup: acquire fence
//            wt        as xcd, payload stored write-through (sc1) so that nobody needs a release fence
//            local     the workgroups of ONE XCC only (they share an L2): no release fence, no top counter
//          G = workgroups (1024 threads each, one per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_sync.hip -o tools/ubench/grid_sync && tools/ubench/grid_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) uint32_t gu32;

// ------------------------------------------------------------------ part 1
__global__ __launch_bounds__(256) void k_link(const uint4 *__restrict__ in, uint4 *__restrict__ out, int n, int work)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 v = in[i];
    for (int w = 0; w < work; ++w) { v.x = v.x * 1664525u + 1013904223u; v.y ^= v.x >> 3; v.z += v.y; v.w ^= v.z << 1; }
    out[i] = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void part1(int K, int wgs, int work)
{
    const int n = wgs * 256;
    uint4 *a, *b;
    CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 16));
    CK(hipMemset(a, 1, (size_t)n * 16));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto chain = [&]() { for (int k = 0; k < K; ++k) k_link<<<wgs, 256, 0, st>>>((k & 1) ? b : a, (k & 1) ? a : b, n, work); };
    for (int w = 0; w < 3; ++w) chain();
    CK(hipStreamSynchronize(st));
    const int reps = 50;
    double t0 = now_us();
    for (int r = 0; r < reps; ++r) { chain(); CK(hipStreamSynchronize(st)); }
    const double eager = (now_us() - t0) / reps;
    // host enqueue time alone
    t0 = now_us();
    chain();
    const double host = now_us() - t0;
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = now_us();
    for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    const double graph = (now_us() - t0) / reps;
    // device-side span of one eager chain by events
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st)); chain(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain  K %3d  wgs %4d work %4d | eager %7.1f us (%.2f us/launch; host enqueue %.2f us/launch; device span %.2f us/launch) | graph replay %7.1f us (%.2f us/launch)\n",
           K, wgs, work, eager, eager / K, host / K, ms * 1e3 / K, graph, graph / K);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(st));
}

// ------------------------------------------------------------------ part 2
struct Bar {
    uint32_t xcc_count[8 * 32];   // one 128-byte line per XCC
    uint32_t xcc_gen[8 * 32];
    uint32_t members[8 * 32];     // workgroups of the launch on each XCC (census at kernel start)
    uint32_t top[32];
    uint32_t flat[32];
    uint32_t census[32];
    uint32_t timeout[32];
    uint32_t nxcc[32];
};

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ uint32_t ld_rlx(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t add_rlx(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr uint32_t SPIN_MAX = 1u << 20;   // x ~0.5 us: half a second, then the timeout word ends the launch

// one lane of the workgroup polls *p until it reaches `want` (monotonic words); false + timeout word on a stuck launch
__device__ __forceinline__ bool poll_ge(uint32_t *p, uint32_t want, uint32_t *tmo)
{
    for (uint32_t s = 0; s < SPIN_MAX; ++s) {
        if (ld_rlx(p) >= want) return true;
        if ((s & 63u) == 63u && ld_rlx(tmo)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    st_rlx(tmo, 1u);
    return false;
}

enum { BAR_FLAT = 0, BAR_XCD = 1, BAR_WT = 2, BAR_LOCAL = 3 };

// epoch = 1, 2, ...; every thread of every participating workgroup calls it; returns false when the launch timed out
template <int KIND> __device__ __forceinline__ bool grid_barrier(Bar *b, uint32_t epoch, uint32_t G, uint32_t xcc)
{
    __shared__ uint32_t ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its own stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = true;
        if (KIND == BAR_FLAT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            add_rlx(&b->flat[0], 1u);
            ok = poll_ge(&b->flat[0], epoch * G, &b->timeout[0]);
        } else if (KIND == BAR_LOCAL) {
            add_rlx(&b->xcc_count[xcc * 32], 1u);
            ok = poll_ge(&b->xcc_count[xcc * 32], epoch * ld_rlx(&b->members[xcc * 32]), &b->timeout[0]);
        } else {
            const uint32_t m = ld_rlx(&b->members[xcc * 32]);
            const uint32_t old = add_rlx(&b->xcc_count[xcc * 32], 1u);
            if (old + 1u == epoch * m) {                       // last of this XCC: the XCC's L2 holds everything its workgroups wrote
                if (KIND == BAR_XCD) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const uint32_t nx = ld_rlx(&b->nxcc[0]);
                const uint32_t t = add_rlx(&b->top[0], 1u);
                if (t + 1u == epoch * nx)
                    for (uint32_t x = 0; x < 8u; ++x) st_rlx(&b->xcc_gen[x * 32], epoch);
            }
            ok = poll_ge(&b->xcc_gen[xcc * 32], epoch, &b->timeout[0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_s = ok ? 1u : 0u;
    }
    __syncthreads();
    return ok_s != 0u;
}

__host__ __device__ inline uint32_t nbr(uint32_t r, uint32_t p, uint32_t n, uint32_t which)
{
    // changes every second phase: phases p and p + 2 read the same rows of the same buffer, rewritten in between
    const uint32_t q = (p >> 1) & 1u;
    return (uint32_t)(((uint64_t)r * (which ? 2654435761ull : 40503ull) + q * 977u + which * 131u + 5u) % n);
}

template <int KIND>
__global__ __launch_bounds__(1024) void k_persist(Bar *b, uint32_t *A, uint32_t *B, uint32_t n, int P, int active_xcc, uint32_t *xcc_of_wg, int empty)
{
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) xcc_of_wg[blockIdx.x] = xcc;
    if (KIND == BAR_LOCAL && (int)xcc != active_xcc) return;
    // census: who is where (flat barrier on its own counter; the members words are read after it)
    __shared__ uint32_t rank_s, G_s;
    if (threadIdx.x == 0) {
        rank_s = add_rlx(&b->members[xcc * 32], 1u);
        const uint32_t part = KIND == BAR_LOCAL ? 0u : gridDim.x;
        if (KIND != BAR_LOCAL) {
            add_rlx(&b->census[0], 1u);
            poll_ge(&b->census[0], part, &b->timeout[0]);
            if (blockIdx.x == 0) { uint32_t nx = 0; for (uint32_t x = 0; x < 8u; ++x) nx += ld_rlx(&b->members[x * 32]) ? 1u : 0u; st_rlx(&b->nxcc[0], nx); }
            add_rlx(&b->census[1], 1u);
            poll_ge(&b->census[1], part, &b->timeout[0]);
        }
        G_s = gridDim.x;
    }
    __syncthreads();
    uint32_t G = G_s, wg = blockIdx.x;
    if (KIND == BAR_LOCAL) { G = gridDim.x / 8u; wg = blockIdx.x / 8u; }   // observed placement: block b on XCC b % 8 (checked on the host from xcc_of_wg; a wrong guess ends in the spin bound)
    const uint32_t rows_per = n / G;          // 64
    const uint32_t c4 = threadIdx.x & 7u;     // 16-byte column of the row
    const uint32_t rloc = threadIdx.x >> 3;   // 128 rows per pass with 1024 threads; rows_per = 64: half the threads idle in the body
    for (int p = 0; p < P; ++p) {
        const uint32_t *in = (p & 1) ? B : A;
        uint32_t *out = (p & 1) ? A : B;
        if (!empty && rloc < rows_per) {
            const uint32_t r = wg * rows_per + rloc;
            const u32x4 x = *reinterpret_cast<const u32x4 *>(in + (size_t)nbr(r, (uint32_t)p, n, 0) * 32 + c4 * 4);
            const u32x4 y = *reinterpret_cast<const u32x4 *>(in + (size_t)nbr(r, (uint32_t)p, n, 1) * 32 + c4 * 4);
            u32x4 v = x * 3u + y + (uint32_t)p;
            uint32_t *dst = out + (size_t)r * 32 + c4 * 4;
            if (KIND == BAR_WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
            else *reinterpret_cast<u32x4 *>(dst) = v;
        }
        bool ok;
        if (KIND == BAR_LOCAL) {
            // same-XCC barrier: monotonic counter of this XCC, G participants
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __shared__ uint32_t ok_l;
            if (threadIdx.x == 0) {
                add_rlx(&b->xcc_count[xcc * 32], 1u);
                const bool o = poll_ge(&b->xcc_count[xcc * 32], (uint32_t)(p + 1) * G, &b->timeout[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                ok_l = o;
            }
            __syncthreads();
            ok = ok_l != 0u;
        } else ok = grid_barrier<KIND>(b, (uint32_t)(p + 1), G, xcc);
        if (!ok) return;
    }
}

static void host_ref(std::vector<uint32_t> &A, std::vector<uint32_t> &B, uint32_t n, int P)
{
    for (int p = 0; p < P; ++p) {
        const std::vector<uint32_t> &in = (p & 1) ? B : A;
        std::vector<uint32_t> &out = (p & 1) ? A : B;
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t a = nbr(r, (uint32_t)p, n, 0), c = nbr(r, (uint32_t)p, n, 1);
            for (int k = 0; k < 32; ++k) out[(size_t)r * 32 + k] = in[(size_t)a * 32 + k] * 3u + in[(size_t)c * 32 + k] + (uint32_t)p;
        }
    }
}

template <int KIND> static void part2(const char *name, int grid, int P, int empty, int polluter)
{
    const uint32_t G = KIND == BAR_LOCAL ? (uint32_t)grid / 8u : (uint32_t)grid;
    const uint32_t n = G * 64u;
    std::vector<uint32_t> hA((size_t)n * 32), hB((size_t)n * 32, 0u);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = (uint32_t)(i * 2246822519u + 374761393u);
    uint32_t *dA, *dB, *dx; Bar *db;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&db, sizeof(Bar))); CK(hipMalloc(&dx, 4 * (size_t)grid));
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // optional load on a second stream: short streaming kernels that come and go (what the codec's side stream does)
    uint4 *pa = nullptr, *pb = nullptr;
    const int pn = 64 * 256;
    if (polluter) { CK(hipMalloc(&pa, (size_t)pn * 16)); CK(hipMalloc(&pb, (size_t)pn * 16)); CK(hipMemset(pa, 3, (size_t)pn * 16)); }
    double best = 1e30, sum = 0;
    int bad_runs = 0, timeouts = 0;
    const int reps = 12;
    std::vector<uint32_t> refA = hA, refB = hB;
    if (!empty) host_ref(refA, refB, n, P);
    std::vector<uint32_t> xs((size_t)grid);
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpyAsync(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(db, 0, sizeof(Bar), st));
        CK(hipStreamSynchronize(st));
        if (polluter) for (int k = 0; k < 40; ++k) k_link<<<64, 256, 0, st2>>>((k & 1) ? pb : pa, (k & 1) ? pa : pb, pn, 200);
        CK(hipEventRecord(e0, st));
        k_persist<KIND><<<grid, 1024, 0, st>>>(db, dA, dB, n, P, 0, dx, empty);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(st2));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        Bar hb; CK(hipMemcpy(&hb, db, sizeof hb, hipMemcpyDeviceToHost));
        if (hb.timeout[0]) { ++timeouts; continue; }
        if (!empty) {
            std::vector<uint32_t> gA(hA.size()), gB(hB.size());
            CK(hipMemcpy(gA.data(), dA, gA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gB.data(), dB, gB.size() * 4, hipMemcpyDeviceToHost));
            if (gA != refA || gB != refB) ++bad_runs;
        }
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    CK(hipMemcpy(xs.data(), dx, 4 * (size_t)grid, hipMemcpyDeviceToHost));
    int off_rule = 0; for (int i = 0; i < grid; ++i) off_rule += xs[(size_t)i] != (uint32_t)(i % 8);
    printf("persist %-5s grid %4d G %3u P %3d %s%s | %7.2f us/phase best, %7.2f mean | wrong results %d/%d, timeouts %d | blocks off the b%%8 rule: %d\n", name, grid, G, P,
           empty ? "barrier only" : "rows+barrier", polluter ? " +load" : "", best * 1e3 / P, sum / (reps - 2) * 1e3 / P, bad_runs, reps, timeouts, off_rule);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(db)); CK(hipFree(dx));
    if (pa) { CK(hipFree(pa)); CK(hipFree(pb)); }
    CK(hipStreamDestroy(st)); CK(hipStreamDestroy(st2));
}

int main(int argc, char **argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 3;
    if (which & 1) {
        part1(60, 160, 0);
        part1(60, 160, 100);    // ~ a few us of ALU per kernel
        part1(60, 160, 1000);
        part1(60, 1024, 0);
        part1(18, 160, 2000);
    }
    if (which & 2) {
        for (int empty = 1; empty >= 0; --empty) {
            for (int g : {32, 64, 128, 256}) {
                part2<BAR_FLAT>("flat", g, 60, empty, 0);
                part2<BAR_XCD>("xcd", g, 60, empty, 0);
                part2<BAR_WT>("wt", g, 60, empty, 0);
            }
            part2<BAR_LOCAL>("local", 256, 60, empty, 0);   // 32 workgroups on XCC 0
            part2<BAR_LOCAL>("local", 128, 60, empty, 0);   // 16
        }
        part2<BAR_XCD>("xcd", 256, 60, 0, 1);
        part2<BAR_WT>("wt", 256, 60, 0, 1);
        part2<BAR_LOCAL>("local", 256, 60, 0, 1);
    }
    return 0;
}
