// Issue rate of the brute-force search's instruction pair, with occupancy set by the launch and co-residency shown.
// 256-thread workgroups (one wave per SIMD each); 256 x W of them put W waves on every SIMD of a 256-CU chip.  Every
// wave stamps the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) when it starts and ends; the
// program prints, per row: the wall time between events, the time per pair and SIMD from it, the same from the in-kernel
// cycle counts (median wave duration x shader clock), and (latest start - earliest start) / median duration, which is
// ~0 when all workgroups are resident together and ~1 when the grid ran as two rounds.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pair_rates.hip -o build/pair_rates && build/pair_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define KINDS(X)                                                                                                      \
    X(0, "v_xor(sgpr) + v_bcnt(acc)", "v_xor_b32 %1, %3, %2\n\tv_bcnt_u32_b32 %0, %1, %0", 2)                           \
    X(1, "v_xor(vgpr) + v_bcnt(acc)", "v_xor_b32 %1, %4, %2\n\tv_bcnt_u32_b32 %0, %1, %0", 2)                           \
    X(2, "v_bcnt(acc) alone", "v_bcnt_u32_b32 %0, %2, %0", 1)                                                         \
    X(3, "v_xor(sgpr) alone", "v_xor_b32 %0, %3, %0", 1)                                                              \
    X(4, "v_xor(vgpr) alone", "v_xor_b32 %0, %4, %0", 1)                                                              \
    X(5, "v_fma_f32 (control)", "v_fma_f32 %0, %4, %0, %0", 1)                                                        \
    X(6, "v_add_u32(vgpr) (control)", "v_add_u32 %0, %4, %0", 1)                                                      \
    X(7, "v_xor(16 sgprs in turn) + v_bcnt", "v_xor_b32 %1, %5, %2\n\tv_bcnt_u32_b32 %0, %1, %0", 2)                     \
    X(8, "v_xor(16 sgprs in turn) alone", "v_xor_b32 %0, %5, %0", 1)                                                  \
    X(9, "v_xor(sgprs) + v_bcnt, 4 chains", "v_xor_b32 %1, %5, %2\n\tv_bcnt_u32_b32 %0, %1, %0", 2)                   \
    X(10, "the same, loop body of 1024 instructions", "v_xor_b32 %1, %5, %2\n\tv_bcnt_u32_b32 %0, %1, %0", 2)

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, Stamp* stamps, int iters, unsigned seed) {
    unsigned acc[16], q[8], tmp, v2 = threadIdx.x * 40503u + seed;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 2654435761u + i + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = acc[i] * 7u + 1u;
    const unsigned s = __builtin_amdgcn_readfirstlane(seed * 77u + 3u);
    unsigned sv[16];            // sixteen different scalar registers, as the search kernel's train rows are
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_readfirstlane(seed * (31u + i) + i); asm volatile("" : "+s"(sv[i])); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int kReps = KIND == 10 ? 32 : 4;         // 16 groups each: 128 or 1024 instructions per trip for a pair row
    for (int it = 0; it < iters * 4 / kReps; ++it) {
#pragma unroll
        for (int r = 0; r < kReps; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#define X(id, name, text, n) if (KIND == id) asm volatile(text : "+v"(acc[KIND >= 9 ? (i & 3) : i]), "=&v"(tmp) : "v"(q[i & 7]), "s"(s), "v"(v2), "s"(sv[i]));
                KINDS(X)
#undef X
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
}

static int g_long = 1;     // argv[1]: run every row this many times longer (does the clock hold under a long launch?)
template <int KIND> void run(const char* name, int perIter, int cus, int W) {
    const int blocks = cus * W, threads = 256, iters = 4000 * g_long;
    unsigned* d; Stamp* ds;
    if (hipMalloc(&d, (size_t)blocks * threads * 4) != hipSuccess || hipMalloc(&ds, (size_t)blocks * 4 * sizeof(Stamp)) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, 400, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h((size_t)blocks * 4);
    hipMemcpy(h.data(), ds, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, dur;
    unsigned long long rmin = ~0ull, rmaxStart = 0;
    for (const Stamp& s : h) { cyc.push_back((double)(s.c1 - s.c0)); dur.push_back((double)(s.r1 - s.r0)); rmin = std::min(rmin, s.r0); rmaxStart = std::max(rmaxStart, s.r0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(dur.begin(), dur.end());
    const double medCyc = cyc[cyc.size() / 2], medDur = dur[dur.size() / 2];          // realtime ticks are 10 ns
    const double ghz = medCyc / (medDur * 10.0);
    const double steps = (double)iters * 64.0;                                        // instruction groups per wave
    const double nsWall = (double)ms * 1e6 / (steps * W);                             // per group and SIMD
    const double nsKernel = medCyc / ghz / (steps * W);
    const double spread = (double)(rmaxStart - rmin) / medDur;
    const double laneOps = 64.0 * perIter / nsWall * 1024.0 * 1e9 / 1e12;             // T lane-ops/s on 1024 SIMDs
    printf("%-28s W=%d  wall %8.3f ms  %6.3f ns/group/SIMD (wall)  %6.3f (in-kernel)  clock %.2f GHz  %5.2f cycles/group  start spread %.3f  -> %5.1f T lane-ops/s\n",
           name, W, ms, nsWall, nsKernel, ghz, nsWall * ghz, spread, laneOps);
    hipFree(d); hipFree(ds); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (argc > 1) g_long = std::max(1, atoi(argv[1]));
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; a group = the row's instruction(s) once per lane\n", p.name, cus);
    for (int W : (g_long > 1 ? std::vector<int>{8} : std::vector<int>{1, 2, 4, 8})) {
#define X(id, name, text, n) run<id>(name, n, cus, W);
        KINDS(X)
#undef X
    }
    return 0;
}
