// How the ORDER of vector instructions of the two issue classes inside one wave changes their cost.  Rows are sequences
// of S = v_bcnt_u32_b32 (issues every 4th cycle), F = v_xor_b32 (every 2nd cycle when four or more waves offer one)
// and s_nop 0, each written as ONE inline-asm block of 8 vector instructions so that the compiler cannot place
// anything between them.  Same harness as pair_rates.hip (256-thread workgroups, 256 x W of them; in-kernel clocks).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_order.hip -o build/issue_order && build/issue_order 2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define S4 "v_bcnt_u32_b32 %0, %2, %0\n\t"      /* a 4-cycle-class instruction */
#define F2 "v_xor_b32 %1, %4, %2\n\t"           /* a 2-cycle-class instruction (result unused: only its issue matters) */
#define FD "v_xor_b32 %1, %4, %0\n\t"           /* the same, reading the result of the instruction before it */
#define NOP "s_nop 0\n\t"
#define KINDS(X)                                                                                                      \
    X(0, "S F        x4 (no idle slot)", S4 F2 S4 F2 S4 F2 S4 F2, 8)                                                   \
    X(1, "S F nop    x4", S4 F2 NOP S4 F2 NOP S4 F2 NOP S4 F2 NOP, 8)                                                  \
    X(2, "S nop F    x4", S4 NOP F2 S4 NOP F2 S4 NOP F2 S4 NOP F2, 8)                                                  \
    X(3, "F S        x4 (no idle slot)", F2 S4 F2 S4 F2 S4 F2 S4, 8)                                                   \
    X(4, "F S nop    x4", F2 S4 NOP F2 S4 NOP F2 S4 NOP F2 S4 NOP, 8)                                                  \
    X(5, "S S S S F F F F", S4 S4 S4 S4 F2 F2 F2 F2, 8)                                                                \
    X(6, "S S S S nop F F F F", S4 S4 S4 S4 NOP F2 F2 F2 F2, 8)                                                        \
    X(7, "S x8", S4 S4 S4 S4 S4 S4 S4 S4, 8)                                                                           \
    X(8, "F x8", F2 F2 F2 F2 F2 F2 F2 F2, 8)                                                                           \
    X(9, "S F F F    x2", S4 F2 F2 F2 S4 F2 F2 F2, 8)                                                                  \
    X(10, "S nop F F F x2", S4 NOP F2 F2 F2 S4 NOP F2 F2 F2, 8)                                                        \
    X(11, "S Fdep    x4 (F reads S's result)", S4 FD S4 FD S4 FD S4 FD, 8)                                             \
    X(12, "S nop Fdep x4", S4 NOP FD S4 NOP FD S4 NOP FD S4 NOP FD, 8)

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
    constexpr int kReps = 4;         // 16 groups each: 128 or 1024 instructions per trip for a pair row
    for (int it = 0; it < iters * 4 / kReps; ++it) {
#pragma unroll
        for (int r = 0; r < kReps; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#define X(id, name, text, n) if (KIND == id) asm volatile(text : "+v"(acc[i]), "=&v"(tmp) : "v"(q[i & 7]), "s"(s), "v"(v2), "s"(sv[i]));
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
