// Micro-benchmark: issue rate of v_bcnt_u32_b32 against v_xor_b32 / v_add_u32 on gfx950.
// Every wave runs `iters` trips of 64 independent instructions of one kind; the grid fills
// all SIMDs with 8 waves each.  Prints wave-instructions per cycle per SIMD (at 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
    unsigned s = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 3) { unsigned t; asm volatile("v_xor_b32 %0, %2, %1\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(a[i]) : "s"(s)); }
                if (KIND == 4) asm volatile("v_sad_u8 %0, %0, 0, %0" : "+v"(a[i]));
                if (KIND == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(s));
            }
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int KIND> void run(const char* name, int per_trip) {
    const int blocks = 256 * 8, iters = 4000;
    unsigned* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 100, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, instr = waves * iters * 64.0 * per_trip;
    const double per_simd = instr / 1024.0;          // wave-instructions per SIMD
    printf("%-28s %8.3f ms  %.3f wave-instr/ns/SIMD  -> %.2f cycles per wave-instr at 2.4 GHz\n", name, ms,
           per_simd / (ms * 1e6), (ms * 1e6 * 2.4) / per_simd);
    hipFree(d);
}

int main() {
    run<2>("v_add_u32", 1);
    run<1>("v_xor_b32 (sgpr src)", 1);
    run<0>("v_bcnt_u32_b32", 1);
    run<3>("v_xor + v_bcnt pair", 2);
    run<4>("v_sad_u8", 1);
    run<5>("v_mad_u32_u24", 1);
    return 0;
}
