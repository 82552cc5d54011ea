// Micro-benchmark: issue rate of the vector instructions the matching and extraction kernels are made of,
// on gfx950, with control rows (v_fma_f32, v_pk_fma_f32) and a measured shader clock.
//
// Every wave runs `iters` trips of 64 independent instructions of one kind (16 accumulators x 4).  The grid is
// 1024 SIMDs x W waves with W in {1, 2, 4, 8}: blocks of 64 * W * 4 threads, one block per CU, so each SIMD
// hosts exactly W waves.  Two clocks are read inside the kernel around the loop:
//   s_memtime      shader cycles
//   s_memrealtime  100 MHz constant clock
// which gives (a) cycles per wave-instruction per SIMD without assuming a frequency and (b) the shader clock
// the chip actually held.  The wall-clock rate (hipEvents) is printed beside it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

enum Kind { FMA_F32, PK_FMA_F32, ADD_U32, AND_B32, XOR_B32, BCNT, XOR_BCNT, PK_ADD_U16, PK_MAX_U16, LERP_U8, SAD_U8,
            DOT4_U8, PERM_B32, ALIGNBYTE, MAD_U24, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_and_b32", "v_xor_b32 (sgpr src)",
                                     "v_bcnt_u32_b32", "v_xor + v_bcnt pair", "v_pk_add_u16", "v_pk_max_u16",
                                     "v_lerp_u8", "v_sad_u8", "v_dot4_u32_u8", "v_perm_b32", "v_alignbyte_b32",
                                     "v_mad_u32_u24"};
static const int kPerTrip[N_KINDS] = {1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1};

struct Stamp { unsigned long long cyc, real; };

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned* out, Stamp* stamps, int iters, unsigned seed) {
    unsigned a[16];
    unsigned long long b[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = ((unsigned long long)a[2 * i] << 32) | a[2 * i + 1];
    unsigned s = seed | 1;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == PK_FMA_F32) {
                // 8 independent 64-bit accumulators, two instructions each per round = 16 per round
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == AND_B32) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == XOR_B32) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == XOR_BCNT) { unsigned t; asm volatile("v_xor_b32 %0, %2, %1\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(a[i]) : "s"(s)); }
                if (KIND == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %0" : "+v"(a[i]));
                if (KIND == PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %0" : "+v"(a[i]));
                if (KIND == LERP_U8) asm volatile("v_lerp_u8 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == SAD_U8) asm volatile("v_sad_u8 %0, %0, 0, %0" : "+v"(a[i]));
                if (KIND == DOT4_U8) asm volatile("v_dot4_u32_u8 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == PERM_B32) asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %0, 1" : "+v"(a[i]));
                if (KIND == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(s));
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= (unsigned)b[i] ^ (unsigned)(b[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        Stamp st; st.cyc = c1 - c0; st.real = r1 - r0;
        stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    }
}

template <int KIND> void run(int wavesPerSimd) {
    const int blocks = 256, threads = 64 * 4 * wavesPerSimd, iters = 4000;
    const int nWaves = blocks * threads / 64;
    unsigned* d; Stamp* ds;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipMalloc(&ds, (size_t)nWaves * sizeof(Stamp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(nWaves);
    hipMemcpy(st.data(), ds, (size_t)nWaves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc(nWaves), mhz(nWaves);
    for (int i = 0; i < nWaves; ++i) { cyc[i] = (double)st[i].cyc; mhz[i] = st[i].real ? 100.0 * st[i].cyc / st[i].real : 0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double instrPerWave = (double)iters * 64.0 * kPerTrip[KIND];
    // every SIMD hosts wavesPerSimd waves that run side by side for the median wave's duration
    const double cycPerInstrSimd = cyc[nWaves / 2] / (instrPerWave * wavesPerSimd);
    const double wallPerSimd = (double)ms * 1e6 / (instrPerWave * wavesPerSimd);   // ns per wave-instr per SIMD
    printf("%-22s W=%d  %7.3f ms  in-kernel: %5.2f cycles/wave-instr/SIMD at %6.0f MHz   wall: %.3f ns/wave-instr/SIMD"
           " -> %5.1f T lane-ops/s chip\n", kNames[KIND], wavesPerSimd, ms, cycPerInstrSimd, mhz[nWaves / 2], wallPerSimd,
           1024.0 * 64.0 / wallPerSimd / 1e3 * (KIND == PK_FMA_F32 ? 2 : 1));
    hipFree(d); hipFree(ds);
}

template <int KIND> void sweep() {
    for (int w : {1, 2, 4}) run<KIND>(w);   // a block is at most 1024 threads = 4 waves per SIMD ...
}

// ... 8 waves per SIMD need two blocks per CU: launch 512 blocks of 1024 threads
template <int KIND> void run8() {
    const int blocks = 512, threads = 1024, iters = 4000;
    const int nWaves = blocks * threads / 64;
    unsigned* d; Stamp* ds;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipMalloc(&ds, (size_t)nWaves * sizeof(Stamp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, ds, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(nWaves);
    hipMemcpy(st.data(), ds, (size_t)nWaves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc(nWaves), mhz(nWaves);
    for (int i = 0; i < nWaves; ++i) { cyc[i] = (double)st[i].cyc; mhz[i] = st[i].real ? 100.0 * st[i].cyc / st[i].real : 0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double instrPerWave = (double)iters * 64.0 * kPerTrip[KIND];
    const double cycPerInstrSimd = cyc[nWaves / 2] / (instrPerWave * 8);
    const double wallPerSimd = (double)ms * 1e6 / (instrPerWave * 8);
    printf("%-22s W=8  %7.3f ms  in-kernel: %5.2f cycles/wave-instr/SIMD at %6.0f MHz   wall: %.3f ns/wave-instr/SIMD"
           " -> %5.1f T lane-ops/s chip\n", kNames[KIND], ms, cycPerInstrSimd, mhz[nWaves / 2], wallPerSimd,
           1024.0 * 64.0 / wallPerSimd / 1e3 * (KIND == PK_FMA_F32 ? 2 : 1));
    hipFree(d); hipFree(ds);
}

template <int KIND> void all() { sweep<KIND>(); run8<KIND>(); }

int main() {
    all<FMA_F32>(); all<PK_FMA_F32>(); all<ADD_U32>(); all<AND_B32>(); all<XOR_B32>(); all<BCNT>(); all<XOR_BCNT>();
    all<PK_ADD_U16>(); all<PK_MAX_U16>(); all<LERP_U8>(); all<SAD_U8>(); all<DOT4_U8>(); all<PERM_B32>();
    all<ALIGNBYTE>(); all<MAD_U24>();
    return 0;
}
