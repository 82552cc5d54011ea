// Micro-benchmark: issue cost of individual gfx950 vector instructions with register-only operands, to tell the
// instructions that a SIMD retires in about 2 cycles per wave64 (binary32 arithmetic) from those that take about 4
// (the byte / packed-16 / multi-operand integer instructions), and the multi-pass ones (32-bit integer multiply,
// transcendental).  Same method as popc_rate.hip: 1024 SIMDs x 8 waves, each wave runs `iters` trips of 64
// independent instructions of one kind; the time per wave-instruction per SIMD comes from hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ROWS(X)                                                         \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %0, %0")                       \
    X(1, "v_mul_f32", "v_mul_f32 %0, %0, %0")                           \
    X(2, "v_add_f32", "v_add_f32 %0, %0, %0")                           \
    X(3, "v_max_f32", "v_max_f32 %0, %0, %0")                           \
    X(4, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0")                 \
    X(5, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0")                       \
    X(6, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0")                       \
    X(7, "v_rcp_f32", "v_rcp_f32 %0, %0")                               \
    X(8, "v_mov_b32", "v_mov_b32 %0, %0")                               \
    X(9, "v_and_b32 vvv", "v_and_b32 %0, %0, %0")                       \
    X(10, "v_or_b32 vvv", "v_or_b32 %0, %0, %0")                        \
    X(11, "v_xor_b32 vvv", "v_xor_b32 %0, %0, %0")                      \
    X(12, "v_not_b32", "v_not_b32 %0, %0")                              \
    X(13, "v_add_u32 vvv", "v_add_u32 %0, %0, %0")                      \
    X(14, "v_sub_u32 vvv", "v_sub_u32 %0, %0, %0")                      \
    X(15, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0")                   \
    X(16, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0")                   \
    X(17, "v_max_u32", "v_max_u32 %0, %0, %0")                          \
    X(18, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %0, vcc")             \
    X(19, "v_cmp_gt_u32", "v_cmp_gt_u32 vcc, %0, %0")                   \
    X(20, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 9")                        \
    X(21, "v_and_or_b32", "v_and_or_b32 %0, %0, %0, %0")                \
    X(22, "v_or3_b32", "v_or3_b32 %0, %0, %0, %0")                      \
    X(23, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %0, %0 bitop3:0xc8")    \
    X(24, "v_add3_u32", "v_add3_u32 %0, %0, %0, %0")                    \
    X(25, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %0")             \
    X(26, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 4, %0")               \
    X(27, "v_max3_u32", "v_max3_u32 %0, %0, %0, %0")                    \
    X(28, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %0")                    \
    X(29, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %0")                    \
    X(30, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %0")                  \
    X(31, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %0, %0")              \
    X(32, "v_dot2_u32_u16", "v_dot2_u32_u16 %0, %0, %0, %0")            \
    X(33, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %0, %0")              \
    X(34, "v_lerp_u8", "v_lerp_u8 %0, %0, %0, %0")                      \
    X(35, "v_msad_u8", "v_msad_u8 %0, %0, %0, %0")                      \
    X(36, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %0")                    \
    X(37, "v_pk_sub_u16", "v_pk_sub_u16 %0, %0, %0")                    \
    X(38, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %0")              \
    X(39, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %0, %0")                \
    X(40, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 3, %0")             \
    X(41, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")   \
    X(42, "v_add_u32 dpp row_shr:1", "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(43, "v_add_u32 sdwa byte", "v_add_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(44, "v_max_u16", "v_max_u16 %0, %0, %0")                          \
    X(45, "v_min3_u32", "v_min3_u32 %0, %0, %0, %0")                    \
    X(46, "v_med3_u32", "v_med3_u32 %0, %0, %0, %0")                    \
    X(47, "v_sad_u8", "v_sad_u8 %0, %0, %0, %0")                        \
    X(48, "v_perm_b32", "v_perm_b32 %0, %0, %0, %0")                    \
    X(49, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %0, 1")           \
    X(50, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %0")                    \
    X(51, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %0")                    \
    X(52, "v_fma_f64", "v_fma_f64 %0, %0, %0, %0")                      \
    X(53, "v_mul_f64", "v_mul_f64 %0, %0, %0")                          \
    X(54, "v_add_f64", "v_add_f64 %0, %0, %0")                          \
    X(55, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %0, %0")

// Operand kinds: %1 is a second vector register, %2 a scalar register.
#define ROWS2(X)                                                        \
    X(100, "v_add_u32 v,v2", "v_add_u32 %0, %1, %0")                    \
    X(101, "v_add_u32 sgpr", "v_add_u32 %0, %2, %0")                    \
    X(102, "v_add_u32 inline 5", "v_add_u32 %0, 5, %0")                 \
    X(103, "v_add_u32 literal", "v_add_u32 %0, 0x7f7f7f7f, %0")         \
    X(104, "v_sub_u32 v,v2", "v_sub_u32 %0, %1, %0")                    \
    X(105, "v_and_b32 v,v2", "v_and_b32 %0, %1, %0")                    \
    X(106, "v_and_b32 literal", "v_and_b32 %0, 0x7f7f7f7f, %0")         \
    X(107, "v_or_b32 v,v2", "v_or_b32 %0, %1, %0")                      \
    X(108, "v_xor_b32 v,v2", "v_xor_b32 %0, %1, %0")                    \
    X(109, "v_lshlrev_b32 v,v2", "v_lshlrev_b32 %0, %1, %0")            \
    X(110, "v_lshrrev_b32 v,v2", "v_lshrrev_b32 %0, %1, %0")            \
    X(111, "v_lshrrev_b32 inline 7", "v_lshrrev_b32 %0, 7, %1")         \
    X(112, "v_lshlrev_b32 inline 4", "v_lshlrev_b32 %0, 4, %1")         \
    X(113, "v_max_u32 v,v2", "v_max_u32 %0, %1, %0")                    \
    X(114, "v_min_u32 v,v2", "v_min_u32 %0, %1, %0")                    \
    X(115, "v_max_u16 v,v2", "v_max_u16 %0, %1, %0")                    \
    X(116, "v_max_f32 v,v2", "v_max_f32 %0, %1, %0")                    \
    X(117, "v_add_f32 v,v2", "v_add_f32 %0, %1, %0")                    \
    X(118, "v_mul_f32 v,v2", "v_mul_f32 %0, %1, %0")                    \
    X(119, "v_fma_f32 v,v2,v", "v_fma_f32 %0, %1, %0, %0")              \
    X(120, "v_mov_b32 v2", "v_mov_b32 %0, %1")                          \
    X(121, "v_not_b32 v2", "v_not_b32 %0, %1")                          \
    X(122, "v_bitop3 v,v2,v", "v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc8") \
    X(123, "v_and_or v,v2,v", "v_and_or_b32 %0, %0, %1, %0")            \
    X(124, "v_lerp_u8 v,v2,sgpr", "v_lerp_u8 %0, %0, %1, %2")           \
    X(125, "v_lerp_u8 v,v2,v", "v_lerp_u8 %0, %0, %1, %0")              \
    X(126, "v_alignbyte v,v2", "v_alignbyte_b32 %0, %0, %1, 2")         \
    X(127, "v_cndmask v,v2", "v_cndmask_b32 %0, %0, %1, vcc")           \
    X(128, "v_sub_u32 v2 - v", "v_sub_u32 %0, %1, %0")                  \
    X(129, "v_subrev_u32 v,v2", "v_subrev_u32 %0, %1, %0")              \
    X(130, "v_addc_co_u32", "v_addc_co_u32 %0, vcc, %0, %1, vcc")       \
    X(131, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1")              \
    X(132, "v_mul_lo_u32 v,v2", "v_mul_lo_u32 %0, %0, %1")              \
    X(133, "v_cvt_f32_ubyte1 v2", "v_cvt_f32_ubyte1 %0, %1")            \
    X(134, "v_bfe_u32 v2", "v_bfe_u32 %0, %1, 8, 8")                    \
    X(135, "v_ashrrev_i32 v,v2", "v_ashrrev_i32 %0, 3, %1")             \
    X(136, "v_and_b32 sgpr", "v_and_b32 %0, %2, %0")                    \
    X(137, "v_sub_u32 sgpr - v2", "v_sub_u32 %0, %2, %1")               \
    X(138, "v_pk_add_u16 v,v2", "v_pk_add_u16 %0, %0, %1")              \
    X(139, "v_pk_sub_i16 v,v2", "v_pk_sub_i16 %0, %0, %1")              \
    X(140, "v_pk_max_i16 v,v2", "v_pk_max_i16 %0, %0, %1")              \
    X(141, "v_sub_u16 v,v2", "v_sub_u16 %0, %0, %1")                    \
    X(142, "v_add_u16 v,v2", "v_add_u16 %0, %0, %1")                    \
    X(143, "v_min_u16 v,v2", "v_min_u16 %0, %0, %1")                    \
    X(144, "v_lshrrev_b16", "v_lshrrev_b16 %0, 1, %1")                  \
    X(145, "v_mad_u16", "v_mad_u16 %0, %0, %1, %0")                     \
    X(146, "v_add_u32 sdwa b0", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
    X(147, "v_max_i32 v,v2", "v_max_i32 %0, %1, %0")                    \
    X(148, "v_cmp_gt_u32 + cndmask", "v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") \
    X(149, "lerp + add alternating", "v_lerp_u8 %0, %0, %1, %0\n\tv_add_u32 %0, %1, %0")           \
    X(150, "add + add (same reg chain)", "v_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0")          \
    X(151, "alignbyte + sub alternating", "v_alignbyte_b32 %0, %0, %1, 2\n\tv_sub_u32 %0, %1, %0") \
    X(152, "add + s_nop 0", "v_add_u32 %0, %1, %0\n\ts_nop 0")                                    \
    X(153, "add + s_mov (salu)", "v_add_u32 %0, %1, %0\n\ts_mov_b32 s20, 0")              \
    X(154, "lerp + s_mov (salu)", "v_lerp_u8 %0, %0, %1, %0\n\ts_mov_b32 s20, 0")

template <int KIND>
__global__ __launch_bounds__(1024) void k2(unsigned* out, int iters, unsigned seed) {
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) | 0x01010101u;
    unsigned s = __builtin_amdgcn_readfirstlane(seed * 77u + 3u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#define X(id, name, text) if (KIND == id) asm volatile(text : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "s"(s) : "vcc", "scc", "s20");
                ROWS2(X)
#undef X
            }
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// fully dependent chain: every instruction reads the result of the one before it
template <int KIND>
__global__ __launch_bounds__(1024) void k3(unsigned* out, int iters, unsigned seed) {
    unsigned a = threadIdx.x * 2654435761u + seed, b = a * 7u + 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
            if (KIND == 1) asm volatile("v_lerp_u8 %0, %0, %1, %0" : "+v"(a) : "v"(b));
            if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int KIND> void run3(const char* name, int blocks) {
    const int threads = 1024, iters = 2000;
    unsigned* d;
    if (hipMalloc(&d, (size_t)blocks * threads * 4) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int w = blocks / 256 * 4;
    const double ns = (double)ms * 1e6 / ((double)iters * 64.0 * w);
    printf("%-26s dependent chain, %d waves per SIMD: %7.3f ms   %.3f ns per wave-instruction per SIMD\n", name, w, ms, ns);
    hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
}

constexpr int kRows = 52;

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[16];
    unsigned long long b[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = ((unsigned long long)a[2 * i] << 32) | a[2 * i + 1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND >= 50) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
#define X(id, name, text) if (KIND == id && id >= 50) asm volatile(text : "+v"(b[i]));
                        ROWS(X)
#undef X
                    }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#define X(id, name, text) if (KIND == id && id < 50) asm volatile(text : "+v"(a[i]) : : "vcc");
                ROWS(X)
#undef X
            }
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= (unsigned)b[i] ^ (unsigned)(b[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int KIND> void run(const char* name) {
    // 512 blocks of 1024 threads: two blocks per CU, 8 waves per SIMD
    const int blocks = 512, threads = 1024, iters = 2000;
    unsigned* d;
    if (hipMalloc(&d, (size_t)blocks * threads * 4) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = (double)iters * 64.0 * 8;
    const double ns = (double)ms * 1e6 / instrPerSimd;
    printf("%-26s %7.3f ms   %.3f ns per wave-instruction per SIMD  (%.2f x v_fma_f32's 1.17)\n", name, ms, ns, ns / 1.17);
    hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KIND> void run2(const char* name, int per) {
    const int blocks = 512, threads = 1024, iters = 2000;
    unsigned* d;
    if (hipMalloc(&d, (size_t)blocks * threads * 4) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = (double)ms * 1e6 / ((double)iters * 64.0 * 8 * per);
    printf("%-26s %7.3f ms   %.3f ns per wave-instruction per SIMD\n", name, ms, ns);
    hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (argc < 2 || argv[1][0] == '1') {
#define X(id, name, text) run<id>(name);
        ROWS(X)
#undef X
    }
    for (int blocks : {256, 512}) { run3<0>("v_add_u32", blocks); run3<1>("v_lerp_u8", blocks); run3<2>("v_add_f32", blocks); }
#define X(id, name, text) run2<id>(name, id >= 148 && id <= 151 ? 2 : 1);
    ROWS2(X)
#undef X
    return 0;
}
