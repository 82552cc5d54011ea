// ORB extraction on gfx950: pyramid, per-cell FAST with threshold fallback, the fork's
// per-cell scale round trip, quadtree distribution, orientation, blurred steered-BRIEF
// descriptors and the lapping-area output order.  Restates ORBextractor::operator()
// (src/ORBextractor.cc:1271-1399) and everything it calls, batched over frames.
//
// Kernel inventory (one launch each per batch, all frames and levels inside the grid,
// except the pyramid whose levels depend on each other):
//   k_resize     level l from level l-1       cv::resize INTER_LINEAR   (:1419)
//   k_fast       one workgroup per FAST cell  cv::FAST + fallback       (:960-970) + round trip (:989-1025)
//   k_quadtree   one workgroup per level      DistributeOctTree         (:661-883)
//   k_layout     one workgroup per frame      output order / monoIndex  (:1329-1398)
//   k_describe   one wave per keypoint        IC_Angle (:90-137), GaussianBlur (:1347),
//                                             computeOrbDescriptor (:150-213)
// All integer / float steps are written so that they round exactly like the scalar C++
// code (no fused multiply-add: this file is compiled with -ffp-contract=off and uses the
// explicit _rn intrinsics where the order of roundings matters).

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "orbx_common.h"
#include "orbx_wavesort.h"

// ======================================================================================
// error string
// ======================================================================================
static thread_local char g_err[512] = "";
void orbx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* orbx_last_error(void) { return g_err; }
extern "C" int orbx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        orbx_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return ORBX_ERR_HIP;
    }
    return n;
}

// ======================================================================================
// device tables
// ======================================================================================
__constant__ __align__(16) signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};
// the same offsets as binary32 (the steered tests rotate them in float): a lane's 8 points are 4 x 16 bytes
__constant__ __align__(16) float c_patternF[1024] = {
#include "orb_pattern.inc"
};
// umax[v]: half width of row v of the radius-15 disc (ORBextractor.cc:520-535)
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin in launch order.
// xcd_remap turns the linear block id into a logical id such that every XCD works on one
// contiguous run of logical ids, so blocks that touch neighbouring data share an L2.
// A bijection for any grid size; placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int lin, int nBlocks) {
    const int q = nBlocks >> 3, r = nBlocks & 7, xcd = lin & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
}

struct OrbxErr {
    int cand_overflow;   // a level produced more FAST candidates than candCap
    int kp_overflow;     // quadtree list / level keypoints exceeded their capacity
    int out_overflow;    // a frame produced more keypoints than the caller's capacity
    int need_cand;       // largest candidate count seen (for regrowing)
};

// ======================================================================================
// k_resize: 8-bit bilinear resize in 11-bit fixed point, 4 output pixels per lane.
// The 8 source bytes that the lane's 4 outputs touch in one source row (scale factors up to 2) are
// cut out of three aligned dwords; one v_perm_b32 per tap gathers the four left / the four right
// neighbours, so a source row costs two permutes instead of eight byte extractions.  The taps come
// as one int4 of source offsets and one uint4 of packed (a0 | a1 << 16) coefficients.
// ======================================================================================
// Threads per workgroup of the level-0 copy and the resize kernels, chosen per image size.  Pipelined benchmark, three
// runs each: 752x480 with 64 / 128 / 256 threads 275 / 288 / 283 k frames/s (one step at a time 248 / 252 / 237 k);
// 1920x1080 with 128 / 256 threads 51.5 / 53.1 k.
constexpr int kPyrThreadsMax = 256;
static inline int pyr_threads_for(int w, int h) { return (long long)w * h <= (1ll << 20) ? 128 : 256; }
constexpr int kResizeRows = 4;   // output rows per lane: taps are fetched once, source rows are reused

typedef unsigned short us2v __attribute__((ext_vector_type(2)));
// the three aligned dwords that hold a lane's 8-byte source window of one row
struct ResizeRaw { uint32_t d0, d1, d2; };
__device__ __forceinline__ ResizeRaw resize_load(const uint8_t* __restrict__ row, int base) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(row + base);
    ResizeRaw r;
    r.d0 = p[0]; r.d1 = p[1]; r.d2 = p[2];
    return r;
}
// Horizontal pass of one source row for the lane's four outputs: (p0 a0 + p1 a1) >> 4 each.  One v_perm_b32 per
// output puts its left and right neighbour into the two halves of a dword ([p0, 0, p1, 0], selector byte 0x0C = zero),
// and v_dot2_u32_u16 against the coefficient pair (a0 | a1 << 16, as the tap table stores it) is the whole sum.
__device__ __forceinline__ void resize_hsum(const ResizeRaw r, int sh, const uint32_t (&sel)[4], const uint32_t (&co)[4],
                                            uint32_t (&T)[4]) {
    const uint32_t w0 = __builtin_amdgcn_alignbyte(r.d1, r.d0, sh), w1 = __builtin_amdgcn_alignbyte(r.d2, r.d1, sh);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        T[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, __builtin_amdgcn_perm(w1, w0, sel[i])),
                                      __builtin_bit_cast(us2v, co[i]), 0u, false) >> 4;
}

__global__ __launch_bounds__(kPyrThreadsMax) void k_resize(const OrbxFrameGeo* __restrict__ geo, int level,
                                                uint8_t* __restrict__ pyr, long long frameStride,
                                                const int* __restrict__ xofs, const uint32_t* __restrict__ xcoef,
                                                const int* __restrict__ yofs, const uint32_t* __restrict__ ycoef,
                                                int lanesPerRow, uint32_t lanesPerRowRcp) {
    const OrbxLevelGeo& S = geo->lv[level - 1];
    const OrbxLevelGeo& D = geo->lv[level];
    const int fz = blockIdx.y;
    // (group of kResizeRows rows, 4-pixel column) as one sequence, so that the workgroups are full whatever the
    // level's width; lanesPerRowRcp = floor(2^32 / lanesPerRow), one correction step makes the quotient exact
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t grp = __umulhi(i, lanesPerRowRcp);
    uint32_t col = i - grp * (uint32_t)lanesPerRow;
    if (col >= (uint32_t)lanesPerRow) { ++grp; col -= (uint32_t)lanesPerRow; }
    const int yFirst = (int)grp * kResizeRows;
    const int x4 = (int)col * 4;
    if (yFirst >= D.h) return;
    const uint8_t* src = pyr + (long long)fz * frameStride + S.off;
    uint8_t* dst = pyr + (long long)fz * frameStride + D.off;
    const int4 o = *reinterpret_cast<const int4*>(xofs + D.tapX + x4);        // padded to a multiple of 4
    const uint4 c = *reinterpret_cast<const uint4*>(xcoef + D.tapX + x4);
    const int so[4] = {o.x, o.y, o.z, o.w};
    const uint32_t co[4] = {c.x, c.y, c.z, c.w};
    int a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0[i] = (int)(co[i] & 0xFFFFu); a1[i] = (int)(co[i] >> 16); }
    // byte positions inside the 8-byte window that starts at source column so[0]
    const int last = min(o.w + 1, S.w - 1);
    const bool window = last - o.x < 8;
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        sel[i] = 0x0C000C00u | (uint32_t)((so[i] - o.x) & 7) | ((uint32_t)((min(so[i] + 1, S.w - 1) - o.x) & 7) << 16);
    const int base = o.x & ~3, sh = o.x & 3;

    // Vertical taps of the lane's rows (wave-uniform), then every source row the rows need is requested before
    // any of them is used: the kernel's time is the memory latency, so the loads of all rows are in flight
    // together.  A row whose upper source row is the previous row's lower one reuses it (no load).
    int sy0[kResizeRows], sy1[kResizeRows], b0[kResizeRows], b1[kResizeRows];
    bool live[kResizeRows], need0[kResizeRows], need1[kResizeRows];
    static_assert(kResizeRows == 4, "the vertical taps of a row group are one int4 / uint4");
    const int4 oy = *reinterpret_cast<const int4*>(yofs + D.tapY + yFirst);       // padded to a multiple of 4
    const uint4 cy = *reinterpret_cast<const uint4*>(ycoef + D.tapY + yFirst);
    const int syv[4] = {oy.x, oy.y, oy.z, oy.w};
    const uint32_t ycv[4] = {cy.x, cy.y, cy.z, cy.w};
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
        live[r] = yFirst + r < D.h;
        sy0[r] = syv[r];
        sy1[r] = min(sy0[r] + 1, S.h - 1);
        b0[r] = (int)(ycv[r] & 0xFFFFu); b1[r] = (int)(ycv[r] >> 16);
        need0[r] = live[r] && (r == 0 || sy0[r] != sy1[r - 1]);
        need1[r] = live[r] && sy1[r] != sy0[r];
    }
    if (window) {
        ResizeRaw raw0[kResizeRows], raw1[kResizeRows];
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            // (a row's window is only read when it was requested: no initialisation)
            if (need0[r]) raw0[r] = resize_load(src + (long long)sy0[r] * S.pitch, base);
            if (need1[r]) raw1[r] = resize_load(src + (long long)sy1[r] * S.pitch, base);
        }
        // horizontal sums (p0 a0 + p1 a1) >> 4 of a source row are computed once: the lower row of one output row is
        // the upper row of the next whenever the vertical taps say so
        uint32_t prevT[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            if (!live[r]) break;
            uint32_t T0[4], T1[4];
            if (need0[r]) {
                resize_hsum(raw0[r], sh, sel, co, T0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) T0[i] = prevT[i];
            }
            if (need1[r]) {
                resize_hsum(raw1[r], sh, sel, co, T1);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) T1[i] = T0[i];
            }
            // (b T) >> 16 as the high half of (b << 16) T
            const uint32_t b0s = (uint32_t)b0[r] << 16, b1s = (uint32_t)b1[r] << 16;
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                prevT[i] = T1[i];
                const uint32_t v = (__umulhi(b0s, T0[i]) + __umulhi(b1s, T1[i]) + 2u) >> 2;     // <= 255
                packed |= v << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(dst + (long long)(yFirst + r) * D.pitch + x4) = packed;   // pitch padding absorbs the tail
        }
    } else {
        for (int r = 0; r < kResizeRows; ++r) {
            if (!live[r]) break;
            const uint8_t* r0 = src + (long long)sy0[r] * S.pitch;
            const uint8_t* r1 = src + (long long)sy1[r] * S.pitch;
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sx0 = so[i], sx1 = min(sx0 + 1, S.w - 1);
                const int t0 = r0[sx0] * a0[i] + r0[sx1] * a1[i];
                const int t1 = r1[sx0] * a0[i] + r1[sx1] * a1[i];
                const int v = (((b0[r] * (t0 >> 4)) >> 16) + ((b1[r] * (t1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 255) << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(dst + (long long)(yFirst + r) * D.pitch + x4) = packed;
        }
    }
}

// Copies n_frames images (row_stride / frame_stride) into level 0 of each pyramid slab,
// 16 bytes per lane when the source is 16-byte aligned, bytes otherwise.  The first workgroup of every frame also
// clears the frame's FAST candidate counters (the only per-batch state that must start at zero), which saves a
// separate memset on the stream.
// The FAST candidate counters, one per (frame, level), lie kCandCounterStride ints apart: every cell of a level adds to
// its counter once, and same-line atomics are served one after the other (about 9 ns each: with the 8 counters of a
// frame -- and those of the next frame -- in one line, the 4620 cells of a 1920x1080 frame took 48 us through that
// queue, two frames 84 us, whatever else the kernel did).
constexpr int kCandCounterStride = 32;
__global__ __launch_bounds__(kPyrThreadsMax) void k_copy_level0(const OrbxFrameGeo* __restrict__ geo,
                                                     const uint8_t* __restrict__ images, long long rowStride,
                                                     long long imgStride, uint8_t* __restrict__ pyr,
                                                     long long frameStride, int aligned16, int chunksPerRow,
                                                     int* __restrict__ candCount) {
    const OrbxLevelGeo& D = geo->lv[0];
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)geo->nlevels) candCount[(blockIdx.y * geo->nlevels + threadIdx.x) * kCandCounterStride] = 0;
    // the 16-byte chunks of a frame as one sequence: full workgroups whatever the image width
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = i / chunksPerRow;
    if (y >= D.h) return;
    const int x = (i - y * chunksPerRow) * 16;
    const uint8_t* src = images + (long long)blockIdx.y * imgStride + (long long)y * rowStride + x;
    uint8_t* dst = pyr + (long long)blockIdx.y * frameStride + D.off + (long long)y * D.pitch + x;
    if (aligned16 && x + 16 <= D.w) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
        const int n = min(16, D.w - x);
        for (int k = 0; k < n; ++k) dst[k] = src[k];
    }
}

// ======================================================================================
// k_fast: FAST-9/16 + 3x3 non-maximum suppression on one ~35-px cell, decided twice
// (initial threshold, then the lower one if the cell stayed empty), followed by the
// fork's repeated level<->image scale round trip of every emitted corner.
// ======================================================================================
// LDS tile geometry is fixed per handle: pitch P (multiple of 4, >= widest cell image + 3 for
// the alignment shift) and R rows; both come from the level geometry (OrbxFrameGeo::tileP/R).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 pk_max(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ us2 pk_min(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ us2 pk_swap(us2 a) { return __builtin_shufflevector(a, a, 1, 0); }

// Arc strength of the pixel at (x, y): with p[k] the 16 circle pixels and v the centre,
//   bright strength = max over the 16 arcs of 9 contiguous pixels of min(p) - v
//   dark strength   = v - min over the arcs of max(p)  =  the bright strength of the inverted values 255 - p, 255 - v
// and cornerScore<16> before its "- 1" is the larger of the two.  Only strengths above `floorTh` matter to the
// caller (anything else is "no corner"), and an arc of 9 contains one pixel of every antipodal pair, so a polarity
// can exceed floorTh only if one of p0 / p8 and one of p4 / p12 differ from the centre by more than floorTh in its
// direction.  That test picks the polarity to evaluate (the inversion makes one code path serve both); a pixel
// that passes for both polarities (about one survivor of the quick reject in 300) is evaluated twice, one that
// passes for neither returns 0.
// Sliding min by doubling (2, 4, 8, then 9) on packed 16-bit pairs: register k holds (p[k], p[k+8]), so one packed
// op advances arcs k and k+8 together; the second half of the cyclic sequence is the same registers with their halves
// swapped.
__device__ __forceinline__ int fast_arc_min_max(const us2 (&P8)[8], uint32_t inv) {
    us2 E[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) E[k] = __builtin_bit_cast(us2, __builtin_bit_cast(uint32_t, P8[k]) ^ inv);
#pragma unroll
    for (int k = 0; k < 8; ++k) E[8 + k] = pk_swap(E[k]);
    us2 lo2[10], lo4[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) lo2[k] = pk_min(E[k], E[k + 1]);
    lo2[8] = pk_swap(lo2[0]); lo2[9] = pk_swap(lo2[1]);
#pragma unroll
    for (int k = 0; k < 8; ++k) lo4[k] = pk_min(lo2[k], lo2[k + 2]);
#pragma unroll
    for (int k = 0; k < 4; ++k) lo4[8 + k] = pk_swap(lo4[k]);
    us2 B = us2{0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const us2 lo9 = pk_min(pk_min(lo4[k], lo4[k + 4]), E[k + 8]);   // min of p[k .. k+8] | p[k+8 .. k+16]
        B = pk_max(B, lo9);
    }
    return max((int)B.x, (int)B.y);
}

__device__ __forceinline__ int fast_strength_lds(const uint8_t* t, int P, int x, int y, int floorTh) {
    // addressed from the top-left corner of the 7x7 neighbourhood: every offset is non-negative
    // (an immediate when the pitch is a compile-time constant)
    const uint8_t* c = t + (y - 3) * P + (x - 3);
#define ORBX_PX(dx, dy) c[(3 + (dy)) * P + 3 + (dx)]
    const int v = ORBX_PX(0, 0);
    us2 E[8];
    E[0] = us2{ORBX_PX(0, 3), ORBX_PX(0, -3)};     E[1] = us2{ORBX_PX(1, 3), ORBX_PX(-1, -3)};
    E[2] = us2{ORBX_PX(2, 2), ORBX_PX(-2, -2)};    E[3] = us2{ORBX_PX(3, 1), ORBX_PX(-3, -1)};
    E[4] = us2{ORBX_PX(3, 0), ORBX_PX(-3, 0)};     E[5] = us2{ORBX_PX(3, -1), ORBX_PX(-3, 1)};
    E[6] = us2{ORBX_PX(2, -2), ORBX_PX(-2, 2)};    E[7] = us2{ORBX_PX(1, -3), ORBX_PX(-1, 3)};
#undef ORBX_PX
    const int hiV = max((int)E[0].x, (int)E[0].y), hiH = max((int)E[4].x, (int)E[4].y);
    const int loV = min((int)E[0].x, (int)E[0].y), loH = min((int)E[4].x, (int)E[4].y);
    const bool bright = min(hiV, hiH) - v > floorTh, dark = v - max(loV, loH) > floorTh;
    int st = 0;
    if (bright || dark) st = fast_arc_min_max(E, bright ? 0u : 0x00FF00FFu) - (bright ? v : 255 - v);
    if (bright && dark) st = max(st, fast_arc_min_max(E, 0x00FF00FFu) - (255 - v));
    return st;
}

// Rows per band when `lanes` lanes share a cell of ncol dword columns and ih testable rows, one lane per
// (band, column).  Host and kernel use the same arithmetic.
__host__ __device__ __forceinline__ int fast_band_rows(int ncol, int ih, int lanes) {
    const int q = lanes / ncol;
    const int bands = q < 1 ? 1 : (q < ih ? q : ih);
    return (ih + bands - 1) / bands;
}

// Sum of a value over the 64 lanes of the wave, returned in every lane (as a scalar): four cross-lane adds inside the
// rows of 16 lanes, two row broadcasts, one read of lane 63 -- data-parallel-primitive moves, no LDS round trips.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Inclusive prefix sum over the 64 lanes of a wave (all lanes must be active): shifts by 1, 2, 4, 8 inside the rows of
// 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ int score_at(int strength, int th) { return strength > th ? strength - 1 : 0; }

// cv::Rect2i::contains(Point2i(pt)): the float point converts with round-half-even.
__device__ __forceinline__ bool in_any_rect(float X, float Y, const orbx_rect* rects, int nrects) {
    const int px = __float2int_rn(X), py = __float2int_rn(Y);
    for (int r = 0; r < nrects; ++r) {
        const orbx_rect R = rects[r];
        if (R.x <= px && px < R.x + R.width && R.y <= py && py < R.y + R.height) return true;
    }
    return false;
}

constexpr int kFastThreads = 128;
#ifndef ORBX_QR_SHIFT
#define ORBX_QR_SHIFT 1
#endif
// The quick reject of k_fast compares pixels without their kQrShift low bits (see there).
constexpr int kQrShift = ORBX_QR_SHIFT;
constexpr uint32_t kQrMask = (0xFFu >> kQrShift) * 0x01010101u;

// Diagnostic build only (-DORBX_FAST_STAMPS, tools/prof_fast_phases.py): wave 0 of every workgroup adds the shader
// cycles it spent between the phase boundaries of k_fast to a device array; the product library contains none of it.
#ifdef ORBX_FAST_STAMPS
constexpr int kStampWgs = 1 << 18;
__device__ unsigned int g_fastStamps[kStampWgs][8];     // per workgroup: 5 phase durations, a "ran" mark, survivors of the quick reject, corners of wave 0
#define ORBX_STAMP(slot)                                                                   \
    do {                                                                                   \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                     \
        if (threadIdx.x == 0 && stampWg < kStampWgs) g_fastStamps[stampWg][slot] = (unsigned int)(now__ - stampPrev); \
        stampPrev = now__;                                                                 \
    } while (0)
#define ORBX_STAMP_BEGIN()                                                                 \
    const int stampWg = blockIdx.y * gridDim.x + blockIdx.x;                               \
    if (threadIdx.x == 0 && stampWg < kStampWgs) g_fastStamps[stampWg][5] = 1u;            \
    unsigned long long stampPrev = __builtin_amdgcn_s_memtime()
// Sums over the workgroups that ran since the last reset: out8[0..4] = cycles per phase, out8[7] = workgroups.
extern "C" int orbx_debug_fast_stamps(unsigned long long* out8, int reset) {
    std::vector<unsigned int> h((size_t)kStampWgs * 8);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fastStamps), h.size() * 4) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (int w = 0; w < kStampWgs; ++w) {
        if (!h[(size_t)w * 8 + 5]) continue;
        for (int k = 0; k < 5; ++k) out8[k] += h[(size_t)w * 8 + k];
        out8[5] += h[(size_t)w * 8 + 6];
        out8[6] += h[(size_t)w * 8 + 7];
        ++out8[7];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0u);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fastStamps), h.data(), h.size() * 4) != hipSuccess) return -1;
    }
    return 0;
}
// k_describe: per wave
__device__ unsigned int g_descStamps[kStampWgs][6];
#define ORBX_DSTAMP(slot)                                                                  \
    do {                                                                                   \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                     \
        if (lane == 0 && dstampId < kStampWgs) g_descStamps[dstampId][slot] = (unsigned int)(now__ - dstampPrev); \
        dstampPrev = now__;                                                                \
    } while (0)
#define ORBX_DSTAMP_BEGIN()                                                                \
    const int dstampId = (blockIdx.y * gridDim.x + blockIdx.x) * kDescWaves + wv;          \
    if (lane == 0 && dstampId < kStampWgs) g_descStamps[dstampId][5] = active ? 1u : 0u;   \
    unsigned long long dstampPrev = __builtin_amdgcn_s_memtime()
extern "C" int orbx_debug_desc_stamps(unsigned long long* out8, int reset) {
    std::vector<unsigned int> h((size_t)kStampWgs * 6);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_descStamps), h.size() * 4) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (int w = 0; w < kStampWgs; ++w) {
        if (!h[(size_t)w * 6 + 5]) continue;
        for (int k = 0; k < 5; ++k) out8[k] += h[(size_t)w * 6 + k];
        ++out8[7];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0u);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_descStamps), h.data(), h.size() * 4) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define ORBX_STAMP(slot) do {} while (0)
#define ORBX_STAMP_BEGIN() do {} while (0)
#define ORBX_DSTAMP(slot) do {} while (0)
#define ORBX_DSTAMP_BEGIN() do {} while (0)
#endif
// Diagnostic builds only (-DORBX_FAST_STOP=n / -DORBX_DESC_STOP=n): the kernel ends after its n-th phase, so that the
// instruction counters of successive builds differ by exactly one phase.  Results are meaningless in such a build.
// (-DORBX_ISA_MARKERS, assembly listings only: a comment line at every phase boundary, for tools/isa_phase_table.py)
#if defined(ORBX_ISA_MARKERS)
#define ORBX_FAST_PHASE_END(n) asm volatile("; orbx-phase-end fast " #n ::: "memory")
#elif defined(ORBX_FAST_STOP)
#define ORBX_FAST_PHASE_END(n) do { if (ORBX_FAST_STOP == (n)) return; } while (0)
#else
#define ORBX_FAST_PHASE_END(n) do {} while (0)
#endif
#if defined(ORBX_ISA_MARKERS)
#define ORBX_DESC_PHASE_END(n, value) asm volatile("; orbx-phase-end describe " #n ::: "memory")
#elif defined(ORBX_DESC_STOP)
// (the value keeps what the phases so far have produced alive: without a consumer the compiler drops them)
#define ORBX_DESC_PHASE_END(n, value) do { if (ORBX_DESC_STOP == (n)) { if (active) outDesc[((long long)frame * capacity + G.slot) * 32 + (lane & 31)] = (uint8_t)(value); return; } } while (0)
#else
#define ORBX_DESC_PHASE_END(n, value) do {} while (0)
#endif

// kP = LDS tile pitch known at compile time (row steps and the 3-row look-ahead become
// immediate offsets), or 0 to take it from the geometry.
// Diagnostic build only (-DORBX_EXTRA_SALU=n): n more scalar moves in every row step of the quick reject, to see
// how the kernel's time follows its scalar instruction count (tools/extra_salu.sh).
#ifdef ORBX_EXTRA_SALU
#define ORBX_EXTRA_SALU_PER_ROW { int sdummy; _Pragma("unroll") for (int e = 0; e < ORBX_EXTRA_SALU; ++e) asm volatile("s_mov_b32 %0, 0x1234" : "=s"(sdummy)); }
#else
#define ORBX_EXTRA_SALU_PER_ROW
#endif
template <int kP>
__global__ __launch_bounds__(kFastThreads) void k_fast(const OrbxFrameGeo* __restrict__ geo,
                                                       const OrbxFastCell* __restrict__ cells,
                                                       const uint8_t* __restrict__ pyr,
                                                       long long frameStride, OrbxCand* __restrict__ cands,
                                                       int* __restrict__ candCount, int iniTh, int minTh,
                                                       const orbx_rect* __restrict__ rects, int nrects,
                                                       OrbxErr* __restrict__ err, int tileP, int tileR, int queueOff,
                                                       int nl, unsigned int cellMagic) {
    // Dynamic LDS: image tile [R][P] (stored from the 4-byte aligned column at or left of the
    // cell so that rows are fetched with dword loads), strength tile [R][P], the image tile again with every
    // pixel shifted right by kQrShift bits (what the quick reject compares), and the queue of
    // pixels that survive the quick reject (one ushort per testable pixel), later compacted in
    // place to those that are corners.  The queue starts where the host put it: on top of the coarse tile when
    // the quick reject reads all of that tile before it queues anything, behind it otherwise.  After the strengths are known the image tile is reused for
    // the list of corners that survive the non-maximum suppression.
    extern __shared__ __align__(16) unsigned char fast_smem[];
    __shared__ int s_nQueue, s_nIni, s_nMin, s_nKeep;
    const int P = kP ? kP : tileP, R = tileR;   // tile geometry and queue offset: OrbxFrameGeo::tileP / tileR / fastQueueOff
    uint8_t* tile = fast_smem;
    uint8_t* strength = tile + P * R;
    uint8_t* coarse = strength + P * R;
    unsigned short* s_queue = reinterpret_cast<unsigned short*>(fast_smem + queueOff);

    // neighbouring cells overlap by 6 pixels: keep them on one XCD (see xcd_remap)
    const int nCellsAll = gridDim.x;
    const int logical = xcd_remap(blockIdx.y * nCellsAll + blockIdx.x, nCellsAll * gridDim.y);
    // logical / nCellsAll through the host's reciprocal (exact for this grid, see enqueue_batch)
    const int frame = cellMagic ? (int)__umulhi((unsigned int)logical, cellMagic) : logical / nCellsAll;
    const int cellAll = logical - frame * nCellsAll;
    const OrbxFastCell CI = cells[cellAll];
    const int x0 = CI.xy & 0xFFFF, y0 = CI.xy >> 16;
    const int tw = CI.dims & 255, th = (CI.dims >> 8) & 255, level = CI.dims >> 16;
    const int cell = CI.index;
    const int levelPitch = CI.pitch;
    const int tid = threadIdx.x;
    const int xa = x0 & ~3, ox = x0 - xa;

    ORBX_STAMP_BEGIN();
    if (tid == 0) { s_nQueue = 0; s_nIni = 0; s_nMin = 0; s_nKeep = 0; }
    const uint8_t* tileSrc = pyr + (long long)frame * frameStride + CI.tileOff;   // row y0, column xa of the level
    {
        // lane -> (row, 16-byte chunk of the row): the tile pitch is a multiple of 16 when it is the compile-time
        // one, and a multi-dword global load only needs dword alignment.  Chunks past the cell's columns are
        // fetched too (inside the level's slab, never used) -- one load and two LDS stores per lane and pass.
        if ((P & 15) == 0) {
            const int RC = P >> 4;                                   // chunks per tile row
            const int rowsPerPass = kFastThreads / RC;
            const int row = tid / RC, xc = tid - row * RC;
            struct __attribute__((packed, aligned(4))) Chunk16 { uint32_t a, b, c, d; };
            if (row < rowsPerPass) {
                const uint8_t* src = tileSrc + (long long)row * levelPitch + 16 * xc;
                uint8_t* dstT = tile + row * P + 16 * xc;
                const int planeBytes = P * R;
                const long long srcStep = (long long)rowsPerPass * levelPitch;
                const int dstStep = rowsPerPass * P;
                for (int y = row; y < th; y += rowsPerPass) {
                    const Chunk16 v = *reinterpret_cast<const Chunk16*>(src);
                    *reinterpret_cast<uint4*>(dstT) = make_uint4(v.a, v.b, v.c, v.d);
                    *reinterpret_cast<uint4*>(dstT + planeBytes) = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(dstT + 2 * planeBytes) =
                        make_uint4((v.a >> kQrShift) & kQrMask, (v.b >> kQrShift) & kQrMask, (v.c >> kQrShift) & kQrMask,
                                   (v.d >> kQrShift) & kQrMask);
                    src += srcStep; dstT += dstStep;
                }
            }
        } else {
            // lane -> (row, dword of the row); kFastThreads / (P / 4) rows per pass
            const int RD = P >> 2;
            const int rowsPerPass = kFastThreads / RD;
            const int row = tid / RD, xd = tid - row * RD;
            // (whole rows, like the 16-byte path: the quick reject needs every byte of a row to be a pixel)
            if (row < rowsPerPass) {
                const uint8_t* src = tileSrc + (long long)row * levelPitch + 4 * xd;
                uint8_t* dstT = tile + row * P + 4 * xd;
                const int planeBytes = P * R;
                const long long srcStep = (long long)rowsPerPass * levelPitch;
                const int dstStep = rowsPerPass * P;
                for (int y = row; y < th; y += rowsPerPass) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(src);
                    *reinterpret_cast<uint32_t*>(dstT) = v;
                    *reinterpret_cast<uint32_t*>(dstT + planeBytes) = 0u;
                    *reinterpret_cast<uint32_t*>(dstT + 2 * planeBytes) = (v >> kQrShift) & kQrMask;
                    src += srcStep; dstT += dstStep;
                }
            }
        }
    }
    __syncthreads();
    ORBX_STAMP(0);   // tile load
    ORBX_FAST_PHASE_END(1);

    const int iw = tw - 6, ih = th - 6;   // pixels FAST can test
    const int lowTh = min(iniTh, minTh);
    uint8_t* T = tile + ox;               // T[y * P + x], x in cell coordinates
    uint8_t* S = strength + ox;
    if (iw > 0 && ih > 0) {
        // (A) quick reject.  An arc of 9 contiguous circle pixels contains one pixel of every
        // antipodal pair, so a corner needs, on one side of the (lower) threshold, a member of each of
        // {0,8} (3 rows below / above), {4,12} (3 columns right / left) and the diagonal pairs {2,10}, {6,14}.
        // (Two pairs let about a quarter of a textured image through, four pairs about a third of that, and
        // every survivor costs a full arc evaluation in (B).)
        // The test runs on the coarse tile, n' = n >> kQrShift per byte: n - c >= t implies n' - c' >= t >> kQrShift
        // (floor is superadditive), so comparing the coarse values against tq = t >> kQrShift loses no corner and
        // lets through a few pixels whose difference is just below the threshold; (B) evaluates survivors exactly.
        // Coarse bytes have spare high bits, which makes a byte-wise comparison ONE 32-bit add or subtract on four
        // pixels with the verdict in bit 7 of each byte and no carry between bytes:
        //   brighter:  n' - c' >= tq  <=>  bit 7 of n' + Cb,  Cb = max(128 - tq - c', 0)
        //   darker:    c' - n' >= tq  <=>  bit 7 of Cd - n',  Cd = max(c' + 128 - tq, 127)
        // (plain 32-bit add / subtract / logic with register or literal operands issue at twice the rate of the byte
        // instructions; the constants therefore live in vector registers).  Cb and Cd are per centre pixel; with
        // kQrShift = 2 no clamp is needed because 6-bit values cannot leave the byte.
        // The testable pixels are cut into horizontal bands; a lane walks down one dword column of one
        // band with the 7 rows (y-3 .. y+3) of that column and of its two neighbours in registers, so each
        // further 4 pixels cost three LDS dword reads.  Survivors are collected in a per-lane bit mask and queued at the end.
        // One wave does the whole cell when its bands fit one 8-row pass (fewer instructions than two waves with
        // shorter bands); cells whose bands would be taller are shared by both waves.
        // (the lane layout comes with the cell record: columns, bands, rows per band, one wave or both)
        const int dFirst = (ox + 3) >> 2;                                     // first tile dword column with testable pixels
        const int ncol = CI.bands & 255;
        const bool bothWaves = (CI.bands >> 24) != 0;
        const int qrLanes = bothWaves ? 128 : 64;
        if (tid < qrLanes) {
            const int t1 = lowTh + 1;                                  // a - b >= t1
            const int tq = min(t1 >> kQrShift, (int)(0xFFu >> kQrShift));
            const bool passAll = tq < 1;                               // thresholds below the coarse step: everything survives
            uint32_t KB0 = (uint32_t)(128 - max(tq, 1)) * 0x01010101u;     // 128 - tq per byte
            uint32_t KB0H = KB0 | 0x80808080u;
            asm volatile("" : "+v"(KB0), "+v"(KB0H));                  // vector registers: a scalar operand halves the issue rate
            const int bands = (CI.bands >> 8) & 255;
            const int bandRows = (CI.bands >> 16) & 255;
            const int band = (tid * CI.colRcp) >> 16, d = dFirst + (tid - band * ncol);
            const bool laneOn = band < bands;
            const int yBeg = 3 + band * bandRows;
            const int yEnd = laneOn ? min(yBeg + bandRows, 3 + ih) : 0;
            // bytes of this lane's dword that are testable pixels (cell x = 4 d + k - ox in [3, 3 + iw))
            uint32_t valid = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * d + k - ox;
                if (x >= 3 && x < 3 + iw) valid |= 0x80u << (8 * k);
            }
            for (int chunk = 0; chunk < bandRows; chunk += 8) {
                const int yc = yBeg + chunk;
                const int rows = min(8, bandRows - chunk);             // uniform
                const uint32_t* c = reinterpret_cast<const uint32_t*>(coarse + (laneOn ? yc : 3) * P) + (laneOn ? d : 1);
                const int PD = P >> 2;                                 // row pitch in dwords
                // the dword columns left and right of the lane's own, at fixed offsets from it.  At the ends of a row
                // they are the neighbouring row's dwords (or, around the tile, the strength tile's last dword and the 16
                // spare bytes behind the coarse tile): such a column is only ever the source of bytes that no testable
                // pixel of this lane uses, since a testable pixel's whole circle lies inside its tile row.  What those
                // bytes hold still matters on the LEFT: they land below the used bytes of the shifted dword, and a byte
                // above 127 there would carry into them.  Every byte the left column can come from is a coarse pixel:
                // the tile load fills whole rows, and the rows whose left column is used start at row 1.
                constexpr int dl = -1, dr = 1;
                // rows y-3 .. y+3 of the three columns, rotated through registers
                uint32_t C0 = c[-3 * PD], C1 = c[-2 * PD], C2 = c[-PD], C3 = c[0], C4 = c[PD], C5 = c[2 * PD], C6 = c[3 * PD];
                uint32_t L0 = c[-3 * PD + dl], L1 = c[-2 * PD + dl], L2 = c[-PD + dl], L3 = c[dl], L4 = c[PD + dl],
                         L5 = c[2 * PD + dl], L6 = c[3 * PD + dl];
                uint32_t R0 = c[-3 * PD + dr], R1 = c[-2 * PD + dr], R2 = c[-PD + dr], R3 = c[dr], R4 = c[PD + dr],
                         R5 = c[2 * PD + dr], R6 = c[3 * PD + dr];
                // Row yc + i of the chunk sets bit 8 - rows + i of byte k for pixel k of the dword: the mask moves down
                // one bit per row and the new verdicts enter at bit 7 of each byte (at most 8 rows, so no bit leaves
                // its byte).
                uint32_t mask = 0;
                // circle pixels used: 0 / 8 (0, +-3), 4 / 12 (+-3, 0), 2 / 10 (+2, +2) / (-2, -2), 6 / 14 (+2, -2) / (-2, +2)
#define ORBX_FAST_ROW4(i0, i1, i2, i3, i4, i5, i6, r)                                                       \
                if ((r) < rows) {                                                                           \
                    uint32_t Cb, Cd;                                                                        \
                    if (kQrShift >= 2) {                                                                    \
                        Cb = KB0 - C##i3;                                                                   \
                        Cd = C##i3 + KB0;                                                                   \
                    } else {                                                                                \
                        const uint32_t V = KB0H - C##i3;              /* 256 - tq - c': bit 7 = "not negative" */ \
                        const uint32_t mB = V & 0x80808080u;                                                \
                        Cb = V & (mB - (mB >> 7));                                                          \
                        const uint32_t W = C##i3 + KB0;                                                     \
                        const uint32_t mD = W & 0x80808080u;                                                \
                        Cd = W | ((mD - (mD >> 7)) ^ 0x7F7F7F7Fu);                                          \
                    }                                                                                       \
                    const uint32_t nU = C##i0, nD = C##i6;                                                  \
                    const uint32_t nL = __builtin_amdgcn_alignbyte(C##i3, L##i3, 1);   /* x - 3 */          \
                    const uint32_t nR = __builtin_amdgcn_alignbyte(R##i3, C##i3, 3);   /* x + 3 */          \
                    const uint32_t nA = __builtin_amdgcn_alignbyte(R##i5, C##i5, 2);   /* (+2, +2) */       \
                    const uint32_t nB = __builtin_amdgcn_alignbyte(C##i1, L##i1, 2);   /* (-2, -2) */       \
                    const uint32_t nP = __builtin_amdgcn_alignbyte(R##i1, C##i1, 2);   /* (+2, -2) */       \
                    const uint32_t nQ = __builtin_amdgcn_alignbyte(C##i5, L##i5, 2);   /* (-2, +2) */       \
                    /* bit 7 of each byte: brighter / darker than the centre by the threshold */            \
                    const uint32_t br = ((nU + Cb) | (nD + Cb)) & ((nL + Cb) | (nR + Cb)) &                 \
                                        ((nA + Cb) | (nB + Cb)) & ((nP + Cb) | (nQ + Cb));                  \
                    const uint32_t dk = ((Cd - nU) | (Cd - nD)) & ((Cd - nL) | (Cd - nR)) &                 \
                                        ((Cd - nA) | (Cd - nB)) & ((Cd - nP) | (Cd - nQ));                  \
                    mask = (mask >> 1) | ((br | dk) & valid);                                               \
                    ORBX_EXTRA_SALU_PER_ROW                                                                 \
                    c += PD;                                                                                \
                    C##i0 = c[3 * PD]; L##i0 = c[3 * PD + dl]; R##i0 = c[3 * PD + dr];                      \
                }
                for (int r = 0; r < rows; r += 7) {
                    ORBX_FAST_ROW4(0, 1, 2, 3, 4, 5, 6, r + 0)
                    ORBX_FAST_ROW4(1, 2, 3, 4, 5, 6, 0, r + 1)
                    ORBX_FAST_ROW4(2, 3, 4, 5, 6, 0, 1, r + 2)
                    ORBX_FAST_ROW4(3, 4, 5, 6, 0, 1, 2, r + 3)
                    ORBX_FAST_ROW4(4, 5, 6, 0, 1, 2, 3, r + 4)
                    ORBX_FAST_ROW4(5, 6, 0, 1, 2, 3, 4, r + 5)
                    ORBX_FAST_ROW4(6, 0, 1, 2, 3, 4, 5, r + 6)
                }
#undef ORBX_FAST_ROW4
                // keep the rows inside the lane's band / the cell: the first `nvalid` rows of the chunk, i.e. the
                // bits below 8 - rows + nvalid of every byte
                const int nvalid = min(yEnd - yc, rows);
                const int rowBit0 = 8 - rows;
                uint32_t m = 0;
                if (nvalid > 0) {
                    const uint32_t keepBits = (((1u << (rowBit0 + nvalid)) - 1u) & 0xFFu) * 0x01010101u;
                    m = passAll ? ((((1u << nvalid) - 1u) << rowBit0) * (((valid >> 7) & 0x01010101u))) : (mask & keepBits);
                }
                // Queue slots: a prefix sum over the wave's lanes (cross-lane moves) and ONE atomic per wave.  (Left as a
                // per-lane atomicAdd, the compiler serialises the lanes: nine scalar instructions for every lane that
                // has a survivor.)
                const int cnt = __popc(m);
                const int incl = wave_incl_scan_i32(cnt);
                const int waveTotal = __builtin_amdgcn_readlane(incl, 63);
                if (bothWaves) __syncthreads();     // the queue may lie on the coarse tile the other wave still reads
                int waveBase = 0;
                if ((tid & 63) == 0 && waveTotal) waveBase = atomicAdd(&s_nQueue, waveTotal);
                waveBase = __builtin_amdgcn_readfirstlane(waveBase);
                if (cnt) {
                    int base = waveBase + incl - cnt;
                    while (m) {
                        const int i = __ffs(m) - 1;       // byte i >> 3 = pixel of the dword, bit i & 7 = rowBit0 + row of the chunk
                        m &= m - 1;
                        const int x = 4 * d + (i >> 3) - ox;
                        s_queue[base++] = (unsigned short)(x | ((yc + (i & 7) - rowBit0) << 7));
                    }
                }
            }
        }
    }
    __syncthreads();
    ORBX_STAMP(1);   // quick reject
    ORBX_FAST_PHASE_END(2);
    const int nQueue = s_nQueue;
    const int lane = tid & 63, waveBase = tid & ~63;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    // (B) arc strength of the survivors; those above the lower threshold are corners (about one survivor in
    // five), and the steps that follow visit only those.  Each wave compacts its corners in place, into the
    // queue slots it has already consumed (its own 64-entry chunks, in order), so no second list is needed.
    // Whole waves iterate together: the ballots below need every lane of the wave.
    int nMine = 0;     // corners of this wave so far (wave-uniform)
    for (int q0 = waveBase; q0 < nQueue; q0 += kFastThreads) {
        const int q = q0 + lane;
        int rec = 0;
        bool corner = false;
        if (q < nQueue) {
            rec = s_queue[q];
            const int x = rec & 127, y = rec >> 7;
            const int st = fast_strength_lds(T, P, x, y, lowTh);
            corner = st > lowTh;
            if (corner) S[y * P + x] = (uint8_t)st;      // the strength tile was cleared by the tile load
        }
        const unsigned long long bal = __ballot(corner);
        if (corner) {
            const int j = nMine + __popcll(bal & lanesBelow);           // j-th corner of this wave
            s_queue[(j >> 6) * kFastThreads + waveBase + (j & 63)] = (unsigned short)rec;
        }
        nMine += __popcll(bal);
    }
    __syncthreads();
    ORBX_STAMP(2);   // arc strength + compaction
#ifdef ORBX_FAST_STAMPS
    if (threadIdx.x == 0 && stampWg < kStampWgs) { g_fastStamps[stampWg][6] = (unsigned int)nQueue; g_fastStamps[stampWg][7] = (unsigned int)nMine; }
#endif
    ORBX_FAST_PHASE_END(3);
    // (C) 3x3 non-maximum suppression of the corners at both thresholds.  The image tile is no longer needed:
    // it receives the list of corners kept at either threshold, (verdicts << 14) | position.
    unsigned short* s_keep = reinterpret_cast<unsigned short*>(tile);
    const int keepCap = (P * R) >> 1;
    for (int j0 = 0; j0 < nMine; j0 += 64) {
        const int j = j0 + lane;
        int rec = 0, flags = 0;
        if (j < nMine) {
            rec = s_queue[(j >> 6) * kFastThreads + waveBase + lane];
            const int x = rec & 127, y = rec >> 7;
            const uint8_t* sp = S + (y - 1) * P + (x - 1);   // top-left of the 3x3 neighbourhood
            const int st = sp[P + 1];
            int nb[8];
            nb[0] = sp[0];         nb[1] = sp[1];         nb[2] = sp[2];
            nb[3] = sp[P];                                nb[4] = sp[P + 2];
            nb[5] = sp[2 * P];     nb[6] = sp[2 * P + 1]; nb[7] = sp[2 * P + 2];
            // kept at threshold th: st > th and score(st) > score(nb) for the 8 neighbours, with score(s) = s > th ?
            // s - 1 : 0.  A neighbour above th needs st > nb, one at or below th needs st - 1 > 0; and a neighbourhood
            // maximum at or below th is below st anyway.  Hence: st > max(th, 1, largest neighbour).
            const int nbMax = max(max(max(nb[0], nb[1]), max(nb[2], nb[3])), max(max(nb[4], nb[5]), max(nb[6], nb[7])));
            const bool keepI = st > max(max(iniTh, 1), nbMax), keepM = st > max(max(minTh, 1), nbMax);
            flags = (keepI ? 1 : 0) | (keepM ? 2 : 0);
        }
        const unsigned long long bI = __ballot(flags & 1), bM = __ballot(flags & 2), bK = bI | bM;
        int base = 0;
        if (lane == 0 && bK) {
            if (bI) atomicAdd(&s_nIni, __popcll(bI));
            if (bM) atomicAdd(&s_nMin, __popcll(bM));
            base = atomicAdd(&s_nKeep, __popcll(bK));
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (flags) {
            // two kept corners are never 8-neighbours, so at most a quarter of the tile's pixels arrive here
            const int pos = base + __popcll(bK & lanesBelow);
            if (pos < keepCap) s_keep[pos] = (unsigned short)(rec | (flags << 14));
            else atomicAdd(&err->cand_overflow, 1);
        }
    }
    __syncthreads();
    ORBX_STAMP(3);   // non-maximum suppression
    ORBX_FAST_PHASE_END(4);
    const bool useIni = s_nIni > 0;
    const int n = useIni ? s_nIni : s_nMin;
    if (n == 0) return;
    const int nKeep = min(s_nKeep, keepCap);
    const int wantFlag = useIni ? 1 : 2;
    const int useTh = useIni ? iniTh : minTh;

    // Round trip: every corner found in the c-th processed cell is scaled to level 0 and
    // back once for each cell processed from its own to the last one of the level.
    const int apps = CI.apps;
    const OrbxLevelGeo& G = geo->lv[level];
    const float s = G.scale, inv = G.invScale;
    const float border = (float)kOrbxMinBorder;
    OrbxCand* out = cands + (long long)frame * geo->candTotal + G.candOff;
    int* counter = candCount + (frame * nl + level) * kCandCounterStride;
    // Without erase rectangles every kept corner is emitted, so the cell reserves its output
    // range with ONE global atomic (the per-level counter is shared by hundreds of cells);
    // with rectangles the survivors are not known in advance and each takes its own slot.
    const bool reserve = nrects == 0;
    if (reserve) {
        __syncthreads();   // everyone has read the counters that are about to be reused
        if (tid == 0) { s_nQueue = atomicAdd(counter, n); s_nIni = 0; }   // s_nQueue / s_nIni reused: base / cursor
        __syncthreads();
    }
    const int base = reserve ? s_nQueue : 0;
    for (int q = tid; q < nKeep; q += kFastThreads) {
        const int rec = s_keep[q];
        if (!((rec >> 14) & wantFlag)) continue;
        const int lx = rec & 127, ly = (rec >> 7) & 127;
        const int sc = score_at(S[ly * P + lx], useTh);
        float x = __fadd_rn((float)lx, (float)(x0 - kOrbxMinBorder));
        float y = __fadd_rn((float)ly, (float)(y0 - kOrbxMinBorder));
        bool erased = false;
        for (int a = 0; a < apps; ++a) {
            const float X = __fmul_rn(__fadd_rn(x, border), s);
            const float Y = __fmul_rn(__fadd_rn(y, border), s);
            if (nrects > 0 && in_any_rect(X, Y, rects, nrects)) { erased = true; break; }
            const float nx = __fsub_rn(__fmul_rn(X, inv), border);
            const float ny = __fsub_rn(__fmul_rn(Y, inv), border);
            const bool fixed = (nx == x) && (ny == y);
            x = nx; y = ny;
            if (fixed) break;   // further applications reproduce the same values and the same rect test
        }
        if (erased) continue;
        const int pos = reserve ? base + atomicAdd(&s_nIni, 1) : atomicAdd(counter, 1);
        if (pos < G.candCap) {
            OrbxCand c;
            c.x = x; c.y = y; c.response = (float)sc;
            c.order = ((uint32_t)cell << 14) | ((uint32_t)ly << 7) | (uint32_t)lx;
            out[pos] = c;
        } else {
            atomicAdd(&err->cand_overflow, 1);
            atomicMax(&err->need_cand, pos + 1);
        }
    }
    ORBX_STAMP(4);   // round trip + emission
}

// ======================================================================================
// k_quadtree: DistributeOctTree for one (frame, level) per workgroup.
//
// The node list lives in LDS in list order; a split replaces the list by
//   [children, newest first] ++ [nodes that were not split, old order]
// exactly as push_front / erase do in the reference.  Keys never move: each carries the
// index of its node, and only per-node key COUNTS drive the algorithm (a node with one
// key is final).  The ordering-sensitive parts are reproduced literally: children are
// created in the order UL, UR, BL, BR; the "expandable" record is in creation order; when
// the list is about to overshoot N the record is sorted with libstdc++'s introsort and
// split from the back until the list holds N nodes.  The per-node winner is the key with
// the highest response, the earliest in emission order among equals.
// ======================================================================================
// Threads per workgroup: a template parameter of the kernel, chosen per handle (qt_threads_for).  Large workgroups
// finish a big level sooner, small ones are placed sooner on compute units that other streams' kernels keep full.
constexpr int kQtThreadsMax = 512;
// keys per thread whose position and node stay in registers (4096 keys per workgroup, at most 16 per thread)
__host__ __device__ constexpr int qt_reg_keys(int threads) { return 4096 / threads < 16 ? 4096 / threads : 16; }
// Measured in the pipelined benchmark (two handle sets, four streams), frames/s with 128 / 256 / 512 threads:
// 752x480 with 1000 features 284 k / 277 k / 274 k, 1920x1080 with 5000 features 46.5 k / 52.5 k / 53.3 k.
// A call with a few frames leaves the chip mostly empty: nothing competes for wave slots and the large workgroup's
// shorter critical path wins (a single 752x480 frame through the host entry point: 0.164 ms with 128 threads, 0.155 with 512).
static inline int qt_threads_for(int nfeatures, int nframes) {
    if (nframes <= 8) return 512;
    return nfeatures <= 1500 ? 128 : (nfeatures <= 3000 ? 256 : 512);
}

struct QtBox { short ulx, uly, brx, bry; };

__device__ __forceinline__ int qt_quadrant(const QtBox b, float x, float y) {
    const int mx = b.ulx + ((b.brx - b.ulx + 1) >> 1);   // UL.x + ceil((UR.x - UL.x) / 2)
    const int my = b.uly + ((b.bry - b.uly + 1) >> 1);
    return (x < (float)mx ? 0 : 1) | (y < (float)my ? 0 : 2);
}
__device__ __forceinline__ QtBox qt_child_box(const QtBox b, int q) {
    const short mx = (short)(b.ulx + ((b.brx - b.ulx + 1) >> 1));
    const short my = (short)(b.uly + ((b.bry - b.uly + 1) >> 1));
    QtBox c;
    c.ulx = (q & 1) ? mx : b.ulx;  c.brx = (q & 1) ? b.brx : mx;
    c.uly = (q & 2) ? my : b.uly;  c.bry = (q & 2) ? b.bry : my;
    return c;
}

// Exclusive scan of v[0..n) (in place) by the whole workgroup; returns the total.
// Per-thread chunk sums, a shuffle scan inside each wave, then the wave totals (tmp: one int per wave).
template <int kQtThreads>
__device__ int qt_block_scan(int* v, int n, int* tmp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + kQtThreads - 1) / kQtThreads;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += v[i];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kQtThreads / 64; ++w) {
        const int t = tmp[w];
        before += (w < wave) ? t : 0;
        total += t;
    }
    int run = before + incl - sum;
    for (int i = b; i < e; ++i) { int t = v[i]; v[i] = run; run += t; }
    __syncthreads();
    return total;
}

template <int kQtThreads>
__global__ __launch_bounds__(kQtThreads) void k_quadtree(const OrbxFrameGeo* __restrict__ geo,
                                                         const OrbxCand* __restrict__ cands,
                                                         const int* __restrict__ candCount,
                                                         uint16_t* __restrict__ keyNode,
                                                         OrbxLevelKp* __restrict__ levelKps,
                                                         int* __restrict__ kpCount, OrbxErr* __restrict__ err) {
    constexpr int kQtRegKeys = qt_reg_keys(kQtThreads);
    extern __shared__ __align__(16) unsigned char qt_smem[];
    const int level = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const int nl = geo->nlevels;
    const OrbxLevelGeo& G = geo->lv[level];
    const int cap = geo->maxNodeCap;

    // LDS carve-up (cap entries each unless noted)
    unsigned long long* sortbuf = reinterpret_cast<unsigned long long*>(qt_smem);   // expandable record / winners
    QtBox* boxA = reinterpret_cast<QtBox*>(sortbuf + cap);
    QtBox* boxB = boxA + cap;
    int* cntA = reinterpret_cast<int*>(boxB + cap);
    int* cntB = cntA + cap;
    int* child = cntB + cap;        // 4 * cap: child key counts, then new indices
    int* scanv = child + 4 * cap;   // cap: scan input / output
    int* base = scanv + cap;        // cap: creation sequence base of a split node's children
    int* tmp = base + cap;          // kQtThreads
    __shared__ int s_size, s_done, s_nExp, s_total, s_careful, s_nE, s_bad;

    const int n = min(candCount[(frame * nl + level) * kCandCounterStride], G.candCap);
    const OrbxCand* C = cands + (long long)frame * geo->candTotal + G.candOff;
    uint16_t* KN = keyNode + (long long)frame * geo->candTotal + G.candOff;
    const int N = G.quota;
    const int H = G.h - 2 * kOrbxMinBorder;
    const int nIni = G.nIni;
    const float hX = G.hX;

    QtBox* box = boxA; QtBox* nbox = boxB;
    int* cnt = cntA;   int* ncnt = cntB;

    // ---- roots ---------------------------------------------------------------------
    for (int i = tid; i < nIni; i += kQtThreads) child[i] = 0;
    if (tid == 0) { s_size = 0; s_done = 0; s_bad = 0; }
    __syncthreads();
    // The position and the node of the first kQtRegKeys keys of every thread stay in registers for the whole
    // kernel (all keys of a 752 x 480 level); the sweeps below then touch global memory only for the keys beyond
    // that, whose node index lives in keyNode.
    float kx[kQtRegKeys], ky[kQtRegKeys];
    int kn[kQtRegKeys];
#pragma unroll
    for (int k = 0; k < kQtRegKeys; ++k) {
        const int i = tid + k * kQtThreads;
        kx[k] = 0.f; ky[k] = 0.f; kn[k] = 0;
        if (i < n) { kx[k] = C[i].x; ky[k] = C[i].y; }
    }
    // fn(x, y, node&) for every key of this thread
    auto for_keys = [&](auto&& fn) {
#pragma unroll
        for (int k = 0; k < kQtRegKeys; ++k)
            if (tid + k * kQtThreads < n) fn(kx[k], ky[k], kn[k]);
        for (int i = tid + kQtRegKeys * kQtThreads; i < n; i += kQtThreads) {
            const int before = KN[i];
            int nd = before;
            fn(C[i].x, C[i].y, nd);
            if (nd != before) KN[i] = (uint16_t)nd;
        }
    };
    for_keys([&](float x, float, int& nd) {
        int r = (int)__fdiv_rn(x, hX);
        r = min(max(r, 0), nIni - 1);
        nd = r + 0x10000;      // marks "written" for the global tail (the stored value is the low 16 bits)
        atomicAdd(&child[r], 1);
    });
    __syncthreads();
    if (tid == 0) {
        int sz = 0;
        for (int r = 0; r < nIni; ++r) {
            if (child[r] > 0) {
                QtBox b;
                b.ulx = (short)(int)__fmul_rn(hX, (float)r);
                b.brx = (short)(int)__fmul_rn(hX, (float)(r + 1));
                b.uly = 0; b.bry = (short)H;
                box[sz] = b; cnt[sz] = child[r];
                scanv[r] = sz++;
            } else scanv[r] = 0;
        }
        s_size = sz;
    }
    __syncthreads();
    for_keys([&](float, float, int& nd) { nd = scanv[nd & 0xFFFF]; });
    __syncthreads();

    // ---- split loop ------------------------------------------------------------------
    int careful = 0;   // uniform across the workgroup
    while (true) {
        const int size = s_size;
        if (s_done) break;
        // (1) per-child key counts of every node that can still be split
        for (int i = tid; i < 4 * size; i += kQtThreads) child[i] = 0;
        __syncthreads();
        for_keys([&](float x, float y, int& nd) {
            if (cnt[nd] > 1) atomicAdd(&child[4 * nd + qt_quadrant(box[nd], x, y)], 1);
        });
        __syncthreads();

        // (2) choose the nodes to split and the creation order of their children
        int total;   // number of children created
        int kept;    // nodes that are not split
        if (!careful) {
            // full sweep: every node with more than one key, in list order.  One scan carries both prefix sums:
            // children created before the node (low half) and unsplit nodes before it (high half); neither exceeds
            // 4 x the list capacity < 65536.
            for (int p = tid; p < size; p += kQtThreads) {
                int v = 1 << 16;
                if (cnt[p] > 1)
                    v = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0);
                scanv[p] = v;
            }
            __syncthreads();
            const int both = qt_block_scan<kQtThreads>(scanv, size, tmp);
            total = both & 0xFFFF; kept = both >> 16;
            for (int p = tid; p < size; p += kQtThreads) {
                const int v = scanv[p];
                base[p] = (cnt[p] > 1) ? (v & 0xFFFF) : -1;
                scanv[p] = v >> 16;      // (3) position of a node that stays, among those that stay
            }
            __syncthreads();
        } else {
            // careful pass: sort the expandable record (all waves), split from the back until N nodes
            // (wave 0).  scanv / base / nbox / ncnt are free at this point and serve as scratch.
            orbx_sort::block_sort(sortbuf, s_nE, ncnt, scanv, base, reinterpret_cast<orbx_sort::elem_t*>(nbox));
            for (int p = tid; p < size; p += kQtThreads) base[p] = -1;
            __syncthreads();
            if (tid < 64) {
                const int nE = s_nE;
                int sz = size, seq = 0;
                bool stop = false;
                for (int j0 = nE - 1; j0 >= 0 && !stop; j0 -= 64) {
                    const int j = j0 - tid;
                    int p = 0, nc = 0;
                    if (j >= 0) {
                        p = (int)orbx_sort::payload_of(sortbuf[j]);
                        nc = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0);
                    }
                    int incl = nc;   // inclusive scan over the lanes (lane 0 = back of the record)
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (tid >= off) incl += o;
                    }
                    // list size after splitting the nodes of lanes 0..tid
                    const int szAfter = sz + incl - (tid + 1);
                    const unsigned long long hit = __ballot(j >= 0 && szAfter >= N);
                    const int lastLane = hit ? (__ffsll((long long)hit) - 1) : min(63, j0);
                    if (tid <= lastLane) base[p] = seq + incl - nc;
                    const int chunkSeq = __shfl(incl, lastLane, 64);
                    seq += chunkSeq;
                    sz += chunkSeq - (lastLane + 1);
                    stop = hit != 0ull;
                }
                if (tid == 0) s_total = seq;
            }
            __syncthreads();
            total = s_total;
        }

        // (3) positions of the nodes that stay: after all children, old relative order
        if (careful) {
            for (int p = tid; p < size; p += kQtThreads) scanv[p] = (base[p] < 0) ? 1 : 0;
            __syncthreads();
            kept = qt_block_scan<kQtThreads>(scanv, size, tmp);
        }
        const int newSize = total + kept;
        if (newSize > cap) {   // cannot happen for a consistent geometry; never write out of bounds
            if (tid == 0) { atomicAdd(&err->kp_overflow, 1); s_bad = 1; s_done = 1; }
            __syncthreads();
            break;
        }
        if (tid == 0) s_nExp = 0;
        __syncthreads();

        // (4) write the new generation
        int nexpMine = 0;
        for (int p = tid; p < size; p += kQtThreads) {
            if (base[p] < 0) {
                const int np = total + scanv[p];
                nbox[np] = box[p]; ncnt[np] = cnt[p];
                child[4 * p] = np;
            } else {
                int seq = base[p], nexp = 0;   // (nexp: children of this node that can be split again)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = child[4 * p + q];
                    if (c > 0) {
                        const int np = total - 1 - seq;
                        ++seq;
                        nbox[np] = qt_child_box(box[p], q); ncnt[np] = c;
                        child[4 * p + q] = np;
                        nexp += (c > 1);
                    }
                }
                nexpMine += nexp;
            }
        }
        {
            // one atomic per wave (a per-lane atomicAdd of differing values is serialised lane by lane by the compiler)
            const int waveExp = wave_sum_i32(nexpMine);
            if ((tid & 63) == 0 && waveExp) atomicAdd(&s_nExp, waveExp);
        }
        __syncthreads();
        // (5) move the keys
        for_keys([&](float x, float y, int& nd) {
            nd = (base[nd] < 0) ? child[4 * nd] : child[4 * nd + qt_quadrant(box[nd], x, y)];
        });
        __syncthreads();
        { QtBox* t = box; box = nbox; nbox = t; int* u = cnt; cnt = ncnt; ncnt = u; }

        // (6) termination / mode switch (ORBextractor.cc:790-858)
        if (tid == 0) {
            const int nExp = s_nExp;
            bool done = (newSize >= N) || (newSize == size);
            bool toCareful = careful != 0;
            if (!done && !careful && newSize + 3 * nExp > N) toCareful = true;
            s_careful = toCareful ? 1 : 0;
            s_done = done ? 1 : 0;
            s_size = newSize;
        }
        __syncthreads();
        careful = s_careful;
        if (!s_done && careful) {
            // expandable record in creation order = new positions total-1 .. 0 (compacted by wave 0)
            if (tid < 64) {
                int nE = 0;
                for (int np0 = total - 1; np0 >= 0; np0 -= 64) {
                    const int np = np0 - tid;
                    const bool exp = np >= 0 && cnt[np] > 1;
                    const unsigned long long b = __ballot(exp);
                    if (exp)
                        sortbuf[nE + __popcll(b & ((1ull << tid) - 1ull))] =
                            orbx_sort::make_elem((uint32_t)cnt[np], (uint32_t)box[np].ulx, (uint32_t)np);
                    nE += __popcll(b);
                }
                if (tid == 0) s_nE = nE;
            }
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- winners -----------------------------------------------------------------------
    const int size = s_bad ? 0 : s_size;
    OrbxLevelKp* out = levelKps + (long long)frame * geo->kpTotal + G.kpOff;
    if (size > G.kpCap) {
        if (tid == 0) { atomicAdd(&err->kp_overflow, 1); kpCount[frame * nl + level] = 0; }
        return;
    }
    for (int p = tid; p < size; p += kQtThreads) sortbuf[p] = 0ull;
    __syncthreads();
    auto node_of = [&](int i) {      // i = tid + k * kQtThreads
        int nd = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < kQtRegKeys; ++k)
            if (i == tid + k * kQtThreads) { nd = kn[k]; found = true; }
        return found ? nd : (int)KN[i];
    };
    for (int i = tid; i < n; i += kQtThreads) {
        const unsigned long long key = ((unsigned long long)(uint32_t)C[i].response << 32) | (uint32_t)(~C[i].order);
        atomicMax(&sortbuf[node_of(i)], key);
    }
    __syncthreads();
    for (int i = tid; i < n; i += kQtThreads) {
        const OrbxCand c = C[i];
        const unsigned long long key = ((unsigned long long)(uint32_t)c.response << 32) | (uint32_t)(~c.order);
        const int nd = node_of(i);
        if (sortbuf[nd] == key) {
            OrbxLevelKp k;
            k.x = __fadd_rn(c.x, (float)kOrbxMinBorder);
            k.y = __fadd_rn(c.y, (float)kOrbxMinBorder);
            k.response = c.response;
            k.slot = -1;
            out[nd] = k;
        }
    }
    if (tid == 0) kpCount[frame * nl + level] = size;
}

// ======================================================================================
// k_layout: row of every keypoint in the frame's output (ORBextractor.cc:1329-1398).
// Keypoints whose level-0 x lies inside [lap0, lap1] fill the arrays from the back,
// the others from the front, both in (level, list) order.
// ======================================================================================
constexpr int kLayThreads = 256;     // small workgroups find room beside the other stream's kernels sooner

__global__ __launch_bounds__(kLayThreads) void k_layout(const OrbxFrameGeo* __restrict__ geo,
                                                        OrbxLevelKp* __restrict__ levelKps,
                                                        const int* __restrict__ kpCount, int lap0, int lap1,
                                                        int capacity, int* __restrict__ counts,
                                                        OrbxDescWork* __restrict__ work,
                                                        OrbxErr* __restrict__ err) {
    __shared__ int s_wave[kLayThreads / 64];
    __shared__ int s_running;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int nl = geo->nlevels;
    OrbxLevelKp* K = levelKps + (long long)frame * geo->kpTotal;
    // keypoints of all levels as one sequence in (level, list) order: first[l] = keypoints before level l
    int first[kOrbxMaxLevels + 1];
    first[0] = 0;
#pragma unroll
    for (int l = 0; l < kOrbxMaxLevels; ++l) first[l + 1] = first[l] + (l < nl ? kpCount[frame * nl + l] : 0);
    const int total = first[kOrbxMaxLevels];
    if (tid == 0) s_running = 0;
    __syncthreads();
    const bool overflow = total > capacity;
    if (overflow && tid == 0) atomicAdd(&err->out_overflow, 1);
    for (int b = 0; b < total; b += kLayThreads) {
        const int j = b + tid;
        int mono = 0, l = 0, i = 0;
        OrbxLevelKp rec;
        rec.x = 0.f; rec.y = 0.f; rec.response = 0.f; rec.slot = -1;
        if (j < total) {
#pragma unroll
            for (int q = 1; q < kOrbxMaxLevels; ++q) l += (j >= first[q]) ? 1 : 0;
            i = j;
#pragma unroll
            for (int q = 1; q < kOrbxMaxLevels; ++q) i = (l == q) ? j - first[q] : i;
            const OrbxLevelGeo& G = geo->lv[l];
            rec = K[G.kpOff + i];
            float x = rec.x;
            if (l != 0) x = __fmul_rn(x, G.scale);
            mono = (x >= (float)lap0 && x <= (float)lap1) ? 0 : 1;
        }
        // ordered rank of the mono flags inside this chunk
        const unsigned long long bal = __ballot(mono);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int wbase = 0, chunkMono = 0;
#pragma unroll
        for (int w = 0; w < kLayThreads / 64; ++w) { const int c = s_wave[w]; wbase += (w < wv) ? c : 0; chunkMono += c; }
        const int running = s_running;          // mono keypoints before this chunk
        if (j < total && !overflow) {
            const int monoBefore = running + wbase + before;
            const int stereoBefore = j - monoBefore;      // j keypoints precede this one in (level, list) order
            const OrbxLevelGeo& G = geo->lv[l];
            const int slot = mono ? monoBefore : (total - 1 - stereoBefore);
            K[G.kpOff + i].slot = slot;
            // everything k_describe needs to know about this keypoint
            OrbxDescWork wk;
            wk.imgOff = (int)G.off; wk.pitch = G.pitch; wk.wh = G.w | (G.h << 16);
            wk.cxy = __float2int_rn(rec.x) | (__float2int_rn(rec.y) << 16);
            wk.slotLevel = slot | (l << 24);
            wk.x = rec.x; wk.y = rec.y; wk.response = rec.response;
            work[(long long)frame * geo->kpTotal + j] = wk;
        }
        __syncthreads();
        if (tid == 0) s_running = running + chunkMono;
        __syncthreads();
    }
    if (tid == 0) {
        counts[2 * frame] = overflow ? 0 : total;
        counts[2 * frame + 1] = overflow ? 0 : s_running;
    }
}

// ======================================================================================
// k_describe: one wave per kept keypoint.
//   raw 43x43 patch -> LDS (REFLECT_101 at the level's own edges)
//   IC_Angle on the raw patch, fastAtan2
//   7x7 Gaussian (8.8 fixed point, taps 18 34 48 56 48 34 18) on the 37x37 core
//   256 steered comparisons -> 32 bytes
// ======================================================================================
constexpr int kDescWaves = 1;     // waves (= keypoints) per workgroup
constexpr int kRawR = 21, kRawW = 43, kRawP = 48;   // raw rows start at a 4-byte aligned column (at most 3 + 43 bytes)
constexpr int kBlurR = 18, kBlurW = 37, kBlurP = 40;
// transposed row sums: one dword = the sums of a row pair; an odd pitch (in dwords) spreads the columns over the banks
constexpr int kHorTPD = 25;

__device__ __forceinline__ int reflect101_dev(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float k = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cos / sin of a binary32 angle in [0, 2 pi], evaluated in binary64 and rounded once.
__device__ __forceinline__ void sincos_f32(float xf, float* s_out, float* c_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double x = (double)xf;
    const int k = (int)__dadd_rn(__dmul_rn(x, two_over_pi), 0.5);
    const double r = __dsub_rn(__dsub_rn(x, __dmul_rn((double)k, pio2_hi)), __dmul_rn((double)k, pio2_lo));
    const double z = __dmul_rn(r, r);
    double sp = 1.0 / 355687428096000.0;
    sp = __dsub_rn(__dmul_rn(sp, z), 1.0 / 1307674368000.0);
    sp = __dadd_rn(__dmul_rn(sp, z), 1.0 / 6227020800.0);
    sp = __dsub_rn(__dmul_rn(sp, z), 1.0 / 39916800.0);
    sp = __dadd_rn(__dmul_rn(sp, z), 1.0 / 362880.0);
    sp = __dsub_rn(__dmul_rn(sp, z), 1.0 / 5040.0);
    sp = __dadd_rn(__dmul_rn(sp, z), 1.0 / 120.0);
    sp = __dsub_rn(__dmul_rn(sp, z), 1.0 / 6.0);
    const double sn = __dadd_rn(r, __dmul_rn(__dmul_rn(r, z), sp));
    double cp = 1.0 / 20922789888000.0;
    cp = __dsub_rn(__dmul_rn(cp, z), 1.0 / 87178291200.0);
    cp = __dadd_rn(__dmul_rn(cp, z), 1.0 / 479001600.0);
    cp = __dsub_rn(__dmul_rn(cp, z), 1.0 / 3628800.0);
    cp = __dadd_rn(__dmul_rn(cp, z), 1.0 / 40320.0);
    cp = __dsub_rn(__dmul_rn(cp, z), 1.0 / 720.0);
    cp = __dadd_rn(__dmul_rn(cp, z), 1.0 / 24.0);
    const double cs = __dsub_rn(1.0, __dsub_rn(__dmul_rn(0.5, z), __dmul_rn(__dmul_rn(z, z), cp)));
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// The same value with fused multiply-adds (half the binary64 instructions; the kernels use this one).  The two
// evaluations can differ in the last bits of the binary64 result, which the rounding to binary32 hides unless the
// result lies within ~1e-16 (relative) of a binary32 rounding boundary: orbx_debug_sincos_diff counts the inputs
// where that happens -- none of the 1 086 918 650 binary32 values in [0, 6.2832].
// (a * b + c with the constant c held in a scalar register pair: left to itself the compiler spends two vector moves
// per constant, as many vector instructions as the fusion saves)
__device__ __forceinline__ double fma_sconst(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ void sincos_f32_fma(float xf, float* s_out, float* c_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double x = (double)xf;
    const int k = (int)fma_sconst(x, two_over_pi, 0.5);
    const double kd = (double)k;
    const double r = __fma_rn(kd, -pio2_lo, __fma_rn(kd, -pio2_hi, x));
    const double z = __dmul_rn(r, r);
    double sp = 1.0 / 355687428096000.0;
    sp = fma_sconst(sp, z, -1.0 / 1307674368000.0);
    sp = fma_sconst(sp, z, 1.0 / 6227020800.0);
    sp = fma_sconst(sp, z, -1.0 / 39916800.0);
    sp = fma_sconst(sp, z, 1.0 / 362880.0);
    sp = fma_sconst(sp, z, -1.0 / 5040.0);
    sp = fma_sconst(sp, z, 1.0 / 120.0);
    sp = fma_sconst(sp, z, -1.0 / 6.0);
    const double sn = __fma_rn(__dmul_rn(r, z), sp, r);
    double cp = 1.0 / 20922789888000.0;
    cp = fma_sconst(cp, z, -1.0 / 87178291200.0);
    cp = fma_sconst(cp, z, 1.0 / 479001600.0);
    cp = fma_sconst(cp, z, -1.0 / 3628800.0);
    cp = fma_sconst(cp, z, 1.0 / 40320.0);
    cp = fma_sconst(cp, z, -1.0 / 720.0);
    cp = fma_sconst(cp, z, 1.0 / 24.0);
    const double cs = __fma_rn(__dmul_rn(z, z), cp, __fma_rn(z, -0.5, 1.0));
    const float snf = (float)sn, csf = (float)cs;          // (negation commutes with the rounding)
    const bool swap = (k & 1) != 0;
    const float s = swap ? csf : snf, c = swap ? snf : csf;
    *s_out = (k & 2) ? -s : s;
    *c_out = ((k + 1) & 2) ? -c : c;
}

__global__ void k_debug_sincos_diff(uint32_t firstBits, unsigned long long count, unsigned long long* nDiff, uint32_t* firstDiffs) {
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = firstBits + (uint32_t)i;
        const float x = __builtin_bit_cast(float, bits);
        float s0, c0, s1, c1;
        sincos_f32(x, &s0, &c0);
        sincos_f32_fma(x, &s1, &c1);
        if (__builtin_bit_cast(uint32_t, s0) != __builtin_bit_cast(uint32_t, s1) ||
            __builtin_bit_cast(uint32_t, c0) != __builtin_bit_cast(uint32_t, c1)) {
            const unsigned long long slot = atomicAdd(nDiff, 1ull);
            if (slot < 16) firstDiffs[slot] = bits;
        }
        ++local;
    }
    atomicAdd(nDiff + 1, local);       // inputs compared
}

// Diagnostics: over the binary32 values with bit patterns first_bits .. first_bits + count - 1, the number of inputs for
// which the two evaluation orders of the binary64 sine / cosine round to different binary32 values (the first 16 of
// them in first16, the number of inputs the kernel went through in *compared).
extern "C" long long orbx_debug_sincos_diff(uint32_t first_bits, unsigned long long count, uint32_t* first16,
                                            unsigned long long* compared) {
    unsigned long long* d_n = nullptr;
    uint32_t* d_first = nullptr;
    if (hipMalloc((void**)&d_n, 16) != hipSuccess || hipMalloc((void**)&d_first, 64) != hipSuccess) return -1;
    hipMemset(d_n, 0, 16);
    hipMemset(d_first, 0, 64);
    hipLaunchKernelGGL(k_debug_sincos_diff, dim3(4096), dim3(256), 0, 0, first_bits, count, d_n, d_first);
    const hipError_t launched = hipGetLastError(), ran = hipDeviceSynchronize();
    if (launched != hipSuccess || ran != hipSuccess) {
        orbx_set_error("sincos comparison kernel: %s / %s", hipGetErrorString(launched), hipGetErrorString(ran));
        hipFree(d_n); hipFree(d_first);
        return -1;
    }
    unsigned long long n[2] = {0, 0};
    const bool ok = hipMemcpy(n, d_n, 16, hipMemcpyDeviceToHost) == hipSuccess &&
                    (!first16 || hipMemcpy(first16, d_first, 64, hipMemcpyDeviceToHost) == hipSuccess);
    hipFree(d_n); hipFree(d_first);
    if (compared) *compared = n[1];
    return ok ? (long long)n[0] : -1;
}

// Weight bytes of the orientation disc: entry [|v| * 4 + g] covers the 8 columns u = 8g - 15 .. 8g - 8 of row v as
// { (u + 15) of the first 4 columns, 1 for the same columns, (u + 15) of the last 4, 1 for those }, a weight being 0
// where |u| > umax[|v|] (outside the disc; u = 16 of the last group never is inside).
struct alignas(16) OrientWeights { uint32_t w[256]; };
constexpr OrientWeights make_orient_weights() {
    OrientWeights t{};
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};   // ORBextractor.cc:520-535
    for (int av = 0; av < 16; ++av)
        for (int g = 0; g < 4; ++g)
            for (int half = 0; half < 2; ++half) {
                uint32_t wu = 0, w1 = 0;
                for (int k = 0; k < 4; ++k) {
                    const int u = 8 * g + 4 * half - 15 + k, au = u < 0 ? -u : u;
                    if (au <= umax[av]) { wu |= (uint32_t)(u + 15) << (8 * k); w1 |= 1u << (8 * k); }
                }
                t.w[4 * (av * 4 + g) + 2 * half] = wu;
                t.w[4 * (av * 4 + g) + 2 * half + 1] = w1;
            }
    return t;
}
__constant__ OrientWeights c_orientW = make_orient_weights();

// Horizontal 7-tap pass of the blur (taps 18 34 48 56 48 34 18) over the raw patch rows of one keypoint.
// item = (pair of raw rows 2 rp, 2 rp + 1; 8 output columns c0 .. c0 + 7).  The 20 source bytes of a row that the 8
// outputs touch are fetched as two 8-byte reads and one dword (c0 is a multiple of 8, so the reads are aligned).
// Output k needs bytes OX + k .. OX + k + 6 of them; instead of shifting the data into place, the TAPS are shifted: for
// every byte offset 0..3 inside a dword there is a set of 2 or 3 weight dwords (zero where no tap falls), and the
// output is 2 or 3 v_dot4_u32_u8 over the aligned source dwords -- 20 dot products per row and no byte shuffles.
// The row sums are stored TRANSPOSED, horT[c][rp] = (sum of row 2 rp) | (sum of row 2 rp + 1) << 16, so that the
// vertical pass reads consecutive rows as packed 16-bit pairs; one dword store per column and row pair.
__device__ __forceinline__ constexpr uint32_t hblur_weight(int byteOffset, int dword) {
    const uint32_t taps[7] = {18u, 34u, 48u, 56u, 48u, 34u, 18u};
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) {
        const int tap = dword * 4 + b - byteOffset;
        if (tap >= 0 && tap < 7) w |= taps[tap] << (8 * b);
    }
    return w;
}
// ======================================================================================
// k_blur_levels: cv::GaussianBlur(7 x 7, sigma 2, BORDER_REFLECT_101) of whole pyramid levels (ORBextractor.cc:1344-1351)
// into a second slab with the pyramid's layout, for handles whose keypoints cover their levels densely enough that
// blurring every level once is cheaper than blurring a 37 x 37 patch per keypoint (see choose_blur_mode).
// One wave = one tile: `nb` bands of (cw + 2) lanes; a band covers cw dword columns x stripRows rows, the bands of
// a tile lie below each other (full-width tiles have one band of 62 columns; the narrow remainder of a level
// stacks many short bands so that its lanes are used as well).  A lane owns 4 pixels of a row (one dword) and
// marches down its band: the row's 7-tap horizontal sums of its 4 pixels come from its own dword and the two
// neighbouring lanes' dwords (two DPP moves) as 10 v_dot4_u32_u8 against taps shifted to the byte position; the
// vertical pass keeps, per pixel column, the last six sums as packed pairs of consecutive rows in registers (the row
// loop is unrolled six times so that the ring needs no moves): three v_dot2_u32_u16 and one multiply-add per output.
// The first and last lane of a band are halo lanes (they load and take part in the exchange, and store nothing).
// REFLECT_101 at the level's own edge: rows by reflecting the row index, columns by byte permutes in the two edge
// lanes (see the kernel).
// ======================================================================================
constexpr int kBlurTileCols = 62;       // dword columns of a full-width tile
struct alignas(16) OrbxBlurTile {
    int imgOff;      // byte offset of the level inside a frame's slab
    int pitchShape;  // pitch | cw << 16 | nb << 24
    int wh;          // width | height << 16
    int xy;          // first stored dword column | first output row << 16
};

__device__ __forceinline__ uint32_t dpp_from_lower_lane(uint32_t v) {   // lane i receives lane i - 1's value (lane 0: zero)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dpp_from_upper_lane(uint32_t v) {   // lane i receives lane i + 1's value (lane 63: zero)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}

// (waves_per_eu: 63 vector registers instead of 68, i.e. 8 waves per SIMD instead of 7, without spills)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_blur_levels(const OrbxBlurTile* __restrict__ tiles, int nTiles, int stripRows, const uint8_t* __restrict__ pyr,
                   uint8_t* __restrict__ blurred, long long frameStride) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // Neighbouring tiles share cache lines (a tile's 256-byte row segments start at multiples of 248 bytes) and halo
    // rows: keep them on one XCD, close in time, so that the shared lines come from that XCD's L2 (see xcd_remap).
    const int logical = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = logical / (int)gridDim.x;
    const int t = (logical - frame * (int)gridDim.x) * 4 + wave;
    if (t >= nTiles) return;
    const OrbxBlurTile T = tiles[t];
    const int w = T.wh & 0xFFFF, h = T.wh >> 16, pitch = T.pitchShape & 0xFFFF;
    const int cw = (T.pitchShape >> 16) & 0xFF, nb = T.pitchShape >> 24;
    const int col0 = T.xy & 0xFFFF, y0 = T.xy >> 16;
    const int bandLanes = cw + 2;
    const int bandRaw = lane / bandLanes, pos = lane - bandRaw * bandLanes;
    const int band = min(bandRaw, nb - 1);                  // (lanes past the last band repeat it and store nothing)
    const int c = col0 + pos - 1;                           // this lane's dword column (-1: left of the level)
    const int yBand = y0 + band * stripRows;
    const int lastCol = (w - 1) >> 2;
    const uint8_t* __restrict__ src = pyr + (long long)frame * frameStride + T.imgOff;
    uint8_t* __restrict__ dst = blurred + (long long)frame * frameStride + T.imgOff;

    // REFLECT_101 at the level's left and right edge.  Every lane loads the dword of its own column (clamped into the
    // level).  The pixels beyond an edge are mirror images of pixels that the edge lane and its inner neighbour hold,
    // so the two edge lanes build them with byte permutes instead of receiving them from a halo lane:
    //   first column (c = 0):     L' = pixels -4..-1 = (4, 3, 2, 1) = (R.0, C.3, C.2, C.1)
    //   last column, m = (w - 1) & 3 the byte of the level's last pixel: C' byte b > m = pixel w - 1 - (b - m), and
    //   R' byte b <= m - 1 (the only ones a stored pixel's window reaches) = pixel w - 1 - (4 + b - m): bytes of C, or
    //   of L when the offset is negative -- v_perm_b32(C, L, selector 4 + offset) serves both.
    // Tiles without an edge column skip all of it (a wave-uniform test).
    const int m = (w - 1) & 3;
    const bool isFirst = c == 0, isLast = c == lastCol;
    // (wave-uniform; "+ cw", not "+ cw - 1": when the last column is a tile's right halo lane its bytes beyond the level's
    // width still feed the windows of the lane before it, and must be mirrored too)
    const bool edgeTile = col0 == 0 || col0 + cw >= lastCol;
    uint32_t selC = 0x07060504u, selR = 0x0c0c0c0cu;
    if (isLast) {
        selC = 0; selR = 0;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            selC |= (uint32_t)(bb <= m ? 4 + bb : 4 + 2 * m - bb) << (8 * bb);
            selR |= (uint32_t)(bb <= m - 1 ? 2 * m - bb : 0x0c) << (8 * bb);
        }
    }
    const int cl = min(max(c, 0), lastCol);
    const bool stores = bandRaw < nb && pos >= 1 && pos <= cw && c <= lastCol;

    uint32_t P[4][6];            // P[k][j % 6] = H_k(row j) | H_k(row j + 1) << 16, H_k = row sum of the lane's k-th pixel
    uint32_t Hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) P[k][j] = 0;
    const us2 w0 = {18, 34}, w1 = {48, 56}, w2 = {48, 34};

    // Byte offset of source row r of the lane's band (mirrored at the level's top and bottom).  Only the trips that
    // touch the top or the bottom of the level (a wave-uniform test over all bands of the tile) pay for the mirror;
    // the others add a scalar to the lane's first row.
    const uint32_t plainIn = (uint32_t)((yBand - 3) * pitch + 4 * cl);
    const uint32_t plainOut = (uint32_t)(yBand * pitch + 4 * c);
    const int lastBandY = y0 + (nb - 1) * stripRows;
    auto mirrored_row = [&](int r) {
        int y = yBand - 3 + r;
        y = max(y, -y);
        y = min(y, 2 * (h - 1) - y);
        return (uint32_t)(max(y, 0) * pitch);               // (rows far below the level: stored nowhere)
    };
    // The six rows of a trip are loaded one trip ahead of their use: a wave waits for memory once per trip, with a
    // trip's arithmetic in between, instead of once per row.
    uint32_t cur[6], nxt[6];
    auto load_trip = [&](int rBase, uint32_t (&A)[6]) {
        const bool edge = y0 - 3 + rBase < 0 || lastBandY - 3 + rBase + 5 >= h;          // wave-uniform
        if (edge) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
                A[u] = *reinterpret_cast<const uint32_t*>(src + (mirrored_row(rBase + u) + 4u * (uint32_t)cl));
        } else {
#pragma unroll
            for (int u = 0; u < 6; ++u)
                A[u] = *reinterpret_cast<const uint32_t*>(src + (plainIn + (uint32_t)((rBase + u) * pitch)));
        }
    };
    load_trip(0, cur);
    const int trips = stripRows / 6 + 1;                    // stripRows + 6 source rows (stripRows is a multiple of 6)
    for (int r6 = 0; r6 < trips; ++r6) {
        if (r6 + 1 < trips) load_trip(6 * (r6 + 1), nxt);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int r = 6 * r6 + u;
            uint32_t C = cur[u], L, R;
            if (edgeTile) {
                const uint32_t Ld = dpp_from_lower_lane(C);
                C = __builtin_amdgcn_perm(C, Ld, selC);                     // (the identity except in the last column)
                const uint32_t Rd = dpp_from_upper_lane(C);
                R = isLast ? __builtin_amdgcn_perm(C, Ld, selR) : Rd;
                L = isFirst ? __builtin_amdgcn_perm(Rd, C, 0x01020304u) : Ld;
            } else {
                L = dpp_from_lower_lane(C);
                R = dpp_from_upper_lane(C);
            }
            // horizontal sums of the lane's 4 pixels: the window of pixel k starts at byte k + 1 of (L, C, R)
            uint32_t H[4];
            H[0] = __builtin_amdgcn_udot4(C, hblur_weight(1, 1), __builtin_amdgcn_udot4(L, hblur_weight(1, 0), 0u, false), false);
            H[1] = __builtin_amdgcn_udot4(R, hblur_weight(2, 2), __builtin_amdgcn_udot4(C, hblur_weight(2, 1), __builtin_amdgcn_udot4(L, hblur_weight(2, 0), 0u, false), false), false);
            H[2] = __builtin_amdgcn_udot4(R, hblur_weight(3, 2), __builtin_amdgcn_udot4(C, hblur_weight(3, 1), __builtin_amdgcn_udot4(L, hblur_weight(3, 0), 0u, false), false), false);
            H[3] = __builtin_amdgcn_udot4(R, hblur_weight(0, 1), __builtin_amdgcn_udot4(C, hblur_weight(0, 0), 0u, false), false);
#pragma unroll
            for (int k = 0; k < 4; ++k) P[k][(u + 5) % 6] = Hprev[k] | (H[k] << 16);      // rows r - 1, r
            if (r6 > 0) {
                const int yo = yBand + r - 6;                // source rows r - 6 .. r give output row yo
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t a = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, P[k][u]), w0, __umul24(H[k], 18u) + 32768u, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, P[k][(u + 2) % 6]), w1, a, false);
                    o[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, P[k][(u + 4) % 6]), w2, a, false);
                }
                // bytes 2 of the four sums (sum >> 16 < 256)
                const uint32_t out4 = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
                if (stores && yo < h) *reinterpret_cast<uint32_t*>(dst + (plainOut + (uint32_t)((r - 6) * pitch))) = out4;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) Hprev[k] = H[k];
        }
        if (r6 + 1 < trips) {
#pragma unroll
            for (int u = 0; u < 6; ++u) cur[u] = nxt[u];
        }
    }
}

template <int OX>
__device__ __forceinline__ void hblur_rows(const uint8_t* __restrict__ rawBase, uint32_t* __restrict__ hor, int lane) {
    constexpr int kRowPairs = (kRawW + 1) / 2;       // 22
    for (int i = lane; i < kRowPairs * 5; i += 64) {
        const int rp = i / 5, j = i - rp * 5, c0 = 8 * j;
        uint32_t sums[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint8_t* rowb = rawBase + (2 * rp + h) * kRawP + c0;     // 8-byte aligned
            const uint2 q0 = *reinterpret_cast<const uint2*>(rowb), q1 = *reinterpret_cast<const uint2*>(rowb + 8);
            const uint32_t D[5] = {q0.x, q0.y, q1.x, q1.y, *reinterpret_cast<const uint32_t*>(rowb + 16)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = OX + k, m0 = t >> 2, r = t & 3;
                uint32_t s = __builtin_amdgcn_udot4(D[m0], hblur_weight(r, 0), 0u, false);
                s = __builtin_amdgcn_udot4(D[m0 + 1], hblur_weight(r, 1), s, false);
                if (r >= 2) s = __builtin_amdgcn_udot4(D[m0 + 2], hblur_weight(r, 2), s, false);
                sums[h][k] = s;
            }
        }
        uint32_t* dst = hor + c0 * kHorTPD + rp;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < kBlurW) dst[k * kHorTPD] = sums[0][k] | (sums[1][k] << 16);     // a row sum is < 65536
    }
}

// Steered BRIEF (ORBextractor.cc:139-179) over a blurred patch in LDS and the keypoint's output record: lane handles
// comparisons 4 lane .. 4 lane + 3.  patch[(18 + rx) * SX + (18 + ry) * SY] is the blurred pixel (rx, ry) from the keypoint.
template <int SX, int SY>
__device__ __forceinline__ void steered_brief_store(const uint8_t* __restrict__ patch, float angle, int lane,
                                                    const OrbxFrameGeo* __restrict__ geo, const OrbxDescWork& W, int frame,
                                                    int capacity, orbx_keypoint* __restrict__ outKps,
                                                    uint8_t* __restrict__ outDesc) {
    const float deg2rad = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    sincos_f32_fma(__fmul_rn(angle, deg2rad), &b, &a);   // (equal to sincos_f32 on every binary32 input in [0, 2 pi])
    // the lane's 8 sample points = 16 consecutive floats of the pattern table.  Both rotated coordinates of a point
    // come from packed binary32 operations (each product and sum rounded on its own, as in the scalar reference), and
    // adding 1.5 * 2^23 + 18 rounds to the nearest-even integer and leaves (coordinate + 18) in the low mantissa bits.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float4* patp = reinterpret_cast<const float4*>(c_patternF + 16 * lane);
    const float4 pq[4] = {patp[0], patp[1], patp[2], patp[3]};
    const f2 ba = {b, a}, anb = {a, -b};
    const float magic = 12582912.0f + (float)kBlurR;
    uint32_t nib = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float pts[4] = {pq[j].x, pq[j].y, pq[j].z, pq[j].w};
        int t[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float px = pts[2 * e], py = pts[2 * e + 1];
            const f2 u = f2{px, px} * ba;                 // (px b, px a)
            const f2 w = f2{py, py} * anb;                // (py a, -(py b)): the sign change is exact
            const f2 r = u + w;                           // (ry, rx) = (px b + py a, px a - py b)
            // (scalar adds on purpose: with the sum written as a 2-vector this compiler folded both lanes into one)
            const float my = r[0] + magic, mx = r[1] + magic;
            const uint32_t iy = __builtin_bit_cast(uint32_t, my) & 0xFFFFu, ix = __builtin_bit_cast(uint32_t, mx) & 0xFFFFu;
            t[e] = patch[ix * SX + iy * SY];
        }
        nib |= (uint32_t)(t[0] < t[1]) << j;
    }
    const uint32_t other = __shfl_xor(nib, 1);
    const long long row = (long long)frame * capacity + (W.slotLevel & 0xFFFFFF);
    if ((lane & 1) == 0) outDesc[row * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
    if (lane == 0) {
        const int level = W.slotLevel >> 24;
        const OrbxLevelGeo& L = geo->lv[level];
        orbx_keypoint k;
        k.x = W.x; k.y = W.y;
        if (level != 0) { k.x = __fmul_rn(W.x, L.scale); k.y = __fmul_rn(W.y, L.scale); }
        k.size = L.patchSize;
        k.angle = angle;
        k.response = W.response;
        k.octave = level;
        k.class_id = -1;
        outKps[row] = k;
    }
}

__global__ __launch_bounds__(64 * kDescWaves) void k_describe(const OrbxFrameGeo* __restrict__ geo,
                                                              const uint8_t* __restrict__ pyr, long long frameStride,
                                                              const OrbxDescWork* __restrict__ work, int kpTotal,
                                                              unsigned int frameMagic,
                                                              const int* __restrict__ counts,
                                                              orbx_keypoint* __restrict__ outKps,
                                                              uint8_t* __restrict__ outDesc, int capacity) {
    // Per wave: the raw patch (later the blurred core) and the transposed row sums.  Nothing is shared between
    // waves, so a workgroup is ONE wave: a finished keypoint frees its 5.8 KB at once instead of waiting for the
    // slowest wave of a larger group (measured per 64 frames of 752 x 480: 109 us with 8 waves per workgroup, 98
    // with 4, 95 with 2, 91 with 1).
    // (+ one row: the horizontal pass works on row pairs and the last pair's second row is read, never used)
    __shared__ __align__(16) uint8_t s_raw[kDescWaves][(kRawW + 1) * kRawP + 16];   // + the dword the last row's window may touch
    __shared__ __align__(16) uint32_t s_hor[kDescWaves][kBlurW * kHorTPD];   // transposed row sums: 37 columns x 22 row pairs

    // keypoints of one frame read the same few level images: keep a frame on one XCD
    const int logicalBlock = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    // frame = logicalBlock / gridDim.x through the host's reciprocal (exact for this grid, see enqueue_batch)
    const int frame = frameMagic ? (int)__umulhi((unsigned int)logicalBlock, frameMagic) : logicalBlock / (int)gridDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ordinal = (logicalBlock - frame * gridDim.x) * kDescWaves + wv;   // wave-uniform: the frame's ordinal-th keypoint
    // k_layout left one record per kept keypoint (and 0 keypoints for a frame that overflowed its output arrays)
    const bool active = ordinal < counts[2 * frame];
    const OrbxDescWork W = work[(long long)frame * kpTotal + (ordinal < kpTotal ? ordinal : 0)];
    struct { int w, h, pitch, slot; } G;
    G.w = W.wh & 0xFFFF; G.h = W.wh >> 16; G.pitch = W.pitch; G.slot = W.slotLevel & 0xFFFFFF;
    const uint8_t* img = pyr + (long long)frame * frameStride + W.imgOff;
    const int cx = W.cxy & 0xFFFF, cy = W.cxy >> 16;
    ORBX_DSTAMP_BEGIN();
    if (kDescWaves == 1 && !active) return;

    uint8_t* rawBase = s_raw[wv];
    uint32_t* hor = s_hor[wv];
    uint8_t* blur = rawBase;   // the raw patch is dead once the horizontal pass is done

    // Patch rows cy-21 .. cy+21, columns cx-21 .. cx+21.  Interior keypoints fetch each row as three 16-byte
    // loads starting at the 4-byte aligned column xa <= cx-21 (ox = offset of the patch inside the tile row;
    // multi-dword global loads need dword alignment only); keypoints whose patch crosses the level edge take
    // the byte path with REFLECT_101 indices.
    struct __attribute__((packed, aligned(4))) Chunk16 { uint32_t a, b, c, d; };
    const int px0 = cx - kRawR, py0 = cy - kRawR;
    const int xa = px0 & ~3;
    const bool interior = px0 >= 0 && py0 >= 0 && cx + kRawR < G.w && cy + kRawR < G.h && xa + kRawP <= G.pitch;
    const int ox = interior ? px0 - xa : 0;
    if (active) {
        if (interior) {
            // 43 rows x 3 chunks of 16 bytes = 129 loads: issued together (the third round is one lane)
            Chunk16 v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = lane + 64 * u, r = i / 3, d = i - 3 * r;
                if (i < kRawW * (kRawP / 16))
                    v[u] = *reinterpret_cast<const Chunk16*>(img + (long long)(py0 + r) * G.pitch + xa + 16 * d);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = lane + 64 * u, r = i / 3, d = i - 3 * r;
                if (i < kRawW * (kRawP / 16))
                    *reinterpret_cast<uint4*>(rawBase + r * kRawP + 16 * d) = make_uint4(v[u].a, v[u].b, v[u].c, v[u].d);
            }
        } else {
            for (int i = lane; i < kRawW * kRawW; i += 64) {
                const int r = i / kRawW, c = i - r * kRawW;
                const int yy = reflect101_dev(py0 + r, G.h), xx = reflect101_dev(px0 + c, G.w);
                rawBase[r * kRawP + c] = img[(long long)yy * G.pitch + xx];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // this wave's own LDS region: wave order is enough
    ORBX_DSTAMP(0);   // patch load
    ORBX_DESC_PHASE_END(1, rawBase[lane * 29]);

    // ---- orientation: m10 = sum u*I, m01 = sum v*I over the radius-15 disc ----
    // item = (row v, group of 8 columns u0 .. u0 + 7, u0 = 8g - 15): the 8 pixels come as two byte windows cut out of
    // three dwords; sum((u + 15) * I) and sum(I) over the columns inside the disc are v_dot4_u32_u8 against the
    // row's weight bytes (zero outside the disc).
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = lane + 64 * it;
        if (i < 31 * 4) {
            const int row = i >> 2, g = i & 3;
            const int v = row - 15, av = v < 0 ? -v : v;
            const uint4 w = *reinterpret_cast<const uint4*>(&c_orientW.w[4 * (av * 4 + g)]);   // 1 KB table, cache resident
            const int b0 = ox + (kRawR - 15) + 8 * g;
            const uint32_t* rowp = reinterpret_cast<const uint32_t*>(rawBase + (kRawR + v) * kRawP) + (b0 >> 2);
            const uint32_t d0 = rowp[0], d1 = rowp[1], d2 = rowp[2];
            const uint32_t pa = __builtin_amdgcn_alignbyte(d1, d0, b0 & 3), pb = __builtin_amdgcn_alignbyte(d2, d1, b0 & 3);
            const int sumI = (int)__builtin_amdgcn_udot4(pb, w.w, __builtin_amdgcn_udot4(pa, w.y, 0u, false), false);
            const int sumUI = (int)__builtin_amdgcn_udot4(pb, w.z, __builtin_amdgcn_udot4(pa, w.x, 0u, false), false);
            m10 += sumUI - 15 * sumI;
            m01 += v * sumI;
        }
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    ORBX_DSTAMP(1);   // orientation
    ORBX_DESC_PHASE_END(2, rawBase[lane * 29] + (int)angle);

    // ---- separable 7-tap blur of the 37x37 core with the dot-product instructions ----
    // horizontal: see hblur_rows; the patch starts `ox` (0..3, the same for the whole wave) bytes into its LDS rows,
    // and each value of ox has its own instance of the loop so that every byte position is a compile-time constant.
    if (active) {
        switch (ox) {
            case 0: hblur_rows<0>(rawBase, hor, lane); break;
            case 1: hblur_rows<1>(rawBase, hor, lane); break;
            case 2: hblur_rows<2>(rawBase, hor, lane); break;
            default: hblur_rows<3>(rawBase, hor, lane); break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // this wave's own LDS region: wave order is enough
    ORBX_DSTAMP(2);   // horizontal blur
    ORBX_DESC_PHASE_END(3, hor[lane * 11] + (int)angle);
    // vertical: item = (column c, 8 output rows r0..r0+7): 14 row sums = 7 packed pairs, four
    // v_dot2_u32_u16 per output; odd rows use the weight pairs shifted by one tap.  The blurred core is stored
    // transposed as well, blurT[c][r]: an item's 8 outputs are 8 consecutive bytes, one 8-byte store.
    for (int i = lane; active && i < kBlurW * 5; i += 64) {
        const int seg = i / kBlurW, c = i - seg * kBlurW, r0 = seg * 8;
        const uint32_t* colp = hor + c * kHorTPD + (r0 >> 1);
        us2 Hh[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) Hh[m] = __builtin_bit_cast(us2, colp[m]);
        const us2 e0 = {18, 34}, e1 = {48, 56}, e2 = {48, 34}, e3 = {18, 0};      // even rows: taps 0..6 from the pair start
        const us2 o0 = {0, 18}, o1 = {34, 48}, o2 = {56, 48}, o3 = {34, 18};      // odd rows: one tap later
        uint32_t sum[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = __builtin_amdgcn_udot2(Hh[k >> 1], (k & 1) ? o0 : e0, 32768u, false);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = __builtin_amdgcn_udot2(Hh[(k >> 1) + 1], (k & 1) ? o1 : e1, sum[k], false);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = __builtin_amdgcn_udot2(Hh[(k >> 1) + 2], (k & 1) ? o2 : e2, sum[k], false);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = __builtin_amdgcn_udot2(Hh[(k >> 1) + 3], (k & 1) ? o3 : e3, sum[k], false);
        // bytes 2 of the eight sums (sum >> 16 < 256), packed four to a dword
        uint2 out8;
        out8.x = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);
        out8.y = __builtin_amdgcn_perm(sum[5], sum[4], 0x0c0c0602u) | __builtin_amdgcn_perm(sum[7], sum[6], 0x06020c0cu);
        *reinterpret_cast<uint2*>(blur + c * kBlurP + r0) = out8;    // rows 37..39 of a column are padding
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // this wave's own LDS region: wave order is enough
    ORBX_DSTAMP(3);   // vertical blur
    ORBX_DESC_PHASE_END(4, blur[lane * 23] + (int)angle);

    if (!active) return;
    steered_brief_store<kBlurP, 1>(blur, angle, lane, geo, W, frame, capacity, outKps, outDesc);
    ORBX_DSTAMP(4);   // steered tests + output
}

// k_describe_pre: k_describe for a handle whose levels k_blur_levels has blurred already.  Orientation straight from
// the level image (item = row v, group of 8 columns: one 12-byte load, two byte windows, four dot products -- the
// raw patch never goes through LDS); the 37 x 37 blurred patch is fetched into LDS as 16-byte chunks and sampled.
template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) void k_describe_pre(const OrbxFrameGeo* __restrict__ geo, const uint8_t* __restrict__ pyr,
                                                     const uint8_t* __restrict__ blurred, long long frameStride,
                                                     const OrbxDescWork* __restrict__ work, int kpTotal,
                                                     unsigned int frameMagic, const int* __restrict__ counts,
                                                     orbx_keypoint* __restrict__ outKps, uint8_t* __restrict__ outDesc,
                                                     int capacity) {
    constexpr int kPatP = 48;                                        // LDS pitch of the blurred patch: three 16-byte chunks
    __shared__ __align__(16) uint8_t s_patches[kWaves][kBlurW * kPatP + 16];
    const int logicalBlock = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = frameMagic ? (int)__umulhi((unsigned int)logicalBlock, frameMagic) : logicalBlock / (int)gridDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint8_t* s_patch = s_patches[wv];                                // nothing is shared between the waves of a workgroup
    const int ordinal = (logicalBlock - frame * gridDim.x) * kWaves + wv;
    if (ordinal >= counts[2 * frame]) return;
    const OrbxDescWork W = work[(long long)frame * kpTotal + ordinal];
    const int w = W.wh & 0xFFFF, h = W.wh >> 16, pitch = W.pitch;
    const long long imgOff = (long long)frame * frameStride + W.imgOff;
    const uint8_t* __restrict__ img = pyr + imgOff;
    const uint8_t* __restrict__ bimg = blurred + imgOff;
    const int cx = W.cxy & 0xFFFF, cy = W.cxy >> 16;

    // ---- the blurred patch: rows cy-18 .. cy+18, columns from the 4-byte aligned xa <= cx-18 (a chunk may reach past
    // the row's pitch into the next row or, for a frame's last row, the next frame or the slab's padding: loaded, never
    // sampled).  A patch that crosses the level's edge takes the byte path: the blur of the mirrored image is the
    // mirror image of the blur.
    struct __attribute__((packed, aligned(4))) Chunk16 { uint32_t a, b, c, d; };
    const int px0 = cx - kBlurR, py0 = cy - kBlurR;
    const int xa = px0 & ~3;
    const bool interior = px0 >= 0 && py0 >= 0 && cx + kBlurR < w && cy + kBlurR < h;
    const int ox = interior ? px0 - xa : 0;
    if (interior) {
        Chunk16 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = lane + 64 * u, r = i / 3, d = i - 3 * r;
            if (i < kBlurW * 3) v[u] = *reinterpret_cast<const Chunk16*>(bimg + (long long)(py0 + r) * pitch + xa + 16 * d);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = lane + 64 * u, r = i / 3, d = i - 3 * r;
            if (i < kBlurW * 3) *reinterpret_cast<uint4*>(s_patch + r * kPatP + 16 * d) = make_uint4(v[u].a, v[u].b, v[u].c, v[u].d);
        }
    } else {
        for (int i = lane; i < kBlurW * kBlurW; i += 64) {
            const int r = i / kBlurW, c = i - r * kBlurW;
            const int yy = reflect101_dev(py0 + r, h), xx = reflect101_dev(px0 + c, w);
            s_patch[r * kPatP + c] = bimg[(long long)yy * pitch + xx];
        }
    }

    // ---- orientation (ORBextractor.cc:90-137): m10 = sum u I, m01 = sum v I over the radius-15 disc ----
    int m10 = 0, m01 = 0;
    const int ux0 = cx - 15;
    const bool discInside = ux0 >= 0 && cy - 15 >= 0 && cx + 15 < w && cy + 15 < h;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = lane + 64 * it;
        if (i < 31 * 4) {
            const int row = i >> 2, g = i & 3;
            const int v = row - 15, av = v < 0 ? -v : v;
            const uint4 wt = *reinterpret_cast<const uint4*>(&c_orientW.w[4 * (av * 4 + g)]);
            uint32_t pa, pb;
            if (discInside) {
                struct __attribute__((packed, aligned(4))) Chunk12 { uint32_t a, b, c; };
                const int b0 = ux0 + 8 * g;                      // first of the item's 8 columns
                const Chunk12 d = *reinterpret_cast<const Chunk12*>(img + (long long)(cy + v) * pitch + (b0 & ~3));
                pa = __builtin_amdgcn_alignbyte(d.b, d.a, b0 & 3);
                pb = __builtin_amdgcn_alignbyte(d.c, d.b, b0 & 3);
            } else {
                pa = pb = 0;
                const int yy = reflect101_dev(cy + v, h);
                for (int k = 0; k < 4; ++k) {
                    pa |= (uint32_t)img[(long long)yy * pitch + reflect101_dev(ux0 + 8 * g + k, w)] << (8 * k);
                    pb |= (uint32_t)img[(long long)yy * pitch + reflect101_dev(ux0 + 8 * g + 4 + k, w)] << (8 * k);
                }
            }
            const int sumI = (int)__builtin_amdgcn_udot4(pb, wt.w, __builtin_amdgcn_udot4(pa, wt.y, 0u, false), false);
            const int sumUI = (int)__builtin_amdgcn_udot4(pb, wt.z, __builtin_amdgcn_udot4(pa, wt.x, 0u, false), false);
            m10 += sumUI - 15 * sumI;
            m01 += v * sumI;
        }
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the wave's own LDS writes: wave order is enough
    steered_brief_store<1, kPatP>(s_patch + ox, angle, lane, geo, W, frame, capacity, outKps, outDesc);
}

// ======================================================================================
// host side
// ======================================================================================
static inline int round_half_even_f(float v) { return (int)std::nearbyintf(v); }

struct orbx_extractor {
    orbx_params prm;
    int device = 0;
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> quota;
    OrbxFrameGeo geo;
    bool geoValid = false;
    int candShift = 3;            // candCap = level pixels >> candShift

    hipStream_t stream = nullptr;
    OrbxFrameGeo* d_geo = nullptr;
    uint8_t* d_pyr = nullptr;      size_t pyrAlloc = 0;
    OrbxCand* d_cands = nullptr;   size_t candAlloc = 0;
    uint16_t* d_keyNode = nullptr;
    OrbxLevelKp* d_levelKps = nullptr; size_t kpAlloc = 0;
    int* d_candCount = nullptr;    // max_batch * nlevels counters, kCandCounterStride ints apart
    int* d_kpCount = nullptr;      // max_batch * nlevels
    OrbxErr* d_err = nullptr;
    OrbxErr* h_err = nullptr;      // pinned
    orbx_rect* d_rects = nullptr;
    OrbxFastCell* d_cells = nullptr;   // FAST cell table
    OrbxDescWork* d_descWork = nullptr;   // per frame: k_layout's records for k_describe
    // whole-level blur (k_blur_levels + k_describe_pre) instead of a blur per keypoint patch: see choose_blur_mode
    bool levelBlur = false;
    uint8_t* d_blur = nullptr;            // the blurred pyramid: max_batch slabs with the pyramid's layout
    OrbxBlurTile* d_blurTiles = nullptr;
    int nBlurTiles = 0, blurStripRows = 36;
    int *d_xofs = nullptr, *d_yofs = nullptr;
    uint32_t *d_xcoef = nullptr, *d_ycoef = nullptr;
    size_t tapAlloc = 0;
    // staging for the host entry points
    orbx_keypoint* d_outKps = nullptr;
    uint8_t* d_outDesc = nullptr;
    int* d_counts = nullptr;
    int outCap = 0;
    int stageOutCap = 0;           // outCap the pinned result arrays were allocated for
    int lastFrames = 0;
    void* d_scratch = nullptr;     size_t scratchAlloc = 0;
    // pinned / device staging of the host entry points
    uint8_t* h_in = nullptr;       uint8_t* d_in = nullptr;   size_t stageInBytes = 0;
    orbx_keypoint* h_outKps = nullptr;
    uint8_t* h_outDesc = nullptr;
    int* h_counts = nullptr;
    // The host entry points replay their whole command sequence (upload, every kernel, downloads) as one
    // HIP graph: a frame is ~20 dependent commands, and launching them one by one costs more than running them.
    // Large host batches run as a pipeline of chunks instead (extract_host_pipelined): staging copies on helper
    // threads, uploads on copyIn, kernels on `stream`, downloads on copyOut, ordered by one event per chunk and stage.
    hipStream_t copyIn = nullptr, copyOut = nullptr;
    std::vector<hipEvent_t> pipeEvents;
    hipGraphExec_t hostGraph = nullptr;
    struct { int nf, w, h, lap0, lap1; long epoch; } hostGraphKey = {0, 0, 0, 0, 0, -1};
    long geoEpoch = 0;             // bumped whenever the device buffers or the geometry change
    // optional per-kernel timing (hipEvents on the launch stream)
    bool timing = false;
    std::vector<hipEvent_t> evPool;          // kTimedStages + 1 events per recorded batch
    size_t evUsed = 0;
    double stageMs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long stageCalls = 0;
};
constexpr int kTimedStages = 6;   // copy-in, pyramid, fast, quadtree, layout, describe
static hipEvent_t next_event(orbx_extractor* ex) {
    if (ex->evUsed == ex->evPool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ex->evPool.push_back(e);
    }
    return ex->evPool[ex->evUsed++];
}
static void mark(orbx_extractor* ex, hipStream_t st) {
    if (!ex->timing) return;
    hipEvent_t e = next_event(ex);
    if (e) hipEventRecord(e, st);
}
// Folds the recorded events into stageMs (synchronises on the last event).
static void resolve_timing(orbx_extractor* ex) {
    const size_t per = kTimedStages + 1;
    const size_t nb = ex->evUsed / per;
    if (nb == 0) { ex->evUsed = 0; return; }
    hipEventSynchronize(ex->evPool[nb * per - 1]);
    for (size_t b = 0; b < nb; ++b) {
        for (int s = 0; s < kTimedStages; ++s) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ex->evPool[b * per + s], ex->evPool[b * per + s + 1]) == hipSuccess) ex->stageMs[s] += ms;
        }
        ++ex->stageCalls;
    }
    ex->evUsed = 0;
}

static void host_linear_taps(int ssize, int dsize, int* ofs, uint32_t* coef) {
    // cv::resize's coefficient set-up for INTER_LINEAR, 8-bit: 11-bit fixed point
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        const uint32_t a0 = (uint32_t)(uint16_t)(short)round_half_even_f((1.f - f) * 2048.f);
        const uint32_t a1 = (uint32_t)(uint16_t)(short)round_half_even_f(f * 2048.f);
        coef[d] = a0 | (a1 << 16);
    }
}

static void free_geo_buffers(orbx_extractor* ex) {
    hipFree(ex->d_pyr); ex->d_pyr = nullptr;
    hipFree(ex->d_cands); ex->d_cands = nullptr;
    hipFree(ex->d_keyNode); ex->d_keyNode = nullptr;
    hipFree(ex->d_levelKps); ex->d_levelKps = nullptr;
    hipFree(ex->d_xofs); ex->d_xofs = nullptr;
    hipFree(ex->d_yofs); ex->d_yofs = nullptr;
    hipFree(ex->d_xcoef); ex->d_xcoef = nullptr;
    hipFree(ex->d_ycoef); ex->d_ycoef = nullptr;
    hipFree(ex->d_cells); ex->d_cells = nullptr;
    hipFree(ex->d_descWork); ex->d_descWork = nullptr;
    hipFree(ex->d_blur); ex->d_blur = nullptr;
    hipFree(ex->d_blurTiles); ex->d_blurTiles = nullptr;
}

// Whether a handle blurs whole levels once (k_blur_levels, then k_describe_pre samples them) or every keypoint's
// 37 x 37 patch on its own (k_describe).  The level blur costs per pixel of the pyramid and is bound by memory
// bandwidth (it reads and writes every level), the patch blur costs instructions per keypoint.  Measured on an MI355X
// with 752 x 480 images, 64 frames a launch: one launch at a time the two are equal near one keypoint per 300 pixels
// of level 0; with several streams keeping the chip full (where the level blur's memory traffic competes with the
// other kernels) near one per 140 pixels -- 2500 keypoints a frame -- and at 5000 the level blur is 12 % ahead.  The
// second figure decides.  ORBX_BLUR_MODE=patch|level overrides the choice (for measurements).
constexpr int kBlurBreakEvenPixels = 140;
static bool choose_blur_mode(const orbx_params& P, int w, int h) {
    if (const char* e = getenv("ORBX_BLUR_MODE")) {
        if (!strcmp(e, "patch")) return false;
        if (!strcmp(e, "level")) return true;
    }
    return (long long)P.nfeatures * kBlurBreakEvenPixels >= (long long)w * h;
}

// The tiles of k_blur_levels for one frame: per level, full-width tiles of 62 dword columns x one strip of rows, and
// for the columns left over a tile of stacked bands (as many as fit 64 lanes).
static std::vector<OrbxBlurTile> blur_tiles(const OrbxFrameGeo& g, int stripRows) {
    std::vector<OrbxBlurTile> tiles;
    for (int l = 0; l < g.nlevels; ++l) {
        const OrbxLevelGeo& L = g.lv[l];
        const int D = ((L.w - 1) >> 2) + 1, full = D / kBlurTileCols, rem = D - full * kBlurTileCols;
        auto add = [&](int col0, int cw, int nb, int y0) {
            OrbxBlurTile T;
            T.imgOff = (int)L.off;
            T.pitchShape = L.pitch | (cw << 16) | (nb << 24);
            T.wh = L.w | (L.h << 16);
            T.xy = col0 | (y0 << 16);
            tiles.push_back(T);
        };
        for (int y0 = 0; y0 < L.h; y0 += stripRows)
            for (int j = 0; j < full; ++j) add(j * kBlurTileCols, kBlurTileCols, 1, y0);
        if (rem > 0) {
            const int bands = 64 / (rem + 2);
            for (int y0 = 0; y0 < L.h; y0 += stripRows * bands)
                add(full * kBlurTileCols, rem, std::min(bands, (L.h - y0 + stripRows - 1) / stripRows), y0);
        }
    }
    return tiles;
}

// Computes the level / cell / quadtree geometry for a w x h image and (re)allocates the
// device slabs.  Called at creation for the maximum size and again when the size changes.
static orbx_status setup_geometry(orbx_extractor* ex, int w, int h) {
    // Until the last upload below has succeeded the handle has no geometry: a failure on the way (an image too
    // small for the FAST grid, an allocation that fails) must send the next call through here again, never
    // to kernels launched against half-built buffers.  The geometry is built in a local and committed at the end.
    ex->geoValid = false;
    ++ex->geoEpoch;
    if (ex->hostGraph) { hipGraphExecDestroy(ex->hostGraph); ex->hostGraph = nullptr; }
    const orbx_params& P = ex->prm;
    OrbxFrameGeo g;
    memset(&g, 0, sizeof(g));
    g.nlevels = P.nlevels; g.w = w; g.h = h;
    long long off = 0, candOff = 0;
    int kpOff = 0, cellBase = 0, tapX = 0, tapY = 0, maxNode = 0;
    for (int l = 0; l < P.nlevels; ++l) {
        OrbxLevelGeo& L = g.lv[l];
        L.w = round_half_even_f((float)w * ex->inv_sf[l]);
        L.h = round_half_even_f((float)h * ex->inv_sf[l]);
        const float width = (float)(L.w - 2 * kOrbxMinBorder), height = (float)(L.h - 2 * kOrbxMinBorder);
        // A level narrower or lower than one 35-px cell has no cells (ORBextractor.cc:914-915: nCols or nRows is 0
        // and the cell loops do not run): it yields no keypoints and its quota is lost, but it is still the source
        // of the next level.  Level 0 must have a cell, or there is nothing to launch.
        const bool hasCells = width >= 35.f && height >= 35.f;
        ORBX_REQUIRE(hasCells || l > 0,
                     "a %dx%d image is too small for the 35-px FAST grid", w, h);
        ORBX_REQUIRE(L.w >= 8 && L.h >= 8, "level %d of a %dx%d image is %dx%d: fewer than 8 pixels a side", l, w, h, L.w, L.h);
        ORBX_REQUIRE(L.w < 32000 && L.h < 32000, "image too large");
        L.pitch = (L.w + 63) & ~63;
        L.off = off;
        off += (long long)L.pitch * L.h;
        off = (off + 255) & ~255ll;
        L.nCols = hasCells ? (int)(width / 35.f) : 0; L.nRows = hasCells ? (int)(height / 35.f) : 0;
        L.wCell = hasCells ? (int)std::ceil(width / L.nCols) : 0; L.hCell = hasCells ? (int)std::ceil(height / L.nRows) : 0;
        const int maxBX = L.w - kOrbxMinBorder, maxBY = L.h - kOrbxMinBorder;
        L.nColsP = 0; L.nRowsP = 0;
        for (int j = 0; j < L.nCols; ++j) if (!((float)(kOrbxMinBorder + j * L.wCell) >= (float)(maxBX - 3))) L.nColsP = j + 1;
        for (int i = 0; i < L.nRows; ++i) if (!((float)(kOrbxMinBorder + i * L.hCell) >= (float)(maxBY - 3))) L.nRowsP = i + 1;
        ORBX_REQUIRE(L.wCell + 6 <= 75 && L.hCell + 6 <= 75, "FAST cell of level %d is %dx%d: unsupported", l, L.wCell, L.hCell);
        L.cellBase = cellBase;
        cellBase += L.nColsP * L.nRowsP;
        L.quota = ex->quota[l];
        // (a level without cells has no keys to distribute: one root that stays empty)
        L.nIni = hasCells ? (int)std::round((float)(maxBX - kOrbxMinBorder) / (maxBY - kOrbxMinBorder)) : 1;
        ORBX_REQUIRE(L.nIni >= 1, "level %d is taller than twice its width: the quadtree has no root", l);
        L.hX = hasCells ? (float)(maxBX - kOrbxMinBorder) / L.nIni : 1.f;
        long long px = (long long)L.w * L.h;
        L.candCap = (int)std::min<long long>(std::max<long long>(px >> ex->candShift, 2048), (px + 3) / 4 + 16);
        L.candOff = candOff;
        candOff += L.candCap;
        L.nodeCap = std::max(L.quota, 4 * L.nIni) + 8;
        L.kpCap = L.nodeCap;
        L.kpOff = kpOff;
        kpOff += L.kpCap;
        maxNode = std::max(maxNode, L.nodeCap);
        L.scale = ex->sf[l];
        L.invScale = 1 / L.scale;
        L.patchSize = (float)(int)(31 * ex->sf[l]);
        L.tapX = tapX; L.tapY = tapY;
        tapX += (L.w + 3) & ~3; tapY += (L.h + 3) & ~3;
    }
    ORBX_REQUIRE(maxNode <= 2800, "nfeatures too large for the on-chip quadtree (%d nodes per level, limit 2800)", maxNode);
    g.totalCells = cellBase;
    g.pyrBytes = off;
    g.candTotal = candOff;
    g.kpTotal = kpOff;
    g.maxNodeCap = maxNode;
    {
        int mw = 0, mh = 0;
        for (int l = 0; l < P.nlevels; ++l) { mw = std::max(mw, g.lv[l].wCell + 6); mh = std::max(mh, g.lv[l].hCell + 6); }
        g.tileP = (mw + 3 + 3) & ~3;     // widest cell image + alignment shift, multiple of 4
        if (g.tileP <= 48) g.tileP = 48; // the common 35..39-px cells share the kernel instance with a constant pitch
        g.tileR = mh;
        // k_fast's survivor queue shares the memory of the coarse tile when every lane of the quick reject is done
        // with that tile before the first survivor is queued, i.e. when no band of any level has more than the 8
        // rows one pass covers (the same arithmetic as in the kernel, for the widest / tallest cell and every
        // alignment shift).
        int maxBandRows = 0;
        for (int l = 0; l < P.nlevels; ++l)
            for (int ox = 0; ox < 4; ++ox) {
                const int iw = g.lv[l].wCell, ih = g.lv[l].hCell;
                if (iw == 0) continue;     // a level without cells
                const int ncol = ((ox + 3 + iw - 1) >> 2) - ((ox + 3) >> 2) + 1;
                const int rows1 = fast_band_rows(ncol, ih, 64);
                maxBandRows = std::max(maxBandRows, rows1 > 8 ? fast_band_rows(ncol, ih, 128) : rows1);
            }
        const int plane = g.tileP * g.tileR, queueBytes = 2 * ((mw - 6) * (mh - 6) + 4);
        // (+ 16 bytes behind the coarse tile: the quick reject reads one dword past a row's end)
        if (maxBandRows <= 8) { g.fastQueueOff = 2 * plane; g.fastLds = 2 * plane + std::max(plane + 16, queueBytes); }
        else { g.fastQueueOff = 3 * plane + 16; g.fastLds = 3 * plane + 16 + queueBytes; }
    }

    {
        const size_t qtLds = (size_t)maxNode * 56 + kQtThreadsMax * sizeof(int);
        int ldsMax = 0;
        ORBX_HIP_TRY(hipDeviceGetAttribute(&ldsMax, hipDeviceAttributeMaxSharedMemoryPerBlock, ex->device));
        ORBX_REQUIRE((long long)qtLds + 2048 <= ldsMax, "quadtree needs %zu bytes of LDS, the device offers %d", qtLds, ldsMax);
        if (qtLds > 32 * 1024) {
            hipError_t e = hipSuccess;
            for (const void* fn : {reinterpret_cast<const void*>(k_quadtree<128>), reinterpret_cast<const void*>(k_quadtree<256>),
                                   reinterpret_cast<const void*>(k_quadtree<512>)})
                if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qtLds);
            if (e != hipSuccess) {
                orbx_set_error("cannot raise the dynamic LDS limit of the quadtree kernel to %zu (device max %d): %s", qtLds,
                               ldsMax, hipGetErrorString(e));
                return ORBX_ERR_HIP;
            }
        }
    }
    free_geo_buffers(ex);
    const size_t B = (size_t)P.max_batch;
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_pyr, B * (size_t)g.pyrBytes + 256));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_cands, B * (size_t)g.candTotal * sizeof(OrbxCand)));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_keyNode, B * (size_t)g.candTotal * sizeof(uint16_t)));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_levelKps, B * (size_t)g.kpTotal * sizeof(OrbxLevelKp)));
    std::vector<int> xo(tapX, 0), yo(tapY, 0);
    std::vector<uint32_t> xc(tapX, 2048u), yc(tapY, 2048u);
    for (int l = 1; l < P.nlevels; ++l) {
        host_linear_taps(g.lv[l - 1].w, g.lv[l].w, &xo[g.lv[l].tapX], &xc[g.lv[l].tapX]);
        host_linear_taps(g.lv[l - 1].h, g.lv[l].h, &yo[g.lv[l].tapY], &yc[g.lv[l].tapY]);
        // padding entries repeat the last column so that the 4-wide loads stay in range
        for (int x = g.lv[l].w; x < ((g.lv[l].w + 3) & ~3); ++x) {
            xo[g.lv[l].tapX + x] = xo[g.lv[l].tapX + g.lv[l].w - 1];
            xc[g.lv[l].tapX + x] = xc[g.lv[l].tapX + g.lv[l].w - 1];
        }
    }
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_xofs, std::max(tapX, 4) * sizeof(int)));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_yofs, std::max(tapY, 4) * sizeof(int)));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_xcoef, std::max(tapX, 4) * sizeof(uint32_t)));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_ycoef, std::max(tapY, 4) * sizeof(uint32_t)));
    ORBX_HIP_TRY(hipMemcpy(ex->d_xofs, xo.data(), tapX * sizeof(int), hipMemcpyHostToDevice));
    ORBX_HIP_TRY(hipMemcpy(ex->d_yofs, yo.data(), tapY * sizeof(int), hipMemcpyHostToDevice));
    ORBX_HIP_TRY(hipMemcpy(ex->d_xcoef, xc.data(), tapX * sizeof(uint32_t), hipMemcpyHostToDevice));
    ORBX_HIP_TRY(hipMemcpy(ex->d_ycoef, yc.data(), tapY * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
        // per-cell constants of the FAST kernel (one record per processed cell, all levels)
        std::vector<OrbxFastCell> cells(std::max(g.totalCells, 1));
        for (int l = 0; l < P.nlevels; ++l) {
            const OrbxLevelGeo& L = g.lv[l];
            const int maxBX = L.w - kOrbxMinBorder, maxBY = L.h - kOrbxMinBorder;
            for (int c = 0; c < L.nColsP * L.nRowsP; ++c) {
                const int ci = c / L.nColsP, cj = c - ci * L.nColsP;
                const int x0 = kOrbxMinBorder + cj * L.wCell, y0 = kOrbxMinBorder + ci * L.hCell;
                const int x1 = std::min(x0 + L.wCell + 6, maxBX), y1 = std::min(y0 + L.hCell + 6, maxBY);
                OrbxFastCell& C = cells[L.cellBase + c];
                C.xy = x0 | (y0 << 16);
                C.dims = (x1 - x0) | ((y1 - y0) << 8) | (l << 16);
                C.index = c;
                C.apps = L.nColsP * L.nRowsP - c;
                const long long tileOff = L.off + (long long)y0 * L.pitch + (x0 & ~3);
                ORBX_REQUIRE(tileOff < (1ll << 31), "pyramid slab of one frame exceeds 2 GiB");
                C.tileOff = (int)tileOff;
                C.pitch = L.pitch;
                // lane layout of the quick reject (k_fast): one lane per (band of rows, dword column)
                const int ox = x0 & 3, iw = x1 - x0 - 6, ih = y1 - y0 - 6;
                if (iw > 0 && ih > 0) {
                    const int ncol = ((ox + 3 + iw - 1) >> 2) - ((ox + 3) >> 2) + 1;
                    const bool both = fast_band_rows(ncol, ih, 64) > 8;
                    const int lanes = both ? 128 : 64;
                    const int q = lanes / ncol, nb = q < 1 ? 1 : (q < ih ? q : ih);
                    C.bands = ncol | (nb << 8) | (fast_band_rows(ncol, ih, lanes) << 16) | ((both ? 1 : 0) << 24);
                    C.colRcp = 65536 / ncol + 1;
                } else {
                    C.bands = 1 | (1 << 8) | (1 << 16);
                    C.colRcp = 65536 + 1;
                }
            }
        }
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_cells, cells.size() * sizeof(OrbxFastCell)));
        ORBX_HIP_TRY(hipMemcpy(ex->d_cells, cells.data(), cells.size() * sizeof(OrbxFastCell), hipMemcpyHostToDevice));
        // k_layout's records for k_describe: one per keypoint record of a frame
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_descWork, B * (size_t)std::max(g.kpTotal, 1) * sizeof(OrbxDescWork)));
    }
    ex->levelBlur = choose_blur_mode(P, w, h);
    if (ex->levelBlur) {
        ORBX_REQUIRE(g.pyrBytes < (1ll << 31), "pyramid slab of one frame exceeds 2 GiB");
        if (const char* e = getenv("ORBX_BLUR_STRIP_ROWS")) {
            const int v = atoi(e);
            if (v >= 6 && v <= 1200 && v % 6 == 0) ex->blurStripRows = v;
        }
        const std::vector<OrbxBlurTile> tiles = blur_tiles(g, ex->blurStripRows);
        ex->nBlurTiles = (int)tiles.size();
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_blur, B * (size_t)g.pyrBytes + 256));
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_blurTiles, tiles.size() * sizeof(OrbxBlurTile)));
        ORBX_HIP_TRY(hipMemcpy(ex->d_blurTiles, tiles.data(), tiles.size() * sizeof(OrbxBlurTile), hipMemcpyHostToDevice));
    }
    ORBX_HIP_TRY(hipMemcpy(ex->d_geo, &g, sizeof(g), hipMemcpyHostToDevice));
    ex->geo = g;
    ex->geoValid = true;
    return ORBX_OK;
}

// Scale tables and per-level quotas: ORBextractor.cc:481-512 (scaleFactor is a double member).  Host arithmetic only.
static void host_scale_tables(int nfeatures, float scale_factor, int nl, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2,
                              int* quota) {
    const double scaleFactor = (double)scale_factor;
    sf[0] = 1.f; sigma2[0] = 1.f;
    for (int i = 1; i < nl; ++i) { sf[i] = (float)(sf[i - 1] * scaleFactor); sigma2[i] = sf[i] * sf[i]; }
    for (int i = 0; i < nl; ++i) { inv_sf[i] = 1.0f / sf[i]; inv_sigma2[i] = 1.0f / sigma2[i]; }
    const float factor = (float)(1.0f / scaleFactor);
    float per = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; ++l) { quota[l] = round_half_even_f(per); sum += quota[l]; per *= factor; }
    quota[nl - 1] = std::max(nfeatures - sum, 0);
}

extern "C" orbx_status orbx_scale_tables(int nfeatures, float scale_factor, int nlevels, float* scale, float* inv_scale,
                                         float* sigma2, float* inv_sigma2, int* features_per_level) {
    ORBX_REQUIRE(nlevels >= 1 && nlevels <= kOrbxMaxLevels, "nlevels must be in [1, %d]", kOrbxMaxLevels);
    ORBX_REQUIRE(nfeatures >= 1 && scale_factor > 1.0f, "bad nfeatures / scale_factor");
    float sf[kOrbxMaxLevels], isf[kOrbxMaxLevels], s2[kOrbxMaxLevels], is2[kOrbxMaxLevels];
    int q[kOrbxMaxLevels];
    host_scale_tables(nfeatures, scale_factor, nlevels, sf, isf, s2, is2, q);
    for (int i = 0; i < nlevels; ++i) {
        if (scale) scale[i] = sf[i];
        if (inv_scale) inv_scale[i] = isf[i];
        if (sigma2) sigma2[i] = s2[i];
        if (inv_sigma2) inv_sigma2[i] = is2[i];
        if (features_per_level) features_per_level[i] = q[i];
    }
    return ORBX_OK;
}

extern "C" orbx_status orbx_extractor_create(const orbx_params* p, orbx_extractor** out) {
    ORBX_REQUIRE(p && out, "null argument");
    ORBX_REQUIRE(p->nlevels >= 1 && p->nlevels <= kOrbxMaxLevels, "nlevels must be in [1, %d]", kOrbxMaxLevels);
    ORBX_REQUIRE(p->nfeatures >= 1, "nfeatures must be positive");
    ORBX_REQUIRE(p->scale_factor > 1.0f, "scale_factor must be > 1");
    ORBX_REQUIRE(p->max_width > 0 && p->max_height > 0 && p->max_batch >= 1, "bad size / batch");
    int ndev = 0;
    ORBX_HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { orbx_set_error("no HIP device"); return ORBX_ERR_HIP; }
    orbx_extractor* ex = new orbx_extractor;
    ex->prm = *p;
    if (p->device >= 0) ORBX_HIP_TRY(hipSetDevice(p->device));
    ORBX_HIP_TRY(hipGetDevice(&ex->device));
    const int nl = p->nlevels;
    ex->sf.resize(nl); ex->inv_sf.resize(nl); ex->sigma2.resize(nl); ex->inv_sigma2.resize(nl); ex->quota.resize(nl);
    host_scale_tables(p->nfeatures, p->scale_factor, nl, ex->sf.data(), ex->inv_sf.data(), ex->sigma2.data(), ex->inv_sigma2.data(),
                      ex->quota.data());

    orbx_status st = ORBX_OK;
    auto fail = [&](orbx_status s) { orbx_extractor_destroy(ex); return s; };
    if (hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking) != hipSuccess) { orbx_set_error("hipStreamCreate failed"); return fail(ORBX_ERR_HIP); }
    const size_t B = (size_t)p->max_batch;
    if (hipMalloc((void**)&ex->d_geo, sizeof(OrbxFrameGeo)) != hipSuccess ||
        hipMalloc((void**)&ex->d_candCount, B * nl * kCandCounterStride * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&ex->d_kpCount, B * nl * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&ex->d_err, sizeof(OrbxErr)) != hipSuccess ||
        hipMalloc((void**)&ex->d_rects, kOrbxMaxRects * sizeof(orbx_rect)) != hipSuccess ||
        hipHostMalloc((void**)&ex->h_err, sizeof(OrbxErr)) != hipSuccess) {
        orbx_set_error("device allocation failed");
        return fail(ORBX_ERR_HIP);
    }
    memset(ex->h_err, 0, sizeof(OrbxErr));
    if (hipMemset(ex->d_err, 0, sizeof(OrbxErr)) != hipSuccess) { orbx_set_error("hipMemset failed"); return fail(ORBX_ERR_HIP); }
    st = setup_geometry(ex, p->max_width, p->max_height);
    if (st != ORBX_OK) return fail(st);
    ex->outCap = orbx_extractor_max_keypoints(ex);
    if (hipMalloc((void**)&ex->d_outKps, B * ex->outCap * sizeof(orbx_keypoint)) != hipSuccess ||
        hipMalloc((void**)&ex->d_outDesc, B * ex->outCap * 32) != hipSuccess ||
        hipMalloc((void**)&ex->d_counts, B * 2 * sizeof(int)) != hipSuccess) {
        orbx_set_error("device allocation failed");
        return fail(ORBX_ERR_HIP);
    }
    *out = ex;
    return ORBX_OK;
}

extern "C" void orbx_extractor_destroy(orbx_extractor* ex) {
    if (!ex) return;
    hipSetDevice(ex->device);
    if (ex->stream) hipStreamSynchronize(ex->stream);
    if (ex->hostGraph) hipGraphExecDestroy(ex->hostGraph);
    if (ex->copyIn) { hipStreamSynchronize(ex->copyIn); hipStreamDestroy(ex->copyIn); }
    if (ex->copyOut) { hipStreamSynchronize(ex->copyOut); hipStreamDestroy(ex->copyOut); }
    if (ex->stream) hipStreamDestroy(ex->stream);
    for (hipEvent_t e : ex->evPool) hipEventDestroy(e);
    for (hipEvent_t e : ex->pipeEvents) hipEventDestroy(e);
    free_geo_buffers(ex);
    hipFree(ex->d_geo); hipFree(ex->d_candCount); hipFree(ex->d_kpCount); hipFree(ex->d_err);
    hipFree(ex->d_rects); hipFree(ex->d_scratch); hipFree(ex->d_outKps); hipFree(ex->d_outDesc); hipFree(ex->d_counts);
    if (ex->h_err) hipHostFree(ex->h_err);
    if (ex->h_in) hipHostFree(ex->h_in);
    if (ex->d_in) hipFree(ex->d_in);
    if (ex->h_outKps) hipHostFree(ex->h_outKps);
    if (ex->h_outDesc) hipHostFree(ex->h_outDesc);
    if (ex->h_counts) hipHostFree(ex->h_counts);
    delete ex;
}

extern "C" orbx_status orbx_extractor_tables(const orbx_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                                             float* inv_sigma2, int* features_per_level) {
    ORBX_REQUIRE(ex, "null handle");
    for (int i = 0; i < ex->prm.nlevels; ++i) {
        if (scale) scale[i] = ex->sf[i];
        if (inv_scale) inv_scale[i] = ex->inv_sf[i];
        if (sigma2) sigma2[i] = ex->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->inv_sigma2[i];
        if (features_per_level) features_per_level[i] = ex->quota[i];
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_max_keypoints(const orbx_extractor* ex) {
    if (!ex) return 0;
    return ex->geo.kpTotal;
}

static orbx_status ensure_geometry(orbx_extractor* ex, int w, int h) {
    ORBX_REQUIRE(w <= ex->prm.max_width && h <= ex->prm.max_height, "image %dx%d exceeds the handle's %dx%d", w, h,
                 ex->prm.max_width, ex->prm.max_height);
    if (ex->geoValid && ex->geo.w == w && ex->geo.h == h) return ORBX_OK;
    ORBX_HIP_TRY(hipStreamSynchronize(ex->stream));
    return setup_geometry(ex, w, h);
}

// Enqueues every kernel of one batch.  Level 0 of each pyramid slab must already hold the image; the candidate
// counters are cleared here unless the caller's copy kernel has done it (countersCleared).
// frame0: the batch occupies the handle's frame slots frame0 .. frame0 + nf - 1 (every per-frame array of the
// handle is indexed linearly by the frame, so the kernels simply receive offset base pointers); d_kps / d_desc /
// d_counts are the rows of the batch's first frame.  dyn == nullptr with ndyn > 0: the rectangles are already on
// the device (a later chunk of the same host call).
static orbx_status enqueue_batch(orbx_extractor* ex, int nf, int lap0, int lap1, const orbx_rect* dyn, int ndyn,
                                 orbx_keypoint* d_kps, uint8_t* d_desc, int capacity, int* d_counts, hipStream_t st,
                                 bool countersCleared = false, int frame0 = 0) {
    const OrbxFrameGeo& g = ex->geo;
    const int nl = g.nlevels;
    ORBX_REQUIRE(ndyn >= 0 && ndyn <= kOrbxMaxRects, "at most %d dynamic rectangles", kOrbxMaxRects);
    ORBX_REQUIRE(frame0 >= 0 && frame0 + nf <= ex->prm.max_batch, "frames %d .. %d outside the handle's %d slots", frame0,
                 frame0 + nf - 1, ex->prm.max_batch);
    if (ndyn > 0 && dyn) ORBX_HIP_TRY(hipMemcpyAsync(ex->d_rects, dyn, ndyn * sizeof(orbx_rect), hipMemcpyHostToDevice, st));
    // the slots' views of the handle's arrays
    struct {
        uint8_t* d_pyr; OrbxCand* d_cands; uint16_t* d_keyNode; OrbxLevelKp* d_levelKps; int *d_candCount, *d_kpCount;
        OrbxDescWork* d_descWork;
    } V;
    V.d_pyr = ex->d_pyr + (size_t)frame0 * g.pyrBytes;
    V.d_cands = ex->d_cands + (size_t)frame0 * g.candTotal;
    V.d_keyNode = ex->d_keyNode + (size_t)frame0 * g.candTotal;
    V.d_levelKps = ex->d_levelKps + (size_t)frame0 * g.kpTotal;
    V.d_candCount = ex->d_candCount + (size_t)frame0 * nl * kCandCounterStride;
    V.d_kpCount = ex->d_kpCount + (size_t)frame0 * nl;
    V.d_descWork = ex->d_descWork + (size_t)frame0 * g.kpTotal;
    // the only per-batch state that must start at zero; the quadtree writes every keypoint count and
    // the error record is cleared when it is read (orbx_extract_status)
    if (!countersCleared) ORBX_HIP_TRY(hipMemsetAsync(V.d_candCount, 0, (size_t)nf * nl * kCandCounterStride * sizeof(int), st));
    if (ex->timing && ex->evUsed + kTimedStages + 1 > 4096 * (size_t)(kTimedStages + 1)) resolve_timing(ex);
    mark(ex, st);   // end of copy-in (its start is marked by the caller)
    for (int l = 1; l < nl; ++l) {
        const int lanes = (g.lv[l].w + 3) / 4;
        const int groups = (g.lv[l].h + kResizeRows - 1) / kResizeRows;
        const int pyrThreads = pyr_threads_for(g.w, g.h);
        dim3 grid((unsigned)(((long long)lanes * groups + pyrThreads - 1) / pyrThreads), nf);
        hipLaunchKernelGGL(k_resize, grid, dim3(pyrThreads), 0, st, ex->d_geo, l, V.d_pyr, g.pyrBytes, ex->d_xofs, ex->d_xcoef,
                           ex->d_yofs, ex->d_ycoef, lanes, lanes == 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / (unsigned)lanes));
    }
    mark(ex, st);
    // the kernel divides its block index by the cells per frame with a reciprocal, m = floor(2^32 / d) + 1, exact
    // while blocks * d < 2^32
    const unsigned int cellMagic = ((unsigned long long)g.totalCells * nf) * g.totalCells < (1ull << 32) && g.totalCells > 1
                                       ? (unsigned int)((1ull << 32) / (unsigned int)g.totalCells) + 1u : 0u;
    if (g.tileP == 48)
        hipLaunchKernelGGL(k_fast<48>, dim3(g.totalCells, nf), dim3(kFastThreads), (size_t)g.fastLds, st, ex->d_geo, ex->d_cells,
                           V.d_pyr, g.pyrBytes, V.d_cands, V.d_candCount, ex->prm.ini_th_fast, ex->prm.min_th_fast,
                           ex->d_rects, ndyn, ex->d_err, g.tileP, g.tileR, g.fastQueueOff, nl, cellMagic);
    else
        hipLaunchKernelGGL(k_fast<0>, dim3(g.totalCells, nf), dim3(kFastThreads), (size_t)g.fastLds, st, ex->d_geo, ex->d_cells,
                           V.d_pyr, g.pyrBytes, V.d_cands, V.d_candCount, ex->prm.ini_th_fast, ex->prm.min_th_fast,
                           ex->d_rects, ndyn, ex->d_err, g.tileP, g.tileR, g.fastQueueOff, nl, cellMagic);
    mark(ex, st);
    const size_t qtLds = (size_t)g.maxNodeCap * 56 + kQtThreadsMax * sizeof(int);
    switch (qt_threads_for(ex->prm.nfeatures, nf)) {
        case 128:
            hipLaunchKernelGGL(k_quadtree<128>, dim3(nl, nf), dim3(128), qtLds, st, ex->d_geo, V.d_cands, V.d_candCount,
                               V.d_keyNode, V.d_levelKps, V.d_kpCount, ex->d_err);
            break;
        case 256:
            hipLaunchKernelGGL(k_quadtree<256>, dim3(nl, nf), dim3(256), qtLds, st, ex->d_geo, V.d_cands, V.d_candCount,
                               V.d_keyNode, V.d_levelKps, V.d_kpCount, ex->d_err);
            break;
        default:
            hipLaunchKernelGGL(k_quadtree<512>, dim3(nl, nf), dim3(512), qtLds, st, ex->d_geo, V.d_cands, V.d_candCount,
                               V.d_keyNode, V.d_levelKps, V.d_kpCount, ex->d_err);
    }
    mark(ex, st);
    hipLaunchKernelGGL(k_layout, dim3(nf), dim3(kLayThreads), 0, st, ex->d_geo, V.d_levelKps, V.d_kpCount, lap0, lap1,
                       capacity, d_counts, V.d_descWork, ex->d_err);
    mark(ex, st);
    // A frame keeps at most min(kpTotal, capacity) keypoints: one wave each.  The kernel divides its block index
    // by the grid width with a reciprocal, m = floor(2^32 / d) + 1, exact while blocks * d < 2^32.
    const unsigned int descGrid = (unsigned int)((std::min(g.kpTotal, capacity) + kDescWaves - 1) / kDescWaves);
    const unsigned int frameMagic =
        ((unsigned long long)descGrid * nf) * descGrid < (1ull << 32) && descGrid > 1 ? (unsigned int)((1ull << 32) / descGrid) + 1u : 0u;
    if (descGrid > 0 && ex->levelBlur) {
        uint8_t* blurView = ex->d_blur + (size_t)frame0 * g.pyrBytes;
        hipLaunchKernelGGL(k_blur_levels, dim3((ex->nBlurTiles + 3) / 4, nf), dim3(256), 0, st, ex->d_blurTiles, ex->nBlurTiles,
                           ex->blurStripRows, V.d_pyr, blurView, g.pyrBytes);
        // (one wave per workgroup, as for k_describe: 2, 4 and 8 waves measured within 3 % of it)
        static_assert(kDescWaves == 1, "k_describe_pre shares k_describe's grid: one keypoint per workgroup");
        hipLaunchKernelGGL(k_describe_pre<1>, dim3(descGrid, nf), dim3(64), 0, st, ex->d_geo, V.d_pyr, blurView, g.pyrBytes,
                           V.d_descWork, g.kpTotal, frameMagic, d_counts, d_kps, d_desc, capacity);
    } else if (descGrid > 0)
        hipLaunchKernelGGL(k_describe, dim3(descGrid, nf), dim3(64 * kDescWaves), 0, st, ex->d_geo, V.d_pyr, g.pyrBytes,
                           V.d_descWork, g.kpTotal, frameMagic, d_counts, d_kps, d_desc, capacity);
    mark(ex, st);
    ORBX_HIP_TRY(hipGetLastError());
    ex->lastFrames = std::max(frame0 + nf, frame0 ? ex->lastFrames : 0);
    return ORBX_OK;
}

// Level-0 copy + every kernel for n_frames images into the handle's frame slots frame0 .. (geometry already ensured).
static orbx_status enqueue_images(orbx_extractor* ex, const uint8_t* d_images, int n_frames, int width, int height,
                                  ptrdiff_t row_stride, ptrdiff_t frame_stride, int lap0, int lap1, const orbx_rect* dyn_host,
                                  int ndyn, orbx_keypoint* d_keypoints, uint8_t* d_descriptors, int capacity, int* d_counts,
                                  hipStream_t st, int frame0) {
    const int chunksPerRow = (width + 15) / 16;
    const int pyrThreads = pyr_threads_for(width, height);
    dim3 grid((unsigned)(((long long)chunksPerRow * height + pyrThreads - 1) / pyrThreads), n_frames);
    const int aligned16 = ((((uintptr_t)d_images) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 15) == 0;
    mark(ex, st);
    hipLaunchKernelGGL(k_copy_level0, grid, dim3(pyrThreads), 0, st, ex->d_geo, d_images, (long long)row_stride,
                       (long long)frame_stride, ex->d_pyr + (size_t)frame0 * ex->geo.pyrBytes, ex->geo.pyrBytes, aligned16,
                       chunksPerRow, ex->d_candCount + (size_t)frame0 * ex->geo.nlevels * kCandCounterStride);
    return enqueue_batch(ex, n_frames, lap0, lap1, dyn_host, ndyn, d_keypoints, d_descriptors, capacity, d_counts, st, true, frame0);
}

extern "C" orbx_status orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, int n_frames, int width,
                                                 int height, ptrdiff_t row_stride, ptrdiff_t frame_stride, int lap0,
                                                 int lap1, const orbx_rect* dyn_host, int ndyn, orbx_keypoint* d_keypoints,
                                                 uint8_t* d_descriptors, int capacity, int* d_counts, void* stream) {
    ORBX_REQUIRE(ex && d_images && d_keypoints && d_descriptors && d_counts, "null argument");
    ORBX_REQUIRE(n_frames >= 1 && n_frames <= ex->prm.max_batch, "n_frames %d outside [1, %d]", n_frames, ex->prm.max_batch);
    ORBX_REQUIRE(width > 0 && height > 0 && row_stride >= width, "bad image size / stride");
    ORBX_REQUIRE(ndyn == 0 || dyn_host, "null rectangle list");
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    orbx_status s = ensure_geometry(ex, width, height);
    if (s != ORBX_OK) return s;
    // stream taken literally: NULL is the device's default stream
    return enqueue_images(ex, d_images, n_frames, width, height, row_stride, frame_stride, lap0, lap1, dyn_host, ndyn,
                          d_keypoints, d_descriptors, capacity, d_counts, (hipStream_t)stream, 0);
}

extern "C" orbx_status orbx_extract_batch_staged_device(orbx_extractor* ex, orbx_stager* stager, const uint8_t* d_src,
                                                        int n_frames, ptrdiff_t src_row_stride,
                                                        ptrdiff_t src_frame_stride, int lap0, int lap1,
                                                        const orbx_rect* dyn_host, int ndyn, orbx_keypoint* d_keypoints,
                                                        uint8_t* d_descriptors, int capacity, int* d_counts,
                                                        void* stream) {
    ORBX_REQUIRE(ex && stager && d_src && d_keypoints && d_descriptors && d_counts, "null argument");
    ORBX_REQUIRE(n_frames >= 1 && n_frames <= ex->prm.max_batch, "n_frames %d outside [1, %d]", n_frames, ex->prm.max_batch);
    int width = 0, height = 0;
    orbx_status s = orbx_stager_output_size(stager, &width, &height);
    if (s != ORBX_OK) return s;
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    s = ensure_geometry(ex, width, height);
    if (s != ORBX_OK) return s;
    hipStream_t st = (hipStream_t)stream;
    mark(ex, st);
    const OrbxLevelGeo& L0 = ex->geo.lv[0];
    s = orbx_stage_enqueue(stager, d_src, n_frames, (long long)src_row_stride, (long long)src_frame_stride,
                           ex->d_pyr + L0.off, L0.pitch, ex->geo.pyrBytes, st);
    if (s != ORBX_OK) return s;
    return enqueue_batch(ex, n_frames, lap0, lap1, dyn_host, ndyn, d_keypoints, d_descriptors, capacity, d_counts, st);
}

// Acts on an error record that has been read back from the device (clears the device copy if it was set).
static orbx_status extract_status_from(orbx_extractor* ex, const OrbxErr e) {
    if (e.cand_overflow || e.kp_overflow || e.out_overflow || e.need_cand)
        ORBX_HIP_TRY(hipMemset(ex->d_err, 0, sizeof(OrbxErr)));
    if (e.cand_overflow) {
        // regrow so that the next call fits
        while (ex->candShift > 0) {
            --ex->candShift;
            bool ok = true;
            for (int l = 0; l < ex->geo.nlevels; ++l)
                if (((long long)ex->geo.lv[l].w * ex->geo.lv[l].h >> ex->candShift) < e.need_cand) ok = false;
            if (ok) break;
        }
        const int w = ex->geo.w, h = ex->geo.h;
        hipStreamSynchronize(ex->stream);
        orbx_status s = setup_geometry(ex, w, h);
        if (s != ORBX_OK) return s;
        orbx_set_error("FAST candidate buffer overflowed (%d needed); buffers regrown, call again", e.need_cand);
        return ORBX_ERR_CAPACITY;
    }
    if (e.kp_overflow) { orbx_set_error("quadtree node capacity exceeded"); return ORBX_ERR_CAPACITY; }
    if (e.out_overflow) { orbx_set_error("output capacity too small for the keypoints of a frame"); return ORBX_ERR_CAPACITY; }
    return ORBX_OK;
}

extern "C" orbx_status orbx_extract_status(orbx_extractor* ex) {
    ORBX_REQUIRE(ex, "null handle");
    // the caller has synchronised the stream the batch was launched on
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    ORBX_HIP_TRY(hipMemcpy(ex->h_err, ex->d_err, sizeof(OrbxErr), hipMemcpyDeviceToHost));
    return extract_status_from(ex, *ex->h_err);
}

// Host entry points: images are packed into a pinned staging buffer, uploaded with one linear copy,
// run through the device path, and the result arrays come back with three linear copies into pinned
// memory.  (Row-pitched copies straight from pageable memory cost milliseconds per image.)
static orbx_status ensure_host_staging(orbx_extractor* ex) {
    const size_t inBytes = (size_t)ex->prm.max_batch * ex->prm.max_width * ex->prm.max_height;
    const size_t B = (size_t)ex->prm.max_batch;
    if (ex->h_in && ex->stageInBytes >= inBytes && ex->stageOutCap >= ex->outCap) return ORBX_OK;
    if (ex->h_in) { hipHostFree(ex->h_in); ex->h_in = nullptr; }
    if (ex->d_in) { hipFree(ex->d_in); ex->d_in = nullptr; }
    if (ex->h_outKps) { hipHostFree(ex->h_outKps); ex->h_outKps = nullptr; }
    if (ex->h_outDesc) { hipHostFree(ex->h_outDesc); ex->h_outDesc = nullptr; }
    if (ex->h_counts) { hipHostFree(ex->h_counts); ex->h_counts = nullptr; }
    ORBX_HIP_TRY(hipHostMalloc((void**)&ex->h_in, inBytes));
    ORBX_HIP_TRY(hipMalloc((void**)&ex->d_in, inBytes + 16));
    ORBX_HIP_TRY(hipHostMalloc((void**)&ex->h_outKps, B * ex->outCap * sizeof(orbx_keypoint)));
    ORBX_HIP_TRY(hipHostMalloc((void**)&ex->h_outDesc, B * ex->outCap * 32));
    ORBX_HIP_TRY(hipHostMalloc((void**)&ex->h_counts, B * 2 * sizeof(int)));
    ex->stageInBytes = inBytes;
    ex->stageOutCap = ex->outCap;
    ++ex->geoEpoch;
    return ORBX_OK;
}

static int host_chunk_frames(size_t frameBytes, int n_frames);
constexpr int kHostDirectOutFrames = 2;   // host calls of up to this many frames let the kernels write the pinned result arrays
static orbx_status extract_host_pipelined(orbx_extractor* ex, const uint8_t* const* images, int n_frames, int width,
                                          int height, ptrdiff_t stride, int lap0, int lap1, orbx_keypoint* keypoints,
                                          uint8_t* descriptors, int capacity, int* n_out, int* mono_index_out);

static orbx_status extract_host(orbx_extractor* ex, const uint8_t* const* images, int n_frames, int width, int height,
                                ptrdiff_t stride, int lap0, int lap1, const orbx_rect* dyn, int ndyn,
                                orbx_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out,
                                int* mono_index_out) {
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    orbx_status s = ensure_geometry(ex, width, height);
    if (s != ORBX_OK) return s;
    if (ex->geo.kpTotal > ex->outCap) {
        // A size whose levels can hold more keypoints than those of the size the handle was created for (a wider
        // aspect ratio has more quadtree roots per level): the result arrays grow with it.
        ORBX_HIP_TRY(hipStreamSynchronize(ex->stream));
        const size_t B = (size_t)ex->prm.max_batch;
        const int want = ex->geo.kpTotal;
        hipFree(ex->d_outKps); ex->d_outKps = nullptr;
        hipFree(ex->d_outDesc); ex->d_outDesc = nullptr;
        ex->outCap = 0;            // until both allocations have succeeded
        ++ex->geoEpoch;
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_outKps, B * want * sizeof(orbx_keypoint)));
        ORBX_HIP_TRY(hipMalloc((void**)&ex->d_outDesc, B * want * 32));
        ex->outCap = want;
    }
    s = ensure_host_staging(ex);
    if (s != ORBX_OK) return s;
    const size_t frameBytes = (size_t)width * height;
    // 20 MB of images or more (measured on MI355X / PCIe 5, 752 x 480 frames per call, single sequence against pipeline:
    // 48 frames 46.7 k against 36 k frames/s, 64 frames 47.8 k against 54 k, 96 frames 48.2 k against 64 k -- the pipeline
    // has about a millisecond of start-up in its helper threads): overlap staging, upload, kernels and download.
    // Rectangles and per-kernel timing keep the single-sequence path below.  (ORBX_HOST_PIPELINE_MIN_MB moves the limit.)
    static const size_t pipelineMinBytes = [] {
        const char* v = getenv("ORBX_HOST_PIPELINE_MIN_MB");
        return (size_t)(v ? std::max(1, atoi(v)) : 20) << 20;
    }();
    if (ndyn == 0 && !ex->timing && n_frames >= 12 && (size_t)n_frames * frameBytes >= pipelineMinBytes &&
        !getenv("ORBX_HOST_NO_PIPELINE"))
        return extract_host_pipelined(ex, images, n_frames, width, height, stride, lap0, lap1, keypoints, descriptors, capacity,
                                      n_out, mono_index_out);
    for (int f = 0; f < n_frames; ++f) {
        if (!images[f]) { orbx_set_error("empty image"); return ORBX_ERR_EMPTY; }
        uint8_t* dst = ex->h_in + (size_t)f * frameBytes;
        if (stride == width) memcpy(dst, images[f], frameBytes);
        else for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * width, images[f] + (ptrdiff_t)y * stride, width);
    }
    const size_t B = (size_t)n_frames;
    // One or two frames: k_describe writes its keypoint and descriptor rows straight into the pinned result arrays
    // (hipHostMalloc memory is visible to the device under the same address), 60 KB per frame in 28- and 32-byte stores,
    // instead of into device arrays that two copy commands then bring over: a call of this size is a chain of ~18
    // dependent commands whose submission costs more than their execution, and each command less counts.
    const bool directOut = n_frames <= kHostDirectOutFrames && !getenv("ORBX_HOST_NO_DIRECT_OUT");
    // upload, every kernel, the downloads and the error record, in stream order
    auto enqueue_all = [&]() -> orbx_status {
        ORBX_HIP_TRY(hipMemcpyAsync(ex->d_in, ex->h_in, B * frameBytes, hipMemcpyHostToDevice, ex->stream));
        orbx_status e = orbx_extract_batch_device(ex, ex->d_in, n_frames, width, height, width, (ptrdiff_t)frameBytes, lap0,
                                                  lap1, dyn, ndyn, directOut ? ex->h_outKps : ex->d_outKps,
                                                  directOut ? ex->h_outDesc : ex->d_outDesc, ex->outCap, ex->d_counts,
                                                  ex->stream);
        if (e != ORBX_OK) return e;
        ORBX_HIP_TRY(hipMemcpyAsync(ex->h_counts, ex->d_counts, B * 2 * sizeof(int), hipMemcpyDeviceToHost, ex->stream));
        if (!directOut) {
            ORBX_HIP_TRY(hipMemcpyAsync(ex->h_outKps, ex->d_outKps, B * ex->outCap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, ex->stream));
            ORBX_HIP_TRY(hipMemcpyAsync(ex->h_outDesc, ex->d_outDesc, B * ex->outCap * 32, hipMemcpyDeviceToHost, ex->stream));
        }
        ORBX_HIP_TRY(hipMemcpyAsync(ex->h_err, ex->d_err, sizeof(OrbxErr), hipMemcpyDeviceToHost, ex->stream));
        return ORBX_OK;
    };
    for (int attempt = 0; attempt < 8; ++attempt) {
        // the sequence is fixed by (frames, size, lapping area) and the handle's buffers: capture it once, replay it
        const bool graphable = ndyn == 0 && !ex->timing;
        if (graphable) {
            auto& k = ex->hostGraphKey;
            const bool hit = ex->hostGraph && k.nf == n_frames && k.w == width && k.h == height && k.lap0 == lap0 &&
                             k.lap1 == lap1 && k.epoch == ex->geoEpoch;
            if (!hit) {
                if (ex->hostGraph) { hipGraphExecDestroy(ex->hostGraph); ex->hostGraph = nullptr; }
                hipGraph_t graph = nullptr;
                ORBX_HIP_TRY(hipStreamBeginCapture(ex->stream, hipStreamCaptureModeThreadLocal));
                s = enqueue_all();
                const hipError_t ce = hipStreamEndCapture(ex->stream, &graph);
                if (s != ORBX_OK) { if (graph) hipGraphDestroy(graph); return s; }
                ORBX_HIP_TRY(ce);
                const hipError_t ie = hipGraphInstantiate(&ex->hostGraph, graph, nullptr, nullptr, 0);
                hipGraphDestroy(graph);
                ORBX_HIP_TRY(ie);
                k.nf = n_frames; k.w = width; k.h = height; k.lap0 = lap0; k.lap1 = lap1; k.epoch = ex->geoEpoch;
            }
            ORBX_HIP_TRY(hipGraphLaunch(ex->hostGraph, ex->stream));
        } else {
            s = enqueue_all();
            if (s != ORBX_OK) return s;
        }
        ORBX_HIP_TRY(hipStreamSynchronize(ex->stream));
        s = extract_status_from(ex, *ex->h_err);
        if (s == ORBX_ERR_CAPACITY && ex->h_err->cand_overflow) continue;   // regrown: run again
        break;
    }
    if (s != ORBX_OK) return s;
    for (int f = 0; f < n_frames; ++f) {
        const int n = ex->h_counts[2 * f];
        n_out[f] = n;
        mono_index_out[f] = ex->h_counts[2 * f + 1];
        if (n > capacity) { orbx_set_error("frame %d has %d keypoints, capacity is %d", f, n, capacity); return ORBX_ERR_CAPACITY; }
        if (n > 0) {
            ORBX_REQUIRE(keypoints && descriptors, "null output buffer");
            memcpy(keypoints + (size_t)f * capacity, ex->h_outKps + (size_t)f * ex->outCap, n * sizeof(orbx_keypoint));
            memcpy(descriptors + (size_t)f * capacity * 32, ex->h_outDesc + (size_t)f * ex->outCap * 32, (size_t)n * 32);
        }
    }
    return ORBX_OK;
}

// A large host batch as a pipeline of chunks.  Helper threads pack the images of whole chunks into the pinned
// buffer, running ahead of the calling thread, which uploads a packed chunk on `copyIn`, enqueues its kernels on
// `stream` into the chunk's own frame slots and its downloads on `copyOut`; as soon as a chunk's downloads are back,
// whichever thread is free hands its rows out to the caller's arrays.  Every chunk owns its range of every staging
// array (they are sized for max_batch frames), so nothing is reused inside a call, and a frame's kernels see only
// that frame's slot: the results are those of the one-image path.
static int host_chunk_frames(size_t frameBytes, int n_frames) {
    size_t want = (size_t)16 << 20;                            // bytes per upload
    if (const char* v = getenv("ORBX_HOST_CHUNK_MB")) want = (size_t)std::max(1, atoi(v)) << 20;
    const int byBytes = (int)std::min<size_t>(32, std::max<size_t>(4, want / std::max<size_t>(frameBytes, 1)));
    return std::max(4, std::min(byBytes, (n_frames + 3) / 4));   // at least four chunks when the batch allows it
}

static orbx_status extract_host_pipelined(orbx_extractor* ex, const uint8_t* const* images, int n_frames, int width,
                                          int height, ptrdiff_t stride, int lap0, int lap1, orbx_keypoint* keypoints,
                                          uint8_t* descriptors, int capacity, int* n_out, int* mono_index_out) {
    const size_t frameBytes = (size_t)width * height;
    const int C = host_chunk_frames(frameBytes, n_frames);
    const int nChunks = (n_frames + C - 1) / C;
    if (!ex->copyIn) ORBX_HIP_TRY(hipStreamCreateWithFlags(&ex->copyIn, hipStreamNonBlocking));
    if (!ex->copyOut) ORBX_HIP_TRY(hipStreamCreateWithFlags(&ex->copyOut, hipStreamNonBlocking));
    while (ex->pipeEvents.size() < (size_t)3 * nChunks) {
        hipEvent_t e;
        ORBX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ex->pipeEvents.push_back(e);
    }
    for (int f = 0; f < n_frames; ++f)
        if (!images[f]) { orbx_set_error("empty image"); return ORBX_ERR_EMPTY; }
    int helpers = 3;
    if (const char* v = getenv("ORBX_HOST_THREADS")) helpers = std::max(1, std::min(15, atoi(v) - 1));
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 1 && (unsigned)helpers > hw - 1) helpers = (int)hw - 1;
    helpers = std::min(helpers, nChunks);

    orbx_status s = ORBX_OK;
    for (int attempt = 0; attempt < 8; ++attempt) {
        // chunk states, written by one thread and polled by others
        std::vector<std::atomic<int>> staged(nChunks), enqueued(nChunks);
        for (int k = 0; k < nChunks; ++k) { staged[k].store(0); enqueued[k].store(0); }
        std::atomic<int> nextOut{0}, abortAll{0}, outError{ORBX_OK};
        std::atomic<int> badFrame{-1};
        auto stage_chunk = [&](int k) {
            const int f0 = k * C, c = std::min(C, n_frames - f0);
            for (int f = f0; f < f0 + c; ++f) {
                uint8_t* dst = ex->h_in + (size_t)f * frameBytes;
                if (stride == width) memcpy(dst, images[f], frameBytes);
                else for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * width, images[f] + (ptrdiff_t)y * stride, width);
            }
            staged[k].store(1, std::memory_order_release);
        };
        // Hands out the chunks whose downloads are complete, one after the other, until none is left (or, with
        // `onlyReady`, until the next one is not back yet).
        auto hand_out = [&](bool onlyReady) {
            for (;;) {
                const int k = nextOut.load();
                if (k >= nChunks || abortAll.load()) return;
                if (!enqueued[k].load(std::memory_order_acquire)) {
                    if (onlyReady) return;
                    std::this_thread::sleep_for(std::chrono::microseconds(20));     // no spinning beside the enqueuing thread
                    continue;
                }
                hipEvent_t evOut = ex->pipeEvents[3 * k + 2];
                if (onlyReady && hipEventQuery(evOut) != hipSuccess) return;
                int claim = k;
                if (!nextOut.compare_exchange_strong(claim, k + 1)) continue;     // another thread took it
                if (hipEventSynchronize(evOut) != hipSuccess) { outError.store(ORBX_ERR_HIP); return; }
                const int f0 = k * C, c = std::min(C, n_frames - f0);
                for (int f = f0; f < f0 + c; ++f) {
                    const int n = ex->h_counts[2 * f];
                    n_out[f] = n;
                    mono_index_out[f] = ex->h_counts[2 * f + 1];
                    if (n > capacity) { int none = -1; badFrame.compare_exchange_strong(none, f); outError.store(ORBX_ERR_CAPACITY); continue; }
                    if (n <= 0) continue;
                    if (!(keypoints && descriptors)) { outError.store(ORBX_ERR_INVALID); continue; }
                    memcpy(keypoints + (size_t)f * capacity, ex->h_outKps + (size_t)f * ex->outCap, n * sizeof(orbx_keypoint));
                    memcpy(descriptors + (size_t)f * capacity * 32, ex->h_outDesc + (size_t)f * ex->outCap * 32, (size_t)n * 32);
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < helpers; ++t)
            pool.emplace_back([&, t] {
                hipSetDevice(ex->device);
                for (int k = t; k < nChunks && !abortAll.load(); k += helpers) stage_chunk(k);
                hand_out(false);
            });
        hipError_t he = hipSuccess;
        auto hip_ok = [&](hipError_t e) { if (e != hipSuccess && he == hipSuccess) he = e; return e == hipSuccess; };
        const bool trace = getenv("ORBX_HOST_TRACE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tStart = now();
        double tWait = 0, tEnq = 0;
        for (int k = 0; k < nChunks && s == ORBX_OK && he == hipSuccess; ++k) {
            const int f0 = k * C, c = std::min(C, n_frames - f0);
            const double w0 = now();
            while (!staged[k].load(std::memory_order_acquire)) std::this_thread::yield();
            const double w1 = now();
            tWait += w1 - w0;
            struct EnqTimer { double& acc; double t0; decltype(now)& clk; ~EnqTimer() { acc += clk() - t0; } } enqTimer{tEnq, w1, now};
            hipEvent_t evIn = ex->pipeEvents[3 * k], evK = ex->pipeEvents[3 * k + 1], evOut = ex->pipeEvents[3 * k + 2];
            if (!hip_ok(hipMemcpyAsync(ex->d_in + (size_t)f0 * frameBytes, ex->h_in + (size_t)f0 * frameBytes, (size_t)c * frameBytes,
                                       hipMemcpyHostToDevice, ex->copyIn)) ||
                !hip_ok(hipEventRecord(evIn, ex->copyIn)) || !hip_ok(hipStreamWaitEvent(ex->stream, evIn, 0)))
                break;
            s = enqueue_images(ex, ex->d_in + (size_t)f0 * frameBytes, c, width, height, width, (ptrdiff_t)frameBytes, lap0, lap1,
                               nullptr, 0, ex->d_outKps + (size_t)f0 * ex->outCap, ex->d_outDesc + (size_t)f0 * ex->outCap * 32,
                               ex->outCap, ex->d_counts + 2 * f0, ex->stream, f0);
            if (s != ORBX_OK) break;
            if (!hip_ok(hipEventRecord(evK, ex->stream)) || !hip_ok(hipStreamWaitEvent(ex->copyOut, evK, 0)) ||
                !hip_ok(hipMemcpyAsync(ex->h_counts + 2 * f0, ex->d_counts + 2 * f0, (size_t)c * 2 * sizeof(int), hipMemcpyDeviceToHost, ex->copyOut)) ||
                !hip_ok(hipMemcpyAsync(ex->h_outKps + (size_t)f0 * ex->outCap, ex->d_outKps + (size_t)f0 * ex->outCap,
                                       (size_t)c * ex->outCap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, ex->copyOut)) ||
                !hip_ok(hipMemcpyAsync(ex->h_outDesc + (size_t)f0 * ex->outCap * 32, ex->d_outDesc + (size_t)f0 * ex->outCap * 32,
                                       (size_t)c * ex->outCap * 32, hipMemcpyDeviceToHost, ex->copyOut)) ||
                (k == nChunks - 1 && !hip_ok(hipMemcpyAsync(ex->h_err, ex->d_err, sizeof(OrbxErr), hipMemcpyDeviceToHost, ex->copyOut))) ||
                !hip_ok(hipEventRecord(evOut, ex->copyOut)))
                break;
            enqueued[k].store(1, std::memory_order_release);
        }
        const double tLoop = now();
        if (s != ORBX_OK || he != hipSuccess) abortAll.store(1);
        else hand_out(false);
        const double tOut = now();
        for (std::thread& t : pool) t.join();
        if (trace)
            fprintf(stderr, "orbx host pipeline: %d frames, %d chunks of %d, %d helpers: waited for staging %.0f us, enqueued %.0f us, "
                            "hand-out tail %.0f us, join %.0f us, total %.0f us\n", n_frames, nChunks, C, helpers, tWait, tEnq,
                    tOut - tLoop, now() - tOut, now() - tStart);
        // nothing may be left in flight on the handle's streams, whatever happened
        const hipError_t e1 = hipStreamSynchronize(ex->copyIn), e2 = hipStreamSynchronize(ex->stream), e3 = hipStreamSynchronize(ex->copyOut);
        if (s != ORBX_OK) return s;
        ORBX_HIP_TRY(he); ORBX_HIP_TRY(e1); ORBX_HIP_TRY(e2); ORBX_HIP_TRY(e3);
        // a candidate-buffer overflow shows only now: the buffers have been regrown, the whole batch runs again and the
        // rows handed out so far are overwritten
        s = extract_status_from(ex, *ex->h_err);
        if (s == ORBX_ERR_CAPACITY && ex->h_err->cand_overflow) { s = ensure_host_staging(ex); if (s != ORBX_OK) return s; continue; }
        if (s != ORBX_OK) return s;
        if (outError.load() == ORBX_ERR_CAPACITY) orbx_set_error("frame %d has %d keypoints, capacity is %d", badFrame.load(), n_out[badFrame.load()], capacity);
        else if (outError.load() == ORBX_ERR_INVALID) orbx_set_error("null output buffer");
        else if (outError.load() == ORBX_ERR_HIP) orbx_set_error("hipEventSynchronize failed");
        return (orbx_status)outError.load();
    }
    return s;
}

extern "C" orbx_status orbx_extract_batch(orbx_extractor* ex, const uint8_t* const* images, int n_frames, int width,
                                          int height, ptrdiff_t stride, int lap0, int lap1, orbx_keypoint* keypoints,
                                          uint8_t* descriptors, int capacity, int* n_out, int* mono_index_out) {
    ORBX_REQUIRE(ex && images && n_out && mono_index_out, "null argument");
    ORBX_REQUIRE(n_frames >= 1 && n_frames <= ex->prm.max_batch, "n_frames %d outside [1, %d]", n_frames, ex->prm.max_batch);
    if (width <= 0 || height <= 0) { orbx_set_error("empty image"); return ORBX_ERR_EMPTY; }
    ORBX_REQUIRE(stride >= width, "stride smaller than width");
    return extract_host(ex, images, n_frames, width, height, stride, lap0, lap1, nullptr, 0, keypoints, descriptors, capacity,
                        n_out, mono_index_out);
}

extern "C" orbx_status orbx_extract(orbx_extractor* ex, const uint8_t* image, int width, int height, ptrdiff_t stride,
                                    int lap0, int lap1, const orbx_rect* dyn, int ndyn, orbx_keypoint* keypoints,
                                    uint8_t* descriptors, int capacity, int* n_out, int* mono_index_out) {
    ORBX_REQUIRE(ex && n_out && mono_index_out, "null argument");
    if (!image || width <= 0 || height <= 0) {
        orbx_set_error("empty image");
        *n_out = 0; *mono_index_out = -1;
        return ORBX_ERR_EMPTY;
    }
    ORBX_REQUIRE(stride >= width, "stride smaller than width");
    ORBX_REQUIRE(ndyn == 0 || (dyn && ndyn > 0 && ndyn <= kOrbxMaxRects), "bad rectangle list");
    return extract_host(ex, &image, 1, width, height, stride, lap0, lap1, dyn, ndyn, keypoints, descriptors, capacity, n_out,
                        mono_index_out);
}

extern "C" orbx_status orbx_extractor_set_timing(orbx_extractor* ex, int enable) {
    ORBX_REQUIRE(ex, "null handle");
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    if (ex->timing) resolve_timing(ex);
    ex->timing = enable != 0;
    for (double& v : ex->stageMs) v = 0;
    ex->stageCalls = 0;
    return ORBX_OK;
}

extern "C" orbx_status orbx_extractor_get_timing(orbx_extractor* ex, double* stage_ms, int n_stages, long* n_calls) {
    ORBX_REQUIRE(ex && stage_ms && n_calls, "null argument");
    ORBX_REQUIRE(n_stages >= kTimedStages, "need room for %d stages", kTimedStages);
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    resolve_timing(ex);
    for (int s = 0; s < kTimedStages; ++s) stage_ms[s] = ex->stageMs[s];
    *n_calls = ex->stageCalls;
    return ORBX_OK;
}

// accessors for the matching translation unit
const OrbxFrameGeo* orbx_internal_geo_device(orbx_extractor* ex) { return ex->d_geo; }
const OrbxFrameGeo* orbx_internal_geo_host(orbx_extractor* ex) { return ex->geoValid ? &ex->geo : nullptr; }
const uint8_t* orbx_internal_pyramid(orbx_extractor* ex) { return ex->d_pyr; }
hipStream_t orbx_internal_stream(orbx_extractor* ex) { return ex->stream; }
int orbx_internal_device(orbx_extractor* ex) { return ex->device; }
int orbx_internal_max_batch(orbx_extractor* ex) { return ex->prm.max_batch; }
void* orbx_internal_scratch(orbx_extractor* ex, size_t bytes) {
    if (bytes > ex->scratchAlloc) {
        hipStreamSynchronize(ex->stream);
        hipDeviceSynchronize();
        if (ex->d_scratch) hipFree(ex->d_scratch);
        ex->d_scratch = nullptr; ex->scratchAlloc = 0;
        if (hipMalloc(&ex->d_scratch, bytes) != hipSuccess) return nullptr;
        ex->scratchAlloc = bytes;
    }
    return ex->d_scratch;
}

extern "C" orbx_status orbx_pyramid_level_device(orbx_extractor* ex, int frame, int level, const uint8_t** d_ptr,
                                                 int* width, int* height, ptrdiff_t* pitch) {
    ORBX_REQUIRE(ex && ex->geoValid, "no geometry yet");
    ORBX_REQUIRE(frame >= 0 && frame < ex->prm.max_batch && level >= 0 && level < ex->geo.nlevels, "bad frame / level");
    const OrbxLevelGeo& L = ex->geo.lv[level];
    if (d_ptr) *d_ptr = ex->d_pyr + (size_t)frame * ex->geo.pyrBytes + L.off;
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (pitch) *pitch = L.pitch;
    return ORBX_OK;
}

extern "C" orbx_status orbx_pyramid_level(orbx_extractor* ex, int frame, int level, uint8_t* out, size_t capacity,
                                          int* width, int* height) {
    const uint8_t* d = nullptr;
    int w = 0, h = 0;
    ptrdiff_t pitch = 0;
    orbx_status s = orbx_pyramid_level_device(ex, frame, level, &d, &w, &h, &pitch);
    if (s != ORBX_OK) return s;
    if (width) *width = w;
    if (height) *height = h;
    if (!out) return ORBX_OK;
    if (capacity < (size_t)w * h) { orbx_set_error("buffer too small for a %dx%d level", w, h); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    ORBX_HIP_TRY(hipStreamSynchronize(ex->stream));
    ORBX_HIP_TRY(hipMemcpy2D(out, w, d, pitch, w, h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// Diagnostics: the blurred level `level` of frame slot `frame` as the last call left it (handles that blur whole levels
// only: ORBX_ERR_INVALID for one that blurs per keypoint patch).  out: width * height bytes, rows packed.
extern "C" orbx_status orbx_debug_blurred_level(orbx_extractor* ex, int frame, int level, uint8_t* out, size_t capacity,
                                                int* width, int* height) {
    ORBX_REQUIRE(ex && ex->geoValid, "no geometry yet");
    ORBX_REQUIRE(ex->levelBlur && ex->d_blur, "this handle blurs per keypoint patch: there is no blurred level");
    ORBX_REQUIRE(frame >= 0 && frame < ex->prm.max_batch && level >= 0 && level < ex->geo.nlevels, "bad frame / level");
    const OrbxLevelGeo& L = ex->geo.lv[level];
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (!out) return ORBX_OK;
    if (capacity < (size_t)L.w * L.h) { orbx_set_error("buffer too small for a %dx%d level", L.w, L.h); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    ORBX_HIP_TRY(hipStreamSynchronize(ex->stream));
    ORBX_HIP_TRY(hipMemcpy2D(out, L.w, ex->d_blur + (size_t)frame * ex->geo.pyrBytes + L.off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" orbx_status orbx_debug_level_candidates(orbx_extractor* ex, int frame, int level, float* xyr,
                                                   uint32_t* order_key, int capacity, int* n_out) {
    ORBX_REQUIRE(ex && ex->geoValid && n_out, "bad argument");
    ORBX_REQUIRE(frame >= 0 && frame < ex->prm.max_batch && level >= 0 && level < ex->geo.nlevels, "bad frame / level");
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    ORBX_HIP_TRY(hipDeviceSynchronize());
    int n = 0;
    ORBX_HIP_TRY(hipMemcpy(&n, ex->d_candCount + ((size_t)frame * ex->geo.nlevels + level) * kCandCounterStride, sizeof(int), hipMemcpyDeviceToHost));
    const OrbxLevelGeo& L = ex->geo.lv[level];
    n = std::min(n, L.candCap);
    *n_out = n;
    if (n > capacity) { orbx_set_error("capacity %d < %d candidates", capacity, n); return ORBX_ERR_CAPACITY; }
    std::vector<OrbxCand> c(n);
    if (n) ORBX_HIP_TRY(hipMemcpy(c.data(), ex->d_cands + (size_t)frame * ex->geo.candTotal + L.candOff, n * sizeof(OrbxCand), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        xyr[3 * i] = c[i].x; xyr[3 * i + 1] = c[i].y; xyr[3 * i + 2] = c[i].response;
        order_key[i] = c[i].order;
    }
    return ORBX_OK;
}

extern "C" orbx_status orbx_debug_level_keypoints(orbx_extractor* ex, int frame, int level, float* xyr, int capacity,
                                                  int* n_out) {
    ORBX_REQUIRE(ex && ex->geoValid && n_out, "bad argument");
    ORBX_REQUIRE(frame >= 0 && frame < ex->prm.max_batch && level >= 0 && level < ex->geo.nlevels, "bad frame / level");
    ORBX_HIP_TRY(hipSetDevice(ex->device));
    ORBX_HIP_TRY(hipDeviceSynchronize());
    int n = 0;
    ORBX_HIP_TRY(hipMemcpy(&n, ex->d_kpCount + frame * ex->geo.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > capacity) { orbx_set_error("capacity %d < %d keypoints", capacity, n); return ORBX_ERR_CAPACITY; }
    const OrbxLevelGeo& L = ex->geo.lv[level];
    std::vector<OrbxLevelKp> k(n);
    if (n) ORBX_HIP_TRY(hipMemcpy(k.data(), ex->d_levelKps + (size_t)frame * ex->geo.kpTotal + L.kpOff, n * sizeof(OrbxLevelKp), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { xyr[3 * i] = k[i].x; xyr[3 * i + 1] = k[i].y; xyr[3 * i + 2] = k[i].response; }
    return ORBX_OK;
}

// ---- stage tap: the workgroup sort on its own -------------------------------------------
constexpr int kSortTapMax = 2048;

__global__ __launch_bounds__(kQtThreadsMax) void k_sort_tap(unsigned long long* __restrict__ data, const int* __restrict__ offsets) {
    extern __shared__ __align__(16) unsigned char st_smem[];
    const int b = blockIdx.x, n = offsets[b + 1] - offsets[b];
    unsigned long long* v = reinterpret_cast<unsigned long long*>(st_smem);
    unsigned long long* out = v + kSortTapMax;
    int* posL = reinterpret_cast<int*>(out + kSortTapMax);
    int* posR = posL + kSortTapMax;
    int* segs = posR + kSortTapMax;     // 6 * (kSortTapMax / 17 + 1) ints
    unsigned long long* g = data + offsets[b];
    for (int i = threadIdx.x; i < n; i += kQtThreadsMax) v[i] = g[i];
    __syncthreads();
    orbx_sort::block_sort(v, n, segs, posL, posR, out);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kQtThreadsMax) g[i] = v[i];
}

extern "C" orbx_status orbx_debug_wave_sort(const uint64_t* elems, const int* offsets, int n_arrays,
                                            uint64_t* device_sorted, uint64_t* host_sorted) {
    ORBX_REQUIRE(elems && offsets && device_sorted && host_sorted && n_arrays > 0, "bad argument");
    ORBX_REQUIRE(offsets[0] == 0, "offsets must start at 0");
    for (int a = 0; a < n_arrays; ++a)
        ORBX_REQUIRE(offsets[a + 1] >= offsets[a] && offsets[a + 1] - offsets[a] <= kSortTapMax,
                     "array %d has %d elements (0..%d allowed)", a, offsets[a + 1] - offsets[a], kSortTapMax);
    const size_t total = (size_t)offsets[n_arrays];
    int stack[3 * orbx_sort::kStackDepth];
    for (size_t i = 0; i < total; ++i) host_sorted[i] = elems[i];
    for (int a = 0; a < n_arrays; ++a)
        orbx_sort::sort(reinterpret_cast<orbx_sort::elem_t*>(host_sorted) + offsets[a], offsets[a + 1] - offsets[a], stack);
    unsigned long long* d_data = nullptr;
    int* d_off = nullptr;
    ORBX_HIP_TRY(hipMalloc(&d_data, std::max<size_t>(total, 1) * 8));
    ORBX_HIP_TRY(hipMalloc(&d_off, (n_arrays + 1) * sizeof(int)));
    hipError_t e = hipMemcpy(d_data, elems, total * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, offsets, (n_arrays + 1) * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sort_tap, dim3(n_arrays), dim3(kQtThreadsMax), kSortTapMax * 24 + 6 * (kSortTapMax / 17 + 1) * sizeof(int), 0, d_data, d_off);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(device_sorted, d_data, total * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_data);
    (void)hipFree(d_off);
    if (e != hipSuccess) { orbx_set_error("orbx_debug_wave_sort: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    return ORBX_OK;
}
