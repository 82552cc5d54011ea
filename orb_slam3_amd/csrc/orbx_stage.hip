// Input staging in front of the extractor on gfx950: what the reference does to a camera image
// between the driver and ORBextractor::operator(), as one kernel that writes the grey image the
// pyramid starts from.  Restates
//   cv::remap(im, out, M1, M2, cv::INTER_LINEAR)      src/System.cc:361-362 (stereo), :437-440 (mono)
//   cv::resize(im, out, settings_->newImSize())        src/System.cc:367-370
//   cvtColor(.., COLOR_{RGB,BGR,RGBA,BGRA}2GRAY)       src/Tracking.cc:1595-1618, :1680-1700, :1760-1780
// The geometric step runs on the image as delivered (1, 3 or 4 interleaved channels, every channel
// rounded to 8 bits as cv::remap / cv::resize do), the grey conversion on its result; nothing but
// the grey image is written.  Integer arithmetic throughout: byte gathers + a few VALU per pixel,
// bound by the map read (6 bytes per output pixel) and the gathers.

#include <algorithm>
#include <cmath>
#include <vector>

#include "orbx_common.h"

namespace {

// cv::cvtColor 8U grey: 15-bit coefficients (R 9798, G 19235, B 3735), rounded.
__device__ __forceinline__ int grey15(int r, int g, int b) { return (r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15; }

template <int CN>
__device__ __forceinline__ int to_grey(const int (&v)[CN], int rgb) {
    if constexpr (CN == 1) return v[0];
    else return rgb ? grey15(v[0], v[1], v[2]) : grey15(v[2], v[1], v[0]);
}

struct StageGeo {
    int sw, sh, dw, dh;
    int rgb;
    int area2;                 // resize: exact 2 x 2 decimation (cv::resize routes it to the INTER_AREA fast path)
    const int* xy;             // remap: (short X) | (short Y) << 16 per output pixel
    const unsigned short* ab;  // remap: fractional index (fy & 31) * 32 + (fx & 31)
    const int* xofs;           // resize: left source column per output column
    const short2* xa;          // resize: 11-bit weights of the two columns
    const int* yofs;
    const short2* ya;
};

// One output pixel.  Offsets inside a frame are 32-bit (a frame is far below 2 GiB), so every gather is a
// uniform base + a lane offset.
template <int CN, int MODE>
__device__ __forceinline__ int stage_pixel(const StageGeo& g, const uint8_t* __restrict__ S, int srcRow, int x, int y,
                                           int xy, int ab) {
    int v[CN];
    if constexpr (MODE == 0) {
        const uint8_t* p = S + (y * srcRow + x * CN);
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = p[c];
    } else if constexpr (MODE == 1) {
        const int sx = (short)(xy & 0xFFFF), sy = xy >> 16;
        const int a = ab & 31, b = ab >> 5;
        // the 15-bit table weights are 32 x these products; the sum below is the table sum / 32
        const int w00 = (32 - a) * (32 - b), w01 = a * (32 - b), w10 = (32 - a) * b, w11 = a * b;
        const uint8_t* p = S + (sy * srcRow + sx * CN);
        if ((unsigned)sx < (unsigned)(g.sw - 1) && (unsigned)sy < (unsigned)(g.sh - 1)) {
#pragma unroll
            for (int c = 0; c < CN; ++c)
                v[c] = (p[c] * w00 + p[CN + c] * w01 + p[srcRow + c] * w10 + p[srcRow + CN + c] * w11 + 512) >> 10;
        } else {
            // constant border 0: taps outside the image contribute nothing
            const bool x0 = (unsigned)sx < (unsigned)g.sw, x1 = (unsigned)(sx + 1) < (unsigned)g.sw;
            const bool y0 = (unsigned)sy < (unsigned)g.sh, y1 = (unsigned)(sy + 1) < (unsigned)g.sh;
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const int v00 = (x0 && y0) ? p[c] : 0, v01 = (x1 && y0) ? p[CN + c] : 0;
                const int v10 = (x0 && y1) ? p[srcRow + c] : 0, v11 = (x1 && y1) ? p[srcRow + CN + c] : 0;
                v[c] = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + 512) >> 10;
            }
        }
    } else {
        if (g.area2) {
            const uint8_t* p = S + (2 * y * srcRow + 2 * x * CN);
#pragma unroll
            for (int c = 0; c < CN; ++c) v[c] = (p[c] + p[CN + c] + p[srcRow + c] + p[srcRow + CN + c] + 2) >> 2;
        } else {
            const int sx0 = g.xofs[x], sx1 = min(sx0 + 1, g.sw - 1);
            const int sy0 = g.yofs[y], sy1 = min(sy0 + 1, g.sh - 1);
            const short2 xa = g.xa[x], ya = g.ya[y];
            const uint8_t* r0 = S + sy0 * srcRow;
            const uint8_t* r1 = S + sy1 * srcRow;
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const int t0 = r0[sx0 * CN + c] * xa.x + r0[sx1 * CN + c] * xa.y;
                const int t1 = r1[sx0 * CN + c] * xa.x + r1[sx1 * CN + c] * xa.y;
                v[c] = (((ya.x * (t0 >> 4)) >> 16) + ((ya.y * (t1 >> 4)) >> 16) + 2) >> 2;
            }
        }
    }
    return to_grey<CN>(v, g.rgb) & 255;
}

// mode 0: grey conversion only; 1: remap; 2: resize.  One lane = 4 consecutive output pixels of a row.
template <int CN, int MODE>
__global__ __launch_bounds__(256) void k_stage(StageGeo g, const uint8_t* __restrict__ src, int srcRow,
                                               long long srcFrame, uint8_t* __restrict__ dst, long long dstRow,
                                               long long dstFrame, int dstAligned, int lanesPerRow,
                                               uint32_t lanesPerRowRcp) {
    // (row, 4-pixel column) as one sequence: full workgroups whatever the width; lanesPerRowRcp =
    // floor(2^32 / lanesPerRow), one correction step makes the quotient exact
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = __umulhi(i, lanesPerRowRcp);
    uint32_t col = i - row * (uint32_t)lanesPerRow;
    if (col >= (uint32_t)lanesPerRow) { ++row; col -= (uint32_t)lanesPerRow; }
    const int y = (int)row, x4 = (int)col * 4;
    if (y >= g.dh) return;
    const uint8_t* S = src + (long long)blockIdx.y * srcFrame;
    uint8_t* D = dst + (long long)blockIdx.y * dstFrame + (long long)y * dstRow + x4;
    const int n = min(4, g.dw - x4);
    int xy[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
    if constexpr (MODE == 1) {
        const int m = y * g.dw + x4;
        if (n == 4 && (m & 3) == 0) {
            const int4 q = *reinterpret_cast<const int4*>(g.xy + m);
            const uint2 r = *reinterpret_cast<const uint2*>(g.ab + m);
            xy[0] = q.x; xy[1] = q.y; xy[2] = q.z; xy[3] = q.w;
            ab[0] = r.x & 0xFFFF; ab[1] = r.x >> 16; ab[2] = r.y & 0xFFFF; ab[3] = r.y >> 16;
        } else {
            for (int i = 0; i < n; ++i) { xy[i] = g.xy[m + i]; ab[i] = g.ab[m + i]; }
        }
    }
    if (n == 4) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) packed |= (uint32_t)stage_pixel<CN, MODE>(g, S, srcRow, x4 + i, y, xy[i], ab[i]) << (8 * i);
        if (dstAligned) {
            *reinterpret_cast<uint32_t*>(D) = packed;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) D[i] = (uint8_t)(packed >> (8 * i));
        }
    } else {
        for (int i = 0; i < n; ++i) D[i] = (uint8_t)stage_pixel<CN, MODE>(g, S, srcRow, x4 + i, y, xy[i], ab[i]);
    }
}

// cvRound
int round_half_even(float v) { return (int)std::nearbyintf(v); }

// cv::resize(INTER_LINEAR) taps for one axis: source offset and the two 11-bit weights per output element.
void linear_taps(int ssize, int dsize, std::vector<int>& ofs, std::vector<short2>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize(dsize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        coef[d].x = (short)round_half_even((1.f - f) * 2048.f);
        coef[d].y = (short)round_half_even(f * 2048.f);
    }
}

template <class T>
hipError_t upload(const std::vector<T>& h, void** d) {
    hipError_t e = hipMalloc(d, std::max<size_t>(h.size(), 1) * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

struct orbx_stager {
    orbx_stage_params prm;
    void *d_xy = nullptr, *d_ab = nullptr, *d_xofs = nullptr, *d_xa = nullptr, *d_yofs = nullptr, *d_ya = nullptr;
    int area2 = 0;
    // staging of the synchronous host entry point
    void *d_src = nullptr, *d_dst = nullptr;
    size_t srcCap = 0, dstCap = 0;
    hipStream_t stream = nullptr;
};

extern "C" orbx_status orbx_stager_create(const orbx_stage_params* p, orbx_stager** out) {
    ORBX_REQUIRE(p && out, "null argument");
    ORBX_REQUIRE(p->src_width > 0 && p->src_height > 0, "bad source size %d x %d", p->src_width, p->src_height);
    ORBX_REQUIRE(p->channels == 1 || p->channels == 3 || p->channels == 4, "channels must be 1, 3 or 4 (got %d)", p->channels);
    ORBX_REQUIRE(p->mode >= ORBX_STAGE_NONE && p->mode <= ORBX_STAGE_RESIZE, "unknown staging mode %d", p->mode);
    ORBX_REQUIRE(p->src_width < 32768 && p->src_height < 32768, "source larger than 32767 pixels");
    orbx_stage_params prm = *p;
    if (prm.mode == ORBX_STAGE_NONE) { prm.dst_width = prm.src_width; prm.dst_height = prm.src_height; }
    ORBX_REQUIRE(prm.dst_width > 0 && prm.dst_height > 0, "bad output size %d x %d", prm.dst_width, prm.dst_height);
    ORBX_REQUIRE(prm.mode != ORBX_STAGE_REMAP || (prm.map_x && prm.map_y), "remap needs both maps");
    ORBX_HIP_TRY(hipSetDevice(prm.device));
    orbx_stager* s = new orbx_stager;
    s->prm = prm;
    s->prm.map_x = s->prm.map_y = nullptr;   // the maps are consumed here
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess && prm.mode == ORBX_STAGE_REMAP) {
        // cv::remap's conversion of floating-point maps: 5 fractional bits, cvRound, integer part saturated to short
        const size_t n = (size_t)prm.dst_width * prm.dst_height;
        std::vector<int> xy(n);
        std::vector<unsigned short> ab(n);
        for (size_t i = 0; i < n; ++i) {
            const int fx = round_half_even(prm.map_x[i] * 32.f), fy = round_half_even(prm.map_y[i] * 32.f);
            const int sx = std::max(-32768, std::min(32767, fx >> 5)), sy = std::max(-32768, std::min(32767, fy >> 5));
            xy[i] = (sx & 0xFFFF) | (int)((unsigned)sy << 16);
            ab[i] = (unsigned short)((fy & 31) * 32 + (fx & 31));
        }
        e = upload(xy, &s->d_xy);
        if (e == hipSuccess) e = upload(ab, &s->d_ab);
    } else if (e == hipSuccess && prm.mode == ORBX_STAGE_RESIZE) {
        s->area2 = prm.src_width == 2 * prm.dst_width && prm.src_height == 2 * prm.dst_height;
        std::vector<int> xo, yo;
        std::vector<short2> xa, ya;
        linear_taps(prm.src_width, prm.dst_width, xo, xa);
        linear_taps(prm.src_height, prm.dst_height, yo, ya);
        e = upload(xo, &s->d_xofs);
        if (e == hipSuccess) e = upload(xa, &s->d_xa);
        if (e == hipSuccess) e = upload(yo, &s->d_yofs);
        if (e == hipSuccess) e = upload(ya, &s->d_ya);
    }
    if (e != hipSuccess) {
        orbx_set_error("stager set-up failed: %s", hipGetErrorString(e));
        orbx_stager_destroy(s);
        return ORBX_ERR_HIP;
    }
    *out = s;
    return ORBX_OK;
}

extern "C" void orbx_stager_destroy(orbx_stager* s) {
    if (!s) return;
    for (void* p : {s->d_xy, s->d_ab, s->d_xofs, s->d_xa, s->d_yofs, s->d_ya, s->d_src, s->d_dst})
        if (p) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" orbx_status orbx_stager_output_size(const orbx_stager* s, int* width, int* height) {
    ORBX_REQUIRE(s && width && height, "null argument");
    *width = s->prm.dst_width; *height = s->prm.dst_height;
    return ORBX_OK;
}

// Shared with the extractor (which stages straight into level 0 of its pyramid).
orbx_status orbx_stage_enqueue(orbx_stager* s, const uint8_t* d_src, int n_frames, long long srcRow, long long srcFrame,
                               uint8_t* d_dst, long long dstRow, long long dstFrame, hipStream_t st) {
    ORBX_REQUIRE(s && d_src && d_dst, "null argument");
    ORBX_REQUIRE(n_frames >= 1, "n_frames %d", n_frames);
    const orbx_stage_params& p = s->prm;
    ORBX_REQUIRE(srcRow >= (long long)p.src_width * p.channels, "source row stride %lld shorter than a row", srcRow);
    ORBX_REQUIRE(dstRow >= p.dst_width, "output row stride %lld shorter than a row", dstRow);
    ORBX_REQUIRE(srcRow * p.src_height < ((long long)1 << 31), "a source frame must be smaller than 2 GiB");
    StageGeo g;
    g.sw = p.src_width; g.sh = p.src_height; g.dw = p.dst_width; g.dh = p.dst_height;
    g.rgb = p.rgb_order; g.area2 = s->area2;
    g.xy = static_cast<const int*>(s->d_xy); g.ab = static_cast<const unsigned short*>(s->d_ab);
    g.xofs = static_cast<const int*>(s->d_xofs); g.xa = static_cast<const short2*>(s->d_xa);
    g.yofs = static_cast<const int*>(s->d_yofs); g.ya = static_cast<const short2*>(s->d_ya);
    const int aligned = ((((uintptr_t)d_dst) | (uintptr_t)dstRow | (uintptr_t)dstFrame) & 3) == 0;
    const int lanes = (p.dst_width + 3) / 4;
    const uint32_t lanesRcp = lanes == 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / (unsigned)lanes);
    const dim3 block(256), grid((unsigned)(((long long)lanes * p.dst_height + 255) / 256), n_frames);
#define ORBX_STAGE_LAUNCH(CN, MODE) \
    hipLaunchKernelGGL((k_stage<CN, MODE>), grid, block, 0, st, g, d_src, (int)srcRow, srcFrame, d_dst, dstRow, dstFrame, aligned, lanes, lanesRcp)
#define ORBX_STAGE_MODE(CN)                                   \
    do {                                                      \
        if (p.mode == ORBX_STAGE_NONE) ORBX_STAGE_LAUNCH(CN, 0);       \
        else if (p.mode == ORBX_STAGE_REMAP) ORBX_STAGE_LAUNCH(CN, 1); \
        else ORBX_STAGE_LAUNCH(CN, 2);                        \
    } while (0)
    if (p.channels == 1) ORBX_STAGE_MODE(1);
    else if (p.channels == 3) ORBX_STAGE_MODE(3);
    else ORBX_STAGE_MODE(4);
#undef ORBX_STAGE_MODE
#undef ORBX_STAGE_LAUNCH
    ORBX_HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" orbx_status orbx_stage_batch_device(orbx_stager* s, const uint8_t* d_src, int n_frames,
                                               ptrdiff_t src_row_stride, ptrdiff_t src_frame_stride, uint8_t* d_dst,
                                               ptrdiff_t dst_row_stride, ptrdiff_t dst_frame_stride, void* stream) {
    ORBX_REQUIRE(s, "null handle");
    ORBX_HIP_TRY(hipSetDevice(s->prm.device));
    return orbx_stage_enqueue(s, d_src, n_frames, (long long)src_row_stride, (long long)src_frame_stride, d_dst,
                              (long long)dst_row_stride, (long long)dst_frame_stride, (hipStream_t)stream);
}

extern "C" orbx_status orbx_stage(orbx_stager* s, const uint8_t* src, ptrdiff_t src_stride, uint8_t* dst,
                                  ptrdiff_t dst_stride) {
    ORBX_REQUIRE(s && src && dst, "null argument");
    const orbx_stage_params& p = s->prm;
    ORBX_REQUIRE(src_stride >= (ptrdiff_t)p.src_width * p.channels && dst_stride >= p.dst_width, "row stride shorter than a row");
    ORBX_HIP_TRY(hipSetDevice(p.device));
    const size_t srcRowBytes = (size_t)p.src_width * p.channels;
    const size_t srcBytes = srcRowBytes * p.src_height, dstBytes = (size_t)p.dst_width * p.dst_height;
    if (s->srcCap < srcBytes) {
        if (s->d_src) (void)hipFree(s->d_src);
        s->d_src = nullptr; s->srcCap = 0;
        ORBX_HIP_TRY(hipMalloc(&s->d_src, srcBytes));
        s->srcCap = srcBytes;
    }
    if (s->dstCap < dstBytes) {
        if (s->d_dst) (void)hipFree(s->d_dst);
        s->d_dst = nullptr; s->dstCap = 0;
        ORBX_HIP_TRY(hipMalloc(&s->d_dst, dstBytes));
        s->dstCap = dstBytes;
    }
    ORBX_HIP_TRY(hipMemcpy2DAsync(s->d_src, srcRowBytes, src, (size_t)src_stride, srcRowBytes, p.src_height,
                                  hipMemcpyHostToDevice, s->stream));
    orbx_status st = orbx_stage_enqueue(s, static_cast<const uint8_t*>(s->d_src), 1, (long long)srcRowBytes,
                                        (long long)srcBytes, static_cast<uint8_t*>(s->d_dst), p.dst_width,
                                        (long long)dstBytes, s->stream);
    if (st != ORBX_OK) return st;
    ORBX_HIP_TRY(hipMemcpy2DAsync(dst, (size_t)dst_stride, s->d_dst, (size_t)p.dst_width, (size_t)p.dst_width,
                                  p.dst_height, hipMemcpyDeviceToHost, s->stream));
    ORBX_HIP_TRY(hipStreamSynchronize(s->stream));
    return ORBX_OK;
}
