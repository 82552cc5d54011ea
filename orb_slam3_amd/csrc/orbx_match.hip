// Hamming descriptor matching on gfx950.  Restates
//   ORBmatcher::DescriptorDistance                      src/ORBmatcher.cc:2141-2157
//   cv::BFMatcher(NORM_HAMMING).knnMatch(.., 2)         src/Frame.cc:46, :1309
//   Frame::ComputeStereoMatches                         src/Frame.cc:962-1133
//   Frame::GetFeaturesInArea + the candidate loop of
//   ORBmatcher::SearchByProjection                      src/Frame.cc:792-858, src/ORBmatcher.cc:84-141
//   distance core of ORBmatcher::SearchForTriangulation src/ORBmatcher.cc:1051-1167
//
// Everything here is 256-bit XOR + popcount and small integer reductions: VALU work
// (v_xor_b32 + v_bcnt_u32_b32), no matrix cores.

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "orbx_common.h"
#include "orbx_camera_gates.h"

// --------------------------------------------------------------------------------------
// small RAII helper for the synchronous host entry points
// --------------------------------------------------------------------------------------
namespace {
// Device blocks of the synchronous entry points are recycled per host thread: the matchers are called
// many times per frame and a hipMalloc / hipFree pair per temporary costs more than the kernels.
// Blocks are rounded up to powers of two and kept until orbx_release_cached_memory() (or process end).
struct DevBlockCache {
    struct Block { void* p; size_t cap; int device; };
    std::vector<Block> free;
    size_t cachedBytes = 0;
};
// Set once the calling thread's state below has been torn down (thread exit, or the main thread's part of process
// exit); a trivially destructible flag, so it stays readable while static destructors run afterwards.
thread_local bool t_threadStateGone = false;
thread_local DevBlockCache t_cache;
constexpr size_t kCacheLimit = (size_t)2 << 30;
// Defined after t_cache, so constructed after it and destroyed before it: gives a thread's cached blocks, pinned
// arena, scratch frames and streams back when the thread ends.
struct ThreadCleanup {
    ThreadCleanup() {}
    ~ThreadCleanup();
};
thread_local ThreadCleanup t_cleanup;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int device = 0;
    hipError_t alloc(size_t bytes) {
        size_t want = 4096;
        while (want < bytes) want <<= 1;
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess) return e;
        if (t_threadStateGone) {
            e = hipMalloc(&p, want);
            if (e == hipSuccess) cap = want; else p = nullptr;
            return e;
        }
        (void)&t_cleanup;
        auto& f = t_cache.free;
        for (size_t i = 0; i < f.size(); ++i)
            if (f[i].cap == want && f[i].device == device) {
                p = f[i].p; cap = want;
                t_cache.cachedBytes -= want;
                f[i] = f.back(); f.pop_back();
                return hipSuccess;
            }
        e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() {
        if (!p) return;
        if (t_threadStateGone) {
            // a destructor running after this thread's state is gone (a static object at process exit): the runtime
            // may be shutting down as well, so the block is left to the process teardown
            p = nullptr; cap = 0;
            return;
        }
        if (t_cache.cachedBytes + cap <= kCacheLimit) {
            t_cache.free.push_back({p, cap, device});
            t_cache.cachedBytes += cap;
        } else {
            (void)hipFree(p);
        }
        p = nullptr; cap = 0;
    }
    // keeps the block when it is already large enough (contents are not preserved otherwise)
    hipError_t ensure(size_t bytes) {
        if (p && cap >= bytes) return hipSuccess;
        release();
        return alloc(bytes);
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    template <class T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

namespace {
// Host memory the device reads and writes in place (queries in, candidate lists out): a search
// is three kernel launches and one stream synchronisation, with no copy commands in between.
// One arena and one stream per host thread; both live until orbx_release_cached_memory().
struct PinnedArena {
    char* host = nullptr;
    char* dev = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = (size_t)1 << 20;
        while (want < bytes) want <<= 1;
        void* h = nullptr;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) return e;
        void* d = nullptr;
        e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { (void)hipHostFree(h); return e; }
        host = static_cast<char*>(h); dev = static_cast<char*>(d); cap = want;
        return hipSuccess;
    }
    void release() {
        if (host) (void)hipHostFree(host);
        host = dev = nullptr; cap = 0;
    }
};
thread_local PinnedArena t_arena;
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int kMaxDevices = 16;
thread_local hipStream_t t_streams[kMaxDevices] = {};

hipError_t thread_stream(hipStream_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) { *out = nullptr; return hipSuccess; }   // the default stream
    (void)&t_cleanup;
    if (!t_streams[dev]) {
        e = hipStreamCreateWithFlags(&t_streams[dev], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = t_streams[dev];
    return hipSuccess;
}
}  // namespace

namespace {
// A copy on the calling thread's stream that has completed, as far as the host can tell, when it returns:
// uploads from pageable memory are staged before the call returns, downloads are waited for.
hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    if (e != hipSuccess || kind != hipMemcpyDeviceToHost) return e;
    return hipStreamSynchronize(st);
}

// Several small host arrays in ONE upload and several small results in ONE download, both through the calling
// thread's pinned arena (a copy from or to pageable memory costs 10-15 us per command whatever its size).
// add() / want() lay the arrays out (256-byte aligned) and return their offsets; upload() packs and enqueues the
// host-to-device copy into `in`; download() enqueues the copy of `out` back, waits for the stream and hands out
// the results with fetch().
struct PackedIo {
    struct Item { const void* src; size_t bytes, off; };
    std::vector<Item> ups;
    size_t upBytes = 0, downBytes = 0;
    DevBuf in, out;
    size_t add(const void* src, size_t bytes) {
        const size_t off = upBytes;
        ups.push_back({src, bytes, off});
        upBytes = align256(upBytes + std::max<size_t>(bytes, 1));
        return off;
    }
    size_t want(size_t bytes) {
        const size_t off = downBytes;
        downBytes = align256(downBytes + std::max<size_t>(bytes, 1));
        return off;
    }
    hipError_t upload(hipStream_t st) {
        hipError_t e = t_arena.ensure(upBytes + downBytes);
        if (e != hipSuccess) return e;
        for (const Item& it : ups)
            if (it.bytes) memcpy(t_arena.host + it.off, it.src, it.bytes);
        if ((e = in.alloc(std::max<size_t>(upBytes, 256))) != hipSuccess) return e;
        if ((e = out.alloc(std::max<size_t>(downBytes, 256))) != hipSuccess) return e;
        return hipMemcpyAsync(in.p, t_arena.host, upBytes, hipMemcpyHostToDevice, st);
    }
    template <class T> T* dev_in(size_t off) { return reinterpret_cast<T*>(in.as<char>() + off); }
    template <class T> T* dev_out(size_t off) { return reinterpret_cast<T*>(out.as<char>() + off); }
    hipError_t download(hipStream_t st) {
        hipError_t e = hipMemcpyAsync(t_arena.host + upBytes, out.p, downBytes, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(st);
    }
    void fetch(void* dst, size_t off, size_t bytes) const { if (bytes) memcpy(dst, t_arena.host + upBytes + off, bytes); }
};
}  // namespace

static void release_thread_staging();   // the pinned arena and the scratch frames of the windowed searches

extern "C" void orbx_release_cached_memory(void) {
    if (t_threadStateGone) return;
    release_thread_staging();
    for (auto& b : t_cache.free) (void)hipFree(b.p);
    t_cache.free.clear();
    t_cache.cachedBytes = 0;
    for (auto& st : t_streams)
        if (st) { (void)hipStreamDestroy(st); st = nullptr; }
}

namespace {
ThreadCleanup::~ThreadCleanup() {
    orbx_release_cached_memory();
    t_threadStateGone = true;
}
}  // namespace

__device__ __forceinline__ int hamming256(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += __popc(a[k] ^ b[k]);
    return d;
}

// ======================================================================================
// DescriptorDistance for n pairs
// ======================================================================================
__global__ void k_pair_distance(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int n, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hamming256(a + 8ll * i, b + 8ll * i);
}

extern "C" orbx_status orbx_descriptor_distance(const uint8_t* a, const uint8_t* b, int n, int* out) {
    ORBX_REQUIRE(n >= 0 && (n == 0 || (a && b && out)), "bad argument");
    if (n == 0) return ORBX_OK;
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    PackedIo io;
    const size_t oA = io.add(a, (size_t)n * 32), oB = io.add(b, (size_t)n * 32), oOut = io.want((size_t)n * 4);
    ORBX_HIP_TRY(io.upload(st));
    hipLaunchKernelGGL(k_pair_distance, dim3((n + 255) / 256), dim3(256), 0, st, io.dev_in<uint32_t>(oA), io.dev_in<uint32_t>(oB), n,
                       io.dev_out<int>(oOut));
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(io.download(st));
    io.fetch(out, oOut, (size_t)n * 4);
    return ORBX_OK;
}

// ======================================================================================
// Brute-force 2-nearest-neighbour search.
//
// Each lane owns kQpl query descriptors in registers (8 VGPRs each) together with their
// running best / second best.  Train descriptors are wave-uniform: they are fetched with
// scalar loads and applied to all 64 x kQpl queries of the wave, so the inner loop is
// 8 x (v_xor_b32 with an SGPR operand + v_bcnt_u32_b32 accumulate) per pair and one compare.
// The train set is cut into spans of kKnnSpanRows rows (fewer for a small set: knn2_span_rows).  grid.y = the workers of a query group, as many as
// fill the chip once; a worker (one wave) takes the next unclaimed span of its query group from a counter until
// none is left, so waves that the SIMD's arbitration favours simply take more spans and all of them finish
// within one span of each other (with one fixed slice per wave the slowest wave of a SIMD ran its last third
// alone, at a third of the issue rate).  Every worker writes one partial top-2 with global indices and
// k_knn2_merge folds the workers' results.
// Ties: a worker scans its spans in ascending order with strict '<', which keeps the lowest index, and the merge
// is lexicographic on (distance, index), which is what a single sequential scan gives.
// ======================================================================================
constexpr int kQpl = 4;        // queries per lane
constexpr int kKnnWaves = 4;
constexpr int kRowsPerIter = 8;   // train rows fetched (scalar) and applied per loop trip
constexpr int kKnnSpanRows = 2048;   // most rows a worker claims at a time (a multiple of kRowsPerIter)
constexpr int kKnnMinSpanRows = 32;  // fewest: a small train set is cut into this many rows per span so that more than one wave per query group works on it

// Per-query register state.  Four named instances (not an array) so that everything stays in
// VGPRs: a runtime-indexed array of these would be placed in scratch memory.
struct KnnQuery {
    uint32_t q[8];
    int best, second, bestIdx, secondIdx;
    int qi;
};

__device__ __forceinline__ void knn_load(KnnQuery& s, const uint32_t* __restrict__ query, int qi, int nq) {
    s.qi = qi;
    const uint4* src = reinterpret_cast<const uint4*>(query + 8ll * (qi < nq ? qi : 0));
    const uint4 a = src[0], b = src[1];
    s.q[0] = a.x; s.q[1] = a.y; s.q[2] = a.z; s.q[3] = a.w;
    s.q[4] = b.x; s.q[5] = b.y; s.q[6] = b.z; s.q[7] = b.w;
    s.best = INT_MAX; s.second = INT_MAX; s.bestIdx = -1; s.secondIdx = -1;
}

__device__ __forceinline__ void knn_update(KnnQuery& s, int d, int gj) {
    if (d < s.second) {
        if (d < s.best) { s.second = s.best; s.secondIdx = s.bestIdx; s.best = d; s.bestIdx = gj; }
        else { s.second = d; s.secondIdx = gj; }
    }
}

// 8 x (v_xor_b32 + v_bcnt_u32_b32 with its accumulator operand) = 16 instructions per pair.  Written
// with the instruction itself: left to the compiler, the eight counts are summed with three extra
// v_add3_u32 per pair.
__device__ __forceinline__ int knn_dist(const KnnQuery& s, const uint32_t (&t)[8]) {
    uint32_t acc;
    {
        const uint32_t x = s.q[0] ^ t[0];
        asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(acc) : "v"(x));
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const uint32_t x = s.q[k] ^ t[k];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
    }
    return (int)acc;
}

// (The s_nop 0 behind every v_bcnt is worth 17 % of the whole search: 22.4 -> 18.6 ms for 5000 x 10 M.  A v_xor_b32
// that directly follows a v_bcnt_u32_b32 of its own wave issues at the slow rate of that instruction, 8.5 cycles per
// pair and SIMD; with one idle issue slot of the wave in between it issues at its own rate, 6.3 cycles per pair --
// which is also what a stream of the bare pair reaches in a micro-benchmark whose compiler-made code happens to hold
// such a no-op per pair.  A no-op in front of the v_bcnt as well loses the gain again: 22.0 ms.)
// Four rows against one query: the bookkeeping is entered only when the smallest of the four
// distances beats the current second best, and then replays the rows in ascending index order
// (strict '<' keeps the lowest index among equal distances).
__device__ __forceinline__ void knn_rows4(KnnQuery& s, const uint32_t (&t)[4][8], int gj) {
    // the four accumulator chains advance together, word by word
    uint32_t acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = s.q[0] ^ t[r][0];
        asm("v_bcnt_u32_b32 %0, %1, 0\n\ts_nop 0" : "=v"(acc[r]) : "v"(x));
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t x = s.q[k] ^ t[r][k];
            asm("v_bcnt_u32_b32 %0, %1, %0\n\ts_nop 0" : "+v"(acc[r]) : "v"(x));
        }
    }
    const int d0 = (int)acc[0], d1 = (int)acc[1], d2 = (int)acc[2], d3 = (int)acc[3];
    const int m = min(min(d0, d1), min(d2, d3));
    if (__builtin_expect(m < s.second, 0)) {
        knn_update(s, d0, gj); knn_update(s, d1, gj + 1); knn_update(s, d2, gj + 2); knn_update(s, d3, gj + 3);
    }
}

// The distances of four rows, without the bookkeeping (knn_rows8 decides once for eight rows).
__device__ __forceinline__ void knn_dist4(const KnnQuery& s, const uint32_t (&t)[4][8], int (&d)[4]) {
    uint32_t acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = s.q[0] ^ t[r][0];
        asm("v_bcnt_u32_b32 %0, %1, 0\n\ts_nop 0" : "=v"(acc[r]) : "v"(x));
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t x = s.q[k] ^ t[r][k];
            asm("v_bcnt_u32_b32 %0, %1, %0\n\ts_nop 0" : "+v"(acc[r]) : "v"(x));
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = (int)acc[r];
}

// Second half of an eight-row step: rows gj + 4 .. gj + 7 are in t, the distances of rows gj .. gj + 3 in lo.  One
// test of the smallest of the eight against the current second best; the replay is in ascending index order.
__device__ __forceinline__ void knn_rows8_finish(KnnQuery& s, const int (&lo)[4], const uint32_t (&t)[4][8], int gj) {
    int hi[4];
    knn_dist4(s, t, hi);
    const int m = min(min(min(lo[0], lo[1]), min(lo[2], lo[3])), min(min(hi[0], hi[1]), min(hi[2], hi[3])));
    if (m < s.second) {
        knn_update(s, lo[0], gj); knn_update(s, lo[1], gj + 1); knn_update(s, lo[2], gj + 2); knn_update(s, lo[3], gj + 3);
        knn_update(s, hi[0], gj + 4); knn_update(s, hi[1], gj + 5); knn_update(s, hi[2], gj + 6); knn_update(s, hi[3], gj + 7);
    }
}

__device__ __forceinline__ void knn_store(const KnnQuery& s, int nq, orbx_match2* __restrict__ part) {
    if (s.qi < nq) {
        orbx_match2 m;
        m.idx0 = s.bestIdx; m.dist0 = s.best; m.idx1 = s.secondIdx; m.dist1 = s.second;
        part[s.qi] = m;
    }
}

__global__ __launch_bounds__(64 * kKnnWaves) void k_knn2(const uint32_t* __restrict__ query, int nq,
                                                          const uint32_t* __restrict__ train, long long nt,
                                                          int nSpans, int spanRows, long long baseIndex,
                                                          orbx_match2* __restrict__ part, int* __restrict__ spanCounter) {
    static_assert(kQpl == 4 && kRowsPerIter == 8, "the body below is written out for 4 queries x 2 groups of 4 rows");
    static_assert(kKnnSpanRows % kRowsPerIter == 0, "a span is a whole number of loop trips");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int group = blockIdx.x * kKnnWaves + wave;
    if (group * 64 * kQpl >= nq) return;

    KnnQuery a, b, c, d;
    const int q0 = group * 64 * kQpl + lane;
    knn_load(a, query, q0, nq);
    knn_load(b, query, q0 + 64, nq);
    knn_load(c, query, q0 + 128, nq);
    knn_load(d, query, q0 + 192, nq);

  // the next unclaimed span of this query group; the counter only grows, so the loop ends for every wave once it
  // has passed nSpans (the host clears the counters before the launch)
  for (;;) {
    // (spans claimed per workgroup instead, so that its four waves read the same rows and share them in the scalar
    // cache, with a barrier per span: 22.93 against 22.40 ms -- the scalar loads are not what the waves wait for)
    int span = 0;
    if (lane == 0) span = atomicAdd(&spanCounter[group], 1);
    span = __builtin_amdgcn_readfirstlane(span);
    if (span >= nSpans) break;
    const long long j0 = (long long)span * spanRows;
    const long long j1 = min(nt, j0 + spanRows);
    long long j = j0;
    // 8 rows per trip, fetched as two groups of 4: the scalar loads of the second group are in flight
    // while the first group is applied, and the next trip's first group while the second is applied
    if (j + kRowsPerIter <= j1) {
        uint32_t ta[4][8], tb[4][8];
        {
            const uint32_t* __restrict__ t = train + 8ll * j;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) ta[r][k] = t[8 * r + k];
        }
        for (; j + kRowsPerIter <= j1; j += kRowsPerIter) {
            const uint32_t* __restrict__ t = train + 8ll * j;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) tb[r][k] = t[32 + 8 * r + k];
            const int gj = (int)(baseIndex + j);
#ifdef ORBX_KNN_CHECK8
            int la[4], lb[4], lc[4], ld[4];
            knn_dist4(a, ta, la); knn_dist4(b, ta, lb); knn_dist4(c, ta, lc); knn_dist4(d, ta, ld);
#else
            knn_rows4(a, ta, gj); knn_rows4(b, ta, gj); knn_rows4(c, ta, gj); knn_rows4(d, ta, gj);
#endif
            // first group of the next trip.  Scalar loads return out of order, so every wait is for all of them:
            // wait for the second group here, before the next loads are issued, not at its first use
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            if (j + 2 * kRowsPerIter <= j1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) ta[r][k] = t[64 + 8 * r + k];
            }
#ifdef ORBX_KNN_CHECK8
            knn_rows8_finish(a, la, tb, gj); knn_rows8_finish(b, lb, tb, gj); knn_rows8_finish(c, lc, tb, gj); knn_rows8_finish(d, ld, tb, gj);
#else
            knn_rows4(a, tb, gj + 4); knn_rows4(b, tb, gj + 4); knn_rows4(c, tb, gj + 4); knn_rows4(d, tb, gj + 4);
#endif
        }
    }
    for (; j < j1; ++j) {
        const uint32_t* __restrict__ t = train + 8ll * j;
        uint32_t t0[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t0[k] = t[k];
        const int gj = (int)(baseIndex + j);
        knn_update(a, knn_dist(a, t0), gj);
        knn_update(b, knn_dist(b, t0), gj);
        knn_update(c, knn_dist(c, t0), gj);
        knn_update(d, knn_dist(d, t0), gj);
    }
  }
    orbx_match2* out = part + (long long)blockIdx.y * nq;
    knn_store(a, nq, out); knn_store(b, nq, out); knn_store(c, nq, out); knn_store(d, nq, out);
}

__device__ __forceinline__ bool lex_less(int d0, int i0, int d1, int i1) { return d0 < d1 || (d0 == d1 && i0 < i1); }

// Top-2 fold of two (distance, index) candidates lists in lexicographic order; a missing entry has index -1.
__device__ __forceinline__ void knn_fold(int& b, int& bi, int& s, int& si, int d, int i) {
    if (i < 0) return;
    if (bi < 0 || lex_less(d, i, b, bi)) { s = b; si = bi; b = d; bi = i; }
    else if (si < 0 || lex_less(d, i, s, si)) { s = d; si = i; }
}

// One wave per query: the lanes stride over the partial results, then a butterfly folds the 64 lane-local
// top-2 pairs.  (The lexicographic top-2 of a union is the top-2 of the parts' top-2s, in any grouping.)
__global__ __launch_bounds__(256) void k_knn2_merge(const orbx_match2* __restrict__ part, int nParts, int nq,
                                                    orbx_match2* __restrict__ out) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= nq) return;
    int b = INT_MAX, bi = -1, s = INT_MAX, si = -1;
    for (int p = lane; p < nParts; p += 64) {
        const orbx_match2 m = part[(long long)p * nq + qi];
        knn_fold(b, bi, s, si, m.dist0, m.idx0);
        knn_fold(b, bi, s, si, m.dist1, m.idx1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ob = __shfl_xor(b, off), obi = __shfl_xor(bi, off), os = __shfl_xor(s, off), osi = __shfl_xor(si, off);
        knn_fold(b, bi, s, si, ob, obi);
        knn_fold(b, bi, s, si, os, osi);
    }
    if (lane == 0) {
        orbx_match2 m;
        m.idx0 = bi; m.dist0 = b; m.idx1 = si; m.dist1 = s;
        out[qi] = m;
    }
}

// Workgroups the current device keeps resident at 8 waves per SIMD (k_knn2 needs < 64 VGPRs): compute units x 8.
static int knn2_resident_blocks() {
    static std::atomic<int> cached[kMaxDevices];          // zero-initialised; several threads may fill an entry, all with the same value
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 2048;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (!v) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = cus * 32 / kKnnWaves;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Workers (grid.y) the chip holds per query group: one resident round and no more -- a grid a few workgroups larger
// than what fits at once leaves those to start after everything else has finished.
static long long knn2_workers_wanted(int nq) {
    const int groups = (nq + 64 * kQpl - 1) / (64 * kQpl);
    const int blocksPerChunk = (groups + kKnnWaves - 1) / kKnnWaves;
    return std::max(1, knn2_resident_blocks() / std::max(blocksPerChunk, 1));
}
// Rows per span: kKnnSpanRows for a large train set; a train set too small to give every wanted worker such a span
// is cut finer (down to kKnnMinSpanRows), or a 1000 x 1000 search would be one wave per 256 queries walking all rows.
static int knn2_span_rows(int nq, long long nt) {
    const long long per = (nt + knn2_workers_wanted(nq) - 1) / knn2_workers_wanted(nq);
    const long long rows = (per + kRowsPerIter - 1) / kRowsPerIter * kRowsPerIter;
    return (int)std::min<long long>(kKnnSpanRows, std::max<long long>(kKnnMinSpanRows, rows));
}
// ... and no more workers than spans.
static int knn2_chunks(int nq, long long nt) {
    const int spanRows = knn2_span_rows(nq, nt);
    long long spans = std::max<long long>(1, (nt + spanRows - 1) / spanRows);
    long long c = std::max<long long>(1, std::min(knn2_workers_wanted(nq), spans));
    return (int)std::min<long long>(c, 4096);
}
static int knn2_groups(int nq) { return (nq + 64 * kQpl - 1) / (64 * kQpl); }
// the span counters (one int per query group) sit behind the workers' partial results
static size_t knn2_counter_bytes(int nq) { return ((size_t)knn2_groups(nq) * sizeof(int) + 255) & ~(size_t)255; }

// Sized for the device that is current when the search runs if it is current now; for any device of up to 512 CUs
// otherwise (the worker count follows the CU count of the current device).
extern "C" size_t orbx_knn2_workspace_bytes(int n_query, int64_t n_train) {
    if (n_query <= 0 || n_train <= 0) return 16;
    return (size_t)knn2_chunks(n_query, n_train) * (size_t)n_query * sizeof(orbx_match2) + knn2_counter_bytes(n_query);
}

// Clears the span counters and launches the search into `parts` (workers x n_query rows) with `counters` behind it.
static void knn2_launch(const uint8_t* d_query, int n_query, const uint8_t* d_train, int64_t n_train, int64_t base_index,
                        int workers, orbx_match2* parts, int* counters, hipStream_t st) {
    const int groups = knn2_groups(n_query);
    const int spanRows = knn2_span_rows(n_query, n_train);
    const int nSpans = (int)((n_train + spanRows - 1) / spanRows);
    (void)hipMemsetAsync(counters, 0, (size_t)groups * sizeof(int), st);
    dim3 grid((groups + kKnnWaves - 1) / kKnnWaves, workers);
    hipLaunchKernelGGL(k_knn2, grid, dim3(64 * kKnnWaves), 0, st, (const uint32_t*)d_query, n_query, (const uint32_t*)d_train,
                       (long long)n_train, nSpans, spanRows, (long long)base_index, parts, counters);
}

extern "C" orbx_status orbx_hamming_knn2_device(const uint8_t* d_query, int n_query, const uint8_t* d_train,
                                                int64_t n_train, int64_t base_index, orbx_match2* d_out,
                                                void* d_workspace, size_t workspace_bytes, void* stream) {
    ORBX_REQUIRE(n_query >= 0 && n_train >= 0, "negative size");
    if (n_query == 0) return ORBX_OK;
    ORBX_REQUIRE(d_query && d_out && (n_train == 0 || d_train), "null argument");
    ORBX_REQUIRE(base_index >= 0 && base_index + n_train <= (int64_t)INT_MAX, "train index does not fit 32 bits");
    ORBX_REQUIRE((((uintptr_t)d_query | (uintptr_t)d_train) & 15) == 0, "descriptor arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = n_train > 0 ? knn2_chunks(n_query, n_train) : 0;
    if (chunks > 0) {
        const size_t partBytes = (size_t)chunks * n_query * sizeof(orbx_match2);
        ORBX_REQUIRE(d_workspace && workspace_bytes >= partBytes + knn2_counter_bytes(n_query),
                     "workspace too small (%zu bytes, %zu needed on this device: ask orbx_knn2_workspace_bytes with the search's "
                     "device current)", workspace_bytes, partBytes + knn2_counter_bytes(n_query));
        ORBX_REQUIRE(((uintptr_t)d_workspace & 15) == 0, "workspace must be 16-byte aligned");
        knn2_launch(d_query, n_query, d_train, n_train, base_index, chunks, (orbx_match2*)d_workspace,
                    (int*)((char*)d_workspace + partBytes), st);
    }
    hipLaunchKernelGGL(k_knn2_merge, dim3((n_query + 3) / 4), dim3(256), 0, st, (const orbx_match2*)d_workspace, chunks,
                       n_query, d_out);
    ORBX_HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

// --------------------------------------------------------------------------------------
// The same search for callers that may not allocate (stream capture, serving loops): the
// per-chunk partial results go to a fixed array in the library's own device image instead of
// a caller workspace.  Every GPU holds its own copy of the array; host code must never use
// the array's name as an address, only what hipGetSymbolAddress returns on that GPU.
// --------------------------------------------------------------------------------------
constexpr size_t kKnnStaticBytes = (size_t)16 << 20;
static __device__ __align__(256) unsigned char g_knnStatic[kKnnStaticBytes];

namespace {
void* g_knnStaticPtr[kMaxDevices] = {};
std::mutex g_knnStaticMutex;

// Address of the current device's copy of g_knnStatic (looked up once per device).
orbx_status knn_static_ptr(void** out) {
    int dev = 0;
    ORBX_HIP_TRY(hipGetDevice(&dev));
    ORBX_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d is beyond the %d this library tracks", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_knnStaticMutex);
    if (!g_knnStaticPtr[dev]) {
        void* p = nullptr;
        ORBX_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_knnStatic)));
        ORBX_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address");
        g_knnStaticPtr[dev] = p;
    }
    *out = g_knnStaticPtr[dev];
    return ORBX_OK;
}
}  // namespace

// the largest n_query whose one partial row set and span counters fit the array.  Also resolves the array's address
// on the current device (when there is one), so that a caller about to capture a graph can do that outside the capture.
extern "C" size_t orbx_knn2_static_max_queries(void) {
    void* warm = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) (void)knn_static_ptr(&warm);
    size_t n = kKnnStaticBytes / sizeof(orbx_match2);
    while (n > 0 && n * sizeof(orbx_match2) + knn2_counter_bytes((int)n) > kKnnStaticBytes) n -= 64 * kQpl;
    return n;
}

extern "C" orbx_status orbx_hamming_knn2_device_static(const uint8_t* d_query, int n_query, const uint8_t* d_train,
                                                       int64_t n_train, int64_t base_index, orbx_match2* d_out,
                                                       void* stream) {
    ORBX_REQUIRE(n_query >= 0 && n_train >= 0, "negative size");
    if (n_query == 0) return ORBX_OK;
    ORBX_REQUIRE(d_query && d_out && (n_train == 0 || d_train), "null argument");
    ORBX_REQUIRE(base_index >= 0 && base_index + n_train <= (int64_t)INT_MAX, "train index does not fit 32 bits");
    ORBX_REQUIRE((((uintptr_t)d_query | (uintptr_t)d_train) & 15) == 0, "descriptor arrays must be 16-byte aligned");
    const size_t perChunk = (size_t)n_query * sizeof(orbx_match2);
    const size_t counterBytes = knn2_counter_bytes(n_query);
    ORBX_REQUIRE(perChunk + counterBytes <= kKnnStaticBytes,
                 "%d queries exceed the %zu the allocation-free search supports (its partial results live in a fixed "
                 "%zu-byte device array); use orbx_hamming_knn2_device with a workspace",
                 n_query, orbx_knn2_static_max_queries(), kKnnStaticBytes);
    void* scratch = nullptr;
    orbx_status s = knn_static_ptr(&scratch);
    if (s != ORBX_OK) return s;
    hipStream_t st = (hipStream_t)stream;
    // as many chunks as the workspace variant would use, capped by what the array holds; the merge is
    // lexicographic on (distance, index), so the result does not depend on the number of chunks
    int chunks = 0;
    if (n_train > 0) chunks = (int)std::min<size_t>((size_t)knn2_chunks(n_query, n_train), (kKnnStaticBytes - counterBytes) / perChunk);
    // the span counters take the array's last bytes and are cleared on the stream before every launch (hipMemsetAsync
    // on device memory allocates nothing); every partial the merge reads is written by the launch
    if (chunks > 0)
        knn2_launch(d_query, n_query, d_train, n_train, base_index, chunks, (orbx_match2*)scratch,
                    (int*)((char*)scratch + kKnnStaticBytes - counterBytes), st);
    hipLaunchKernelGGL(k_knn2_merge, dim3((n_query + 3) / 4), dim3(256), 0, st, (const orbx_match2*)scratch, chunks,
                       n_query, d_out);
    ORBX_HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" orbx_status orbx_knn2_merge_device(const orbx_match2* d_parts, int n_parts, int n_query, orbx_match2* d_out,
                                              void* stream) {
    ORBX_REQUIRE(n_parts >= 0 && n_query >= 0, "negative size");
    if (n_query == 0) return ORBX_OK;
    ORBX_REQUIRE(d_parts && d_out, "null argument");
    hipLaunchKernelGGL(k_knn2_merge, dim3((n_query + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_parts, n_parts,
                       n_query, d_out);
    ORBX_HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

// ======================================================================================
// Frame::ComputeStereoMatches: one wave per left keypoint.
//   (a) best right keypoint on the same row band / neighbouring octave / disparity range
//   (b) 11x11 SAD over 11 horizontal shifts on the keypoint's pyramid level, parabola fit
// then one workgroup per frame rejects matches whose SAD is >= 2.1 x the median SAD.
// ======================================================================================
struct OrbxPyrView {
    const uint8_t* left;
    const uint8_t* right;
    long long frameStride;      // bytes between frames (same for both: same geometry)
};

// Waves per workgroup.  The waves share nothing, and a small workgroup finds room on a compute unit that other
// streams' kernels keep full (8 waves per workgroup: 266 k frames/s in the pipelined benchmark, 2 or 1: 270 k).
constexpr int kStWaves = 2;
constexpr int kStPerWave = 4;        // left keypoints handled by one wave, one after the other
constexpr int kStBuckets = 256;      // row buckets of the staged keypoints (multiple of 64)

// Right keypoints of one frame bucketed by image row (counting sort), once per frame: for every entry the row band
// minr | (maxr + 1) << 16 of Frame.cc:980-990, its x, and (octave << 20) | index.  A left keypoint then reads only
// the buckets its row can touch.  Order inside a bucket is arbitrary (the match is a lexicographic minimum).
struct OrbxStereoSorted {
    uint32_t* rows;     // capacity entries per frame
    float* x;
    uint32_t* octIdx;
    int* start;         // kStBuckets + 1 per frame
};

__device__ __forceinline__ int stereo_bucket_shift(int nRows) {
    int shift = 3;
    while (((nRows - 1) >> shift) >= kStBuckets) ++shift;
    return shift;
}

constexpr int kStPrepThreads = 128;   // one small workgroup per frame (a thousand keypoints): placed at once beside other kernels
__global__ __launch_bounds__(kStPrepThreads) void k_stereo_prep(const OrbxFrameGeo* __restrict__ geo,
                                                     const orbx_keypoint* __restrict__ kpsR,
                                                     const int* __restrict__ countsR, int capacity, OrbxStereoSorted out) {
    __shared__ int s_start[kStBuckets + 1];
    __shared__ int s_fill[kStBuckets];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int nR = min(countsR[2 * frame], capacity);
    const int nRows = geo->lv[0].h;
    const int shift = stereo_bucket_shift(nRows);
    const orbx_keypoint* KR = kpsR + (long long)frame * capacity;
    for (int i = tid; i < kStBuckets; i += kStPrepThreads) { s_start[i + 1] = 0; s_fill[i] = 0; }
    if (tid == 0) s_start[0] = 0;
    __syncthreads();
    for (int j = tid; j < nR; j += kStPrepThreads) {
        const int b = min(max((int)KR[j].y, 0), nRows - 1) >> shift;
        atomicAdd(&s_start[b + 1], 1);
    }
    __syncthreads();
    if (tid < 64) {   // inclusive scan of the bucket counts: kStBuckets / 64 per lane
        constexpr int per = kStBuckets / 64;
        int v[per], sum = 0;
#pragma unroll
        for (int q = 0; q < per; ++q) { v[q] = s_start[1 + lane * per + q]; sum += v[q]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        int run = incl - sum;
#pragma unroll
        for (int q = 0; q < per; ++q) { run += v[q]; s_start[1 + lane * per + q] = run; }
    }
    __syncthreads();
    uint32_t* rows = out.rows + (long long)frame * capacity;
    float* xs = out.x + (long long)frame * capacity;
    uint32_t* oi = out.octIdx + (long long)frame * capacity;
    for (int j = tid; j < nR; j += kStPrepThreads) {
        const orbx_keypoint kr = KR[j];
        const int b = min(max((int)kr.y, 0), nRows - 1) >> shift;
        const int slot = s_start[b] + atomicAdd(&s_fill[b], 1);
        const float r = __fmul_rn(2.0f, geo->lv[kr.octave].scale);
        const int maxr = min(max((int)ceilf(__fadd_rn(kr.y, r)), -1), 65534);
        const int minr = min(max((int)floorf(__fsub_rn(kr.y, r)), 0), 65535);   // rows are >= 0: clamping keeps the test
        rows[slot] = (uint32_t)minr | ((uint32_t)(maxr + 1) << 16);
        xs[slot] = kr.x;
        oi[slot] = ((uint32_t)kr.octave << 20) | (uint32_t)j;
    }
    for (int i = tid; i <= kStBuckets; i += kStPrepThreads) out.start[(long long)frame * (kStBuckets + 1) + i] = s_start[i];
}

__global__ __launch_bounds__(64 * kStWaves) void k_stereo_rows(const OrbxFrameGeo* __restrict__ geo, OrbxPyrView pv,
                                                     const orbx_keypoint* __restrict__ kpsL, const uint32_t* __restrict__ descL,
                                                     const int* __restrict__ countsL,
                                                     const orbx_keypoint* __restrict__ kpsR, const uint32_t* __restrict__ descR,
                                                     const int* __restrict__ countsR, int capacity, float mbf, float mb,
                                                     float* __restrict__ uRight, float* __restrict__ depth,
                                                     int* __restrict__ sadOut, OrbxStereoSorted sorted) {
    __shared__ uint32_t s_patch[kStWaves][11 * 4 + 11 * 8];   // left rows: 4 dwords each, right rows: 8 dwords each
    __shared__ int s_part[kStWaves][128];
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nL = min(countsL[2 * frame], capacity);
    const int nRows = geo->lv[0].h;
    const float minZ = mb, minD = 0.f;
    const float maxD = __fdiv_rn(mbf, minZ);
    const int iL0 = (blockIdx.x * kStWaves + wv) * kStPerWave;
    if (iL0 >= nL) return;     // nothing shared between the waves of a workgroup: idle waves leave
    // bucket = row >> shift, at most kStBuckets buckets; a right keypoint at row y can only match left
    // rows within reach = ceil(2 * largest scale) + 1 of y
    const int shift = stereo_bucket_shift(nRows);
    const int reach = (int)ceilf(__fmul_rn(2.0f, geo->lv[geo->nlevels - 1].scale)) + 1;
    const uint32_t* sRows = sorted.rows + (long long)frame * capacity;
    const float* sX = sorted.x + (long long)frame * capacity;
    const uint32_t* sOctIdx = sorted.octIdx + (long long)frame * capacity;
    const int* sStart = sorted.start + (long long)frame * (kStBuckets + 1);

    // per left keypoint of this wave (all wave-uniform)
    int rowOf[kStPerWave], levelOf[kStPerWave];
    float minUOf[kStPerWave], maxUOf[kStPerWave];
    unsigned int bestKey[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        const int iL = iL0 + k;
        rowOf[k] = -1; levelOf[k] = 0; minUOf[k] = 0.f; maxUOf[k] = 0.f; bestKey[k] = 0xFFFFFFFFu;
        if (iL < nL) {
            const long long rowL = (long long)frame * capacity + iL;
            const orbx_keypoint kp = kpsL[rowL];
            if (lane == 0) { uRight[rowL] = -1.f; depth[rowL] = -1.f; sadOut[rowL] = -1; }
            const long long rowInt = (long long)kp.y;          // vRowIndices[vL]: float -> size_t
            const float maxU = __fsub_rn(kp.x, minD);
            if (rowInt >= 0 && rowInt < nRows && !(maxU < 0)) {
                rowOf[k] = (int)rowInt; levelOf[k] = kp.octave;
                minUOf[k] = __fsub_rn(kp.x, maxD); maxUOf[k] = maxU;
            }
        }
    }

    // (a) lexicographic min over (distance, right index) of the gated candidates.  The wave's keypoints go through
    // the two dependent reads (bucket entries, then the descriptors of the entries that pass the gates) side by
    // side: first 64 entries of every keypoint's range at once, longer ranges continue one keypoint at a time.
    int jBeg[kStPerWave], jEnd[kStPerWave];
    uint32_t rrOf[kStPerWave], oiOf[kStPerWave];
    float xOf[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        jBeg[k] = 0; jEnd[k] = 0; rrOf[k] = 0u; oiOf[k] = 0u; xOf[k] = 0.f;
        if (rowOf[k] < 0) continue;
        const int bLo = max(rowOf[k] - reach, 0) >> shift, bHi = min(rowOf[k] + reach, nRows - 1) >> shift;
        jBeg[k] = sStart[bLo]; jEnd[k] = sStart[bHi + 1];
        const int t = jBeg[k] + lane;
        if (t < jEnd[k]) { rrOf[k] = sRows[t]; oiOf[k] = sOctIdx[t]; xOf[k] = sX[t]; }
    }
    uint32_t dr[kStPerWave][8];
    bool pass[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        const int row = rowOf[k], lvl = levelOf[k], oc = (int)(oiOf[k] >> 20);
        pass[k] = jBeg[k] + lane < jEnd[k] && !(row < (int)(rrOf[k] & 0xFFFFu) || row + 1 > (int)(rrOf[k] >> 16)) &&
                  !(oc < lvl - 1 || oc > lvl + 1) && (xOf[k] >= minUOf[k] && xOf[k] <= maxUOf[k]);
#pragma unroll
        for (int q = 0; q < 8; ++q) dr[k][q] = 0u;
        if (pass[k]) {
            const uint32_t* dR = descR + ((long long)frame * capacity + (int)(oiOf[k] & 0xFFFFFu)) * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) dr[k][q] = dR[q];
        }
    }
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        if (rowOf[k] < 0) continue;
        const uint32_t* dL = descL + ((long long)frame * capacity + iL0 + k) * 8;
        const int row = rowOf[k], lvl = levelOf[k];
        unsigned int key = bestKey[k];
        if (pass[k]) {
            int d = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) d += __popc(dL[q] ^ dr[k][q]);
            if (d < 100) key = min(key, ((unsigned int)d << 20) | (oiOf[k] & 0xFFFFFu));
        }
        for (int t = jBeg[k] + 64 + lane; t < jEnd[k]; t += 64) {
            const uint32_t rr = sRows[t];
            if (row < (int)(rr & 0xFFFFu) || row + 1 > (int)(rr >> 16)) continue;
            const uint32_t oi = sOctIdx[t];
            const int oc = (int)(oi >> 20);
            if (oc < lvl - 1 || oc > lvl + 1) continue;
            const float x = sX[t];
            if (!(x >= minUOf[k] && x <= maxUOf[k])) continue;
            const int j = (int)(oi & 0xFFFFFu);
            const uint32_t* dR = descR + ((long long)frame * capacity + j) * 8;
            int d = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) d += __popc(dL[q] ^ dR[q]);
            if (d < 100) key = min(key, ((unsigned int)d << 20) | (unsigned int)j);
        }
        bestKey[k] = key;
    }

    // (b) sub-pixel refinement by correlation on the pyramid level of the left keypoint.  The global reads of the
    // wave's keypoints are issued together, stage by stage (keypoint records, then the pixel windows), so that the
    // wave pays each memory round trip once and not once per keypoint.
    uint32_t* patch = s_patch[wv];
    int* part = s_part[wv];
    unsigned int keyOf[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        unsigned int key = bestKey[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) key = min(key, (unsigned int)__shfl_xor((int)key, off));
        keyOf[k] = (unsigned int)__builtin_amdgcn_readfirstlane((int)key);
    }
    // stage 1: the left keypoint and the x of its best right keypoint (wave-uniform addresses)
    bool ok[kStPerWave];
    orbx_keypoint kpOf[kStPerWave];
    float uR0Of[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        ok[k] = keyOf[k] != 0xFFFFFFFFu && (int)(keyOf[k] >> 20) < 75;   // thOrbDist = (TH_HIGH + TH_LOW) / 2
        const long long rowL = (long long)frame * capacity + iL0 + k;
        const int bestIdxR = ok[k] ? (int)(keyOf[k] & 0xFFFFFu) : 0;
        kpOf[k] = kpsL[ok[k] ? rowL : (long long)frame * capacity];
        uR0Of[k] = kpsR[(long long)frame * capacity + bestIdxR].x;
    }
    // stage 2: window geometry, then 11 x 4 dwords of the left patch and 11 x 7 dwords of the right strip
    // (121 dwords: lane i takes dwords i and i + 64)
    float suR0Of[kStPerWave];
    int shLOf[kStPerWave], offROf[kStPerWave];
    uint32_t pa[kStPerWave], pb[kStPerWave];
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        pa[k] = pb[k] = 0u; suR0Of[k] = 0.f; shLOf[k] = 0; offROf[k] = 0;
        if (!ok[k]) continue;
        const orbx_keypoint kp = kpOf[k];
        const OrbxLevelGeo& G = geo->lv[kp.octave];
        const float sInv = G.invScale;
        const float suL = roundf(__fmul_rn(kp.x, sInv)), svL = roundf(__fmul_rn(kp.y, sInv)), suR0 = roundf(__fmul_rn(uR0Of[k], sInv));
        const int w = 5, L = 5;
        const float iniu = __fsub_rn(__fadd_rn(suR0, (float)L), (float)w);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(suR0, (float)L), (float)w), 1.f);
        if (iniu < 0 || endu >= (float)G.w) { ok[k] = false; continue; }
        const int y0 = (int)(svL - w), xl0 = (int)(suL - w), xr0 = (int)(suR0 - w) - L;   // xr0: first column of the right strip
        // the windows of keypoints produced by the extractor always lie inside the level; anything else is not matched
        if (y0 < 0 || y0 + 11 > G.h || xl0 < 0 || xl0 + 11 > G.w || xr0 < 0 || xr0 + 21 > G.w) { ok[k] = false; continue; }
        const uint8_t* imL = pv.left + (long long)frame * pv.frameStride + G.off;
        const uint8_t* imR = pv.right + (long long)frame * pv.frameStride + G.off;
        const int xla = xl0 & ~3, xra = xr0 & ~3;
        suR0Of[k] = suR0; shLOf[k] = xl0 - xla; offROf[k] = xr0 - xra;
        auto fetch = [&](int i) -> uint32_t {
            if (i < 44) {
                const int r = i >> 2, d = i & 3;
                return *reinterpret_cast<const uint32_t*>(imL + (long long)(y0 + r) * G.pitch + xla + 4 * d);
            }
            const int t = i - 44, r = t / 7, d = t - r * 7;
            return *reinterpret_cast<const uint32_t*>(imR + (long long)(y0 + r) * G.pitch + xra + 4 * d);
        };
        pa[k] = fetch(lane);
        if (lane + 64 < 11 * 4 + 11 * 7) pb[k] = fetch(lane + 64);
    }
    // stage 3: correlation, one keypoint after the other through the wave's LDS patch
#pragma unroll
    for (int k = 0; k < kStPerWave; ++k) {
        if (!ok[k]) continue;
        const int L = 5;
        const long long rowL = (long long)frame * capacity + iL0 + k;
        const float uL = kpOf[k].x;
        const OrbxLevelGeo& G = geo->lv[kpOf[k].octave];
        const float suR0 = suR0Of[k];
        {
            const int i0 = lane, i1 = lane + 64;
            patch[i0 < 44 ? i0 : 44 + ((i0 - 44) / 7) * 8 + (i0 - 44) % 7] = pa[k];
            if (i1 < 11 * 4 + 11 * 7) patch[44 + ((i1 - 44) / 7) * 8 + (i1 - 44) % 7] = pb[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // item = (row dy, shift s): SAD of 11 pixels = three v_sad_u8 over byte windows
        const int shL = shLOf[k], offR = offROf[k];
        for (int p = lane; p < 121; p += 64) {
            const int dy = p / 11, sft = p - dy * 11;
            const uint32_t* lp = patch + dy * 4;
            const uint32_t a0 = __builtin_amdgcn_alignbyte(lp[1], lp[0], shL), a1 = __builtin_amdgcn_alignbyte(lp[2], lp[1], shL);
            const uint32_t a2 = __builtin_amdgcn_alignbyte(lp[3], lp[2], shL) & 0x00FFFFFFu;
            const int sb = offR + sft;
            const uint32_t* rp = patch + 44 + dy * 8 + (sb >> 2);
            const int sh = sb & 3;
            const uint32_t b0 = __builtin_amdgcn_alignbyte(rp[1], rp[0], sh), b1 = __builtin_amdgcn_alignbyte(rp[2], rp[1], sh);
            const uint32_t b2 = __builtin_amdgcn_alignbyte(rp[3], rp[2], sh) & 0x00FFFFFFu;
            part[p] = (int)__builtin_amdgcn_sad_u8(a2, b2, __builtin_amdgcn_sad_u8(a1, b1, __builtin_amdgcn_sad_u8(a0, b0, 0u)));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        int mine = 0;
        if (lane < 11) {
#pragma unroll
            for (int dy = 0; dy < 11; ++dy) mine += part[dy * 11 + lane];
        }
        int sad[11];
#pragma unroll
        for (int sft = 0; sft < 11; ++sft) sad[sft] = __shfl(mine, sft);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // patch / part are reused by the next keypoint
        if (lane != 0) continue;
        int bestSad = INT_MAX, bestInc = 0;
#pragma unroll
        for (int sft = 0; sft < 11; ++sft)
            if (sad[sft] < bestSad) { bestSad = sad[sft]; bestInc = sft - L; }
        if (bestInc == -L || bestInc == L) continue;
        float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int sft = 1; sft < 10; ++sft)
            if (sft - L == bestInc) { d1 = (float)sad[sft - 1]; d2 = (float)sad[sft]; d3 = (float)sad[sft + 1]; }
        const float delta = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (delta < -1 || delta > 1) continue;
        float bestuR = __fmul_rn(G.scale, __fadd_rn(__fadd_rn(suR0, (float)bestInc), delta));
        float disparity = __fsub_rn(uL, bestuR);
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01f;                              // float(0.01)
                bestuR = (float)((double)uL - 0.01);            // uL - 0.01 in double, then to float
            }
            depth[rowL] = __fdiv_rn(mbf, disparity);
            uRight[rowL] = bestuR;
            sadOut[rowL] = bestSad;
        }
    }
}

// Median cut: threshold = 1.5f * 1.4f * (SAD of the (count/2)-th match in (SAD, index) order).
__global__ __launch_bounds__(256) void k_stereo_median(const int* __restrict__ countsL, int capacity,
                                                       float* __restrict__ uRight, float* __restrict__ depth,
                                                       const int* __restrict__ sad) {
    __shared__ int hist[256];
    __shared__ int s_bin, s_rank, s_cnt, s_median;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int nL = min(countsL[2 * frame], capacity);
    const long long base = (long long)frame * capacity;
    // pass 1: high part of the SAD (121 * 255 < 2^15)
    hist[tid] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < nL; i += 256) {
        const int v = sad[base + i];
        if (v >= 0) { atomicAdd(&hist[v >> 7], 1); ++local; }
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt == 0) return;
    if (tid == 0) {
        int k = cnt / 2, acc = 0, b = 0;
        for (; b < 256; ++b) { if (acc + hist[b] > k) break; acc += hist[b]; }
        s_bin = b; s_rank = k - acc;
    }
    __syncthreads();
    const int bin = s_bin, rank = s_rank;
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nL; i += 256) {
        const int v = sad[base + i];
        if (v >= 0 && (v >> 7) == bin) atomicAdd(&hist[v & 127], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 128; ++b) { if (acc + hist[b] > rank) break; acc += hist[b]; }
        s_median = (bin << 7) | b;
    }
    __syncthreads();
    const float median = (float)s_median;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    for (int i = tid; i < nL; i += 256) {
        const int v = sad[base + i];
        if (v >= 0 && !((float)v < thDist)) { uRight[base + i] = -1.f; depth[base + i] = -1.f; }
    }
}

// geometry accessors implemented in orbx_extract.hip
extern "C" orbx_status orbx_pyramid_level_device(orbx_extractor*, int, int, const uint8_t**, int*, int*, ptrdiff_t*);
const OrbxFrameGeo* orbx_internal_geo_device(orbx_extractor* ex);
const OrbxFrameGeo* orbx_internal_geo_host(orbx_extractor* ex);
const uint8_t* orbx_internal_pyramid(orbx_extractor* ex);
hipStream_t orbx_internal_stream(orbx_extractor* ex);
int orbx_internal_device(orbx_extractor* ex);
int orbx_internal_max_batch(orbx_extractor* ex);
void* orbx_internal_scratch(orbx_extractor* ex, size_t bytes);

extern "C" orbx_status orbx_stereo_match_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames,
                                                      const orbx_keypoint* d_kps_left, const uint8_t* d_desc_left,
                                                      const int* d_counts_left, const orbx_keypoint* d_kps_right,
                                                      const uint8_t* d_desc_right, const int* d_counts_right, int capacity,
                                                      float mbf, float mb, float* d_u_right, float* d_depth, void* stream) {
    ORBX_REQUIRE(left && right && d_kps_left && d_desc_left && d_counts_left && d_kps_right && d_desc_right &&
                 d_counts_right && d_u_right && d_depth, "null argument");
    const OrbxFrameGeo* gl = orbx_internal_geo_host(left);
    const OrbxFrameGeo* gr = orbx_internal_geo_host(right);
    ORBX_REQUIRE(gl && gr, "extractors have not processed an image yet");
    ORBX_REQUIRE(gl->w == gr->w && gl->h == gr->h && gl->nlevels == gr->nlevels && gl->pyrBytes == gr->pyrBytes,
                 "left and right extractors must have the same geometry");
    ORBX_REQUIRE(n_frames >= 1 && n_frames <= orbx_internal_max_batch(left) && n_frames <= orbx_internal_max_batch(right),
                 "bad n_frames");
    ORBX_REQUIRE(capacity >= 1 && capacity < (1 << 20), "bad capacity");
    ORBX_REQUIRE(mb > 0.f, "baseline must be positive");
    ORBX_HIP_TRY(hipSetDevice(orbx_internal_device(left)));
    hipStream_t st = (hipStream_t)stream;   // taken literally: NULL is the device's default stream
    OrbxPyrView pv;
    pv.left = orbx_internal_pyramid(left);
    pv.right = orbx_internal_pyramid(right);
    pv.frameStride = gl->pyrBytes;
    // scratch: the SAD of every left keypoint, then the row-sorted right keypoints of every frame
    const size_t perFrame = (size_t)capacity * 4;
    const size_t startBytes = (((size_t)n_frames * (kStBuckets + 1) * sizeof(int)) + 255) & ~(size_t)255;
    char* scratch = (char*)orbx_internal_scratch(left, (size_t)n_frames * perFrame * 4 + startBytes);
    if (!scratch) { orbx_set_error("cannot allocate the stereo scratch buffer"); return ORBX_ERR_HIP; }
    int* d_sad = (int*)scratch;
    OrbxStereoSorted sorted;
    sorted.rows = (uint32_t*)(scratch + (size_t)n_frames * perFrame);
    sorted.x = (float*)(scratch + (size_t)n_frames * perFrame * 2);
    sorted.octIdx = (uint32_t*)(scratch + (size_t)n_frames * perFrame * 3);
    sorted.start = (int*)(scratch + (size_t)n_frames * perFrame * 4);
    hipLaunchKernelGGL(k_stereo_prep, dim3(n_frames), dim3(kStPrepThreads), 0, st, orbx_internal_geo_device(left), d_kps_right,
                       d_counts_right, capacity, sorted);
    hipLaunchKernelGGL(k_stereo_rows, dim3((capacity + kStWaves * kStPerWave - 1) / (kStWaves * kStPerWave), n_frames),
                       dim3(64 * kStWaves), 0, st, orbx_internal_geo_device(left), pv,
                       d_kps_left, (const uint32_t*)d_desc_left, d_counts_left, d_kps_right, (const uint32_t*)d_desc_right,
                       d_counts_right, capacity, mbf, mb, d_u_right, d_depth, d_sad, sorted);
    hipLaunchKernelGGL(k_stereo_median, dim3(n_frames), dim3(256), 0, st, d_counts_left, capacity, d_u_right, d_depth, d_sad);
    ORBX_HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" orbx_status orbx_stereo_match(orbx_extractor* left, orbx_extractor* right, int frame,
                                         const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left,
                                         const orbx_keypoint* kps_right, const uint8_t* desc_right, int n_right, float mbf,
                                         float mb, float* u_right, float* depth) {
    ORBX_REQUIRE(left && right && n_left >= 0 && n_right >= 0, "bad argument");
    ORBX_REQUIRE(frame == 0, "the host entry point matches batch slot 0");
    if (n_left == 0) return ORBX_OK;
    ORBX_REQUIRE(kps_left && desc_left && u_right && depth && (n_right == 0 || (kps_right && desc_right)), "null argument");
    ORBX_HIP_TRY(hipSetDevice(orbx_internal_device(left)));
    const int cap = std::max(std::max(n_left, n_right), 1);
    // One upload and one download: the inputs are packed into the calling thread's pinned arena (pageable copies of a
    // few dozen KB each cost 10-15 us apiece; eight of them were two thirds of this call), the results come back
    // through it.
    const size_t kb = (size_t)cap * sizeof(orbx_keypoint), db = (size_t)cap * 32, fb = align256((size_t)cap * 4);
    const size_t oKL = 0, oDL = align256(kb), oKR = align256(oDL + db), oDR = align256(oKR + kb), oC = align256(oDR + db);
    const size_t inBytes = oC + 256, oOut = inBytes;
    ORBX_HIP_TRY(t_arena.ensure(oOut + 2 * fb));
    DevBuf in, out;
    ORBX_HIP_TRY(in.alloc(inBytes)); ORBX_HIP_TRY(out.alloc(2 * fb));
    char* h = t_arena.host;
    memcpy(h + oKL, kps_left, (size_t)n_left * sizeof(orbx_keypoint));
    memcpy(h + oDL, desc_left, (size_t)n_left * 32);
    if (n_right) {
        memcpy(h + oKR, kps_right, (size_t)n_right * sizeof(orbx_keypoint));
        memcpy(h + oDR, desc_right, (size_t)n_right * 32);
    }
    const int counts[4] = {n_left, 0, n_right, 0};
    memcpy(h + oC, counts, sizeof(counts));
    hipStream_t st = orbx_internal_stream(left);      // upload, kernels and download in one stream
    ORBX_HIP_TRY(hipMemcpyAsync(in.p, h, inBytes, hipMemcpyHostToDevice, st));
    ORBX_HIP_TRY(hipStreamSynchronize(orbx_internal_stream(right)));   // the right pyramid was built on its own stream
    char* d = in.as<char>();
    float* dUr = out.as<float>();
    float* dDp = reinterpret_cast<float*>(out.as<char>() + fb);
    orbx_status s = orbx_stereo_match_batch_device(left, right, 1, reinterpret_cast<orbx_keypoint*>(d + oKL),
                                                   reinterpret_cast<uint8_t*>(d + oDL), reinterpret_cast<int*>(d + oC),
                                                   reinterpret_cast<orbx_keypoint*>(d + oKR), reinterpret_cast<uint8_t*>(d + oDR),
                                                   reinterpret_cast<int*>(d + oC) + 2, cap, mbf, mb, dUr, dDp, st);
    if (s != ORBX_OK) return s;
    ORBX_HIP_TRY(hipMemcpyAsync(h + oOut, out.p, 2 * fb, hipMemcpyDeviceToHost, st));
    ORBX_HIP_TRY(hipStreamSynchronize(st));
    memcpy(u_right, h + oOut, (size_t)n_left * 4);
    memcpy(depth, h + oOut + fb, (size_t)n_left * 4);
    return ORBX_OK;
}

// ======================================================================================
// Windowed candidate search (GetFeaturesInArea + per-candidate distance)
// ======================================================================================
constexpr int kGridCols = 64, kGridRows = 48;   // FRAME_GRID_COLS / FRAME_GRID_ROWS (Frame.h:43-44)

struct OrbxGridDev {
    const int* cellStart;     // kGridCols * kGridRows + 1, cell = ix * kGridRows + iy
    const int* cellIdx;       // keypoint indices, insertion (ascending) order inside a cell
    float minX, minY, invW, invH;
};

// One wave per query.  mode 0: count candidates; mode 1: write them at candOffset[q].
__global__ __launch_bounds__(256) void k_window(OrbxGridDev g, const orbx_keypoint* __restrict__ kps,
                                                const uint32_t* __restrict__ desc, const float* __restrict__ uright,
                                                const orbx_query* __restrict__ queries, const uint32_t* __restrict__ qdesc,
                                                int nq, int mode, int* __restrict__ candCount,
                                                const int* __restrict__ candOffset, int* __restrict__ candIdx,
                                                int* __restrict__ candDist, int candCap,
                                                const float* __restrict__ invSigma2) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    const orbx_query Q = queries[q];
    const float r = Q.radius;
    int total = 0;
    const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.x, g.minX), r), g.invW)));
    const int maxCX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.x, g.minX), r), g.invW)));
    const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.y, g.minY), r), g.invH)));
    const int maxCY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.y, g.minY), r), g.invH)));
    const bool empty = (minCX >= kGridCols) || (maxCX < 0) || (minCY >= kGridRows) || (maxCY < 0);
    const bool checkLevels = (Q.min_level > 0) || (Q.max_level >= 0);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = qdesc[8ll * q + k];
    const int outBase = (mode == 1) ? candOffset[q] : 0;
    if (!empty) {
        for (int ix = minCX; ix <= maxCX; ++ix) {
            // cells (ix, minCY..maxCY) are contiguous in the CSR
            const int b = g.cellStart[ix * kGridRows + minCY], e = g.cellStart[ix * kGridRows + maxCY + 1];
            for (int c0 = b; c0 < e; c0 += 64) {
                const int c = c0 + lane;
                bool ok = false;
                int idx = -1;
                if (c < e) {
                    idx = g.cellIdx[c];
                    const orbx_keypoint kp = kps[idx];
                    ok = true;
                    if (checkLevels) {
                        if (kp.octave < Q.min_level) ok = false;
                        if (Q.max_level >= 0 && kp.octave > Q.max_level) ok = false;
                    }
                    const float dx = __fsub_rn(kp.x, Q.x), dy = __fsub_rn(kp.y, Q.y);
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) ok = false;
                    if (ok && invSigma2 != nullptr) {
                        // Fuse: reprojection error against the keypoint, chi-square gates 7.8 (stereo
                        // keypoint) / 5.99 (ORBmatcher.cc:1341-1365)
                        const float ur = uright ? uright[idx] : -1.f;
                        float e2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));   // (u - kpx)^2 == (kpx - u)^2
                        if (ur >= 0) {
                            const float er = __fsub_rn(Q.xr, ur);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            if ((double)__fmul_rn(e2, invSigma2[kp.octave]) > 7.8) ok = false;
                        } else if ((double)__fmul_rn(e2, invSigma2[kp.octave]) > 5.99) ok = false;
                    } else if (ok && uright != nullptr) {
                        const float ur = uright[idx];
                        if (ur > 0 && fabsf(__fsub_rn(Q.xr, ur)) > r) ok = false;
                    }
                }
                const unsigned long long bal = __ballot(ok);
                if (mode == 1 && ok) {
                    const int pos = outBase + total + __popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < candCap) {
                        int d = 0;
#pragma unroll
                        for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ desc[8ll * idx + k]);
                        candIdx[pos] = idx;
                        candDist[pos] = d;
                    }
                }
                total += __popcll(bal);
            }
        }
    }
    if (mode == 0 && lane == 0) candCount[q] = total;
}

// ======================================================================================
// Frame handle: the features of one image resident on the device, with the 64 x 48 grid of
// Frame::AssignFeaturesToGrid built there.  Every windowed search of the frame reuses it.
// ======================================================================================
constexpr int kGridCells = kGridCols * kGridRows;
constexpr int kGridThreads = 1024;
constexpr int kCellsPerThread = kGridCells / kGridThreads;
static_assert(kCellsPerThread * kGridThreads == kGridCells, "each thread owns a whole number of grid cells");

// Exclusive prefix of `v` over the workgroup (kGridThreads lanes); *total = the sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave /* kGridThreads / 64 */, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < kGridThreads / 64; ++w) {
        const int t = s_wave[w];
        if (w < wave) before += t;
        all += t;
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

// Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:468-506, :860-870) in one workgroup:
// cell = (round((x - minX) * invW), round((y - minY) * invH)); keypoints outside the 64 x 48
// cells are left out; inside a cell the indices ascend (the reference appends in index order).
// Output is CSR: cellStart[kGridCells + 1] with cell = column * 48 + row, and cellIdx[].
__global__ __launch_bounds__(kGridThreads) void k_assign_grid(const orbx_keypoint* __restrict__ kps, int n, float minX,
                                                              float minY, float invW, float invH,
                                                              int* __restrict__ cellStart, int* __restrict__ cellIdx) {
    __shared__ int s_cell[kGridCells];   // counts, then fill cursors
    __shared__ int s_wave[kGridThreads / 64];
    const int tid = threadIdx.x;
    for (int c = tid; c < kGridCells; c += kGridThreads) s_cell[c] = 0;
    __syncthreads();
    auto cell_of = [&](int i) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, minX), invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, minY), invH));
        return (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? -1 : px * kGridRows + py;
    };
    for (int i = tid; i < n; i += kGridThreads) {
        const int c = cell_of(i);
        if (c >= 0) atomicAdd(&s_cell[c], 1);
    }
    __syncthreads();
    int cnt[kCellsPerThread], sum = 0;
#pragma unroll
    for (int k = 0; k < kCellsPerThread; ++k) { cnt[k] = s_cell[tid * kCellsPerThread + k]; sum += cnt[k]; }
    int total;
    int start = block_exclusive_scan(sum, s_wave, &total);
    int first[kCellsPerThread];
#pragma unroll
    for (int k = 0; k < kCellsPerThread; ++k) {
        first[k] = start;
        cellStart[tid * kCellsPerThread + k] = start;
        s_cell[tid * kCellsPerThread + k] = start;
        start += cnt[k];
    }
    if (tid == 0) cellStart[kGridCells] = total;
    __syncthreads();
    for (int i = tid; i < n; i += kGridThreads) {
        const int c = cell_of(i);
        if (c >= 0) cellIdx[atomicAdd(&s_cell[c], 1)] = i;
    }
    __syncthreads();   // the fill is complete and visible to the whole workgroup
    // the atomics filled each cell in arbitrary order: sort the (short) cells ascending
#pragma unroll
    for (int k = 0; k < kCellsPerThread; ++k) {
        const int b = first[k], e = b + cnt[k];
        for (int i = b + 1; i < e; ++i) {
            const int v = cellIdx[i];
            int j = i - 1;
            while (j >= b && cellIdx[j] > v) { cellIdx[j + 1] = cellIdx[j]; --j; }
            cellIdx[j + 1] = v;
        }
    }
}

// off[0..n] = exclusive prefix sums of count[0..n), written to two places (a device copy for
// the kernel that follows and a host-visible copy).  One workgroup.
__global__ __launch_bounds__(kGridThreads) void k_exclusive_scan(const int* __restrict__ count, int n,
                                                                 int* __restrict__ offA, int* __restrict__ offB) {
    __shared__ int s_wave[kGridThreads / 64];
    const int per = (n + kGridThreads - 1) / kGridThreads;
    const int b = min(n, (int)threadIdx.x * per), e = min(n, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += count[i];
    int total;
    int run = block_exclusive_scan(sum, s_wave, &total);
    for (int i = b; i < e; ++i) { offA[i] = run; offB[i] = run; run += count[i]; }
    if (threadIdx.x == 0) { offA[n] = total; offB[n] = total; }
}


// Host-array brute-force search (Frame::ComputeStereoFishEyeMatches' knnMatch, Frame.cc:1309).  Per-frame sizes
// (a thousand descriptors a side) go through the calling thread's pinned arena: one staged upload, the two
// kernels, results written straight to host memory, one synchronisation of the thread's own stream.
extern "C" orbx_status orbx_hamming_knn2(const uint8_t* query, int n_query, const uint8_t* train, int64_t n_train,
                                         orbx_match2* out) {
    ORBX_REQUIRE(n_query >= 0 && n_train >= 0, "negative size");
    if (n_query == 0) return ORBX_OK;
    ORBX_REQUIRE(query && out && (n_train == 0 || train), "null argument");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    DevBuf dq, dt, dws;
    const size_t ws = orbx_knn2_workspace_bytes(n_query, n_train);
    const size_t qBytes = (size_t)n_query * 32, tBytes = (size_t)n_train * 32, oBytes = (size_t)n_query * sizeof(orbx_match2);
    ORBX_HIP_TRY(dq.alloc(qBytes)); ORBX_HIP_TRY(dt.alloc(std::max<size_t>(tBytes, 32))); ORBX_HIP_TRY(dws.alloc(ws));
    const bool staged = qBytes + tBytes + oBytes <= ((size_t)8 << 20);
    const size_t oT = align256(qBytes), oOut = align256(oT + tBytes);
    ORBX_HIP_TRY(t_arena.ensure(staged ? oOut + oBytes : align256(oBytes)));
    orbx_match2* hOut = reinterpret_cast<orbx_match2*>(t_arena.host + (staged ? oOut : 0));
    orbx_match2* dOut = reinterpret_cast<orbx_match2*>(t_arena.dev + (staged ? oOut : 0));
    if (staged) {
        memcpy(t_arena.host, query, qBytes);
        if (tBytes) memcpy(t_arena.host + oT, train, tBytes);
        ORBX_HIP_TRY(hipMemcpyAsync(dq.p, t_arena.host, qBytes, hipMemcpyHostToDevice, st));
        if (tBytes) ORBX_HIP_TRY(hipMemcpyAsync(dt.p, t_arena.host + oT, tBytes, hipMemcpyHostToDevice, st));
    } else {
        ORBX_HIP_TRY(hipMemcpyAsync(dq.p, query, qBytes, hipMemcpyHostToDevice, st));
        ORBX_HIP_TRY(hipMemcpyAsync(dt.p, train, tBytes, hipMemcpyHostToDevice, st));
    }
    orbx_status s = orbx_hamming_knn2_device(dq.as<uint8_t>(), n_query, dt.as<uint8_t>(), n_train, 0, dOut, dws.p, ws, st);
    if (s != ORBX_OK) return s;
    ORBX_HIP_TRY(hipStreamSynchronize(st));
    memcpy(out, hOut, oBytes);
    return ORBX_OK;
}

extern "C" orbx_status orbx_stereo_fisheye_matches(const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left,
                                                   int mono_left, const orbx_keypoint* kps_right,
                                                   const uint8_t* desc_right, int n_right, int mono_right,
                                                   const orbx_camera_model* cam_left, const orbx_camera_model* cam_right,
                                                   const float* R_lr, const float* t_lr, const float* level_sigma2,
                                                   int n_levels, int* left_to_right, int* right_to_left, float* depth,
                                                   float* points3d, int* n_matches) {
    ORBX_REQUIRE(n_left >= 0 && n_right >= 0 && mono_left >= 0 && mono_left <= n_left && mono_right >= 0 && mono_right <= n_right,
                 "bad keypoint counts / lapping indices");
    ORBX_REQUIRE(n_matches && (n_left == 0 || (left_to_right && depth)) && (n_right == 0 || right_to_left), "null argument");
    for (int i = 0; i < n_left; ++i) { left_to_right[i] = -1; depth[i] = -1.f; }
    for (int i = 0; i < n_right; ++i) right_to_left[i] = -1;
    *n_matches = 0;
    const int nq = n_left - mono_left, nt = n_right - mono_right;
    if (nq == 0 || nt == 0) return ORBX_OK;
    ORBX_REQUIRE(kps_left && desc_left && kps_right && desc_right && cam_left && cam_right && R_lr && t_lr && level_sigma2 &&
                 n_levels >= 1, "null argument");
    std::vector<orbx_match2> m((size_t)nq);
    orbx_status s = orbx_hamming_knn2(desc_left + (size_t)mono_left * 32, nq, desc_right + (size_t)mono_right * 32, nt, m.data());
    if (s != ORBX_OK) return s;
    orbx_gates::Camera c1, c2;
    c1.type = cam_left->type; c2.type = cam_right->type;
    for (int k = 0; k < 8; ++k) { c1.p[k] = cam_left->params[k]; c2.p[k] = cam_right->params[k]; }
    int nm = 0;
    for (int i = 0; i < nq; ++i) {
        if (m[i].idx1 < 0) continue;                                            // fewer than two neighbours (:1318)
        if (!((double)(float)m[i].dist0 < (double)(float)m[i].dist1 * 0.7)) continue;
        const int iL = i + mono_left, iR = m[i].idx0 + mono_right;
        const orbx_keypoint& a = kps_left[iL];
        const orbx_keypoint& b = kps_right[iR];
        ORBX_REQUIRE(a.octave >= 0 && a.octave < n_levels && b.octave >= 0 && b.octave < n_levels, "keypoint level outside the sigma table");
        orbx_gates::Vec3 P{0.f, 0.f, 0.f};
        const float z = orbx_gates::triangulate_matches(c1, c2, a.x, a.y, b.x, b.y, R_lr, t_lr, level_sigma2[a.octave],
                                                        level_sigma2[b.octave], &P);
        if (z > 0.0001f) {
            left_to_right[iL] = iR;
            right_to_left[iR] = iL;
            depth[iL] = z;
            if (points3d) { points3d[3 * iL] = P.x; points3d[3 * iL + 1] = P.y; points3d[3 * iL + 2] = P.z; }
            ++nm;
        }
    }
    *n_matches = nm;
    return ORBX_OK;
}

// ---- the camera-model calls on their own (host arithmetic only) ------------------------------------------
static orbx_gates::Camera gate_camera(const orbx_camera_model* m) {
    orbx_gates::Camera c;
    c.type = m->type;
    for (int k = 0; k < 8; ++k) c.p[k] = m->params[k];
    return c;
}
static bool gate_camera_ok(const orbx_camera_model* m) {
    return m && (m->type == ORBX_CAMERA_PINHOLE || m->type == ORBX_CAMERA_KANNALA_BRANDT8);
}

extern "C" orbx_status orbx_camera_project(const orbx_camera_model* cam, const float* xyz, int n, float* uv) {
    ORBX_REQUIRE(gate_camera_ok(cam) && n >= 0 && (n == 0 || (xyz && uv)), "bad camera model or null argument");
    const orbx_gates::Camera c = gate_camera(cam);
    for (int i = 0; i < n; ++i)
        orbx_gates::project(c, orbx_gates::Vec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, &uv[2 * i], &uv[2 * i + 1]);
    return ORBX_OK;
}

extern "C" orbx_status orbx_camera_unproject(const orbx_camera_model* cam, const float* uv, int n, float* rays) {
    ORBX_REQUIRE(gate_camera_ok(cam) && n >= 0 && (n == 0 || (uv && rays)), "bad camera model or null argument");
    const orbx_gates::Camera c = gate_camera(cam);
    for (int i = 0; i < n; ++i) {
        const orbx_gates::Vec3 r = orbx_gates::unproject(c, uv[2 * i], uv[2 * i + 1]);
        rays[3 * i] = r.x; rays[3 * i + 1] = r.y; rays[3 * i + 2] = r.z;
    }
    return ORBX_OK;
}

extern "C" orbx_status orbx_camera_triangulate_matches(const orbx_camera_model* cam1, const orbx_camera_model* cam2,
                                                       const float* uv1, const float* uv2, int n, const float* R12,
                                                       const float* t12, const float* sigma_level, const float* unc,
                                                       float* depth, float* points3d) {
    ORBX_REQUIRE(gate_camera_ok(cam1) && gate_camera_ok(cam2) && R12 && t12 && n >= 0, "bad camera model or null argument");
    ORBX_REQUIRE(n == 0 || (uv1 && uv2 && sigma_level && unc && depth), "null argument");
    const orbx_gates::Camera c1 = gate_camera(cam1), c2 = gate_camera(cam2);
    for (int i = 0; i < n; ++i) {
        orbx_gates::Vec3 P{0.f, 0.f, 0.f};
        depth[i] = orbx_gates::triangulate_matches(c1, c2, uv1[2 * i], uv1[2 * i + 1], uv2[2 * i], uv2[2 * i + 1], R12, t12,
                                                   sigma_level[i], unc[i], &P);
        if (points3d && depth[i] > 0) { points3d[3 * i] = P.x; points3d[3 * i + 1] = P.y; points3d[3 * i + 2] = P.z; }
    }
    return ORBX_OK;
}

extern "C" orbx_status orbx_camera_epipolar_constrain(const orbx_camera_model* cam1, const orbx_camera_model* cam2,
                                                      const float* uv1, const float* uv2, int n, const float* R12,
                                                      const float* t12, const float* sigma_level, const float* unc,
                                                      uint8_t* accepted) {
    ORBX_REQUIRE(gate_camera_ok(cam1) && gate_camera_ok(cam2) && R12 && t12 && n >= 0, "bad camera model or null argument");
    ORBX_REQUIRE(n == 0 || (uv1 && uv2 && sigma_level && unc && accepted), "null argument");
    const orbx_gates::Camera c1 = gate_camera(cam1), c2 = gate_camera(cam2);
    for (int i = 0; i < n; ++i)
        accepted[i] = orbx_gates::epipolar_constrain(c1, c2, uv1[2 * i], uv1[2 * i + 1], uv2[2 * i], uv2[2 * i + 1], R12, t12,
                                                     sigma_level[i], unc[i]) ? 1 : 0;
    return ORBX_OK;
}

struct orbx_frame {
    int device = -1;       // where the features live; searches switch the calling thread to it
    int n = 0;
    bool hasUright = false;
    float minX = 0, maxX = 0, minY = 0, maxY = 0, invW = 0, invH = 0;
    DevBuf kps, desc, uright, cellStart, cellIdx;
    std::vector<orbx_keypoint> hostKps;   // what the in-order loops read (level, angle, position)
    std::vector<float> hostUright;        // mvuRight for the stereo flags of SearchForTriangulation
};

// cv::undistortPoints(src, dst, K, D, R = I, P = K) for one point, as Frame::UndistortKeyPoints and
// ComputeImageBounds call it (Frame.cc:901, :932): normalise, five fixed-point iterations of the inverse
// radial-tangential model in binary64 (k = k1 k2 p1 p2 k3 k4 k5 k6), project with the same K (given as
// floats, used as doubles), round to float.  Written once for the kernel and the host (the four image corners).
struct CamD { double fx, fy, cx, cy, k[8]; };
__host__ __device__ inline void undistort_point(const CamD& c, float uf, float vf, float* ox, float* oy) {
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    const double u = uf, v = vf;
    double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
    const double x0 = x, y0 = y;
    const double* k = c.k;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = c.fx * x + 0. * y + c.cx, yy = 0. * x + c.fy * y + c.cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}

// in == out is allowed (each thread reads its own record before writing it)
__global__ __launch_bounds__(256) void k_undistort(CamD c, const orbx_keypoint* in, int n, orbx_keypoint* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    orbx_keypoint kp = in[i];
    undistort_point(c, kp.x, kp.y, &kp.x, &kp.y);
    out[i] = kp;
}

namespace {
orbx_status camera_to_double(const orbx_camera* cam, CamD& c, bool& distorted) {
    ORBX_REQUIRE(cam, "null camera");
    ORBX_REQUIRE(cam->n_dist >= 0 && cam->n_dist <= 8, "n_dist %d outside [0, 8]", cam->n_dist);
    ORBX_REQUIRE(cam->fx != 0.f && cam->fy != 0.f, "zero focal length");
    c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy;
    for (int i = 0; i < 8; ++i) c.k[i] = i < cam->n_dist ? (double)cam->dist[i] : 0.0;
    distorted = cam->n_dist > 0 && cam->dist[0] != 0.0f;      // the reference's test (Frame.cc:884, :918)
    return ORBX_OK;
}
}  // namespace

extern "C" orbx_status orbx_image_bounds(const orbx_camera* cam, int width, int height, float bounds[4]) {
    ORBX_REQUIRE(bounds && width > 0 && height > 0, "bad argument");
    CamD c;
    bool distorted;
    orbx_status s = camera_to_double(cam, c, distorted);
    if (s != ORBX_OK) return s;
    if (!distorted) { bounds[0] = 0.f; bounds[1] = (float)width; bounds[2] = 0.f; bounds[3] = (float)height; return ORBX_OK; }
    float x[4], y[4];
    const float cu[4] = {0.f, (float)width, 0.f, (float)width}, cv[4] = {0.f, 0.f, (float)height, (float)height};
    for (int i = 0; i < 4; ++i) undistort_point(c, cu[i], cv[i], &x[i], &y[i]);
    bounds[0] = std::min(x[0], x[2]); bounds[1] = std::max(x[1], x[3]);
    bounds[2] = std::min(y[0], y[1]); bounds[3] = std::max(y[2], y[3]);
    return ORBX_OK;
}

namespace {
// Candidate lists of one search; they point into the calling thread's arena and stay valid until
// that thread's next search.
struct WindowLists {
    const int* off = nullptr;
    const int* idx = nullptr;
    const int* dist = nullptr;
    std::vector<int> none;   // offsets of a search without queries or keypoints
    void set_empty(int nq) { none.assign((size_t)nq + 1, 0); off = none.data(); idx = dist = nullptr; }
};

orbx_status frame_set_bounds(orbx_frame& F, float min_x, float max_x, float min_y, float max_y) {
    ORBX_REQUIRE(max_x > min_x && max_y > min_y, "empty image bounds");
    F.minX = min_x; F.maxX = max_x; F.minY = min_y; F.maxY = max_y;
    F.invW = (float)kGridCols / (max_x - min_x);     // mfGridElementWidthInv (Frame.cc:127-128)
    F.invH = (float)kGridRows / (max_y - min_y);
    return ORBX_OK;
}

// Uploads (or copies, for device sources) the features and builds the grid.
orbx_status frame_assign(orbx_frame& F, const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n,
                         float min_x, float max_x, float min_y, float max_y, bool fromDevice,
                         const CamD* undistort = nullptr) {
    ORBX_REQUIRE(n >= 0, "negative size");
    ORBX_REQUIRE(n == 0 || (kps && desc), "null argument");
    orbx_status s = frame_set_bounds(F, min_x, max_x, min_y, max_y);
    if (s != ORBX_OK) return s;
    F.n = n;
    F.hasUright = uright != nullptr;
    F.hostKps.resize(n);
    int dev = 0;
    ORBX_HIP_TRY(hipGetDevice(&dev));
    if (F.device >= 0 && F.device != dev) {
        // the handle moves to the calling thread's device: its old blocks go back to the cache
        F.kps.release(); F.desc.release(); F.uright.release(); F.cellStart.release(); F.cellIdx.release();
    }
    F.device = dev;
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    ORBX_HIP_TRY(F.cellStart.ensure((size_t)(kGridCells + 1) * 4));
    ORBX_HIP_TRY(F.cellIdx.ensure((size_t)std::max(n, 1) * 4));
    ORBX_HIP_TRY(F.kps.ensure((size_t)std::max(n, 1) * sizeof(orbx_keypoint)));
    ORBX_HIP_TRY(F.desc.ensure((size_t)std::max(n, 1) * 32));
    if (uright) ORBX_HIP_TRY(F.uright.ensure((size_t)std::max(n, 1) * 4));
    if (n > 0) {
        const hipMemcpyKind kind = fromDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        const void *srcK = kps, *srcD = desc, *srcU = uright;
        if (!fromDevice) {
            // host arrays go through the thread's pinned arena: an upload from pageable memory is staged by the
            // runtime at 10-15 us per command, one from pinned memory is enqueued in 2-3
            const size_t oD = align256((size_t)n * sizeof(orbx_keypoint)), oU = align256(oD + (size_t)n * 32);
            ORBX_HIP_TRY(t_arena.ensure(oU + (size_t)n * 4));
            memcpy(t_arena.host, kps, (size_t)n * sizeof(orbx_keypoint));
            memcpy(t_arena.host + oD, desc, (size_t)n * 32);
            if (uright) memcpy(t_arena.host + oU, uright, (size_t)n * 4);
            srcK = t_arena.host; srcD = t_arena.host + oD; srcU = t_arena.host + oU;
        }
        ORBX_HIP_TRY(hipMemcpyAsync(F.kps.p, srcK, (size_t)n * sizeof(orbx_keypoint), kind, st));
        ORBX_HIP_TRY(hipMemcpyAsync(F.desc.p, srcD, (size_t)n * 32, kind, st));
        if (uright) ORBX_HIP_TRY(hipMemcpyAsync(F.uright.p, srcU, (size_t)n * 4, kind, st));
        F.hostUright.resize(uright ? (size_t)n : 0);
        if (uright && fromDevice) ORBX_HIP_TRY(hipMemcpyAsync(F.hostUright.data(), uright, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        else if (uright) memcpy(F.hostUright.data(), uright, (size_t)n * 4);
        if (undistort) {
            // Frame::UndistortKeyPoints: the frame keeps mvKeysUn; the host mirror is read back
            hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, st, *undistort,
                               F.kps.as<orbx_keypoint>(), n, F.kps.as<orbx_keypoint>());
            ORBX_HIP_TRY(hipMemcpyAsync(F.hostKps.data(), F.kps.p, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
        } else if (fromDevice) {
            ORBX_HIP_TRY(hipMemcpyAsync(F.hostKps.data(), kps, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
        } else {
            memcpy(F.hostKps.data(), kps, (size_t)n * sizeof(orbx_keypoint));
        }
    }
    hipLaunchKernelGGL(k_assign_grid, dim3(1), dim3(kGridThreads), 0, st, F.kps.as<orbx_keypoint>(), n, F.minX, F.minY,
                       F.invW, F.invH, F.cellStart.as<int>(), F.cellIdx.as<int>());
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(hipStreamSynchronize(st));
    return ORBX_OK;
}


// GetFeaturesInArea for every query + the Hamming distance of every returned keypoint.
orbx_status frame_window_lists(const orbx_frame& F, const orbx_query* queries, const uint8_t* query_desc, int n_query,
                               WindowLists& L, const float* inv_sigma2 = nullptr, int n_levels = 0) {
    ORBX_REQUIRE(n_query >= 0, "negative size");
    if (n_query == 0 || F.n == 0) { L.set_empty(n_query); return ORBX_OK; }
    ORBX_REQUIRE(queries && query_desc, "null argument");
    int dev = 0;
    ORBX_HIP_TRY(hipGetDevice(&dev));
    if (dev != F.device) ORBX_HIP_TRY(hipSetDevice(F.device));
    if (inv_sigma2) {
        ORBX_REQUIRE(n_levels > 0 && n_levels <= 64, "inv_level_sigma2 needs 1..64 levels");
        for (int i = 0; i < F.n; ++i)
            ORBX_REQUIRE(F.hostKps[i].octave >= 0 && F.hostKps[i].octave < n_levels,
                         "keypoint %d: octave %d outside the %d levels", i, F.hostKps[i].octave, n_levels);
    }
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    const size_t oQ = 0, oQd = align256(oQ + (size_t)n_query * sizeof(orbx_query));
    const size_t oSig = align256(oQd + (size_t)n_query * 32), oOff = oSig + 256;
    const size_t oCand = align256(oOff + (size_t)(n_query + 1) * 4);
    DevBuf dCount, dOff;
    ORBX_HIP_TRY(dCount.alloc((size_t)n_query * 4));
    ORBX_HIP_TRY(dOff.alloc((size_t)(n_query + 1) * 4));
    OrbxGridDev g;
    g.cellStart = F.cellStart.p ? static_cast<const int*>(F.cellStart.p) : nullptr;
    g.cellIdx = static_cast<const int*>(F.cellIdx.p);
    g.minX = F.minX; g.minY = F.minY; g.invW = F.invW; g.invH = F.invH;
    const orbx_keypoint* dKps = static_cast<const orbx_keypoint*>(F.kps.p);
    const uint32_t* dDesc = static_cast<const uint32_t*>(F.desc.p);
    const float* dUr = F.hasUright ? static_cast<const float*>(F.uright.p) : nullptr;
    const dim3 grid((n_query + 3) / 4), block(256);
    size_t want = oCand + (size_t)n_query * 32 * 8;   // first guess: 32 candidates per query
    for (int attempt = 0; attempt < 2; ++attempt) {
        ORBX_HIP_TRY(t_arena.ensure(want));
        char* h = t_arena.host;
        char* d = t_arena.dev;
        const int cap = (int)std::min<size_t>((t_arena.cap - oCand) / 8, (size_t)INT_MAX / 2);
        memcpy(h + oQ, queries, (size_t)n_query * sizeof(orbx_query));
        memcpy(h + oQd, query_desc, (size_t)n_query * 32);
        if (inv_sigma2) memcpy(h + oSig, inv_sigma2, (size_t)n_levels * 4);
        const orbx_query* dQ = reinterpret_cast<const orbx_query*>(d + oQ);
        const uint32_t* dQd = reinterpret_cast<const uint32_t*>(d + oQd);
        const float* dSig = inv_sigma2 ? reinterpret_cast<const float*>(d + oSig) : nullptr;
        int* dCi = reinterpret_cast<int*>(d + oCand);
        int* dCd = dCi + cap;
        hipLaunchKernelGGL(k_window, grid, block, 0, st, g, dKps, dDesc, dUr, dQ, dQd, n_query, 0, dCount.as<int>(),
                           nullptr, nullptr, nullptr, 0, dSig);
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(kGridThreads), 0, st, dCount.as<int>(), n_query,
                           dOff.as<int>(), reinterpret_cast<int*>(d + oOff));
        hipLaunchKernelGGL(k_window, grid, block, 0, st, g, dKps, dDesc, dUr, dQ, dQd, n_query, 1, nullptr,
                           dOff.as<int>(), dCi, dCd, cap, dSig);
        ORBX_HIP_TRY(hipGetLastError());
        ORBX_HIP_TRY(hipStreamSynchronize(st));
        const int* off = reinterpret_cast<const int*>(h + oOff);
        const int total = off[n_query];
        if (total <= cap) {
            L.off = off;
            L.idx = reinterpret_cast<const int*>(h + oCand);
            L.dist = L.idx + cap;
            return ORBX_OK;
        }
        want = oCand + (size_t)total * 8;   // the lists did not fit: once more with room for all of them
    }
    orbx_set_error("candidate lists did not fit after growing the staging arena");
    return ORBX_ERR_CAPACITY;
}

thread_local orbx_frame* t_scratchFrame[2] = {nullptr, nullptr};

// The searches that take plain host arrays build a frame of the calling thread for the duration of the call.
orbx_status scratch_frame(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps, float min_x,
                          float max_x, float min_y, float max_y, orbx_frame** out, int slot = 0) {
    if (!t_scratchFrame[slot]) t_scratchFrame[slot] = new orbx_frame;
    *out = t_scratchFrame[slot];
    return frame_assign(**out, kps, desc, uright, n_kps, min_x, max_x, min_y, max_y, false);
}

orbx_status copy_lists_out(const WindowLists& L, int n_query, int* cand_offset, int* cand_idx, int* cand_dist,
                           int capacity, int* n_cand_out) {
    memcpy(cand_offset, L.off, ((size_t)n_query + 1) * sizeof(int));
    const int total = L.off[n_query];
    *n_cand_out = total;
    if (total > capacity) { orbx_set_error("%d candidates, capacity %d", total, capacity); return ORBX_ERR_CAPACITY; }
    if (total > 0) {
        ORBX_REQUIRE(cand_idx && cand_dist, "null argument");
        memcpy(cand_idx, L.idx, (size_t)total * sizeof(int));
        memcpy(cand_dist, L.dist, (size_t)total * sizeof(int));
    }
    return ORBX_OK;
}
}  // namespace

static void release_thread_staging() {
    for (auto*& f : t_scratchFrame) { delete f; f = nullptr; }
    t_arena.release();
}

extern "C" orbx_status orbx_frame_create(orbx_frame** out) {
    ORBX_REQUIRE(out, "null argument");
    *out = new orbx_frame;
    return ORBX_OK;
}

extern "C" void orbx_frame_destroy(orbx_frame* f) { delete f; }

extern "C" orbx_status orbx_frame_assign(orbx_frame* f, const orbx_keypoint* kps, const uint8_t* desc, const float* uright,
                                         int n, float min_x, float max_x, float min_y, float max_y) {
    ORBX_REQUIRE(f, "null frame");
    return frame_assign(*f, kps, desc, uright, n, min_x, max_x, min_y, max_y, false);
}

extern "C" orbx_status orbx_frame_assign_device(orbx_frame* f, const orbx_keypoint* d_kps, const uint8_t* d_desc,
                                                const float* d_uright, int n, float min_x, float max_x, float min_y,
                                                float max_y) {
    ORBX_REQUIRE(f, "null frame");
    return frame_assign(*f, d_kps, d_desc, d_uright, n, min_x, max_x, min_y, max_y, true);
}

static orbx_status frame_assign_camera(orbx_frame* f, const orbx_camera* cam, int width, int height,
                                       const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n,
                                       bool fromDevice) {
    ORBX_REQUIRE(f, "null frame");
    CamD c;
    bool distorted;
    orbx_status s = camera_to_double(cam, c, distorted);
    if (s != ORBX_OK) return s;
    float b[4];
    s = orbx_image_bounds(cam, width, height, b);
    if (s != ORBX_OK) return s;
    return frame_assign(*f, kps, desc, uright, n, b[0], b[1], b[2], b[3], fromDevice, distorted ? &c : nullptr);
}

extern "C" orbx_status orbx_frame_assign_distorted(orbx_frame* f, const orbx_camera* cam, int width, int height,
                                                   const orbx_keypoint* kps, const uint8_t* desc, const float* uright,
                                                   int n) {
    return frame_assign_camera(f, cam, width, height, kps, desc, uright, n, false);
}

extern "C" orbx_status orbx_frame_assign_distorted_device(orbx_frame* f, const orbx_camera* cam, int width, int height,
                                                          const orbx_keypoint* d_kps, const uint8_t* d_desc,
                                                          const float* d_uright, int n) {
    return frame_assign_camera(f, cam, width, height, d_kps, d_desc, d_uright, n, true);
}

extern "C" orbx_status orbx_frame_keypoints(const orbx_frame* f, orbx_keypoint* kps_un, int capacity, float bounds[4]) {
    ORBX_REQUIRE(f, "null frame");
    ORBX_REQUIRE(f->n <= capacity, "%d keypoints, capacity %d", f->n, capacity);
    if (f->n > 0) { ORBX_REQUIRE(kps_un, "null argument"); memcpy(kps_un, f->hostKps.data(), (size_t)f->n * sizeof(orbx_keypoint)); }
    if (bounds) { bounds[0] = f->minX; bounds[1] = f->maxX; bounds[2] = f->minY; bounds[3] = f->maxY; }
    return ORBX_OK;
}

extern "C" int orbx_frame_size(const orbx_frame* f) { return f ? f->n : 0; }

extern "C" orbx_status orbx_frame_grid(const orbx_frame* f, int* cell_start, int* cell_idx, int capacity,
                                       int* n_in_grid) {
    ORBX_REQUIRE(f && cell_start && n_in_grid, "null argument");
    ORBX_REQUIRE(f->cellStart.p, "frame has no features yet");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    ORBX_HIP_TRY(copy_sync(cell_start, f->cellStart.p, (size_t)(kGridCells + 1) * 4, hipMemcpyDeviceToHost, st));
    *n_in_grid = cell_start[kGridCells];
    if (*n_in_grid > capacity) { orbx_set_error("%d keypoints in the grid, capacity %d", *n_in_grid, capacity); return ORBX_ERR_CAPACITY; }
    if (*n_in_grid > 0) {
        ORBX_REQUIRE(cell_idx, "null argument");
        ORBX_HIP_TRY(copy_sync(cell_idx, f->cellIdx.p, (size_t)*n_in_grid * 4, hipMemcpyDeviceToHost, st));
    }
    return ORBX_OK;
}

extern "C" orbx_status orbx_frame_window_candidates(const orbx_frame* f, const orbx_query* queries,
                                                    const uint8_t* query_desc, int n_query, int* cand_offset,
                                                    int* cand_idx, int* cand_dist, int capacity, int* n_cand_out) {
    ORBX_REQUIRE(f && cand_offset && n_cand_out, "null argument");
    WindowLists L;
    orbx_status s = frame_window_lists(*f, queries, query_desc, n_query, L);
    if (s != ORBX_OK) return s;
    return copy_lists_out(L, n_query, cand_offset, cand_idx, cand_dist, capacity, n_cand_out);
}

extern "C" orbx_status orbx_window_candidates(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                                              float min_x, float max_x, float min_y, float max_y,
                                              const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                              int* cand_offset, int* cand_idx, int* cand_dist, int capacity,
                                              int* n_cand_out) {
    orbx_frame* F;
    orbx_status s = scratch_frame(kps, desc, uright, n_kps, min_x, max_x, min_y, max_y, &F);
    if (s != ORBX_OK) return s;
    return orbx_frame_window_candidates(F, queries, query_desc, n_query, cand_offset, cand_idx, cand_dist, capacity,
                                        n_cand_out);
}

extern "C" orbx_status orbx_frame_search_by_projection(const orbx_frame* f, const orbx_query* queries,
                                                       const uint8_t* query_desc, int n_query, float nn_ratio,
                                                       uint8_t* taken, int* match_out, int* n_matches) {
    ORBX_REQUIRE(f && n_matches && (n_query == 0 || match_out) && (f->n == 0 || taken), "null argument");
    const orbx_keypoint* kps = f->hostKps.data();
    WindowLists L;
    orbx_status s = frame_window_lists(*f, queries, query_desc, n_query, L);
    if (s != ORBX_OK) return s;
    const int *off = L.off, *idx = L.idx, *dist = L.dist;
    // Greedy resolve in query order (ORBmatcher.cc:84-141): a keypoint matched by an
    // earlier map point is skipped by the later ones.
    int nm = 0;
    for (int q = 0; q < n_query; ++q) {
        match_out[q] = -1;
        int best = 256, bestLevel = -1, best2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = off[q]; c < off[q + 1]; ++c) {
            const int i = idx[c];
            if (taken[i]) continue;
            const int d = dist[c];
            if (d < best) { best2 = best; best = d; bestLevel2 = bestLevel; bestLevel = kps[i].octave; bestIdx = i; }
            else if (d < best2) { bestLevel2 = kps[i].octave; best2 = d; }
        }
        if (bestIdx < 0) continue;
        if (best <= 100) {
            if (bestLevel == bestLevel2 && best > nn_ratio * best2) continue;
            if (bestLevel != bestLevel2 || best <= nn_ratio * best2) {
                match_out[q] = bestIdx;
                taken[bestIdx] = 1;
                ++nm;
            }
        }
    }
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_by_projection(const orbx_keypoint* kps, const uint8_t* desc, const float* uright,
                                                 int n_kps, float min_x, float max_x, float min_y, float max_y,
                                                 const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                                 float nn_ratio, uint8_t* taken, int* match_out, int* n_matches) {
    orbx_frame* F;
    orbx_status s = scratch_frame(kps, desc, uright, n_kps, min_x, max_x, min_y, max_y, &F);
    if (s != ORBX_OK) return s;
    return orbx_frame_search_by_projection(F, queries, query_desc, n_query, nn_ratio, taken, match_out, n_matches);
}

// ======================================================================================
// Bucketed all-pairs best match (SearchForTriangulation's distance core)
// ======================================================================================
__global__ __launch_bounds__(256) void k_bucket_best(const uint32_t* __restrict__ d1, const int* __restrict__ idx1,
                                                     const int* __restrict__ off1, const uint32_t* __restrict__ d2,
                                                     const int* __restrict__ idx2, const int* __restrict__ off2,
                                                     const int* __restrict__ bucketOf, int n1, int* __restrict__ bestIdx2,
                                                     int* __restrict__ bestDist) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n1) return;
    const int b = bucketOf[i];
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = d1[8ll * idx1[i] + k];
    // smallest distance <= 50; among equals the LAST one in bucket order
    unsigned int key = 0xFFFFFFFFu;
    const int jb = off2[b], je = off2[b + 1];
    for (int j = jb + lane; j < je; j += 64) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(a[k] ^ d2[8ll * idx2[j] + k]);
        if (d <= 50) key = min(key, ((unsigned int)d << 22) | (unsigned int)(0x3FFFFF - (j - jb)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) key = min(key, (unsigned int)__shfl_xor((int)key, off));
    if (lane == 0) {
        if (key == 0xFFFFFFFFu) { bestIdx2[i] = -1; bestDist[i] = 50; }
        else { bestDist[i] = (int)(key >> 22); bestIdx2[i] = idx2[jb + (0x3FFFFF - (int)(key & 0x3FFFFF))]; }
    }
}

extern "C" orbx_status orbx_bucket_best(const uint8_t* desc1, int n_desc1, const int* idx1, const int* off1,
                                        const uint8_t* desc2, int n_desc2, const int* idx2, const int* off2, int n_buckets,
                                        int* best_idx2, int* best_dist) {
    ORBX_REQUIRE(n_buckets >= 0 && n_desc1 >= 0 && n_desc2 >= 0, "negative size");
    if (n_buckets == 0) return ORBX_OK;
    ORBX_REQUIRE(off1 && off2, "null argument");
    const int n1 = off1[n_buckets], n2 = off2[n_buckets];
    if (n1 == 0) return ORBX_OK;
    ORBX_REQUIRE(desc1 && idx1 && best_idx2 && best_dist && (n2 == 0 || (desc2 && idx2)), "null argument");
    std::vector<int> bucketOf(n1);
    for (int b = 0; b < n_buckets; ++b) {
        ORBX_REQUIRE(off1[b + 1] >= off1[b] && off2[b + 1] >= off2[b] && off2[b + 1] - off2[b] < 0x3FFFFF, "bad bucket offsets");
        for (int i = off1[b]; i < off1[b + 1]; ++i) bucketOf[i] = b;
    }
    for (int i = 0; i < n1; ++i) ORBX_REQUIRE(idx1[i] >= 0 && idx1[i] < n_desc1, "idx1 out of range");
    for (int j = 0; j < n2; ++j) ORBX_REQUIRE(idx2[j] >= 0 && idx2[j] < n_desc2, "idx2 out of range");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    PackedIo io;
    const size_t oD1 = io.add(desc1, (size_t)n_desc1 * 32), oD2 = io.add(desc2, (size_t)n_desc2 * 32);
    const size_t oI1 = io.add(idx1, (size_t)n1 * 4), oI2 = io.add(idx2, (size_t)n2 * 4);
    const size_t oO1 = io.add(off1, (size_t)(n_buckets + 1) * 4), oO2 = io.add(off2, (size_t)(n_buckets + 1) * 4);
    const size_t oBO = io.add(bucketOf.data(), (size_t)n1 * 4);
    const size_t oBI = io.want((size_t)n1 * 4), oBD = io.want((size_t)n1 * 4);
    ORBX_HIP_TRY(io.upload(st));
    hipLaunchKernelGGL(k_bucket_best, dim3((n1 + 3) / 4), dim3(256), 0, st, io.dev_in<uint32_t>(oD1), io.dev_in<int>(oI1),
                       io.dev_in<int>(oO1), io.dev_in<uint32_t>(oD2), io.dev_in<int>(oI2), io.dev_in<int>(oO2),
                       io.dev_in<int>(oBO), n1, io.dev_out<int>(oBI), io.dev_out<int>(oBD));
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(io.download(st));
    io.fetch(best_idx2, oBI, (size_t)n1 * 4);
    io.fetch(best_dist, oBD, (size_t)n1 * 4);
    return ORBX_OK;
}

// ======================================================================================
// Host-side replay of the matcher loops over the device-built candidate lists.
// The loops carry state from one query to the next ("this keypoint already has a map
// point"), so they run in the reference's order on the host; every Hamming distance and
// every stateless gate was evaluated on the device.
// ======================================================================================
namespace {
constexpr int kHistoLength = 30;   // HISTO_LENGTH (ORBmatcher.cc:36)

// Rotation-consistency histogram (ORBmatcher.cc:1869-1876 and its twins, :2096-2137).
struct RotHist {
    std::vector<int> bins[kHistoLength];
    float factor;
    explicit RotHist(float f) : factor(f) {}
    bool add(float angle1, float angle2, int payload) {
        float rot = angle1 - angle2;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == kHistoLength) bin = 0;
        if (bin < 0 || bin >= kHistoLength) return false;   // the reference asserts
        bins[bin].push_back(payload);
        return true;
    }
    // ComputeThreeMaxima
    void three_maxima(int& ind1, int& ind2, int& ind3) const {
        int max1 = 0, max2 = 0, max3 = 0;
        ind1 = ind2 = ind3 = -1;
        for (int i = 0; i < kHistoLength; ++i) {
            const int s = (int)bins[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    }
    template <class F> void for_each_rejected(F f) const {
        int i1, i2, i3;
        three_maxima(i1, i2, i3);
        for (int i = 0; i < kHistoLength; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int v : bins[i]) f(v);
        }
    }
};
}  // namespace

extern "C" orbx_status orbx_frame_search_window_best(const orbx_frame* f, const orbx_query* queries,
                                                     const uint8_t* query_desc, int n_query, float max_dist,
                                                     uint8_t* taken, const uint8_t* query_marks,
                                                     const float* query_angle, float rot_factor, int* match_out,
                                                     int* kp_match, int* n_matches) {
    ORBX_REQUIRE(f && n_matches && (n_query == 0 || match_out) && (f->n == 0 || kp_match), "null argument");
    const orbx_keypoint* kps = f->hostKps.data();
    const int n_kps = f->n;
    WindowLists L;
    orbx_status s = frame_window_lists(*f, queries, query_desc, n_query, L);
    if (s != ORBX_OK) return s;
    const int *off = L.off, *idx = L.idx, *dist = L.dist;
    for (int i = 0; i < n_kps; ++i) kp_match[i] = -1;
    RotHist hist(rot_factor);
    int nm = 0;
    for (int q = 0; q < n_query; ++q) {
        match_out[q] = -1;
        int best = 256, bestIdx = -1;
        for (int c = off[q]; c < off[q + 1]; ++c) {
            const int i = idx[c];
            if (taken && taken[i]) continue;
            if (dist[c] < best) { best = dist[c]; bestIdx = i; }
        }
        if (bestIdx < 0 || !((float)best <= max_dist)) continue;
        match_out[q] = bestIdx;
        kp_match[bestIdx] = q;
        if (taken) taken[bestIdx] = query_marks ? query_marks[q] : 1;
        ++nm;
        if (query_angle)
            ORBX_REQUIRE(hist.add(query_angle[q], kps[bestIdx].angle, bestIdx), "rotation bin out of range (query %d)", q);
    }
    if (query_angle) hist.for_each_rejected([&](int i) { kp_match[i] = -1; --nm; });
    // a query keeps its keypoint only if it is still the keypoint's owner
    for (int q = 0; q < n_query; ++q)
        if (match_out[q] >= 0 && kp_match[match_out[q]] != q) match_out[q] = -1;
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_window_best(const orbx_keypoint* kps, const uint8_t* desc, const float* uright,
                                               int n_kps, float min_x, float max_x, float min_y, float max_y,
                                               const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                               float max_dist, uint8_t* taken, const uint8_t* query_marks,
                                               const float* query_angle, float rot_factor, int* match_out,
                                               int* kp_match, int* n_matches) {
    orbx_frame* F;
    orbx_status s = scratch_frame(kps, desc, uright, n_kps, min_x, max_x, min_y, max_y, &F);
    if (s != ORBX_OK) return s;
    return orbx_frame_search_window_best(F, queries, query_desc, n_query, max_dist, taken, query_marks, query_angle,
                                         rot_factor, match_out, kp_match, n_matches);
}

// ---- two-camera frames (Frame::Nleft != -1) -------------------------------------------------------------
namespace {
// Candidate lists copied out of the calling thread's arena (the next search on the thread reuses it).
struct OwnedLists {
    std::vector<int> off, idx, dist;
    orbx_status fill(const orbx_frame* f, const orbx_query* queries, const uint8_t* query_desc, int n_query) {
        if (!f || f->n == 0) { off.assign((size_t)n_query + 1, 0); idx.clear(); dist.clear(); return ORBX_OK; }
        WindowLists L;
        orbx_status s = frame_window_lists(*f, queries, query_desc, n_query, L);
        if (s != ORBX_OK) return s;
        off.assign(L.off, L.off + n_query + 1);
        const int total = L.off[n_query];
        idx.assign(L.idx, L.idx + total);
        dist.assign(L.dist, L.dist + total);
        return ORBX_OK;
    }
};
}  // namespace

extern "C" orbx_status orbx_frame_search_by_projection_stereo(const orbx_frame* left, const orbx_frame* right,
                                                              const orbx_query* queries_left,
                                                              const orbx_query* queries_right, const uint8_t* query_desc,
                                                              int n_query, float nn_ratio, const int* left_to_right,
                                                              const int* right_to_left, uint8_t* taken,
                                                              const uint8_t* query_marks, int* kp_query, int* n_matches) {
    ORBX_REQUIRE(left && n_matches && n_query >= 0, "null argument");
    const int nL = left->n, nR = right ? right->n : 0;
    ORBX_REQUIRE(nL + nR == 0 || (taken && kp_query), "null argument");
    ORBX_REQUIRE(n_query == 0 || (queries_left && query_desc && (nR == 0 || queries_right)), "null argument");
    OwnedLists LL, LR;
    orbx_status s = LL.fill(left, queries_left, query_desc, n_query);
    if (s == ORBX_OK && nR > 0) s = LR.fill(right, queries_right, query_desc, n_query);
    if (s != ORBX_OK) return s;
    const orbx_keypoint* kL = left->hostKps.data();
    const orbx_keypoint* kR = nR ? right->hostKps.data() : nullptr;
    for (int i = 0; i < nL + nR; ++i) kp_query[i] = -1;
    int nm = 0;
    for (int q = 0; q < n_query; ++q) {
        const uint8_t mark = query_marks ? query_marks[q] : 1;
        if (queries_left[q].radius > 0 && LL.off[q + 1] > LL.off[q]) {            // ORBmatcher.cc:60-144
            int best = 256, bestLevel = -1, best2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int c = LL.off[q]; c < LL.off[q + 1]; ++c) {
                const int i = LL.idx[c];
                if (taken[i]) continue;
                const int d = LL.dist[c];
                if (d < best) { best2 = best; best = d; bestLevel2 = bestLevel; bestLevel = kL[i].octave; bestIdx = i; }
                else if (d < best2) { bestLevel2 = kL[i].octave; best2 = d; }
            }
            // :123-126: a left best within TH_HIGH that fails the ratio test ends this map point -- the reference's
            // `continue` sits in the map-point loop, so the right-camera pass below is skipped as well
            if (best <= 100 && bestLevel == bestLevel2 && best > nn_ratio * best2) continue;
            if (best <= 100) {
                kp_query[bestIdx] = q; taken[bestIdx] = mark;
                if (nR > 0 && left_to_right && left_to_right[bestIdx] != -1) {       // the stereo observation in the right camera
                    const int r = left_to_right[bestIdx];
                    ORBX_REQUIRE(r >= 0 && r < nR, "left_to_right[%d] out of range", bestIdx);
                    kp_query[r + nL] = q; taken[r + nL] = mark;
                    ++nm;
                }
                ++nm;
            }
        }
        if (nR > 0 && queries_right[q].radius > 0 && LR.off[q + 1] > LR.off[q]) {  // :146-213
            int best = 256, bestLevel = -1, best2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int c = LR.off[q]; c < LR.off[q + 1]; ++c) {
                const int i = LR.idx[c];
                if (taken[i + nL]) continue;
                const int d = LR.dist[c];
                if (d < best) { best2 = best; best = d; bestLevel2 = bestLevel; bestLevel = kR[i].octave; bestIdx = i; }
                else if (d < best2) { bestLevel2 = kR[i].octave; best2 = d; }
            }
            if (best <= 100 && !(bestLevel == bestLevel2 && best > nn_ratio * best2)) {
                if (right_to_left && right_to_left[bestIdx] != -1) {
                    const int l = right_to_left[bestIdx];
                    ORBX_REQUIRE(l >= 0 && l < nL, "right_to_left[%d] out of range", bestIdx);
                    kp_query[l] = q; taken[l] = mark;
                    ++nm;
                }
                kp_query[bestIdx + nL] = q; taken[bestIdx + nL] = mark;
                ++nm;
            }
        }
    }
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_frame_search_window_best_stereo(const orbx_frame* left, const orbx_frame* right,
                                                            const orbx_query* queries_left, const orbx_query* queries_right,
                                                            const uint8_t* query_desc, int n_query, float max_dist,
                                                            uint8_t* taken, const uint8_t* query_marks,
                                                            const float* query_angle, float rot_factor, int* kp_query,
                                                            int* n_matches) {
    ORBX_REQUIRE(left && n_matches && n_query >= 0, "null argument");
    const int nL = left->n, nR = right ? right->n : 0;
    ORBX_REQUIRE(nL + nR == 0 || (taken && kp_query), "null argument");
    ORBX_REQUIRE(n_query == 0 || (queries_left && query_desc && (nR == 0 || queries_right)), "null argument");
    OwnedLists LL, LR;
    orbx_status s = LL.fill(left, queries_left, query_desc, n_query);
    if (s == ORBX_OK && nR > 0) s = LR.fill(right, queries_right, query_desc, n_query);
    if (s != ORBX_OK) return s;
    const orbx_keypoint* kL = left->hostKps.data();
    const orbx_keypoint* kR = nR ? right->hostKps.data() : nullptr;
    for (int i = 0; i < nL + nR; ++i) kp_query[i] = -1;
    RotHist hist(rot_factor);
    int nm = 0;
    for (int q = 0; q < n_query; ++q) {
        if (!(queries_left[q].radius > 0)) continue;                 // not projected into the left image (:1795-1803)
        if (LL.off[q + 1] == LL.off[q]) continue;                    // empty left window: no right pass either (:1820)
        const uint8_t mark = query_marks ? query_marks[q] : 1;
        int best = 256, bestIdx = -1;
        for (int c = LL.off[q]; c < LL.off[q + 1]; ++c) {
            const int i = LL.idx[c];
            if (taken[i]) continue;
            if (LL.dist[c] < best) { best = LL.dist[c]; bestIdx = i; }
        }
        if (bestIdx >= 0 && (float)best <= max_dist) {
            kp_query[bestIdx] = q; taken[bestIdx] = mark;
            ++nm;
            if (query_angle) ORBX_REQUIRE(hist.add(query_angle[q], kL[bestIdx].angle, bestIdx), "rotation bin out of range (query %d)", q);
        }
        if (nR > 0) {                                                 // :1879-1944
            best = 256; bestIdx = -1;
            for (int c = LR.off[q]; c < LR.off[q + 1]; ++c) {
                const int i = LR.idx[c];
                if (taken[i + nL]) continue;
                if (LR.dist[c] < best) { best = LR.dist[c]; bestIdx = i; }
            }
            if (bestIdx >= 0 && (float)best <= max_dist) {
                kp_query[bestIdx + nL] = q; taken[bestIdx + nL] = mark;
                ++nm;
                if (query_angle)
                    ORBX_REQUIRE(hist.add(query_angle[q], kR[bestIdx].angle, bestIdx + nL), "rotation bin out of range (query %d)", q);
            }
        }
    }
    if (query_angle) hist.for_each_rejected([&](int i) { kp_query[i] = -1; --nm; });
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_frame_fuse_search(const orbx_frame* f, const orbx_query* queries, const uint8_t* query_desc,
                                              int n_query, const float* inv_level_sigma2, int n_levels, int* match_out,
                                              int* n_matches) {
    ORBX_REQUIRE(f && n_matches && (n_query == 0 || match_out) && inv_level_sigma2, "null argument");
    WindowLists L;
    orbx_status s = frame_window_lists(*f, queries, query_desc, n_query, L, inv_level_sigma2, n_levels);
    if (s != ORBX_OK) return s;
    const int *off = L.off, *idx = L.idx, *dist = L.dist;
    int nm = 0;
    for (int q = 0; q < n_query; ++q) {
        int best = 256, bestIdx = -1;
        for (int c = off[q]; c < off[q + 1]; ++c)
            if (dist[c] < best) { best = dist[c]; bestIdx = idx[c]; }
        match_out[q] = (bestIdx >= 0 && best <= 50) ? bestIdx : -1;
        nm += match_out[q] >= 0;
    }
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_fuse_search(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                                        float min_x, float max_x, float min_y, float max_y, const orbx_query* queries,
                                        const uint8_t* query_desc, int n_query, const float* inv_level_sigma2,
                                        int n_levels, int* match_out, int* n_matches) {
    orbx_frame* F;
    orbx_status s = scratch_frame(kps, desc, uright, n_kps, min_x, max_x, min_y, max_y, &F);
    if (s != ORBX_OK) return s;
    return orbx_frame_fuse_search(F, queries, query_desc, n_query, inv_level_sigma2, n_levels, match_out, n_matches);
}

extern "C" orbx_status orbx_search_by_sim3(const orbx_keypoint* kps1, const uint8_t* desc1, int n1, const float bounds1[4],
                                           const orbx_keypoint* kps2, const uint8_t* desc2, int n2, const float bounds2[4],
                                           const orbx_query* queries_into2, const uint8_t* query_desc_into2,
                                           const int* source1, int nq1, const orbx_query* queries_into1,
                                           const uint8_t* query_desc_into1, const int* source2, int nq2,
                                           int* matches12, int* n_found) {
    ORBX_REQUIRE(bounds1 && bounds2 && n_found && (n1 == 0 || matches12), "null argument");
    ORBX_REQUIRE((nq1 == 0 || source1) && (nq2 == 0 || source2), "null argument");
    std::vector<int> vnMatch1(n1, -1), vnMatch2(n2, -1);
    WindowLists L;
    const int *off, *idx, *dist;
    orbx_frame* F;
    // KF1's points searched in KF2 (ORBmatcher.cc:1565-1642)
    orbx_status s = scratch_frame(kps2, desc2, nullptr, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3], &F);
    if (s == ORBX_OK) s = frame_window_lists(*F, queries_into2, query_desc_into2, nq1, L);
    if (s != ORBX_OK) return s;
    off = L.off; idx = L.idx; dist = L.dist;
    for (int q = 0; q < nq1; ++q) {
        ORBX_REQUIRE(source1[q] >= 0 && source1[q] < n1, "source1[%d] out of range", q);
        int best = 0x7FFFFFFF, bestIdx = -1;
        for (int c = off[q]; c < off[q + 1]; ++c)
            if (dist[c] < best) { best = dist[c]; bestIdx = idx[c]; }
        if (best <= 100) vnMatch1[source1[q]] = bestIdx;
    }
    // KF2's points searched in KF1 (:1645-1722)
    s = scratch_frame(kps1, desc1, nullptr, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3], &F);
    if (s == ORBX_OK) s = frame_window_lists(*F, queries_into1, query_desc_into1, nq2, L);
    if (s != ORBX_OK) return s;
    off = L.off; idx = L.idx; dist = L.dist;
    for (int q = 0; q < nq2; ++q) {
        ORBX_REQUIRE(source2[q] >= 0 && source2[q] < n2, "source2[%d] out of range", q);
        int best = 0x7FFFFFFF, bestIdx = -1;
        for (int c = off[q]; c < off[q + 1]; ++c)
            if (dist[c] < best) { best = dist[c]; bestIdx = idx[c]; }
        if (best <= 100) vnMatch2[source2[q]] = bestIdx;
    }
    // agreement (:1725-1740)
    int nf = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; ++nf; }
    }
    *n_found = nf;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_for_initialization(const orbx_keypoint* kps1, const uint8_t* desc1, int n1,
                                                      const orbx_keypoint* kps2, const uint8_t* desc2, int n2,
                                                      float min_x, float max_x, float min_y, float max_y,
                                                      float* prev_matched, int window_size, float nn_ratio,
                                                      int check_orientation, int* matches12, int* n_matches) {
    ORBX_REQUIRE(n_matches && (n1 == 0 || (kps1 && desc1 && prev_matched && matches12)), "null argument");
    // queries: level-0 keypoints of the first frame around their previous match (ORBmatcher.cc:708-715)
    std::vector<orbx_query> qs;
    std::vector<uint8_t> qd;
    std::vector<int> src;
    for (int i1 = 0; i1 < n1; ++i1) {
        matches12[i1] = -1;
        if (kps1[i1].octave > 0) continue;
        orbx_query q;
        q.x = prev_matched[2 * i1]; q.y = prev_matched[2 * i1 + 1]; q.xr = 0.f;
        q.radius = (float)window_size;
        q.min_level = 0; q.max_level = 0;
        qs.push_back(q);
        qd.insert(qd.end(), desc1 + 32ll * i1, desc1 + 32ll * i1 + 32);
        src.push_back(i1);
    }
    WindowLists L;
    orbx_frame* F;
    orbx_status s = scratch_frame(kps2, desc2, nullptr, n2, min_x, max_x, min_y, max_y, &F);
    if (s == ORBX_OK) s = frame_window_lists(*F, qs.data(), qd.data(), (int)qs.size(), L);
    if (s != ORBX_OK) return s;
    const int *off = L.off, *idx = L.idx, *dist = L.dist;
    std::vector<int> matchedDistance(n2, 0x7FFFFFFF), matches21(n2, -1);
    RotHist hist(1.0f / kHistoLength);
    int nm = 0;
    for (int q = 0; q < (int)qs.size(); ++q) {
        const int i1 = src[q];
        int best = 0x7FFFFFFF, best2 = 0x7FFFFFFF, bestIdx2 = -1;
        for (int c = off[q]; c < off[q + 1]; ++c) {
            const int i2 = idx[c], d = dist[c];
            if (matchedDistance[i2] <= d) continue;
            if (d < best) { best2 = best; best = d; bestIdx2 = i2; }
            else if (d < best2) best2 = d;
        }
        if (best <= 50 && (float)best < (float)best2 * nn_ratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; --nm; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = best;
            ++nm;
            if (check_orientation)
                ORBX_REQUIRE(hist.add(kps1[i1].angle, kps2[bestIdx2].angle, i1), "rotation bin out of range (keypoint %d)", i1);
        }
    }
    if (check_orientation)
        hist.for_each_rejected([&](int i1) { if (matches12[i1] >= 0) { matches12[i1] = -1; --nm; } });
    for (int i1 = 0; i1 < n1; ++i1)
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[matches12[i1]].x; prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y; }
    *n_matches = nm;
    return ORBX_OK;
}

// ======================================================================================
// Bucketed all-pairs distances (the inner loops of SearchByBoW / SearchForTriangulation)
// ======================================================================================
// One wave per idx1 entry; row i of the output holds the distances to the idx2 entries of
// the same bucket, in bucket order: out[rowStart[i] + j].
__global__ __launch_bounds__(256) void k_bucket_allpairs(const uint32_t* __restrict__ d1, const int* __restrict__ idx1,
                                                         const uint32_t* __restrict__ d2, const int* __restrict__ idx2,
                                                         const int* __restrict__ off2, const int* __restrict__ bucketOf,
                                                         const int* __restrict__ rowStart, int n1,
                                                         uint16_t* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n1) return;
    const int b = bucketOf[i];
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = d1[8ll * idx1[i] + k];
    const int jb = off2[b], je = off2[b + 1];
    uint16_t* row = out + rowStart[i];
    for (int j = jb + lane; j < je; j += 64) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(a[k] ^ d2[8ll * idx2[j] + k]);
        row[j - jb] = (uint16_t)d;
    }
}

namespace {
struct BucketPairs {
    std::vector<int> bucketOf;     // per idx1 entry
    std::vector<int> rowStart;     // per idx1 entry (+1)
    const uint16_t* dist = nullptr;   // rows of distances, in the calling thread's pinned arena (valid until its next search)
    size_t nPairs = 0;
};
}  // namespace

// Distances of all bucket pairs with both descriptor sets already on the device: the five index arrays go in and the
// distance rows come out through the calling thread's mapped pinned arena, so a search is one launch and one
// synchronisation, without copy commands.
static orbx_status bucket_pairs_dev(const uint32_t* dDesc1, int n_desc1, const int* idx1, const int* off1,
                                    const uint32_t* dDesc2, int n_desc2, const int* idx2, const int* off2, int n_buckets,
                                    BucketPairs& bp) {
    ORBX_REQUIRE(n_buckets >= 0 && n_desc1 >= 0 && n_desc2 >= 0, "negative size");
    bp.bucketOf.clear(); bp.rowStart.assign(1, 0); bp.dist = nullptr; bp.nPairs = 0;
    if (n_buckets == 0) return ORBX_OK;
    ORBX_REQUIRE(off1 && off2, "null argument");
    const int n1 = off1[n_buckets], n2 = off2[n_buckets];
    ORBX_REQUIRE(off1[0] == 0 && off2[0] == 0, "bucket offsets must start at 0");
    bp.bucketOf.resize(n1);
    bp.rowStart.resize(n1 + 1);
    long long total = 0;
    for (int b = 0; b < n_buckets; ++b) {
        ORBX_REQUIRE(off1[b + 1] >= off1[b] && off2[b + 1] >= off2[b], "bad bucket offsets");
        for (int i = off1[b]; i < off1[b + 1]; ++i) {
            bp.bucketOf[i] = b;
            bp.rowStart[i] = (int)total;
            total += off2[b + 1] - off2[b];
            ORBX_REQUIRE(total < 0x7FFFFFFF, "too many bucket pairs");
        }
    }
    bp.rowStart[n1] = (int)total;
    bp.nPairs = (size_t)total;
    if (n1 == 0 || total == 0) return ORBX_OK;
    ORBX_REQUIRE(dDesc1 && idx1 && dDesc2 && idx2, "null argument");
    for (int i = 0; i < n1; ++i) ORBX_REQUIRE(idx1[i] >= 0 && idx1[i] < n_desc1, "idx1 out of range");
    for (int j = 0; j < n2; ++j) ORBX_REQUIRE(idx2[j] >= 0 && idx2[j] < n_desc2, "idx2 out of range");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    const size_t oI1 = 0, oI2 = align256(oI1 + (size_t)n1 * 4), oO2 = align256(oI2 + (size_t)n2 * 4);
    const size_t oBO = align256(oO2 + (size_t)(n_buckets + 1) * 4), oRS = align256(oBO + (size_t)n1 * 4);
    const size_t oOut = align256(oRS + (size_t)(n1 + 1) * 4);
    ORBX_HIP_TRY(t_arena.ensure(oOut + (size_t)total * 2));
    char* h = t_arena.host;
    char* d = t_arena.dev;
    memcpy(h + oI1, idx1, (size_t)n1 * 4);
    memcpy(h + oI2, idx2, (size_t)n2 * 4);
    memcpy(h + oO2, off2, (size_t)(n_buckets + 1) * 4);
    memcpy(h + oBO, bp.bucketOf.data(), (size_t)n1 * 4);
    memcpy(h + oRS, bp.rowStart.data(), (size_t)(n1 + 1) * 4);
    hipLaunchKernelGGL(k_bucket_allpairs, dim3((n1 + 3) / 4), dim3(256), 0, st, dDesc1, reinterpret_cast<const int*>(d + oI1),
                       dDesc2, reinterpret_cast<const int*>(d + oI2), reinterpret_cast<const int*>(d + oO2),
                       reinterpret_cast<const int*>(d + oBO), reinterpret_cast<const int*>(d + oRS), n1,
                       reinterpret_cast<uint16_t*>(d + oOut));
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(hipStreamSynchronize(st));
    bp.dist = reinterpret_cast<const uint16_t*>(h + oOut);
    return ORBX_OK;
}

// The same for descriptor sets in host memory: two uploads in front.
static orbx_status bucket_allpairs_impl(const uint8_t* desc1, int n_desc1, const int* idx1, const int* off1,
                                        const uint8_t* desc2, int n_desc2, const int* idx2, const int* off2, int n_buckets,
                                        BucketPairs& bp) {
    ORBX_REQUIRE(n_desc1 >= 0 && n_desc2 >= 0, "negative size");
    ORBX_REQUIRE((n_desc1 == 0 || desc1) && (n_desc2 == 0 || desc2), "null argument");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    DevBuf D1, D2;
    ORBX_HIP_TRY(D1.alloc((size_t)std::max(n_desc1, 1) * 32)); ORBX_HIP_TRY(D2.alloc((size_t)std::max(n_desc2, 1) * 32));
    if (n_desc1) ORBX_HIP_TRY(copy_sync(D1.p, desc1, (size_t)n_desc1 * 32, hipMemcpyHostToDevice, st));
    if (n_desc2) ORBX_HIP_TRY(copy_sync(D2.p, desc2, (size_t)n_desc2 * 32, hipMemcpyHostToDevice, st));
    return bucket_pairs_dev(D1.as<uint32_t>(), n_desc1, idx1, off1, D2.as<uint32_t>(), n_desc2, idx2, off2, n_buckets, bp);
}

// Two frames' descriptor sets as they lie on the device.
static orbx_status bucket_pairs_frames(const orbx_frame* f1, const int* idx1, const int* off1, const orbx_frame* f2,
                                       const int* idx2, const int* off2, int n_buckets, BucketPairs& bp) {
    ORBX_REQUIRE(f1 && f2, "null frame");
    ORBX_REQUIRE(f1->n == 0 || f2->n == 0 || f1->device == f2->device, "the two frames live on different devices");
    int dev = 0;
    ORBX_HIP_TRY(hipGetDevice(&dev));
    const int want = f1->n ? f1->device : f2->device;
    if (want >= 0 && dev != want) ORBX_HIP_TRY(hipSetDevice(want));
    return bucket_pairs_dev(static_cast<const uint32_t*>(f1->desc.p), f1->n, idx1, off1,
                            static_cast<const uint32_t*>(f2->desc.p), f2->n, idx2, off2, n_buckets, bp);
}

static orbx_status search_by_bow_replay(const BucketPairs& bp, const float* angle_kf, const uint8_t* valid_kf, const int* idx_kf,
                                        const int* off_kf, const float* angle_f, int n_f, int n_left_f, const int* idx_f,
                                        const int* off_f, int n_buckets, float nn_ratio, int check_orientation, int* match_f,
                                        int* n_matches);
static orbx_status search_by_bow_keyframes_replay(const BucketPairs& bp, const float* angle1, const uint8_t* valid1, int n1,
                                                  const int* idx1, const int* off1, const float* angle2,
                                                  const uint8_t* valid2, int n2, const int* idx2, const int* off2,
                                                  int n_buckets, float nn_ratio, int check_orientation, int* matches12,
                                                  int* n_matches);
static orbx_status triangulation_check(const orbx_keypoint* kps1, int n1, const orbx_keypoint* kps2, int n2,
                                       const orbx_triangulation_geometry* geom);
static orbx_status triangulation_cameras_replay(const BucketPairs& bp, const orbx_keypoint* kps1, const uint8_t* free1,
                                                const float* uright1, int n1, const int* idx1, const int* off1,
                                                const orbx_keypoint* kps2, const uint8_t* free2, const float* uright2,
                                                const int* idx2, const int* off2, int n_buckets,
                                                const orbx_triangulation_geometry* geom, int only_stereo, int coarse,
                                                int check_orientation, int* matches12, int* n_matches);

extern "C" orbx_status orbx_bucket_allpairs(const uint8_t* desc1, int n_desc1, const int* idx1, const int* off1,
                                            const uint8_t* desc2, int n_desc2, const int* idx2, const int* off2,
                                            int n_buckets, int* row_start, uint16_t* dist, size_t capacity,
                                            size_t* n_pairs_out) {
    ORBX_REQUIRE(n_pairs_out, "null argument");
    BucketPairs bp;
    orbx_status s = bucket_allpairs_impl(desc1, n_desc1, idx1, off1, desc2, n_desc2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    *n_pairs_out = bp.nPairs;
    if (n_buckets == 0) return ORBX_OK;
    ORBX_REQUIRE(row_start, "null argument");
    memcpy(row_start, bp.rowStart.data(), bp.rowStart.size() * sizeof(int));
    if (bp.nPairs > capacity) { orbx_set_error("%zu pairs, capacity %zu", bp.nPairs, capacity); return ORBX_ERR_CAPACITY; }
    if (bp.nPairs && bp.dist) { ORBX_REQUIRE(dist, "null argument"); memcpy(dist, bp.dist, bp.nPairs * 2); }
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_by_bow(const uint8_t* desc_kf, const float* angle_kf, const uint8_t* valid_kf, int n_kf,
                                          const int* idx_kf, const int* off_kf, const uint8_t* desc_f,
                                          const float* angle_f, int n_f, int n_left_f, const int* idx_f, const int* off_f,
                                          int n_buckets, float nn_ratio, int check_orientation, int* match_f,
                                          int* n_matches) {
    ORBX_REQUIRE(n_matches && (n_f == 0 || match_f) && (n_kf == 0 || valid_kf), "null argument");
    ORBX_REQUIRE(!check_orientation || ((n_kf == 0 || angle_kf) && (n_f == 0 || angle_f)), "orientation check needs the angles");
    BucketPairs bp;
    orbx_status s = bucket_allpairs_impl(desc_kf, n_kf, idx_kf, off_kf, desc_f, n_f, idx_f, off_f, n_buckets, bp);
    if (s != ORBX_OK) return s;
    return search_by_bow_replay(bp, angle_kf, valid_kf, idx_kf, off_kf, angle_f, n_f, n_left_f, idx_f, off_f, n_buckets,
                                nn_ratio, check_orientation, match_f, n_matches);
}

// The loop of SearchByBoW(KeyFrame*, Frame&, ...) over the distance rows (ORBmatcher.cc:273-470).
static orbx_status search_by_bow_replay(const BucketPairs& bp, const float* angle_kf, const uint8_t* valid_kf, const int* idx_kf,
                                        const int* off_kf, const float* angle_f, int n_f, int n_left_f, const int* idx_f,
                                        const int* off_f, int n_buckets, float nn_ratio, int check_orientation, int* match_f,
                                        int* n_matches) {
    for (int i = 0; i < n_f; ++i) match_f[i] = -1;
    RotHist hist((float)kHistoLength / 360.0f);   // this fork's factor for this function (ORBmatcher.cc:255)
    int nm = 0;
    for (int b = 0; b < n_buckets; ++b) {
        const int jb = off_f[b], je = off_f[b + 1];
        for (int e = off_kf[b]; e < off_kf[b + 1]; ++e) {
            const int realIdxKF = idx_kf[e];
            if (!valid_kf[realIdxKF]) continue;
            const uint16_t* row = bp.dist + bp.rowStart[e];
            int best1 = 256, bestIdx = -1, best2 = 256, best1R = 256, bestIdxR = -1, best2R = 256;
            for (int j = jb; j < je; ++j) {
                const int realIdxF = idx_f[j];
                if (match_f[realIdxF] >= 0) continue;
                const int d = row[j - jb];
                if (n_left_f == -1 || realIdxF < n_left_f) {
                    if (d < best1) { best2 = best1; best1 = d; bestIdx = realIdxF; }
                    else if (d < best2) best2 = d;
                } else {
                    if (d < best1R) { best2R = best1R; best1R = d; bestIdxR = realIdxF; }
                    else if (d < best2R) best2R = d;
                }
            }
            if (best1 <= 50) {
                if ((float)best1 < nn_ratio * (float)best2) {
                    match_f[bestIdx] = realIdxKF;
                    if (check_orientation)
                        ORBX_REQUIRE(hist.add(angle_kf[realIdxKF], angle_f[bestIdx], bestIdx), "rotation bin out of range");
                    ++nm;
                }
                if (best1R <= 50) {   // right-camera track: the ratio test is disabled in the reference (:404)
                    match_f[bestIdxR] = realIdxKF;
                    if (check_orientation)
                        ORBX_REQUIRE(hist.add(angle_kf[realIdxKF], angle_f[bestIdxR], bestIdxR), "rotation bin out of range");
                    ++nm;
                }
            }
        }
    }
    if (check_orientation) hist.for_each_rejected([&](int i) { match_f[i] = -1; --nm; });
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_by_bow_keyframes(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                                                    const int* idx1, const int* off1, const uint8_t* desc2,
                                                    const float* angle2, const uint8_t* valid2, int n2, const int* idx2,
                                                    const int* off2, int n_buckets, float nn_ratio, int check_orientation,
                                                    int* matches12, int* n_matches) {
    ORBX_REQUIRE(n_matches && (n1 == 0 || (matches12 && valid1)) && (n2 == 0 || valid2), "null argument");
    ORBX_REQUIRE(!check_orientation || ((n1 == 0 || angle1) && (n2 == 0 || angle2)), "orientation check needs the angles");
    BucketPairs bp;
    orbx_status s = bucket_allpairs_impl(desc1, n1, idx1, off1, desc2, n2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    return search_by_bow_keyframes_replay(bp, angle1, valid1, n1, idx1, off1, angle2, valid2, n2, idx2, off2, n_buckets,
                                          nn_ratio, check_orientation, matches12, n_matches);
}

// The loop of SearchByBoW(KeyFrame*, KeyFrame*, ...) over the distance rows (ORBmatcher.cc:856-964).
static orbx_status search_by_bow_keyframes_replay(const BucketPairs& bp, const float* angle1, const uint8_t* valid1, int n1,
                                                  const int* idx1, const int* off1, const float* angle2,
                                                  const uint8_t* valid2, int n2, const int* idx2, const int* off2,
                                                  int n_buckets, float nn_ratio, int check_orientation, int* matches12,
                                                  int* n_matches) {
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<uint8_t> matched2(n2, 0);
    RotHist hist(1.0f / kHistoLength);
    int nm = 0;
    for (int b = 0; b < n_buckets; ++b) {
        const int jb = off2[b], je = off2[b + 1];
        for (int e = off1[b]; e < off1[b + 1]; ++e) {
            const int i1 = idx1[e];
            if (!valid1[i1]) continue;
            const uint16_t* row = bp.dist + bp.rowStart[e];
            int best1 = 256, bestIdx2 = -1, best2 = 256;
            for (int j = jb; j < je; ++j) {
                const int i2 = idx2[j];
                if (matched2[i2] || !valid2[i2]) continue;
                const int d = row[j - jb];
                if (d < best1) { best2 = best1; best1 = d; bestIdx2 = i2; }
                else if (d < best2) best2 = d;
            }
            if (best1 < 50 && (float)best1 < nn_ratio * (float)best2) {
                matches12[i1] = bestIdx2;
                matched2[bestIdx2] = 1;
                if (check_orientation)
                    ORBX_REQUIRE(hist.add(angle1[i1], angle2[bestIdx2], i1), "rotation bin out of range");
                ++nm;
            }
        }
    }
    if (check_orientation) hist.for_each_rejected([&](int i1) { matches12[i1] = -1; --nm; });
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_for_triangulation(const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* free1,
                                                     const float* uright1, int n1, const int* idx1, const int* off1,
                                                     const orbx_keypoint* kps2, const uint8_t* desc2, const uint8_t* free2,
                                                     const float* uright2, int n2, const int* idx2, const int* off2,
                                                     int n_buckets, const orbx_epipolar* geom, int only_stereo, int coarse,
                                                     int check_orientation, int* matches12, int* n_matches) {
    ORBX_REQUIRE(n_matches && geom && (n1 == 0 || (kps1 && free1 && matches12)) && (n2 == 0 || (kps2 && free2)), "null argument");
    ORBX_REQUIRE(geom->n_levels > 0 && geom->scale_factors2 && geom->level_sigma2_2, "level tables missing");
    for (int i = 0; i < n2; ++i)
        ORBX_REQUIRE(kps2[i].octave >= 0 && kps2[i].octave < geom->n_levels, "keypoint %d of the second keyframe: octave out of range", i);
    BucketPairs bp;
    orbx_status s = bucket_allpairs_impl(desc1, n1, idx1, off1, desc2, n2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    RotHist hist(1.0f / kHistoLength);
    const float* F = geom->F12;
    int nm = 0;
    for (int b = 0; b < n_buckets; ++b) {
        const int jb = off2[b], je = off2[b + 1];
        for (int e = off1[b]; e < off1[b + 1]; ++e) {
            const int i1 = idx1[e];
            if (!free1[i1]) continue;                                   // already has a map point (:1037)
            const bool stereo1 = uright1 && uright1[i1] >= 0;
            if (only_stereo && !stereo1) continue;
            const orbx_keypoint& kp1 = kps1[i1];
            const uint16_t* row = bp.dist + bp.rowStart[e];
            int bestDist = 50, bestIdx2 = -1;
            for (int j = jb; j < je; ++j) {
                const int i2 = idx2[j];
                if (!free2[i2]) continue;                               // vbMatched2 is never set in the reference
                const bool stereo2 = uright2 && uright2[i2] >= 0;
                if (only_stereo && !stereo2) continue;
                const int d = row[j - jb];
                if (d > 50 || d > bestDist) continue;
                const orbx_keypoint& kp2 = kps2[i2];
                if (!stereo1 && !stereo2) {                              // too close to the epipole (:1089-1097)
                    const float distex = geom->epipole_x - kp2.x, distey = geom->epipole_y - kp2.y;
                    if (distex * distex + distey * distey < 100 * geom->scale_factors2[kp2.octave]) continue;
                }
                bool ok = coarse != 0;
                if (!ok) {                                               // Pinhole::epipolarConstrain (Pinhole.cpp:126-140)
                    const float a = kp1.x * F[0] + kp1.y * F[3] + F[6];
                    const float bb = kp1.x * F[1] + kp1.y * F[4] + F[7];
                    const float c = kp1.x * F[2] + kp1.y * F[5] + F[8];
                    const float num = a * kp2.x + bb * kp2.y + c;
                    const float den = a * a + bb * bb;
                    if (den != 0) {
                        const float dsqr = num * num / den;
                        ok = dsqr < 3.84 * geom->level_sigma2_2[kp2.octave];
                    }
                }
                if (ok) { bestIdx2 = i2; bestDist = d; }
            }
            if (bestIdx2 >= 0) {
                matches12[i1] = bestIdx2;
                ++nm;
                if (check_orientation)
                    ORBX_REQUIRE(hist.add(kp1.angle, kps2[bestIdx2].angle, i1), "rotation bin out of range");
            }
        }
    }
    if (check_orientation) hist.for_each_rejected([&](int i1) { matches12[i1] = -1; --nm; });
    *n_matches = nm;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_for_triangulation_cameras(const orbx_keypoint* kps1, const uint8_t* desc1,
                                                             const uint8_t* free1, const float* uright1, int n1,
                                                             const int* idx1, const int* off1,
                                                             const orbx_keypoint* kps2, const uint8_t* desc2,
                                                             const uint8_t* free2, const float* uright2, int n2,
                                                             const int* idx2, const int* off2, int n_buckets,
                                                             const orbx_triangulation_geometry* geom, int only_stereo,
                                                             int coarse, int check_orientation, int* matches12,
                                                             int* n_matches) {
    ORBX_REQUIRE(n_matches && geom && (n1 == 0 || (kps1 && free1 && matches12)) && (n2 == 0 || (kps2 && free2)), "null argument");
    orbx_status chk = triangulation_check(kps1, n1, kps2, n2, geom);
    if (chk != ORBX_OK) return chk;
    BucketPairs bp;
    orbx_status s = bucket_allpairs_impl(desc1, n1, idx1, off1, desc2, n2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    return triangulation_cameras_replay(bp, kps1, free1, uright1, n1, idx1, off1, kps2, free2, uright2, idx2, off2, n_buckets,
                                        geom, only_stereo, coarse, check_orientation, matches12, n_matches);
}

static orbx_status triangulation_check(const orbx_keypoint* kps1, int n1, const orbx_keypoint* kps2, int n2,
                                       const orbx_triangulation_geometry* geom) {
    ORBX_REQUIRE(geom->n_levels > 0 && geom->scale_factors2 && geom->level_sigma2_1 && geom->level_sigma2_2, "level tables missing");
    for (int i = 0; i < n1; ++i)
        ORBX_REQUIRE(kps1[i].octave >= 0 && kps1[i].octave < geom->n_levels, "keypoint %d of the first keyframe: octave out of range", i);
    for (int i = 0; i < n2; ++i)
        ORBX_REQUIRE(kps2[i].octave >= 0 && kps2[i].octave < geom->n_levels, "keypoint %d of the second keyframe: octave out of range", i);
    return ORBX_OK;
}

// The loop of SearchForTriangulation over the distance rows (ORBmatcher.cc:1026-1199).
static orbx_status triangulation_cameras_replay(const BucketPairs& bp, const orbx_keypoint* kps1, const uint8_t* free1,
                                                const float* uright1, int n1, const int* idx1, const int* off1,
                                                const orbx_keypoint* kps2, const uint8_t* free2, const float* uright2,
                                                const int* idx2, const int* off2, int n_buckets,
                                                const orbx_triangulation_geometry* geom, int only_stereo, int coarse,
                                                int check_orientation, int* matches12, int* n_matches) {
    const int nLeft1 = geom->n_left1, nLeft2 = geom->n_left2;
    const bool twoCams = nLeft1 != -1 && nLeft2 != -1;
    orbx_gates::Camera cams[4];
    for (int c = 0; c < 4; ++c) {
        const orbx_camera_model& m = c < 2 ? geom->cam1[c] : geom->cam2[c - 2];
        ORBX_REQUIRE(m.type == ORBX_CAMERA_PINHOLE || m.type == ORBX_CAMERA_KANNALA_BRANDT8 || (!twoCams && (c & 1)),
                     "unknown camera model type %d", m.type);
        cams[c].type = m.type;
        for (int k = 0; k < 8; ++k) cams[c].p[k] = m.params[k];
    }
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    RotHist hist(1.0f / kHistoLength);
    int nm = 0;
    for (int b = 0; b < n_buckets; ++b) {
        const int jb = off2[b], je = off2[b + 1];
        for (int e = off1[b]; e < off1[b + 1]; ++e) {
            const int i1 = idx1[e];
            if (!free1[i1]) continue;                                   // already has a map point (:1037)
            const bool stereo1 = nLeft1 == -1 && uright1 && uright1[i1] >= 0;   // !mpCamera2 && mvuRight >= 0 (:1042)
            if (only_stereo && !stereo1) continue;
            const orbx_keypoint& kp1 = kps1[i1];
            const bool right1 = !(nLeft1 == -1 || i1 < nLeft1);
            const uint16_t* row = bp.dist + bp.rowStart[e];
            int bestDist = 50, bestIdx2 = -1;
            for (int j = jb; j < je; ++j) {
                const int i2 = idx2[j];
                if (!free2[i2]) continue;
                const bool stereo2 = nLeft2 == -1 && uright2 && uright2[i2] >= 0;
                if (only_stereo && !stereo2) continue;
                const int d = row[j - jb];
                if (d > 50 || d > bestDist) continue;
                const orbx_keypoint& kp2 = kps2[i2];
                const bool right2 = !(nLeft2 == -1 || i2 < nLeft2);
                if (!stereo1 && !stereo2 && nLeft1 == -1) {             // too close to the epipole (:1089-1097)
                    const float distex = geom->epipole_x - kp2.x, distey = geom->epipole_y - kp2.y;
                    if (distex * distex + distey * distey < 100 * geom->scale_factors2[kp2.octave]) continue;
                }
                bool ok = coarse != 0;
                if (!ok) {
                    if (geom->gate) {
                        ok = geom->gate(geom->gate_user, i1, i2) != 0;
                    } else {
                        // camera pair and relative pose of this candidate (:1099-1137)
                        const int sel = twoCams ? (right1 ? 2 : 0) + (right2 ? 1 : 0) : 0;
                        const orbx_rel_pose& T = geom->pose[sel >> 1][sel & 1];
                        ok = orbx_gates::epipolar_constrain(cams[twoCams && right1 ? 1 : 0], cams[2 + (twoCams && right2 ? 1 : 0)],
                                                            kp1.x, kp1.y, kp2.x, kp2.y, T.R12, T.t12,
                                                            geom->level_sigma2_1[kp1.octave], geom->level_sigma2_2[kp2.octave]);
                    }
                }
                if (ok) { bestIdx2 = i2; bestDist = d; }
            }
            if (bestIdx2 >= 0) {
                matches12[i1] = bestIdx2;
                ++nm;
                if (check_orientation)
                    ORBX_REQUIRE(hist.add(kp1.angle, kps2[bestIdx2].angle, i1), "rotation bin out of range");
            }
        }
    }
    if (check_orientation) hist.for_each_rejected([&](int i1) { matches12[i1] = -1; --nm; });
    *n_matches = nm;
    return ORBX_OK;
}

// ---- the bucketed searches on resident frames: descriptors stay on the device, one launch + one synchronisation ----
namespace {
std::vector<float> frame_angles(const orbx_frame* f) {
    std::vector<float> a((size_t)f->n);
    for (int i = 0; i < f->n; ++i) a[i] = f->hostKps[i].angle;
    return a;
}
}  // namespace

extern "C" orbx_status orbx_frame_search_by_bow(const orbx_frame* kf, const uint8_t* valid_kf, const int* idx_kf,
                                                const int* off_kf, const orbx_frame* f, int n_left_f, const int* idx_f,
                                                const int* off_f, int n_buckets, float nn_ratio, int check_orientation,
                                                int* match_f, int* n_matches) {
    ORBX_REQUIRE(kf && f && n_matches && (f->n == 0 || match_f) && (kf->n == 0 || valid_kf), "null argument");
    BucketPairs bp;
    orbx_status s = bucket_pairs_frames(kf, idx_kf, off_kf, f, idx_f, off_f, n_buckets, bp);
    if (s != ORBX_OK) return s;
    const std::vector<float> aKF = frame_angles(kf), aF = frame_angles(f);
    return search_by_bow_replay(bp, aKF.data(), valid_kf, idx_kf, off_kf, aF.data(), f->n, n_left_f, idx_f, off_f, n_buckets,
                                nn_ratio, check_orientation, match_f, n_matches);
}

extern "C" orbx_status orbx_frame_search_by_bow_keyframes(const orbx_frame* kf1, const uint8_t* valid1, const int* idx1,
                                                          const int* off1, const orbx_frame* kf2, const uint8_t* valid2,
                                                          const int* idx2, const int* off2, int n_buckets, float nn_ratio,
                                                          int check_orientation, int* matches12, int* n_matches) {
    ORBX_REQUIRE(kf1 && kf2 && n_matches && (kf1->n == 0 || (matches12 && valid1)) && (kf2->n == 0 || valid2), "null argument");
    BucketPairs bp;
    orbx_status s = bucket_pairs_frames(kf1, idx1, off1, kf2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    const std::vector<float> a1 = frame_angles(kf1), a2 = frame_angles(kf2);
    return search_by_bow_keyframes_replay(bp, a1.data(), valid1, kf1->n, idx1, off1, a2.data(), valid2, kf2->n, idx2, off2,
                                          n_buckets, nn_ratio, check_orientation, matches12, n_matches);
}

extern "C" orbx_status orbx_frame_search_for_triangulation(const orbx_frame* kf1, const uint8_t* free1, const int* idx1,
                                                           const int* off1, const orbx_frame* kf2, const uint8_t* free2,
                                                           const int* idx2, const int* off2, int n_buckets,
                                                           const orbx_triangulation_geometry* geom, int only_stereo,
                                                           int coarse, int check_orientation, int* matches12,
                                                           int* n_matches) {
    ORBX_REQUIRE(kf1 && kf2 && n_matches && geom && (kf1->n == 0 || (free1 && matches12)) && (kf2->n == 0 || free2), "null argument");
    const orbx_keypoint* kps1 = kf1->hostKps.data();
    const orbx_keypoint* kps2 = kf2->hostKps.data();
    orbx_status s = triangulation_check(kps1, kf1->n, kps2, kf2->n, geom);
    if (s != ORBX_OK) return s;
    BucketPairs bp;
    s = bucket_pairs_frames(kf1, idx1, off1, kf2, idx2, off2, n_buckets, bp);
    if (s != ORBX_OK) return s;
    return triangulation_cameras_replay(bp, kps1, free1, kf1->hasUright ? kf1->hostUright.data() : nullptr, kf1->n, idx1, off1,
                                        kps2, free2, kf2->hasUright ? kf2->hostUright.data() : nullptr, idx2, off2, n_buckets,
                                        geom, only_stereo, coarse, check_orientation, matches12, n_matches);
}

// ======================================================================================
// Bag-of-words transform: DBoW2::TemplatedVocabulary<FORB>::transform
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1112-1243, called from Frame::ComputeBoW,
// Frame.cc:872-879, with levelsup = 4).  One wave per descriptor: at every level the lanes
// take one child each, the wave keeps the child with the smallest distance (the first among
// equals, as the reference's strict '<' scan does) and descends until a leaf.
// ======================================================================================
struct orbx_vocabulary {
    int device = 0;
    int nNodes = 0, depth = 0;
    int* d_childOff = nullptr;      // nNodes + 1
    int* d_childIdx = nullptr;
    uint32_t* d_desc = nullptr;     // nNodes x 8
    std::vector<int> wordId;        // per node, -1 for inner nodes
    std::vector<double> weight;     // per node
};

__global__ __launch_bounds__(256) void k_bow_descend(const int* __restrict__ childOff, const int* __restrict__ childIdx,
                                                     const uint32_t* __restrict__ nodeDesc, const uint32_t* __restrict__ desc,
                                                     int n, int nidLevel, int* __restrict__ leafOut, int* __restrict__ nidOut) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = desc[8ll * i + k];
    int node = 0, level = 0, nid = 0;
    while (true) {
        const int b = childOff[node], e = childOff[node + 1];
        if (b == e) break;                                    // leaf
        ++level;
        unsigned int key = 0xFFFFFFFFu;                       // (distance << 22) | position in the child list
        for (int c = b + lane; c < e; c += 64) {
            const uint32_t* d = nodeDesc + 8ll * childIdx[c];
            int dist = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dist += __popc(q[k] ^ d[k]);
            key = min(key, ((unsigned int)dist << 22) | (unsigned int)(c - b));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) key = min(key, (unsigned int)__shfl_xor((int)key, off));
        node = childIdx[b + (int)(key & 0x3FFFFFu)];
        if (level == nidLevel) nid = node;
    }
    if (lane == 0) { leafOut[i] = node; nidOut[i] = nid; }
}

extern "C" orbx_status orbx_vocabulary_create(int n_nodes, const int* child_off, const int* child_idx,
                                              const uint8_t* node_descriptors, const int* word_id, const double* weight,
                                              int depth, int device, orbx_vocabulary** out) {
    ORBX_REQUIRE(out && n_nodes >= 1 && child_off && node_descriptors && word_id && weight && depth >= 1, "bad argument");
    ORBX_REQUIRE(child_off[0] == 0, "child offsets must start at 0");
    const int nChildren = child_off[n_nodes];
    ORBX_REQUIRE(nChildren == 0 || child_idx, "null child list");
    for (int i = 0; i < n_nodes; ++i) {
        ORBX_REQUIRE(child_off[i + 1] >= child_off[i] && child_off[i + 1] - child_off[i] < (1 << 22), "bad child offsets at node %d", i);
        ORBX_REQUIRE((child_off[i + 1] == child_off[i]) == (word_id[i] >= 0), "node %d: exactly the leaves carry a word id", i);
    }
    // every child index in range and deeper than its parent (no cycles): a child list may only name
    // nodes that are not ancestors; checked by a bounded breadth-first walk from the root
    std::vector<int> levelOf(n_nodes, -1);
    std::vector<int> frontier{0};
    levelOf[0] = 0;
    while (!frontier.empty()) {
        std::vector<int> next;
        for (int p : frontier)
            for (int c = child_off[p]; c < child_off[p + 1]; ++c) {
                const int ch = child_idx[c];
                ORBX_REQUIRE(ch > 0 && ch < n_nodes, "child index %d of node %d out of range", ch, p);
                ORBX_REQUIRE(levelOf[ch] < 0, "node %d has two parents (or the tree has a cycle)", ch);
                levelOf[ch] = levelOf[p] + 1;
                next.push_back(ch);
            }
        frontier.swap(next);
    }
    ORBX_HIP_TRY(hipSetDevice(device));
    orbx_vocabulary* v = new orbx_vocabulary;
    v->device = device; v->nNodes = n_nodes; v->depth = depth;
    v->wordId.assign(word_id, word_id + n_nodes);
    v->weight.assign(weight, weight + n_nodes);
    hipError_t e = hipMalloc((void**)&v->d_childOff, (size_t)(n_nodes + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_childIdx, (size_t)std::max(nChildren, 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_desc, (size_t)n_nodes * 32);
    if (e == hipSuccess) e = hipMemcpy(v->d_childOff, child_off, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nChildren) e = hipMemcpy(v->d_childIdx, child_idx, (size_t)nChildren * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, node_descriptors, (size_t)n_nodes * 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        orbx_set_error("vocabulary upload failed: %s", hipGetErrorString(e));
        hipFree(v->d_childOff); hipFree(v->d_childIdx); hipFree(v->d_desc);
        delete v;
        return ORBX_ERR_HIP;
    }
    *out = v;
    return ORBX_OK;
}

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary* v) {
    if (!v) return;
    hipFree(v->d_childOff); hipFree(v->d_childIdx); hipFree(v->d_desc);
    delete v;
}

extern "C" orbx_status orbx_bow_transform(orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int* word_id,
                                          double* weight, int* node_id) {
    ORBX_REQUIRE(voc && n >= 0 && levelsup >= 0, "bad argument");
    if (n == 0) return ORBX_OK;
    ORBX_REQUIRE(desc && word_id && weight && node_id, "null argument");
    ORBX_HIP_TRY(hipSetDevice(voc->device));
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    PackedIo io;
    const size_t oD = io.add(desc, (size_t)n * 32), oL = io.want((size_t)n * 4), oN = io.want((size_t)n * 4);
    ORBX_HIP_TRY(io.upload(st));
    const int nidLevel = voc->depth - levelsup;     // <= 0: the feature vector node is the root
    hipLaunchKernelGGL(k_bow_descend, dim3((n + 3) / 4), dim3(256), 0, st, voc->d_childOff, voc->d_childIdx, voc->d_desc,
                       io.dev_in<uint32_t>(oD), n, nidLevel, io.dev_out<int>(oL), io.dev_out<int>(oN));
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(io.download(st));
    std::vector<int> leaf(n);
    io.fetch(leaf.data(), oL, (size_t)n * 4);
    io.fetch(node_id, oN, (size_t)n * 4);
    for (int i = 0; i < n; ++i) { word_id[i] = voc->wordId[leaf[i]]; weight[i] = voc->weight[leaf[i]]; }
    return ORBX_OK;
}

extern "C" orbx_status orbx_bow_vectors(orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int l1_normalize,
                                        int* bow_word, double* bow_value, int* n_bow, int* fv_node, int* fv_off,
                                        int* fv_idx, int* n_fv_nodes) {
    ORBX_REQUIRE(n_bow && n_fv_nodes && n >= 0, "bad argument");
    *n_bow = 0; *n_fv_nodes = 0;
    if (n == 0) { if (fv_off) fv_off[0] = 0; return ORBX_OK; }
    ORBX_REQUIRE(bow_word && bow_value && fv_node && fv_off && fv_idx, "null argument");
    std::vector<int> word(n), node(n);
    std::vector<double> w(n);
    orbx_status s = orbx_bow_transform(voc, desc, n, levelsup, word.data(), w.data(), node.data());
    if (s != ORBX_OK) return s;
    // BowVector::addWeight in feature order (TF_IDF weighting), FeatureVector::addFeature; both are
    // std::map in the reference, i.e. ascending key order with features in insertion order
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return word[a] < word[b]; });
    int nb = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        if (!(w[i] > 0)) continue;                     // stopped word
        if (nb > 0 && bow_word[nb - 1] == word[i]) bow_value[nb - 1] += w[i];
        else { bow_word[nb] = word[i]; bow_value[nb] = w[i]; ++nb; }
    }
    if (l1_normalize) {                                // BowVector::normalize(L1)
        double norm = 0.0;
        for (int k = 0; k < nb; ++k) norm += std::fabs(bow_value[k]);
        if (norm > 0.0) for (int k = 0; k < nb; ++k) bow_value[k] /= norm;
    } else if (nb > 0) {                               // TemplatedVocabulary.h:1149-1155
        const double nd = nb;
        for (int k = 0; k < nb; ++k) bow_value[k] /= nd;
    }
    *n_bow = nb;
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return node[a] < node[b]; });
    int nn = 0, nf = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        if (!(w[i] > 0)) continue;
        if (nn == 0 || fv_node[nn - 1] != node[i]) { fv_node[nn] = node[i]; fv_off[nn] = nf; ++nn; }
        fv_idx[nf++] = i;
    }
    fv_off[nn] = nf;
    *n_fv_nodes = nn;
    return ORBX_OK;
}

// ======================================================================================
// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:416-533) for many map points at once:
// per point the observation whose median Hamming distance to all observations (itself
// included) is smallest; median = element floor(0.5 * (N - 1)) of the sorted row; the first
// row among equal medians.  One wave per map point, one lane per row of the distance matrix.
// ======================================================================================
constexpr int kDistinctMaxObs = 384;

__global__ __launch_bounds__(64) void k_distinctive(const uint32_t* __restrict__ desc, const int* __restrict__ off,
                                                    int* __restrict__ bestIdx, int* __restrict__ bestMedian, int maxN) {
    extern __shared__ __align__(16) unsigned char dd_smem[];
    uint32_t* s_desc = reinterpret_cast<uint32_t*>(dd_smem);                       // maxN x 8
    uint16_t* s_row = reinterpret_cast<uint16_t*>(s_desc + 8 * maxN);              // 64 rows x maxN
    const int p = blockIdx.x, lane = threadIdx.x;
    const int b = off[p], N = off[p + 1] - b;
    if (N <= 0) { if (lane == 0) { bestIdx[p] = -1; bestMedian[p] = INT_MAX; } return; }
    for (int i = lane; i < 8 * N; i += 64) s_desc[i] = desc[8ll * b + i];
    __syncthreads();
    const int kth = (int)(0.5 * (N - 1));            // vDists[0.5 * (N - 1)]
    unsigned int best = 0xFFFFFFFFu;                 // (median << 16) | row
    uint16_t* row = s_row + lane * maxN;
    for (int i = lane; i < N; i += 64) {
        uint32_t q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = s_desc[8 * i + k];
        for (int j = 0; j < N; ++j) {
            int d = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += __popc(q[k] ^ s_desc[8 * j + k]);
            row[j] = (uint16_t)d;
        }
        // smallest v with #{j : row[j] <= v} > kth  (the kth element of the sorted row)
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; ++j) cnt += (row[j] <= mid) ? 1 : 0;
            if (cnt > kth) hi = mid; else lo = mid + 1;
        }
        best = min(best, ((unsigned int)lo << 16) | (unsigned int)i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned int)__shfl_xor((int)best, o));
    if (lane == 0) { bestIdx[p] = (int)(best & 0xFFFFu); bestMedian[p] = (int)(best >> 16); }
}

extern "C" orbx_status orbx_distinctive_descriptors(const uint8_t* desc, const int* off, int n_points, int* best_idx,
                                                    int* best_median) {
    ORBX_REQUIRE(n_points >= 0, "negative size");
    if (n_points == 0) return ORBX_OK;
    ORBX_REQUIRE(off && best_idx && best_median && off[0] == 0, "bad argument");
    int maxN = 1;
    for (int p = 0; p < n_points; ++p) {
        ORBX_REQUIRE(off[p + 1] >= off[p], "bad offsets");
        ORBX_REQUIRE(off[p + 1] - off[p] <= kDistinctMaxObs, "map point %d has %d observations, at most %d are supported", p,
                     off[p + 1] - off[p], kDistinctMaxObs);
        maxN = std::max(maxN, off[p + 1] - off[p]);
    }
    const int total = off[n_points];
    ORBX_REQUIRE(total == 0 || desc, "null descriptors");
    hipStream_t st;
    ORBX_HIP_TRY(thread_stream(&st));
    PackedIo io;
    const size_t oD = io.add(desc, (size_t)total * 32), oO = io.add(off, (size_t)(n_points + 1) * 4);
    const size_t oI = io.want((size_t)n_points * 4), oM = io.want((size_t)n_points * 4);
    ORBX_HIP_TRY(io.upload(st));
    maxN = (maxN + 1) & ~1;
    const size_t lds = (size_t)maxN * 32 + (size_t)64 * maxN * 2;      // <= 61440 bytes at 384 observations
    hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(64), lds, st, io.dev_in<uint32_t>(oD), io.dev_in<int>(oO),
                       io.dev_out<int>(oI), io.dev_out<int>(oM), maxN);
    ORBX_HIP_TRY(hipGetLastError());
    ORBX_HIP_TRY(io.download(st));
    io.fetch(best_idx, oI, (size_t)n_points * 4);
    io.fetch(best_median, oM, (size_t)n_points * 4);
    return ORBX_OK;
}
