// The multi-GPU entry points of include/orbx_dist.h on however many GPUs the machine shows (one on the test box):
// the sharded k = 2 search through a real RCCL communicator of one rank, split into two calls' worth of shards by
// hand to exercise base_index; with two or more GPUs the same search with one rank (thread) per device; and the
// extractor pool (all devices) against a single extractor.  Prints one line per check.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "orbx_dist.h"

#define CHECK(expr)                                                                              \
    do {                                                                                         \
        if ((expr) != 0) { std::fprintf(stderr, "%s failed: %s\n", #expr, orbx_last_error()); return 1; } \
    } while (0)
#define HIPCHECK(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } \
    } while (0)

static uint32_t g_seed = 12345u;
static uint32_t rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }

static std::vector<uint8_t> scene(int w, int h, int salt) {
    std::vector<uint8_t> img((size_t)w * h, 90);
    g_seed = 777u + (uint32_t)salt;
    for (int r = 0; r < 400; ++r) {
        const int x0 = (int)(rnd() % (uint32_t)w), y0 = (int)(rnd() % (uint32_t)h);
        const int rw = 6 + (int)(rnd() % 50u), rh = 6 + (int)(rnd() % 50u);
        const uint8_t v = (uint8_t)(rnd() % 256u);
        for (int y = y0; y < y0 + rh && y < h; ++y)
            for (int x = x0; x < x0 + rw && x < w; ++x) img[(size_t)y * w + x] = v;
    }
    return img;
}

// Two or more GPUs in one process: one thread per device, each with its own rank of one RCCL communicator and its
// contiguous shard of the map; every rank must end up with the result of the single-GPU search.  Prints
// "multi_gpu=skipped_one_gpu" on a machine with one GPU.
static int multi_gpu_search(const std::vector<uint8_t>& q, int nq, const std::vector<uint8_t>& t, int64_t nt,
                            const std::vector<orbx_match2>& want) {
    int nd = 0;
    HIPCHECK(hipGetDeviceCount(&nd));
    if (nd < 2) { std::printf("multi_gpu=skipped_one_gpu\n"); return 0; }
    uint8_t id[128];
    CHECK(orbx_rccl_unique_id(id));
    struct Rank {
        void* comm = nullptr;
        uint8_t *dq = nullptr, *dt = nullptr;
        orbx_match2* dout = nullptr;
        void* ws = nullptr;
        size_t wsBytes = 0;
        hipStream_t st = nullptr;
        int64_t base = 0, nLocal = 0;
        bool ready = false;
    };
    std::vector<Rank> R((size_t)nd);
    std::vector<std::string> verdict((size_t)nd, "not run");
    // Step 1, every rank: device buffers, its shard, its communicator rank.  The collective of step 2 only starts when
    // all ranks got this far (a rank that stayed out of an all-gather would leave the others waiting for ever).
    auto set_up = [&](int rank) {
        Rank& k = R[rank];
        if (hipSetDevice(rank) != hipSuccess) { verdict[rank] = "hipSetDevice failed"; return; }
        const int64_t shard = nt / nd;               // equal contiguous shards, the last rank takes the remainder
        k.base = rank * shard;
        k.nLocal = rank < nd - 1 ? shard : nt - k.base;
        k.wsBytes = orbx_knn2_sharded_workspace_bytes(nq, k.nLocal, nd);
        const bool ok =
            hipMalloc((void**)&k.dq, q.size()) == hipSuccess && hipMalloc((void**)&k.dt, (size_t)k.nLocal * 32 + 32) == hipSuccess &&
            hipMalloc((void**)&k.dout, (size_t)nq * sizeof(orbx_match2)) == hipSuccess && hipMalloc(&k.ws, k.wsBytes) == hipSuccess &&
            hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking) == hipSuccess &&
            hipMemcpyAsync(k.dq, q.data(), q.size(), hipMemcpyHostToDevice, k.st) == hipSuccess &&
            hipMemcpyAsync(k.dt, t.data() + (size_t)k.base * 32, (size_t)k.nLocal * 32, hipMemcpyHostToDevice, k.st) == hipSuccess &&
            hipStreamSynchronize(k.st) == hipSuccess;
        if (!ok) { verdict[rank] = "device set-up failed"; return; }
        if (orbx_rccl_comm_create(id, rank, nd, &k.comm) != ORBX_OK) { verdict[rank] = std::string("comm_create: ") + orbx_last_error(); return; }
        k.ready = true;
    };
    auto search = [&](int rank) {
        Rank& k = R[rank];
        if (hipSetDevice(rank) != hipSuccess) { verdict[rank] = "hipSetDevice failed"; return; }
        std::vector<orbx_match2> got((size_t)nq);
        if (orbx_hamming_knn2_sharded_device(k.dq, nq, k.dt, k.nLocal, k.base, k.comm, k.dout, k.ws, k.wsBytes, k.st) != ORBX_OK)
            verdict[rank] = std::string("sharded search: ") + orbx_last_error();
        else if (hipMemcpyAsync(got.data(), k.dout, (size_t)nq * sizeof(orbx_match2), hipMemcpyDeviceToHost, k.st) != hipSuccess ||
                 hipStreamSynchronize(k.st) != hipSuccess)
            verdict[rank] = "download failed";
        else
            verdict[rank] = std::memcmp(got.data(), want.data(), (size_t)nq * sizeof(orbx_match2)) == 0 ? "OK" : "MISMATCH";
    };
    auto run_all = [&](auto&& fn) {
        std::vector<std::thread> threads;
        for (int r = 0; r < nd; ++r) threads.emplace_back(fn, r);
        for (std::thread& th : threads) th.join();
    };
    run_all(set_up);                                   // (ncclCommInitRank is itself a collective: all ranks call it together)
    bool allReady = true;
    for (const Rank& k : R) allReady = allReady && k.ready;
    if (allReady) run_all(search);
    for (int r = 0; r < nd; ++r) {
        Rank& k = R[r];
        (void)hipSetDevice(r);
        if (k.comm) orbx_rccl_comm_destroy(k.comm);
        (void)hipFree(k.dq); (void)hipFree(k.dt); (void)hipFree(k.dout); (void)hipFree(k.ws);
        if (k.st) (void)hipStreamDestroy(k.st);
    }
    bool all = true;
    for (int r = 0; r < nd; ++r) {
        if (verdict[r] != "OK") { all = false; std::printf("multi_gpu_rank%d=%s\n", r, verdict[r].c_str()); }
    }
    std::printf("multi_gpu=%s ranks=%d\n", all ? "OK" : "MISMATCH", nd);
    HIPCHECK(hipSetDevice(0));
    return 0;
}

int main() {
    // ---- sharded search, world size 1 ----
    const int nq = 700;
    const int64_t nt = 50000;
    std::vector<uint8_t> q((size_t)nq * 32), t((size_t)nt * 32);
    for (auto& b : q) b = (uint8_t)rnd();
    for (auto& b : t) b = (uint8_t)rnd();
    for (int i = 0; i < nq; ++i) {                                   // plant near copies so that the winners are not noise
        uint8_t* row = &t[(size_t)((i * 71) % nt) * 32];
        std::memcpy(row, &q[(size_t)i * 32], 32);
        row[i % 32] ^= (uint8_t)(1u << (i % 8));
    }
    std::memcpy(&t[(size_t)40000 * 32], &t[(size_t)71 * 32], 32);   // a duplicate row: the lower index must win
    std::vector<orbx_match2> want(nq), got(nq), gotEmpty(nq);
    CHECK(orbx_hamming_knn2(q.data(), nq, t.data(), nt, want.data()));

    uint8_t id[128];
    void* comm = nullptr;
    CHECK(orbx_rccl_unique_id(id));
    CHECK(orbx_rccl_comm_create(id, 0, 1, &comm));
    uint8_t *dq = nullptr, *dt = nullptr;
    orbx_match2* dout = nullptr;
    void* ws = nullptr;
    const size_t wsBytes = orbx_knn2_sharded_workspace_bytes(nq, nt, 1);
    HIPCHECK(hipMalloc((void**)&dq, q.size()));
    HIPCHECK(hipMalloc((void**)&dt, t.size()));
    HIPCHECK(hipMalloc((void**)&dout, (size_t)nq * sizeof(orbx_match2)));
    HIPCHECK(hipMalloc(&ws, wsBytes));
    hipStream_t st;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCHECK(hipMemcpyAsync(dq, q.data(), q.size(), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(dt, t.data(), t.size(), hipMemcpyHostToDevice, st));
    CHECK(orbx_hamming_knn2_sharded_device(dq, nq, dt, nt, 0, comm, dout, ws, wsBytes, st));
    HIPCHECK(hipMemcpyAsync(got.data(), dout, (size_t)nq * sizeof(orbx_match2), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::printf("sharded_world1=%s\n", std::memcmp(got.data(), want.data(), (size_t)nq * sizeof(orbx_match2)) == 0 ? "OK" : "MISMATCH");
    // the rows this rank would hold as rank 1 of 2: global indices start at base_index
    const int64_t half = nt / 2;
    std::vector<orbx_match2> wantHi(nq);
    CHECK(orbx_hamming_knn2(q.data(), nq, t.data() + (size_t)half * 32, nt - half, wantHi.data()));
    CHECK(orbx_hamming_knn2_sharded_device(dq, nq, dt + (size_t)half * 32, nt - half, half, comm, dout, ws, wsBytes, st));
    HIPCHECK(hipMemcpyAsync(got.data(), dout, (size_t)nq * sizeof(orbx_match2), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    bool ok = true;
    for (int i = 0; i < nq; ++i)
        ok = ok && got[i].idx0 == wantHi[i].idx0 + half && got[i].idx1 == wantHi[i].idx1 + half &&
             got[i].dist0 == wantHi[i].dist0 && got[i].dist1 == wantHi[i].dist1;
    std::printf("sharded_base_index=%s\n", ok ? "OK" : "MISMATCH");
    // an empty shard and a short workspace
    CHECK(orbx_hamming_knn2_sharded_device(dq, nq, dt, 0, 0, comm, dout, ws, wsBytes, st));
    HIPCHECK(hipMemcpyAsync(gotEmpty.data(), dout, (size_t)nq * sizeof(orbx_match2), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::printf("sharded_empty=%s\n", gotEmpty[0].idx0 == -1 && gotEmpty[nq - 1].idx1 == -1 ? "OK" : "MISMATCH");
    std::printf("sharded_short_workspace=%s\n",
                orbx_hamming_knn2_sharded_device(dq, nq, dt, nt, 0, comm, dout, ws, 1024, st) == ORBX_ERR_INVALID ? "refused" : "accepted");
    orbx_rccl_comm_destroy(comm);
    if (multi_gpu_search(q, nq, t, nt, want) != 0) return 1;

    // ---- extractor pool ----
    const int W = 640, H = 400, nFrames = 5;
    std::vector<std::vector<uint8_t>> frames;
    std::vector<const uint8_t*> ptrs;
    for (int f = 0; f < nFrames; ++f) { frames.push_back(scene(W, H, f)); ptrs.push_back(frames.back().data()); }
    orbx_params p{1000, 1.2f, 8, 20, 7, W, H, 2, -1};
    orbx_extract_pool* pool = nullptr;
    CHECK(orbx_extract_pool_create(&p, nullptr, 0, &pool));
    orbx_extractor* single = nullptr;
    CHECK(orbx_extractor_create(&p, &single));
    const int cap = orbx_extractor_max_keypoints(single);
    std::vector<orbx_keypoint> kp((size_t)nFrames * cap), kp1(cap);
    std::vector<uint8_t> de((size_t)nFrames * cap * 32), de1((size_t)cap * 32);
    std::vector<int> n(nFrames), mono(nFrames);
    CHECK(orbx_extract_pool_batch(pool, ptrs.data(), nFrames, W, H, W, 0, 0, kp.data(), de.data(), cap, n.data(), mono.data()));
    ok = true;
    int total = 0;
    for (int f = 0; f < nFrames; ++f) {
        int n1 = 0, mono1 = 0;
        CHECK(orbx_extract(single, ptrs[f], W, H, W, 0, 0, nullptr, 0, kp1.data(), de1.data(), cap, &n1, &mono1));
        ok = ok && n1 == n[f] && mono1 == mono[f] &&
             std::memcmp(kp1.data(), &kp[(size_t)f * cap], (size_t)n1 * sizeof(orbx_keypoint)) == 0 &&
             std::memcmp(de1.data(), &de[(size_t)f * cap * 32], (size_t)n1 * 32) == 0;
        total += n1;
    }
    std::printf("pool_devices=%d pool_frames=%s keypoints=%d\n", orbx_extract_pool_devices(pool), ok ? "OK" : "MISMATCH", total);
    CHECK(orbx_extract_pool_batch(pool, ptrs.data(), 0, W, H, W, 0, 0, kp.data(), de.data(), cap, n.data(), mono.data()));
    int bad = 99;
    orbx_extract_pool* none = nullptr;
    std::printf("pool_bad_device=%s\n", orbx_extract_pool_create(&p, &bad, 1, &none) == ORBX_ERR_INVALID ? "refused" : "accepted");
    orbx_extract_pool_destroy(pool);
    orbx_extractor_destroy(single);
    return 0;
}
