// liborbx_dist.so: the multi-GPU entry points (include/orbx_dist.h).  Host code only; the kernels are
// liborbx.so's, reached through its C ABI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbx_dist.h"

void orbx_set_error(const char* fmt, ...);      // liborbx.so: the calling thread's message for orbx_last_error()

namespace {
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

#define DIST_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) { orbx_set_error(__VA_ARGS__); return ORBX_ERR_INVALID; } \
    } while (0)
#define DIST_NCCL_TRY(expr)                                                                      \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) { orbx_set_error("%s: %s", #expr, ncclGetErrorString(r_)); return ORBX_ERR_HIP; } \
    } while (0)
}  // namespace

extern "C" size_t orbx_knn2_sharded_workspace_bytes(int n_query, int64_t n_train_shard, int world_size) {
    if (n_query < 0 || n_train_shard < 0 || world_size < 1) return 0;
    const size_t rows = align256((size_t)std::max(n_query, 1) * sizeof(orbx_match2));
    // [local result][gathered parts: world_size x n_query][workspace of the local search]
    return rows + align256(rows * (size_t)world_size) + align256(orbx_knn2_workspace_bytes(n_query, n_train_shard));
}

extern "C" orbx_status orbx_hamming_knn2_sharded_device(const uint8_t* d_query, int n_query,
                                                        const uint8_t* d_train_shard, int64_t n_train_shard,
                                                        int64_t base_index, void* rccl_comm, orbx_match2* d_out,
                                                        void* d_workspace, size_t workspace_bytes, void* stream) {
    DIST_REQUIRE(rccl_comm, "null communicator");
    DIST_REQUIRE(n_query >= 0 && n_train_shard >= 0 && base_index >= 0, "negative size");
    ncclComm_t comm = static_cast<ncclComm_t>(rccl_comm);
    int world = 0;
    DIST_NCCL_TRY(ncclCommCount(comm, &world));
    DIST_REQUIRE(world >= 1, "communicator of %d ranks", world);
    if (n_query == 0) return ORBX_OK;              // nothing to gather: every rank sees the same n_query
    DIST_REQUIRE(d_query && d_out && d_workspace, "null argument");
    DIST_REQUIRE(workspace_bytes >= orbx_knn2_sharded_workspace_bytes(n_query, n_train_shard, world),
                 "workspace of %zu bytes, %zu needed", workspace_bytes,
                 orbx_knn2_sharded_workspace_bytes(n_query, n_train_shard, world));
    const size_t rows = align256((size_t)n_query * sizeof(orbx_match2));
    char* ws = static_cast<char*>(d_workspace);
    orbx_match2* local = reinterpret_cast<orbx_match2*>(ws);
    orbx_match2* parts = reinterpret_cast<orbx_match2*>(ws + rows);
    char* inner = ws + rows + align256(rows * (size_t)world);
    // Everything that depends on the arguments has been checked above, before anything is enqueued: from here on the
    // local search can only fail with a HIP error.  A rank to which that happens returns without entering the
    // all-gather, and the other ranks wait in it -- as after any failed rank of a collective, the caller has to tear
    // the communicator down (ncclCommAbort / orbx_rccl_comm_destroy) on all ranks.
    DIST_REQUIRE(n_train_shard == 0 || d_train_shard, "null shard of %lld rows", (long long)n_train_shard);
    DIST_REQUIRE(base_index + n_train_shard <= (int64_t)INT_MAX, "train index does not fit 32 bits");
    DIST_REQUIRE((((uintptr_t)d_query | (uintptr_t)d_train_shard | (uintptr_t)d_workspace) & 15) == 0,
                 "descriptor arrays and workspace must be 16-byte aligned");
    orbx_status s = orbx_hamming_knn2_device(d_query, n_query, d_train_shard, n_train_shard, base_index, local, inner,
                                             workspace_bytes - (size_t)(inner - ws), stream);
    if (s != ORBX_OK) return s;
    // 16 bytes per query and rank: the only exchange of the search
    DIST_NCCL_TRY(ncclAllGather(local, parts, (size_t)n_query * sizeof(orbx_match2), ncclUint8, comm,
                                static_cast<hipStream_t>(stream)));
    return orbx_knn2_merge_device(parts, world, n_query, d_out, stream);
}

extern "C" orbx_status orbx_rccl_unique_id(uint8_t unique_id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    DIST_REQUIRE(unique_id, "null argument");
    ncclUniqueId id;
    DIST_NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(unique_id, &id, sizeof(id));
    return ORBX_OK;
}

extern "C" orbx_status orbx_rccl_comm_create(const uint8_t unique_id[128], int rank, int world_size, void** out) {
    DIST_REQUIRE(unique_id && out && world_size >= 1 && rank >= 0 && rank < world_size, "bad rank %d of %d", rank, world_size);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    DIST_NCCL_TRY(ncclCommInitRank(&comm, world_size, id, rank));
    *out = comm;
    return ORBX_OK;
}

extern "C" void orbx_rccl_comm_destroy(void* rccl_comm) {
    if (rccl_comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(rccl_comm));
}

// ---------------------------------------------------------------------------------------------------------------
struct orbx_extract_pool {
    std::vector<int> devices;
    std::vector<orbx_extractor*> extractors;
    int maxBatch = 1;
};

extern "C" orbx_status orbx_extract_pool_create(const orbx_params* params, const int* devices, int n_devices,
                                                orbx_extract_pool** out) {
    DIST_REQUIRE(params && out, "null argument");
    const int visible = orbx_device_count();
    if (visible < 1) { orbx_set_error("no HIP device"); return ORBX_ERR_HIP; }
    std::vector<int> devs;
    if (devices) {
        DIST_REQUIRE(n_devices >= 1, "empty device list");
        for (int i = 0; i < n_devices; ++i) {
            DIST_REQUIRE(devices[i] >= 0 && devices[i] < visible, "device %d of %d visible", devices[i], visible);
            DIST_REQUIRE(std::find(devs.begin(), devs.end(), devices[i]) == devs.end(), "device %d listed twice", devices[i]);
            devs.push_back(devices[i]);
        }
    } else {
        for (int d = 0; d < visible; ++d) devs.push_back(d);
    }
    int before = 0;
    (void)hipGetDevice(&before);
    orbx_extract_pool* pool = new orbx_extract_pool;
    pool->devices = devs;
    pool->maxBatch = std::max(params->max_batch, 1);
    orbx_status s = ORBX_OK;
    for (int d : devs) {
        orbx_params p = *params;
        p.device = d;
        orbx_extractor* ex = nullptr;
        s = orbx_extractor_create(&p, &ex);
        if (s != ORBX_OK) break;
        pool->extractors.push_back(ex);
    }
    (void)hipSetDevice(before);
    if (s != ORBX_OK) { orbx_extract_pool_destroy(pool); return s; }
    *out = pool;
    return ORBX_OK;
}

extern "C" void orbx_extract_pool_destroy(orbx_extract_pool* pool) {
    if (!pool) return;
    for (orbx_extractor* ex : pool->extractors) orbx_extractor_destroy(ex);
    delete pool;
}

extern "C" int orbx_extract_pool_devices(const orbx_extract_pool* pool) { return pool ? (int)pool->devices.size() : 0; }

extern "C" orbx_status orbx_extract_pool_batch(orbx_extract_pool* pool, const uint8_t* const* images, int n_frames,
                                               int width, int height, ptrdiff_t stride, int lap0, int lap1,
                                               orbx_keypoint* keypoints, uint8_t* descriptors, int capacity,
                                               int* n_out, int* mono_index_out) {
    DIST_REQUIRE(pool && n_frames >= 0, "null argument");
    if (n_frames == 0) return ORBX_OK;
    DIST_REQUIRE(images && keypoints && descriptors && n_out && mono_index_out && capacity >= 1, "null argument");
    const int nd = (int)pool->devices.size();
    const int per = (n_frames + nd - 1) / nd;
    std::vector<orbx_status> status((size_t)nd, ORBX_OK);
    std::vector<std::string> message((size_t)nd);
    auto work = [&](int k) {
        const int first = std::min(k * per, n_frames), last = std::min(first + per, n_frames);
        for (int f = first; f < last && status[k] == ORBX_OK; f += pool->maxBatch) {
            const int n = std::min(pool->maxBatch, last - f);
            status[k] = orbx_extract_batch(pool->extractors[k], images + f, n, width, height, stride, lap0, lap1,
                                           keypoints + (size_t)f * capacity, descriptors + (size_t)f * capacity * 32,
                                           capacity, n_out + f, mono_index_out + f);
        }
        if (status[k] != ORBX_OK) message[k] = orbx_last_error();     // the worker thread's message
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < nd; ++k)
        if (k * per < n_frames) threads.emplace_back(work, k);
    int before = 0;
    (void)hipGetDevice(&before);
    work(0);                                                            // the caller's thread drives the first device
    (void)hipSetDevice(before);
    for (std::thread& t : threads) t.join();
    for (int k = 0; k < nd; ++k)
        if (status[k] != ORBX_OK) {
            orbx_set_error("device %d: %s", pool->devices[k], message[k].c_str());
            return status[k];
        }
    return ORBX_OK;
}
