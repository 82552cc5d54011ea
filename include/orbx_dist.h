/*
 * orbx_dist -- the multi-GPU entry points of the ORB front-end, a second shared library
 * (liborbx_dist.so) on top of liborbx.so so that single-GPU users carry no RCCL dependency.
 * Links librccl.so.1 (the collectives run over xGMI between the GPUs of a node).
 *
 * The reference is a single-process CPU system and has no counterpart for either entry point; they
 * are how its two data-parallel loads spread over GPUs:
 *   - a map too large or too slow for one GPU (relocalisation / loop-closing candidates searched with
 *     cv::BFMatcher-style k = 2, Frame.cc:1309) is split into contiguous shards, one per rank;
 *   - independent frames (sequences, cameras) are split over the GPUs of one process.
 */
#ifndef ORBX_DIST_H_
#define ORBX_DIST_H_

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Sharded orbx_hamming_knn2_device.  One process per GPU; every rank passes the same queries and its own
 * contiguous shard of the map, rows [base_index, base_index + n_train_shard) of the whole.  Enqueued on `stream`:
 * the local top-2 with global indices, one ncclAllGather of the n_query 16-byte rows over `rccl_comm`
 * (an ncclComm_t of this rank), and the lexicographic merge of the gathered parts; every rank ends with the
 * result over the whole map in d_out, identical to a single-GPU search of the concatenated shards (ties go to the
 * lower global index).  A shard may be empty.  `d_workspace` holds orbx_knn2_sharded_workspace_bytes().
 * Does not synchronise; as with any collective, all ranks of the communicator must make the call.  Every check that
 * depends on the arguments (null pointers, alignment, the 32-bit index range of this rank's shard, the workspace size)
 * is made before anything is enqueued, so a rank that returns ORBX_ERR_INVALID has enqueued nothing; whenever any
 * rank returns a status other than ORBX_OK it has not entered the all-gather, the other ranks wait in it, and the
 * caller must abort the communicator on all ranks.  Verified with one rank on real RCCL and with 2 and 3 ranks over gloo through the
 * torch.distributed binding of the same protocol; not yet run with more than one RCCL rank. */
size_t orbx_knn2_sharded_workspace_bytes(int n_query, int64_t n_train_shard, int world_size);
orbx_status orbx_hamming_knn2_sharded_device(const uint8_t* d_query, int n_query, const uint8_t* d_train_shard,
                                             int64_t n_train_shard, int64_t base_index, void* rccl_comm,
                                             orbx_match2* d_out, void* d_workspace, size_t workspace_bytes,
                                             void* stream);

/* Communicator helpers for hosts that do not use RCCL themselves (thin wrappers of ncclGetUniqueId /
 * ncclCommInitRank / ncclCommDestroy).  unique_id: 128 bytes, created on rank 0 and handed to the other
 * ranks by the host's own means; the calling thread's current device is the rank's GPU. */
orbx_status orbx_rccl_unique_id(uint8_t unique_id[128]);
orbx_status orbx_rccl_comm_create(const uint8_t unique_id[128], int rank, int world_size, void** rccl_comm_out);
void orbx_rccl_comm_destroy(void* rccl_comm);

/* Frames spread over the GPUs of one process: one extractor, one host thread and one stream per device. */
typedef struct orbx_extract_pool orbx_extract_pool;
/* devices: HIP ordinals, or NULL for all visible devices (n_devices ignored).  params->device is ignored;
 * params->max_batch is the per-device batch. */
orbx_status orbx_extract_pool_create(const orbx_params* params, const int* devices, int n_devices,
                                     orbx_extract_pool** out);
void orbx_extract_pool_destroy(orbx_extract_pool* pool);
int orbx_extract_pool_devices(const orbx_extract_pool* pool);
/* The contract of orbx_extract_batch for n_frames host images (any n_frames >= 0): the frames are cut into
 * contiguous blocks, one per device, each block runs through that device's extractor in chunks of max_batch on
 * the device's own thread, all devices at once; results land in the caller's arrays in frame order, so they do
 * not depend on the number of devices. */
orbx_status orbx_extract_pool_batch(orbx_extract_pool* pool, const uint8_t* const* images, int n_frames, int width,
                                    int height, ptrdiff_t stride, int lap0, int lap1, orbx_keypoint* keypoints,
                                    uint8_t* descriptors, int capacity, int* n_out, int* mono_index_out);

#ifdef __cplusplus
}
#endif
#endif
