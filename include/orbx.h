/*
 * orbx -- C ABI of the MI355X (gfx950) ORB front-end: ORB extraction and Hamming
 * descriptor matching as HIP kernels.  This is the drop-in boundary for the two C++
 * classes the reference links into libORB_SLAM3.so:
 *
 *   ORB_SLAM3::ORBextractor   include/ORBextractor.h:42-121, src/ORBextractor.cc
 *   ORB_SLAM3::ORBmatcher     include/ORBmatcher.h:34-226,   src/ORBmatcher.cc
 *   + the Frame members that are part of the same data-parallel path
 *     (Frame::ComputeStereoMatches src/Frame.cc:962-1133,
 *      Frame::ComputeStereoFishEyeMatches src/Frame.cc:1286-1338,
 *      Frame::AssignFeaturesToGrid / GetFeaturesInArea src/Frame.cc:468-506, :792-858).
 *
 * Conventions
 *  - plain C: opaque handles, POD structs, pointers + sizes; no exceptions cross it;
 *  - every function returns an orbx_status (0 = OK, negative = error) unless noted;
 *    orbx_last_error() gives the message of the calling thread's last failure;
 *  - "host" entry points take host pointers and are synchronous; "_device" entry points
 *    take device pointers, enqueue on the given hipStream_t (passed as void* and taken
 *    literally: NULL is the device's default stream) and do not synchronise; work that
 *    consumes the results of two handles (stereo matching) must be ordered after both by
 *    the caller (same stream, or an event wait);
 *  - one handle may be used by one thread at a time; distinct handles are independent
 *    (the reference runs one extractor per camera thread, Frame.cc:143-147);
 *  - orbx_keypoint has the memory layout of cv::KeyPoint (28 bytes) and descriptors are
 *    row-major N x 32 bytes, so std::vector<cv::KeyPoint>::data() and cv::Mat::data can be
 *    passed straight through.
 */
#ifndef ORBX_H_
#define ORBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int orbx_status;
enum {
    ORBX_OK = 0,
    ORBX_ERR_INVALID = -1,   /* bad argument / unsupported geometry                      */
    ORBX_ERR_CAPACITY = -2,  /* a caller- or handle-provided capacity was too small      */
    ORBX_ERR_HIP = -3,       /* a HIP runtime call failed (no device, out of memory ...) */
    ORBX_ERR_EMPTY = -4      /* empty image (the reference's operator() returns -1)      */
};

/* cv::KeyPoint layout. */
typedef struct orbx_keypoint {
    float x, y;      /* pt, level-0 pixel coordinates                          */
    float size;      /* (int)(31 * scale[octave])                              */
    float angle;     /* degrees, [0, 360)                                      */
    float response;  /* FAST score                                             */
    int octave;      /* pyramid level                                          */
    int class_id;    /* always -1                                              */
} orbx_keypoint;

/* cv::Rect2i layout (ORBextractor::mvDynamicArea, ORBextractor.h:95-96). */
typedef struct orbx_rect { int x, y, width, height; } orbx_rect;

/* Two nearest neighbours of one query (cv::DMatch pair of knnMatch(k=2)).
 * Missing neighbours have idx = -1 and dist = INT32_MAX. */
typedef struct orbx_match2 { int32_t idx0, dist0, idx1, dist1; } orbx_match2;

/* Constructor arguments of ORBextractor (ORBextractor.cc:477-479) plus buffer sizing. */
typedef struct orbx_params {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int max_width;    /* largest image the handle will see                    */
    int max_height;
    int max_batch;    /* frames per batched call (>= 1)                       */
    int device;       /* HIP device ordinal, -1 = current device              */
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

const char* orbx_last_error(void);
/* Number of visible HIP devices, or a negative status. */
int orbx_device_count(void);

/* ---------------------------------------------------------------- extractor -------- */

/* ORBextractor::ORBextractor (ORBextractor.cc:477-536). */
orbx_status orbx_extractor_create(const orbx_params* params, orbx_extractor** out);
void orbx_extractor_destroy(orbx_extractor* ex);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (ORBextractor.h:73-91) and mnFeaturesPerLevel;
 * each array has nlevels entries; any pointer may be NULL. */
orbx_status orbx_extractor_tables(const orbx_extractor* ex, float* scale, float* inv_scale,
                                  float* sigma2, float* inv_sigma2, int* features_per_level);

/* The same tables from the constructor arguments alone (ORBextractor.cc:481-512): host arithmetic, no device and
 * no handle needed.  Any pointer may be NULL. */
orbx_status orbx_scale_tables(int nfeatures, float scale_factor, int nlevels, float* scale, float* inv_scale,
                              float* sigma2, float* inv_sigma2, int* features_per_level);

/* Largest number of keypoints one frame of the current image size can yield (>= nfeatures; at creation the
 * size is max_width x max_height); size output buffers with it.  DistributeOctTree keeps one keypoint per
 * non-empty root node even when a level's quota is smaller (ORBextractor.cc:665-700), and a wider aspect ratio has
 * more roots: after a call with another size the value may be larger, and a call whose `capacity` was too small
 * returns ORBX_ERR_CAPACITY; ask again and repeat the call with larger buffers. */
int orbx_extractor_max_keypoints(const orbx_extractor* ex);

/* ORBextractor::operator() (ORBextractor.cc:1271-1399) for one host image.
 * image: 8-bit single channel, `stride` bytes per row.  lap0/lap1 = vLappingArea.
 * dyn/ndyn = mvDynamicArea (may be NULL/0).  keypoints/descriptors: caller buffers of
 * `capacity` rows.  *n_out = number of keypoints, *mono_index_out = the return value of
 * operator().  Returns ORBX_ERR_EMPTY for a NULL / zero-sized image. */
orbx_status orbx_extract(orbx_extractor* ex, const uint8_t* image, int width, int height, ptrdiff_t stride,
                         int lap0, int lap1, const orbx_rect* dyn, int ndyn,
                         orbx_keypoint* keypoints, uint8_t* descriptors, int capacity,
                         int* n_out, int* mono_index_out);

/* Batched operator() over n_frames host images of identical size (one call = one
 * pass of every kernel over the whole batch).  images[i] points at frame i.  Outputs are
 * n_frames x capacity rows; n_out / mono_index_out have n_frames entries. */
orbx_status orbx_extract_batch(orbx_extractor* ex, const uint8_t* const* images, int n_frames,
                               int width, int height, ptrdiff_t stride, int lap0, int lap1,
                               orbx_keypoint* keypoints, uint8_t* descriptors, int capacity,
                               int* n_out, int* mono_index_out);

/* Same, device resident and asynchronous: d_images = n_frames images, `frame_stride`
 * bytes apart; d_keypoints / d_descriptors = n_frames x capacity rows; d_counts =
 * n_frames x {n, mono_index}.  Check orbx_extract_status() after synchronising. */
orbx_status orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, int n_frames,
                                      int width, int height, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                      int lap0, int lap1, const orbx_rect* dyn_host, int ndyn,
                                      orbx_keypoint* d_keypoints, uint8_t* d_descriptors, int capacity,
                                      int* d_counts, void* stream);

/* ---- input staging in front of the extractor ---------------------------------------
 * What the reference does to a camera image before ORBextractor::operator() sees it:
 *   cv::remap(im, out, M1, M2, cv::INTER_LINEAR)   System.cc:361-362, when Settings asks for
 *                                                  rectification (maps from initUndistortRectifyMap,
 *                                                  CV_32FC1, border constant 0), OR
 *   cv::resize(im, out, newImSize)                 System.cc:367-370 (INTER_LINEAR), OR nothing,
 * on the image as delivered (1, 3 or 4 interleaved 8-bit channels), followed by
 *   cvtColor(im, im, COLOR_{RGB|BGR}[A]2GRAY)      Tracking.cc:1595-1618 (mbRGB picks the order).
 * A stager holds the per-camera constants (the maps converted to cv::remap's fixed-point form,
 * or the resize taps) on the device; one kernel produces the grey image.  map_x / map_y are
 * host arrays of dst_width x dst_height floats (row-major) and are only read by
 * orbx_stager_create.  For ORBX_STAGE_NONE the output size is the source size. */
enum { ORBX_STAGE_NONE = 0, ORBX_STAGE_REMAP = 1, ORBX_STAGE_RESIZE = 2 };
typedef struct orbx_stage_params {
    int src_width, src_height;
    int channels;           /* 1, 3 or 4                                            */
    int rgb_order;          /* Tracking::mbRGB: 1 = R first, 0 = B first            */
    int mode;               /* ORBX_STAGE_*                                         */
    int dst_width, dst_height;
    const float* map_x;     /* ORBX_STAGE_REMAP only                                */
    const float* map_y;
    int device;
} orbx_stage_params;
typedef struct orbx_stager orbx_stager;
orbx_status orbx_stager_create(const orbx_stage_params* params, orbx_stager** out);
void orbx_stager_destroy(orbx_stager* stager);
orbx_status orbx_stager_output_size(const orbx_stager* stager, int* width, int* height);
/* One host image in, the grey image out (host pointers, strides in bytes; synchronous). */
orbx_status orbx_stage(orbx_stager* stager, const uint8_t* src, ptrdiff_t src_stride, uint8_t* dst,
                       ptrdiff_t dst_stride);
/* n_frames device images in, grey device images out; asynchronous on `stream`. */
orbx_status orbx_stage_batch_device(orbx_stager* stager, const uint8_t* d_src, int n_frames,
                                    ptrdiff_t src_row_stride, ptrdiff_t src_frame_stride, uint8_t* d_dst,
                                    ptrdiff_t dst_row_stride, ptrdiff_t dst_frame_stride, void* stream);
/* orbx_extract_batch_device on staged input: the stager writes level 0 of the pyramid directly
 * (no intermediate grey image, no level-0 copy); d_src are the images as delivered.  The
 * extractor works at the stager's output size. */
orbx_status orbx_extract_batch_staged_device(orbx_extractor* ex, orbx_stager* stager, const uint8_t* d_src,
                                             int n_frames, ptrdiff_t src_row_stride, ptrdiff_t src_frame_stride,
                                             int lap0, int lap1, const orbx_rect* dyn_host, int ndyn,
                                             orbx_keypoint* d_keypoints, uint8_t* d_descriptors, int capacity,
                                             int* d_counts, void* stream);

/* After the stream of the last batched device call has been synchronised: ORBX_OK, or
 * ORBX_ERR_CAPACITY when an internal candidate / output buffer overflowed (results of
 * that call are then invalid; the handle has grown its buffers, call again). */
orbx_status orbx_extract_status(orbx_extractor* ex);

/* ORBextractor::mvImagePyramid (ORBextractor.h:93) of the last call: copies level
 * `level` of batch slot `frame` (borderless, w x h, tightly packed) to a host buffer of
 * `capacity` bytes.  out may be NULL to query the size only. */
orbx_status orbx_pyramid_level(orbx_extractor* ex, int frame, int level, uint8_t* out, size_t capacity,
                               int* width, int* height);
/* Device view of the same level: pointer and row pitch in bytes. */
orbx_status orbx_pyramid_level_device(orbx_extractor* ex, int frame, int level, const uint8_t** d_ptr,
                                      int* width, int* height, ptrdiff_t* pitch);

/* Per-kernel timing of the extractor with hipEvents recorded on the launch stream.
 * set_timing(1) starts (and clears) the accumulation, get_timing() synchronises on the last
 * recorded event and returns, summed over n_calls batches, the milliseconds of the 6 stages
 * { copy-in, pyramid (all resize launches), FAST cells, quadtree, layout, describe }. */
orbx_status orbx_extractor_set_timing(orbx_extractor* ex, int enable);
orbx_status orbx_extractor_get_timing(orbx_extractor* ex, double* stage_ms, int n_stages, long* n_calls);

/* Stage taps for tests (device results copied to the host, last call, batch slot `frame`):
 * FAST candidates of one level after the per-cell round trip, as rows of
 * {x, y, response, order_key} floats / ints (order_key sorts them in the reference's
 * cell-by-cell emission order); returns the count through *n_out. */
orbx_status orbx_debug_level_candidates(orbx_extractor* ex, int frame, int level, float* xyr, uint32_t* order_key,
                                        int capacity, int* n_out);
/* cv::GaussianBlur(mvImagePyramid[level], 7x7, sigma 2, BORDER_REFLECT_101) (ORBextractor.cc:1344-1351) as the last call
 * left it in frame slot `frame`: width * height bytes, rows packed.  Only a handle whose keypoint quota is dense enough
 * that it blurs whole levels keeps such an image (ORBX_BLUR_MODE=level forces that); ORBX_ERR_INVALID otherwise. */
orbx_status orbx_debug_blurred_level(orbx_extractor* ex, int frame, int level, uint8_t* out, size_t capacity,
                                     int* width, int* height);
/* Keypoints of one level after quadtree distribution, in node-list order, level coordinates. */
orbx_status orbx_debug_level_keypoints(orbx_extractor* ex, int frame, int level, float* xyr, int capacity,
                                       int* n_out);

/* The sort of DistributeOctTree's careful phase (ORBextractor.cc:805, std::sort over nodes that
 * tie) on its own: n_arrays arrays of 64-bit elements (key = bits 63..24, payload = bits 23..0),
 * array a = elems[offsets[a] .. offsets[a+1]), at most 2048 elements each.  device_sorted receives
 * the result of the device sort (one workgroup, pending ranges spread over its waves), host_sorted the result of the sequential restatement
 * of libstdc++'s algorithm run on the host; the two must be identical.  Host pointers. */
orbx_status orbx_debug_wave_sort(const uint64_t* elems, const int* offsets, int n_arrays,
                                 uint64_t* device_sorted, uint64_t* host_sorted);

/* ---------------------------------------------------------------- matcher ---------- */

/* The synchronous matcher entry points below keep, per calling thread, their temporary device blocks
 * (at most 2 GiB), a pinned staging arena, scratch frames and one stream per device instead of
 * creating them per call; this gives all of them back for the calling thread.  The same happens by
 * itself when a thread ends.  No counterpart in the reference. */
void orbx_release_cached_memory(void);

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2141-2157) for n pairs: out[i] =
 * popcount(a[i] xor b[i]) over 256 bits.  Host pointers. */
orbx_status orbx_descriptor_distance(const uint8_t* a, const uint8_t* b, int n, int* out);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) (Frame.cc:46, :1309): the two
 * smallest distances per query, ties towards the lower train index.  Host pointers. */
orbx_status orbx_hamming_knn2(const uint8_t* query, int n_query, const uint8_t* train, int64_t n_train,
                              orbx_match2* out);
/* Device-resident variant; train indices are reported as base_index + row, so a shard of a
 * larger map reports global indices.  `workspace` (16-byte aligned) must hold orbx_knn2_workspace_bytes(), which
 * depends on the compute-unit count of the device that is CURRENT when it is asked: ask with the search's device
 * current (a smaller figure from another device makes the search return ORBX_ERR_INVALID, never overrun). */
size_t orbx_knn2_workspace_bytes(int n_query, int64_t n_train);
orbx_status orbx_hamming_knn2_device(const uint8_t* d_query, int n_query, const uint8_t* d_train, int64_t n_train,
                                     int64_t base_index, orbx_match2* d_out, void* d_workspace, size_t workspace_bytes,
                                     void* stream);
/* orbx_hamming_knn2_device for callers that cannot allocate device memory (code under HIP graph
 * capture, serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * partial results and span counters live in a fixed 16 MiB array inside the library's device image (one copy
 * per GPU, held until the process ends) instead of a caller workspace, so the call performs no
 * hipMalloc / hipFree.  n_query is limited to orbx_knn2_static_max_queries() (1 047 552); larger
 * inputs return ORBX_ERR_INVALID.  Results are identical to orbx_hamming_knn2_device.  All calls on
 * one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two threads)
 * race; order them.  The array's address is looked up on the first call per device (or by
 * orbx_knn2_static_max_queries(), which a caller about to capture a graph calls beforehand). */
size_t orbx_knn2_static_max_queries(void);
orbx_status orbx_hamming_knn2_device_static(const uint8_t* d_query, int n_query, const uint8_t* d_train,
                                            int64_t n_train, int64_t base_index, orbx_match2* d_out, void* stream);
/* Merge n_parts per-shard results (n_parts x n_query, shard-major) into the global top-2,
 * lexicographic on (dist, idx): the step after an all-gather of per-GPU shards. */
orbx_status orbx_knn2_merge_device(const orbx_match2* d_parts, int n_parts, int n_query, orbx_match2* d_out,
                                   void* stream);

/* Frame::ComputeStereoMatches (Frame.cc:962-1133) on the pyramids held by two extractor
 * handles (batch slot `frame` of each; left/right keypoints as returned by orbx_extract*).
 * mbf = baseline * fx, mb = baseline (the reference's minZ; maxD = mbf / mb).
 * u_right / depth: n_left floats, -1 where no match.  Host pointers, synchronous. */
orbx_status orbx_stereo_match(orbx_extractor* left, orbx_extractor* right, int frame,
                              const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left,
                              const orbx_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                              float mbf, float mb, float* u_right, float* depth);
/* Batched device variant over the frames of the last orbx_extract_batch_device calls on
 * both handles: inputs are the d_keypoints / d_descriptors / d_counts buffers of those
 * calls (same capacity); outputs n_frames x capacity floats. */
orbx_status orbx_stereo_match_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames,
                                           const orbx_keypoint* d_kps_left, const uint8_t* d_desc_left,
                                           const int* d_counts_left,
                                           const orbx_keypoint* d_kps_right, const uint8_t* d_desc_right,
                                           const int* d_counts_right, int capacity,
                                           float mbf, float mb, float* d_u_right, float* d_depth, void* stream);

/* Windowed projection search core shared by ORBmatcher::SearchByProjection
 * (ORBmatcher.cc:42-216) and its siblings: for every query (projected map point) the
 * keypoints returned by Frame::GetFeaturesInArea(x, y, radius, min_level, max_level)
 * (Frame.cc:792-858) over the 64 x 48 grid built by AssignFeaturesToGrid with the image
 * bounds {min_x, max_x, min_y, max_y}, in the reference's (cell-x, cell-y, insertion)
 * order, each with its Hamming distance to the query descriptor.  A candidate is dropped
 * when uright != NULL, uright[idx] > 0 and |query_xr - uright[idx]| > radius (the stereo
 * gate of ORBmatcher.cc:92-97).  Output is CSR: cand_offset[n_query+1], then cand_idx /
 * cand_dist with `capacity` entries; *n_cand_out = total (ORBX_ERR_CAPACITY if larger).
 * Host pointers. */
typedef struct orbx_query {
    float x, y;         /* projection                                          */
    float xr;           /* projected right-image u (stereo gate), unused if no uright */
    float radius;       /* search radius in pixels                             */
    int min_level;      /* as GetFeaturesInArea: -1 / <=0 and max<0 = no check */
    int max_level;
} orbx_query;
orbx_status orbx_window_candidates(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                                   float min_x, float max_x, float min_y, float max_y,
                                   const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                   int* cand_offset, int* cand_idx, int* cand_dist, int capacity, int* n_cand_out);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) for the mono /
 * rectified case (ORBmatcher.cc:42-147): device candidates + the greedy in-order resolve
 * (best / second best with levels, TH_HIGH = 100, ratio test only when both are on the same
 * level).  taken[idx] != 0 marks keypoints that already hold a map point with observations;
 * accepted matches set it.  match_out[q] = keypoint index or -1.  Returns the number of
 * matches through *n_matches. */
orbx_status orbx_search_by_projection(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                                      float min_x, float max_x, float min_y, float max_y,
                                      const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                      float nn_ratio, uint8_t* taken, int* match_out, int* n_matches);

/* Distance core of ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1051-1167): for two
 * keyframes whose descriptors are bucketed by shared vocabulary node (CSR: off[n_buckets+1]
 * into idx arrays), for every idx1 the LAST idx2 of the same bucket with the smallest
 * distance <= TH_LOW (50); best_idx2 = -1 when none.  Host pointers. */
orbx_status orbx_bucket_best(const uint8_t* desc1, int n_desc1, const int* idx1, const int* off1,
                             const uint8_t* desc2, int n_desc2, const int* idx2, const int* off2, int n_buckets,
                             int* best_idx2, int* best_dist);

/* ---- order-dependent matcher loops: device candidate lists + in-order replay ----------
 * The reference's matchers are loops whose iterations depend on each other (a keypoint that
 * received a map point is skipped by later queries).  Each function below evaluates every
 * window test, stateless gate and Hamming distance on the device and replays the loop of the
 * cited reference function on the host, in its order, over the resulting short lists.  The
 * caller supplies what the reference computes from its object graph before the loop: the
 * projections, predicted levels and radii as orbx_query records, the map points' descriptors,
 * and the "skip" flags.  Host pointers; synchronous. */

/* Best-only windowed search.  For every query in order: the closest candidate (strict <,
 * first seen wins ties) among those with taken[idx] == 0; accepted when dist <= max_dist.
 * An accepted keypoint gets taken[idx] = query_marks[q] (1 when query_marks == NULL; pass 0
 * for a map point without observations, which the reference does not treat as occupying the
 * keypoint, ORBmatcher.cc:1832-1834).  taken == NULL: stateless.  With query_angle != NULL the
 * rotation-consistency histogram (30 bins, bin = round(rot * rot_factor), 3 dominant bins with
 * the 10 % rule, ORBmatcher.cc:2096-2137) prunes the matches afterwards.
 *   kp_match[n_kps]  query that owns each keypoint at the end (-1 none)
 *   match_out[q]     keypoint of query q if it still owns it, else -1
 *   *n_matches       the reference's return value (counts overwritten assignments too)
 * This is the loop of
 *   SearchByProjection(Frame&, const Frame&, th, bMono)          ORBmatcher.cc:1761-1972  max_dist 100, factor 1/30
 *   SearchByProjection(Frame&, KeyFrame*, sFound, th, ORBdist)   :1974-2094               max_dist ORBdist, factor 1/30
 *   SearchByProjection(KeyFrame*, Sim3, points, matched, th, r)  :476-580, :582-693        max_dist 50 * r, no angles
 *   Fuse(KeyFrame*, Sim3, points, th, replace)                   :1409-1524               taken NULL, max_dist 50
 * for the mono / rectified camera (Nleft == -1). */
orbx_status orbx_search_window_best(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                                    float min_x, float max_x, float min_y, float max_y,
                                    const orbx_query* queries, const uint8_t* query_desc, int n_query,
                                    float max_dist, uint8_t* taken, const uint8_t* query_marks,
                                    const float* query_angle, float rot_factor,
                                    int* match_out, int* kp_match, int* n_matches);

/* Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) (ORBmatcher.cc:1214-1407), matching
 * part: per projected map point {x, y, xr = projected right u, radius, pred-1, pred} the closest
 * keypoint that passes the reprojection gate e2 * inv_level_sigma2[octave] <= 7.8 (keypoint with
 * uright >= 0, e2 includes the right coordinate) or <= 5.99; accepted when dist <= 50.
 * match_out[q] = keypoint or -1; what happens to the map (Replace / AddObservation) stays with
 * the caller. */
orbx_status orbx_fuse_search(const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps,
                             float min_x, float max_x, float min_y, float max_y,
                             const orbx_query* queries, const uint8_t* query_desc, int n_query,
                             const float* inv_level_sigma2, int n_levels, int* match_out, int* n_matches);

/* SearchBySim3 (ORBmatcher.cc:1526-1743): the map points of keyframe 1 searched in keyframe 2
 * (queries_into2[q] belongs to keypoint source1[q] of keyframe 1) and vice versa, best-only,
 * accepted when dist <= 100, then mutual agreement.  bounds = {min_x, max_x, min_y, max_y} of
 * each keyframe's grid.  matches12[n1] = keypoint of keyframe 2 or -1. */
orbx_status orbx_search_by_sim3(const orbx_keypoint* kps1, const uint8_t* desc1, int n1, const float bounds1[4],
                                const orbx_keypoint* kps2, const uint8_t* desc2, int n2, const float bounds2[4],
                                const orbx_query* queries_into2, const uint8_t* query_desc_into2, const int* source1, int nq1,
                                const orbx_query* queries_into1, const uint8_t* query_desc_into1, const int* source2, int nq2,
                                int* matches12, int* n_found);

/* SearchForInitialization (ORBmatcher.cc:695-808): level-0 keypoints of frame 1 searched in
 * frame 2 inside window_size around prev_matched[i1] ({x, y} pairs, updated on return), best /
 * second best with the per-keypoint matched-distance rule (:734), accepted when best <= 50 and
 * best < second * nn_ratio, re-assignment of keypoints of frame 2 (:753-760), rotation histogram
 * with factor 1/30.  matches12[n1] = keypoint of frame 2 or -1. */
orbx_status orbx_search_for_initialization(const orbx_keypoint* kps1, const uint8_t* desc1, int n1,
                                           const orbx_keypoint* kps2, const uint8_t* desc2, int n2,
                                           float min_x, float max_x, float min_y, float max_y,
                                           float* prev_matched, int window_size, float nn_ratio, int check_orientation,
                                           int* matches12, int* n_matches);

/* ---- frame handle: features resident on the device --------------------------------------
 * The reference's Frame / KeyFrame keep mvKeysUn, mDescriptors, mvuRight and the 64 x 48 mGrid
 * for as long as the object lives, and every search of that frame reads them.  An orbx_frame is
 * the same state on the device: orbx_frame_assign uploads the (undistorted) keypoints,
 * descriptors and right coordinates once and builds the grid there (Frame::AssignFeaturesToGrid,
 * Frame.cc:468-506 with PosInGrid :860-870: cell = round((x - min_x) * 64 / (max_x - min_x)),
 * likewise y; keypoints outside the cells are left out; indices ascend inside a cell).  The
 * orbx_frame_* searches below are the loops of the functions of the same name without the
 * "frame" (same arguments, results and order), minus the per-call upload.  The host-array
 * functions above are these, run on a frame that lives for the call.
 * A frame is immutable between assigns and may be searched from several threads at once;
 * assign / destroy need exclusive access.  uright may be NULL (monocular).  The features live on
 * the device that is current when assign is called; a search makes that device current for the
 * calling thread.
 * orbx_frame_assign_device takes device pointers (for instance the output of
 * orbx_extract_batch_device and orbx_stereo_match_batch_device for a camera whose keypoints need
 * no undistortion, Frame.cc:884-888) and copies them; it returns when the copy is complete. */
typedef struct orbx_frame orbx_frame;
orbx_status orbx_frame_create(orbx_frame** out);
void orbx_frame_destroy(orbx_frame* frame);
orbx_status orbx_frame_assign(orbx_frame* frame, const orbx_keypoint* kps, const uint8_t* desc, const float* uright,
                              int n_kps, float min_x, float max_x, float min_y, float max_y);
orbx_status orbx_frame_assign_device(orbx_frame* frame, const orbx_keypoint* d_kps, const uint8_t* d_desc,
                                     const float* d_uright, int n_kps, float min_x, float max_x, float min_y,
                                     float max_y);
/* Frame::UndistortKeyPoints (Frame.cc:882-913) and Frame::ComputeImageBounds (:916-947) in front of the grid, for
 * a pinhole camera with radial-tangential distortion: cv::undistortPoints(pts, pts, K, D, Mat(), K), i.e. five
 * iterations of the inverse model in binary64 and re-projection with the same K.  dist = {k1, k2, p1, p2, k3, k4,
 * k5, k6} (n_dist of them, the rest 0).  As in the reference, dist[0] == 0 means "no distortion": keypoints are
 * taken as they are and the bounds are the image.  The frame then holds mvKeysUn; orbx_frame_keypoints returns
 * them together with {mnMinX, mnMaxX, mnMinY, mnMaxY}.  orbx_image_bounds is ComputeImageBounds on its own. */
typedef struct orbx_camera {
    float fx, fy, cx, cy;
    float dist[8];
    int n_dist;
} orbx_camera;
orbx_status orbx_image_bounds(const orbx_camera* cam, int width, int height, float bounds[4]);
orbx_status orbx_frame_assign_distorted(orbx_frame* frame, const orbx_camera* cam, int width, int height,
                                        const orbx_keypoint* kps, const uint8_t* desc, const float* uright, int n_kps);
orbx_status orbx_frame_assign_distorted_device(orbx_frame* frame, const orbx_camera* cam, int width, int height,
                                               const orbx_keypoint* d_kps, const uint8_t* d_desc,
                                               const float* d_uright, int n_kps);
orbx_status orbx_frame_keypoints(const orbx_frame* frame, orbx_keypoint* kps_un, int capacity, float bounds[4]);
int orbx_frame_size(const orbx_frame* frame);
/* mGrid as CSR (Frame.h:257): cell_start[64 * 48 + 1] with cell = column * 48 + row, cell_idx[]
 * = keypoint indices, capacity entries; *n_in_grid = keypoints inside the cells. */
orbx_status orbx_frame_grid(const orbx_frame* frame, int* cell_start, int* cell_idx, int capacity, int* n_in_grid);
orbx_status orbx_frame_window_candidates(const orbx_frame* frame, const orbx_query* queries, const uint8_t* query_desc,
                                         int n_query, int* cand_offset, int* cand_idx, int* cand_dist, int capacity,
                                         int* n_cand_out);
orbx_status orbx_frame_search_by_projection(const orbx_frame* frame, const orbx_query* queries,
                                            const uint8_t* query_desc, int n_query, float nn_ratio, uint8_t* taken,
                                            int* match_out, int* n_matches);
orbx_status orbx_frame_search_window_best(const orbx_frame* frame, const orbx_query* queries, const uint8_t* query_desc,
                                          int n_query, float max_dist, uint8_t* taken, const uint8_t* query_marks,
                                          const float* query_angle, float rot_factor, int* match_out, int* kp_match,
                                          int* n_matches);
orbx_status orbx_frame_fuse_search(const orbx_frame* frame, const orbx_query* queries, const uint8_t* query_desc,
                                   int n_query, const float* inv_level_sigma2, int n_levels, int* match_out,
                                   int* n_matches);

/* ---- frames with a second camera (Frame::Nleft != -1: fisheye stereo) -----------------------
 * Such a Frame keeps the left keypoints in mvKeys / mGrid, the right ones in mvKeysRight / mGridRight, and indexes
 * mvpMapPoints / mDescriptors with left keypoints first and right keypoints at Nleft + i (Frame.cc:1265-1283).
 * Here each camera is an orbx_frame of its own (assigned without uright: the rectified-stereo gate does not apply,
 * ORBmatcher.cc:92); `taken` and `kp_query` run over the combined index (n_left + n_right entries).
 * `right` may be NULL (one camera).
 *
 * SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) with both passes (ORBmatcher.cc:42-216).
 * queries_left[q]: projection into the left image, radius = RadiusByViewingCos * th * scale[level],
 * min_level / max_level = level - 1 / level; radius <= 0 = not in the left view (mbTrackInView false).
 * queries_right[q]: the same for the right image with radius = RadiusByViewingCos * scale[level] (no th, :151-154);
 * radius <= 0 = mbTrackInViewR false or level -1.  left_to_right / right_to_left = mvLeftToRightMatch /
 * mvRightToLeftMatch (-1 none; may be NULL): an accepted keypoint hands the map point to its stereo partner too.
 * query_marks[q] != 0: the map point has observations (only then do later queries skip a keypoint it occupies,
 * :88-90); NULL = all 1.  kp_query[idx] = the last query assigned to keypoint idx (-1 none); *n_matches = the
 * reference's return value. */
orbx_status orbx_frame_search_by_projection_stereo(const orbx_frame* left, const orbx_frame* right,
                                                   const orbx_query* queries_left, const orbx_query* queries_right,
                                                   const uint8_t* query_desc, int n_query, float nn_ratio,
                                                   const int* left_to_right, const int* right_to_left,
                                                   uint8_t* taken, const uint8_t* query_marks, int* kp_query, int* n_matches);

/* SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) with the right-image pass and the rotation histogram
 * shared by both passes (ORBmatcher.cc:1761-1972).  queries_left[q].radius <= 0 = the map point is skipped before
 * either pass (:1795-1803); a point whose left window holds no keypoint skips the right pass as well (:1820-1821).
 * Accept per pass: best <= max_dist (TH_HIGH = 100 in the reference).  With query_angle the matches of the
 * non-dominant rotation bins are removed afterwards (kp_query[idx] = -1, counted down). */
orbx_status orbx_frame_search_window_best_stereo(const orbx_frame* left, const orbx_frame* right,
                                                 const orbx_query* queries_left, const orbx_query* queries_right,
                                                 const uint8_t* query_desc, int n_query, float max_dist,
                                                 uint8_t* taken, const uint8_t* query_marks, const float* query_angle,
                                                 float rot_factor, int* kp_query, int* n_matches);

/* All Hamming distances between the entries of the same vocabulary node of two feature vectors
 * (the inner loops of SearchByBoW / SearchForTriangulation, ORBmatcher.cc:273-363, :856-907,
 * :1030-1078).  Buckets are CSR: entries off[b] .. off[b+1] of idx1 / idx2 are descriptor rows.
 * Row e (one per idx1 entry) = distances to the idx2 entries of its bucket in bucket order:
 * dist[row_start[e] + j]; row_start has off1[n_buckets] + 1 entries. */
orbx_status orbx_bucket_allpairs(const uint8_t* desc1, int n_desc1, const int* idx1, const int* off1,
                                 const uint8_t* desc2, int n_desc2, const int* idx2, const int* off2, int n_buckets,
                                 int* row_start, uint16_t* dist, size_t capacity, size_t* n_pairs_out);

/* SearchByBoW(KeyFrame*, Frame&, matches) (ORBmatcher.cc:235-474).  valid_kf[i] != 0: keypoint i
 * of the keyframe has a map point that is not bad.  n_left_f = Frame::Nleft (-1: one camera;
 * otherwise frame keypoints >= n_left_f form the right-camera track, whose ratio test the
 * reference disables, :404).  Accept: best <= 50 and best < nn_ratio * second; rotation
 * histogram with this fork's factor 30/360 (:255).  match_f[n_f] = keyframe keypoint whose map
 * point the frame keypoint receives, or -1. */
orbx_status orbx_search_by_bow(const uint8_t* desc_kf, const float* angle_kf, const uint8_t* valid_kf, int n_kf,
                               const int* idx_kf, const int* off_kf,
                               const uint8_t* desc_f, const float* angle_f, int n_f, int n_left_f,
                               const int* idx_f, const int* off_f, int n_buckets,
                               float nn_ratio, int check_orientation, int* match_f, int* n_matches);

/* SearchByBoW(KeyFrame*, KeyFrame*, matches12) (ORBmatcher.cc:824-966): valid1 / valid2 mark the
 * keypoints with a good map point; a keypoint of keyframe 2 is used once; accept: best < 50
 * (strict) and the ratio test; rotation histogram with factor 1/30.  matches12[n1] = keypoint of
 * keyframe 2 or -1. */
orbx_status orbx_search_by_bow_keyframes(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                                         const int* idx1, const int* off1,
                                         const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                                         const int* idx2, const int* off2, int n_buckets,
                                         float nn_ratio, int check_orientation, int* matches12, int* n_matches);

/* SearchForTriangulation (ORBmatcher.cc:968-1212) for pinhole cameras without a second camera
 * model.  free1 / free2 != 0: the keypoint has no map point yet; uright (may be NULL) >= 0 marks
 * stereo keypoints.  Per candidate, in bucket order: dist <= 50 and <= running best, then the
 * epipole-distance gate (:1089-1097), then (unless coarse) Pinhole::epipolarConstrain
 * (Pinhole.cpp:118-141) with the caller's F12 (row-major 3x3, = K1^-T [t12]x R12 K2^-1); a later
 * candidate with an equal distance replaces an earlier one.  Rotation histogram (factor 1/30)
 * when check_orientation.  matches12[n1] = keypoint of keyframe 2 or -1. */
typedef struct orbx_epipolar {
    float F12[9];
    float epipole_x, epipole_y;        /* projection of camera centre 1 into image 2 */
    const float* scale_factors2;       /* mvScaleFactors of keyframe 2 */
    const float* level_sigma2_2;       /* mvLevelSigma2 of keyframe 2 */
    int n_levels;
} orbx_epipolar;
orbx_status orbx_search_for_triangulation(const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* free1,
                                          const float* uright1, int n1, const int* idx1, const int* off1,
                                          const orbx_keypoint* kps2, const uint8_t* desc2, const uint8_t* free2,
                                          const float* uright2, int n2, const int* idx2, const int* off2, int n_buckets,
                                          const orbx_epipolar* geom, int only_stereo, int coarse, int check_orientation,
                                          int* matches12, int* n_matches);

/* SearchForTriangulation with the per-candidate camera selection of keyframes that carry a second camera
 * (ORBmatcher.cc:989-1006, :1099-1139).  Keypoint / descriptor arrays are combined (left first, right at n_left + i).
 * n_left1 / n_left2 = KeyFrame::NLeft (-1: one camera; then cam[0], pose[0][0] and the epipole gate are used, and
 * uright marks stereo keypoints).  pose[right1][right2] = {R12, t12} of the camera pair (Tll, Tlr, Trl, Trr),
 * cam1[0/1] / cam2[0/1] = mpCamera / mpCamera2 of each keyframe.  The gate is pCamera1->epipolarConstrain(pCamera2, kp1,
 * kp2, R12, t12, sigma1[kp1.octave], sigma2[kp2.octave]): Pinhole (Pinhole.cpp:118-141, with F12 = K1^-T [t12]x R12
 * K2^-1) or KannalaBrandt8 (KannalaBrandt8.cpp:227-232, :323-397: unproject, parallax < 0.9998, triangulation,
 * positive depths, reprojection errors <= 5.991 sigma).  If `gate` is set it replaces the built-in camera models:
 * gate(gate_user, idx1, idx2) != 0 accepts -- an adapter passes the reference's own GeometricCamera objects through
 * it, which keeps the decision bit-identical to the reference's whatever Eigen / libm it is built with. */
enum { ORBX_CAMERA_PINHOLE = 0, ORBX_CAMERA_KANNALA_BRANDT8 = 1 };
typedef struct orbx_camera_model {
    int type;            /* ORBX_CAMERA_*                                              */
    float params[8];     /* fx, fy, cx, cy (+ k0..k3 for KannalaBrandt8)               */
} orbx_camera_model;
typedef struct orbx_rel_pose { float R12[9]; float t12[3]; } orbx_rel_pose;   /* row-major rotation */
typedef struct orbx_triangulation_geometry {
    int n_left1, n_left2;
    orbx_camera_model cam1[2], cam2[2];
    orbx_rel_pose pose[2][2];
    float epipole_x, epipole_y;        /* projection of camera centre 1 into image 2 (single-camera case) */
    const float* scale_factors2;       /* mvScaleFactors of keyframe 2 */
    const float* level_sigma2_1;       /* mvLevelSigma2 of keyframe 1 */
    const float* level_sigma2_2;       /* mvLevelSigma2 of keyframe 2 */
    int n_levels;
    int (*gate)(void* user, int idx1, int idx2);
    void* gate_user;
} orbx_triangulation_geometry;
orbx_status orbx_search_for_triangulation_cameras(const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* free1,
                                                  const float* uright1, int n1, const int* idx1, const int* off1,
                                                  const orbx_keypoint* kps2, const uint8_t* desc2, const uint8_t* free2,
                                                  const float* uright2, int n2, const int* idx2, const int* off2,
                                                  int n_buckets, const orbx_triangulation_geometry* geom, int only_stereo,
                                                  int coarse, int check_orientation, int* matches12, int* n_matches);

/* The camera-model calls that the search above and orbx_stereo_fisheye_matches evaluate per candidate, on their own for
 * n points / pairs at once (host arithmetic in binary32 as in the reference, no device involved):
 *   orbx_camera_project              GeometricCamera::project(const Eigen::Vector3f&)   Pinhole.cpp:46-53, KannalaBrandt8.cpp:66-83
 *   orbx_camera_unproject            GeometricCamera::unprojectEig                      Pinhole.cpp:66-70, KannalaBrandt8.cpp:111-146
 *   orbx_camera_triangulate_matches  KannalaBrandt8::TriangulateMatches                 KannalaBrandt8.cpp:323-397 (Triangulate :418-430)
 *   orbx_camera_epipolar_constrain   GeometricCamera::epipolarConstrain                 Pinhole.cpp:118-141, KannalaBrandt8.cpp:227-232
 * xyz / rays: n x 3, uv: n x 2, R12 row-major 3x3 with x1 = R12 x2 + t12; sigma_level[i] / unc[i] = mvLevelSigma2 of the two
 * keypoints' octaves.  depth[i] = TriangulateMatches' return value (depth in camera 1, or -1 parallax, -2 / -3 behind a
 * camera, -4 / -5 reprojection error); points3d (n x 3, may be NULL) is written for accepted pairs only. */
orbx_status orbx_camera_project(const orbx_camera_model* cam, const float* xyz, int n, float* uv);
orbx_status orbx_camera_unproject(const orbx_camera_model* cam, const float* uv, int n, float* rays);
orbx_status orbx_camera_triangulate_matches(const orbx_camera_model* cam1, const orbx_camera_model* cam2, const float* uv1,
                                            const float* uv2, int n, const float* R12, const float* t12,
                                            const float* sigma_level, const float* unc, float* depth, float* points3d);
orbx_status orbx_camera_epipolar_constrain(const orbx_camera_model* cam1, const orbx_camera_model* cam2, const float* uv1,
                                           const float* uv2, int n, const float* R12, const float* t12,
                                           const float* sigma_level, const float* unc, uint8_t* accepted);

/* The three bucketed searches on resident frames (orbx_frame holds mDescriptors on the device and the keypoints'
 * angles, levels and mvuRight on the host): the same loops, results and order as the functions of the same name
 * without "frame", minus the upload of both descriptor sets -- the FeatureVector CSR arrays go in and the distance
 * rows come out through the calling thread's mapped pinned memory: one kernel launch and one synchronisation per
 * call.  A two-camera Frame / KeyFrame is assigned with its combined arrays (left keypoints first; the grid of such a
 * frame is not used by these searches).  orbx_frame_search_for_triangulation takes the geometry of
 * orbx_search_for_triangulation_cameras (n_left = -1 and cam[0] / pose[0][0] for single-camera keyframes). */
orbx_status orbx_frame_search_by_bow(const orbx_frame* kf, const uint8_t* valid_kf, const int* idx_kf, const int* off_kf,
                                     const orbx_frame* f, int n_left_f, const int* idx_f, const int* off_f, int n_buckets,
                                     float nn_ratio, int check_orientation, int* match_f, int* n_matches);
orbx_status orbx_frame_search_by_bow_keyframes(const orbx_frame* kf1, const uint8_t* valid1, const int* idx1, const int* off1,
                                               const orbx_frame* kf2, const uint8_t* valid2, const int* idx2, const int* off2,
                                               int n_buckets, float nn_ratio, int check_orientation, int* matches12,
                                               int* n_matches);
orbx_status orbx_frame_search_for_triangulation(const orbx_frame* kf1, const uint8_t* free1, const int* idx1, const int* off1,
                                                const orbx_frame* kf2, const uint8_t* free2, const int* idx2, const int* off2,
                                                int n_buckets, const orbx_triangulation_geometry* geom, int only_stereo,
                                                int coarse, int check_orientation, int* matches12, int* n_matches);

/* ---- bag-of-words transform (the step between extraction and the BoW matchers) ----------
 * DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1112-1243; Frame::ComputeBoW, Frame.cc:872-879,
 * KeyFrame::ComputeBoW, KeyFrame.cc:100-109, both with levelsup = 4).
 * The vocabulary tree is handed over flattened: node 0 is the root, the children of node i are
 * child_idx[child_off[i] .. child_off[i+1]) in the order of DBoW2's Node::children, every node has
 * a 32-byte descriptor (the root's is ignored), leaves carry word_id >= 0 and their weight
 * (idf for TF_IDF), inner nodes word_id = -1.  depth = m_L. */
typedef struct orbx_vocabulary orbx_vocabulary;
orbx_status orbx_vocabulary_create(int n_nodes, const int* child_off, const int* child_idx,
                                   const uint8_t* node_descriptors, const int* word_id, const double* weight,
                                   int depth, int device, orbx_vocabulary** out);
void orbx_vocabulary_destroy(orbx_vocabulary* voc);

/* transform(feature, word_id, weight, &nid, levelsup) for n descriptors: descent from the root,
 * at every level the child with the smallest FORB::distance, the first among equals; node_id is
 * the node passed at level depth - levelsup (the root when that is <= 0; also the root when the
 * leaf lies above that level, where the reference leaves nid unset).  Host pointers. */
orbx_status orbx_bow_transform(orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup,
                               int* word_id, double* weight, int* node_id);

/* transform(features, BowVector&, FeatureVector&, levelsup) for TF_IDF weighting: the BowVector as
 * ascending (word, value) pairs (weights summed in feature order, then L1-normalised when
 * l1_normalize != 0, else divided by the number of words as TemplatedVocabulary.h:1149-1155 does),
 * the FeatureVector as CSR over ascending node ids: features fv_idx[fv_off[b] .. fv_off[b+1]) of
 * node fv_node[b] in ascending feature order -- the layout orbx_search_by_bow* consume.  Outputs
 * hold up to n entries (fv_off: n + 1). */
orbx_status orbx_bow_vectors(orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int l1_normalize,
                             int* bow_word, double* bow_value, int* n_bow,
                             int* fv_node, int* fv_off, int* fv_idx, int* n_fv_nodes);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:416-533) for n_points map points at once.
 * The observed descriptors of point p are rows off[p] .. off[p+1] of desc (at most 384 per point).
 * best_idx[p] = the observation (relative to off[p]) with the smallest median distance to all
 * observations of the point, the first among equal medians (-1 for a point without observations);
 * best_median[p] = that median = element floor(0.5 * (N - 1)) of the sorted distance row. */
orbx_status orbx_distinctive_descriptors(const uint8_t* desc, const int* off, int n_points,
                                         int* best_idx, int* best_median);

/* Frame::ComputeStereoFishEyeMatches (Frame.cc:1286-1338), the whole member: brute-force k = 2 search of the
 * left descriptors [mono_left, n_left) in the right descriptors [mono_right, n_right) (Frame.cc:1290-1309), Lowe's
 * ratio test `best < 0.7 * second` in double as the reference evaluates it (:1318), KannalaBrandt8::TriangulateMatches
 * (KannalaBrandt8.cpp:323-397) of every match that passes with R_lr / t_lr = Frame::mRlr / mtlr (row-major 3x3, 3) and
 * sigma = level_sigma2[octave] of the two keypoints, accepted when the depth exceeds 0.0001 (:1328).  Matches are
 * visited in ascending left index, so a right keypoint claimed twice keeps the later left one, as in the
 * reference.  Outputs (host): left_to_right[n_left] / right_to_left[n_right] = mvLeftToRightMatch /
 * mvRightToLeftMatch (-1 none), depth[n_left] = mvDepth (-1 none), points3d[n_left x 3] = mvStereo3Dpoints (rows
 * of unmatched keypoints are left untouched; may be NULL), *n_matches = nMatches.  The search runs on the GPU, the
 * per-match triangulation (a few hundred 4x4 null vectors) on the calling thread. */
orbx_status orbx_stereo_fisheye_matches(const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left, int mono_left,
                                        const orbx_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                                        int mono_right, const orbx_camera_model* cam_left,
                                        const orbx_camera_model* cam_right, const float* R_lr, const float* t_lr,
                                        const float* level_sigma2, int n_levels, int* left_to_right, int* right_to_left,
                                        float* depth, float* points3d, int* n_matches);

/* ---- saved-map records (the source of large map-descriptor sets) ------------------------------
 * The reference writes keypoints and descriptors into its Atlas archives (System.cc:1568-1672, boost text or binary
 * archive) through two templates: serializeMatrix (include/SerializationUtils.h:76-108: cols, rows, type,
 * continuous, then the bytes) and serializeVectorKeyPoints (:125-169: NumEl, then per keypoint angle, response, size,
 * pt.x, pt.y, class_id, octave); KeyFrame::serialize (include/KeyFrame.h:118-126) stores mvKeys, mvKeysUn and
 * mDescriptors with them.  These functions read and write exactly those records inside an archive image held in
 * memory (text != 0: boost text archive, every primitive a decimal token; else boost binary archive, native little
 * endian).  Host memory, no device involved.  Parity unpinned: no saved map and no boost in the reference tree.
 * read_*: the record starts at buf; *consumed = its length.  out may be NULL to query sizes.
 * write_*: out may be NULL to query *written. */
orbx_status orbx_archive_read_matrix(const uint8_t* buf, size_t len, int text, size_t* consumed, int* rows, int* cols,
                                     int* cv_type, uint8_t* out, size_t capacity);
orbx_status orbx_archive_write_matrix(const uint8_t* data, int rows, int cols, int cv_type, int text, uint8_t* out,
                                      size_t capacity, size_t* written);
orbx_status orbx_archive_read_keypoints(const uint8_t* buf, size_t len, int text, size_t* consumed, orbx_keypoint* out,
                                        int capacity, int* n_out);
orbx_status orbx_archive_write_keypoints(const orbx_keypoint* kps, int n, int text, uint8_t* out, size_t capacity,
                                         size_t* written);
/* Every descriptor matrix of an archive image, found by the header serializeMatrix gives it (32 columns, CV_8UC1,
 * continuous, at least min_rows rows, payload inside the image) without interpreting the rest of the archive:
 * KeyFrame::mDescriptors (rows = N) and, with min_rows = 1, MapPoint::mDescriptor.  payload_offset[i] / rows_out[i]
 * (capacity entries, may be NULL to count) locate matrix i; *total_rows = descriptors in all of them.  In a binary
 * archive the rows can be used in place (buf + payload_offset, rows x 32 bytes, e.g. uploaded shard by shard into
 * orbx_hamming_knn2_device); orbx_archive_copy_descriptors decodes them for either format. */
orbx_status orbx_archive_scan_descriptors(const uint8_t* buf, size_t len, int text, int min_rows, size_t* payload_offset,
                                          int* rows_out, int capacity, int* n_found, long long* total_rows);
orbx_status orbx_archive_copy_descriptors(const uint8_t* buf, size_t len, int text, size_t payload_offset, int rows,
                                          uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H_ */
