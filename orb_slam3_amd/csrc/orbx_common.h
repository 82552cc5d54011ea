// Shared declarations of the orbx library (host side + device structs).
#ifndef ORBX_COMMON_H_
#define ORBX_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/orbx.h"

void orbx_set_error(const char* fmt, ...);

// orbx_stage.hip: enqueue the staging kernel (strides in bytes); used by the extractor to stage into its pyramid.
orbx_status orbx_stage_enqueue(orbx_stager* s, const uint8_t* d_src, int n_frames, long long srcRow, long long srcFrame,
                               uint8_t* d_dst, long long dstRow, long long dstFrame, hipStream_t st);

#define ORBX_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            orbx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                        \
            return ORBX_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define ORBX_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            orbx_set_error(__VA_ARGS__); \
            return ORBX_ERR_INVALID;     \
        }                                \
    } while (0)

constexpr int kOrbxMaxLevels = 16;
constexpr int kOrbxEdge = 19;        // EDGE_THRESHOLD (ORBextractor.cc:76)
constexpr int kOrbxMinBorder = 16;   // EDGE_THRESHOLD - 3 (ORBextractor.cc:899)
constexpr int kOrbxMaxRects = 64;

// Geometry of one pyramid level; everything here is fixed by (params, width, height).
struct OrbxLevelGeo {
    int w, h, pitch;         // level image, pitch in bytes (multiple of 64)
    long long off;           // byte offset of the level inside one frame's pyramid slab
    int nCols, nRows;        // FAST cell grid (ORBextractor.cc:914-918)
    int wCell, hCell;
    int nColsP, nRowsP;      // cells that are not skipped by the border tests (:931, :949)
    int cellBase;            // first FAST block of this level
    int quota;               // mnFeaturesPerLevel[level]
    int nIni;                // quadtree roots (ORBextractor.cc:665)
    float hX;
    int candCap;             // candidate records per frame
    long long candOff;       // record offset of the level inside one frame's candidate slab
    int kpCap;               // distributed keypoints per frame
    int kpOff;
    int nodeCap;             // quadtree list capacity
    float scale;             // mvScaleFactor[level]
    float invScale;          // 1 / scale in binary32 (ORBextractor.cc:1019)
    float patchSize;         // (int)(31 * scale)
    int tapX, tapY;          // offsets into the resize tap tables
};

struct OrbxFrameGeo {
    int nlevels;
    int w, h;
    int totalCells;
    long long pyrBytes;      // one frame's pyramid slab
    long long candTotal;     // one frame's candidate records
    int kpTotal;             // one frame's distributed keypoint records
    int maxNodeCap;
    int tileP, tileR;        // LDS tile of the FAST kernel: pitch (bytes) and rows
    int fastLds;             // its dynamic LDS size in bytes
    int fastQueueOff;        // byte offset of its survivor queue in that memory
    OrbxLevelGeo lv[kOrbxMaxLevels];
};

// FAST candidate (level coordinates relative to the 16-px border, after the round trip).
struct OrbxCand {
    float x, y;
    float response;
    uint32_t order;          // (processed cell index << 14) | (y in cell << 7) | x in cell
};

// Keypoint kept by the quadtree (level coordinates, +16 applied).
// Per-cell constants of k_fast, built on the host (one scalar load of 32 bytes per workgroup).
struct alignas(32) OrbxFastCell {
    int xy;                  // x0 | y0 << 16: top-left of the cell image in level coordinates
    int dims;                // width | height << 8 | level << 16
    int index;               // index among the level's processed cells (row-major)
    int apps;                // cells from this one to the level's last
    int tileOff;             // byte offset, inside a frame's pyramid slab, of the tile's first (4-byte aligned) column
    int pitch;               // row pitch of the level
    int bands;               // quick-reject lane layout: dword columns | bands << 8 | rows per band << 16 | both waves << 24
    int colRcp;              // floor(65536 / columns) + 1: lane / columns = (lane * colRcp) >> 16 for lane < 128
};

// What k_describe needs to know about one kept keypoint, written by k_layout for the j-th keypoint of a frame in
// (level, list) order: one 32-byte scalar load replaces the walk level table -> level geometry -> keypoint record.
struct alignas(32) OrbxDescWork {
    int imgOff;              // byte offset of the level image inside the frame's pyramid slab
    int pitch;               // of that level
    int wh;                  // its width | height << 16
    int cxy;                 // rounded level coordinates of the keypoint, cx | cy << 16
    int slotLevel;           // row in the frame's output arrays | level << 24
    float x, y, response;    // the keypoint record (level coordinates)
};

struct OrbxLevelKp {
    float x, y;
    float response;
    int slot;                // row in the frame's output arrays (set by the layout kernel)
};

#endif
