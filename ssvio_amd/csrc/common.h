// common.h -- internal definitions shared by the host side and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ssvio_gpu.h"

namespace ssvio {

constexpr int kMaxLevels = SSVIO_MAX_LEVELS;
constexpr int kEdge = 19;          // EDGE_THRESHOLD, src/Feature.cpp:14
constexpr int kHalfPatch = 15;     // HALF_PATCH_SIZE, src/Feature.cpp:13
constexpr int kMinBorder = 16;     // EDGE_THRESHOLD - 3, src/Feature.cpp:835
constexpr int kPadX = 4;          // reflect-101 columns kept on both sides of every level row (for k_blur)
constexpr int kFastX0 = 16;        // x of the first k_fast tile column (dword-aligned; the tested region starts at kEdge)
constexpr int kMaxRoots = 8;       // quadtree roots per level = round(width / height) of the tested region (src/Feature.cpp:574)
// Keypoints one level can deliver: the quadtree stops at >= quota nodes and a split adds at most 3 (quota + 3), but its
// FIRST round splits every root whatever the quota (src/Feature.cpp:621-703: the size test follows the round), so a
// level with a tiny quota can still end with 4 nodes per root.
constexpr int level_sel_cap(int quota) { return quota + 4 > 4 * kMaxRoots ? quota + 4 : 4 * kMaxRoots; }
constexpr int kMaxCellDim = 64;    // interior of a FAST cell window never exceeds 59 px (see geometry.cpp)

void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SSVIO_HIP(expr)                                                        \
    do {                                                                       \
        hipError_t e__ = (expr);                                               \
        if (e__ != hipSuccess) return ::ssvio::hip_fail(e__, #expr, __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------- geometry (host)
struct LevelGeom {
    int w, h;                 // level size (src/Feature.cpp:805)
    int pitch;                // bytes per row in HBM (multiple of 64)
    size_t offset;            // byte offset of the level inside one frame's pyramid
    int nCols, nRows, wCell, hCell;   // FAST cell grid (src/Feature.cpp:843-849)
    int maxBX, maxBY;         // w-16, h-16
    int cellBase;             // first cell index of this level within a frame
    int slotCap;              // worst-case keypoints per cell after NMS
    size_t slotBase;          // first raw-list entry of this level within a frame (= sum of cells * slotCap)
    int selBase;              // first selected-keypoint entry of this level within a frame
    int nIni; float hX; int rootX[kMaxRoots + 1];   // quadtree roots
    int quota;                // mnFeaturesPerLevel
    float scale;              // mvScaleFactor
    int kpSize;               // (int)(15*scale)
    int fastTileBase, fastTilesX, fastTilesY, fastTileH;   // k_fast tiles over the tested region: 64 px wide, fastTileH <= 32 rows
    int fastTileBaseT, fastTilesYT, fastTileHT;            // ... and the tall form (<= 64 rows) used for large batches
    int blurStripBase, blurStripsX, blurStripsY, blurIters;   // k_blur strips: 120 px wide, 7*blurIters-6 rows tall
    int blurPanelStride;      // the blurred level is stored as 32-px-wide column panels of this many bytes (its own layout: only k_blur and k_describe see it)
    size_t tabOffset;         // offset (in shorts) of this level's resize tables
};

struct PyramidGeom {
    int nlevels = 0;
    int width = 0, height = 0;
    LevelGeom lv[kMaxLevels];
    size_t frameBytes = 0;    // pyramid bytes per frame
    int cellsPerFrame = 0;
    size_t slotsPerFrame = 0;
    int blurStripsPerFrame = 0;
    int fastTilesPerFrame = 0, fastTilesPerFrameT = 0;
    int maxQuotaSum = 0;      // sum of level_sel_cap(quota): keypoint slots per frame
};

// parameters handed to kernels by value: one record per level (a wave-uniform level index turns
// into a handful of scalar loads), plus a sentinel record whose *Base fields hold the totals
struct KLevel {
    int w, h, pitch;
    int nCols, nRows, wCell, hCell;
    int cellBase;
    int fastTileBase, fastTilesX, fastTileBaseT, fastTileH, fastTileHT;
    int blurStripBase, blurStripsX, blurIters;
    int blurPanelStride;              // blurred pixel (x, y) of the level: panel base (offset - 64) + (x >> 5) * blurPanelStride + 32 * y + (x & 31)
    int quota, kpSize;
    float scale;
    unsigned long long offset;        // byte offset of pixel (0,0) inside a frame's pyramid
    unsigned int rawBase, rawCap;     // this level's region of the raw candidate list of a frame (entries)
    int selBase;                      // this level's region of the selected-keypoint list of a frame
    int nIni; float hX; int rootX[kMaxRoots + 1]; // quadtree roots (src/Feature.cpp:574-593)
    int wCellMagic, hCellMagic;       // ceil(2^20 / cell size): (n * magic) >> 20 == n / cell for n < 4096
};
struct KLevels {
    int nlevels;
    int pad[3];
    KLevel lv[kMaxLevels + 1];
};

// arguments of the quadtree kernel
struct QtArgs {
    KLevels kl;
    const uint2* raw; const int* rawCount; const int* cellIni;
    size_t rawPerFrame; int cellsPerFrame;
    uint32_t* kval; uint32_t* kord; uint32_t* knode;     // per-key scratch, same indexing as raw
    uint2* sel; int* selCount; int selPerFrame;
    int batch;
    unsigned char* nodeScratch; size_t nodeScratchStride;   // per-workgroup node arrays in HBM (quotas beyond the LDS forms), else null
};

static_assert(sizeof(QtArgs) <= 4096, "kernel arguments are passed by value: 4 KB limit");

struct ExtractorTables {
    int nfeatures, nlevels, iniTh, minTh;
    float scale[kMaxLevels], invScale[kMaxLevels], sigma2[kMaxLevels], invSigma2[kMaxLevels];
    int quota[kMaxLevels];
};

int  build_tables(const ssvio_orb_params& p, ExtractorTables& t);
// returns SSVIO_OK or SSVIO_ERR_GEOMETRY
int  build_geometry(const ExtractorTables& t, int width, int height, PyramidGeom& g);
// resize coefficient tables for level l (from level l-1): per dst x: {sx, a0, a1}, per dst y: {sy0, sy1, b0, b1}
struct ResizeTabs { std::vector<int16_t> xs, ys, quads; bool quadOk; bool area2; int rows; };   // rows: output rows per wave of k_resize_quad   // area2: exact 2:1 shrink -> INTER_AREA   // xs: 4 shorts per dx; ys: 4 per dy; quads: 16 shorts per 4 dx
void build_resize_tabs(int sw, int sh, int dw, int dh, ResizeTabs& t);
// per output quad (4 adjacent dx) entry of the streaming resize kernel, 32 bytes
struct ResizeQuad { uint32_t bOff; uint32_t sel0, sel1; int16_t a0[4], a1[4]; uint32_t pad; };
static_assert(sizeof(ResizeQuad) == 32, "ResizeQuad is 16 shorts");

// host quadtree (src/Feature.cpp:570-798)
int  distribute_octtree(const float* xs, const float* ys, const float* resp, int n,
                        int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out);

// ---------------------------------------------------------------- profiler
struct KernelProfiler {
    struct Rec { const char* name; hipEvent_t a, b; };
    struct Stat { const char* name; double ms; int launches; };
    bool enabled = false;
    const char* only = nullptr;   // when set, only the kernel of this name is bracketed
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    std::vector<Stat> stats;
    hipEvent_t get();
    void begin(const char* name, hipStream_t s);
    void end(hipStream_t s);
    void resolve();          // call after the stream is synchronised
    void clear();
    void destroy();
};

struct ProfScope {
    KernelProfiler& p; hipStream_t s;
    bool on;
    ProfScope(KernelProfiler& p_, const char* name, hipStream_t s_) : p(p_), s(s_), on(p_.enabled && (!p_.only || strcmp(p_.only, name) == 0)) { if (on) p.begin(name, s); }
    ~ProfScope() { if (on) p.end(s); }
};

#if defined(__HIPCC__)
// Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  This mapping sends all
// workgroups of a frame to ONE XCD (frame f -> XCD f % 8), so that a frame's pyramid is fetched into one L2 only and
// neighbouring tiles / keypoints of the frame find each other's lines there.  Frames beyond the last multiple of 8 keep
// the plain order.  (Affinity is for speed only: nothing depends on where a workgroup runs.)
__device__ __forceinline__ void xcd_frame_block(unsigned id, int blocksPerFrame, int batch, int& frame, int& blk)
{
    const unsigned full = (unsigned)(batch & ~7) * (unsigned)blocksPerFrame;      // workgroups of the frames in complete groups of 8
    if (id < full) {
        const unsigned xcd = id & 7u, slot = id >> 3;
        const unsigned g = slot / (unsigned)blocksPerFrame;
        blk = (int)(slot - g * blocksPerFrame);
        frame = (int)(g * 8u + xcd);
    } else {
        const unsigned r = id - full, f = r / (unsigned)blocksPerFrame;
        blk = (int)(r - f * blocksPerFrame);
        frame = (batch & ~7) + (int)f;
    }
}

#endif

// ---------------------------------------------------------------- kernel launchers (defined in *.hip)
// candidates are packed as x | y<<12 | score<<24 (x,y relative to minBorder)
void launch_rgb2gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                     uint8_t* pyr, size_t pyrFrameBytes, int pitch0, int* zeroA, int nA, int* zeroB, int nB, hipStream_t s);
void launch_copy_gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                      uint8_t* pyr, size_t pyrFrameBytes, int pitch0, int* zeroA, int nA, int* zeroB, int nB, hipStream_t s);
bool resize2_fits(const int16_t* xsB, const int16_t* ysB, int wA, int hA, int wB, int hB);
void launch_resize2(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int levelA, const int16_t* tabs,
                    size_t xTabOffA, size_t yTabOffA, size_t xTabOffB, size_t yTabOffB, int batch, hipStream_t s);
void launch_resize(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int level, const int16_t* tabs,
                   size_t xTabOff, size_t yTabOff, size_t quadTabOff, int mode /*0 generic, 1 quad, 2 area 2x2*/, int rows, int batch, hipStream_t s);
// FAST: appends {key, val} per NMS survivor to the level's raw list; key = flags<<30 | cell<<12 | iy<<6 | ix
// (flags bit0: passes at iniTh, bit1: at minTh), val = x | y<<12 | score<<24 relative to minBorder
void launch_fast(const uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int tilesPerFrame /* of the chosen form */, bool tall, int cellsPerFrame, size_t rawPerFrame,
                 int iniTh, int minTh, uint2* raw, int* rawCount, int* cellIni, int batch, hipStream_t s);
int  launch_quadtree(const QtArgs& a, int maxAlive, int nlevels, int batch, hipStream_t s);
void quadtree_prepare();
size_t quadtree_lds_bytes(int maxa);   // bytes of node arrays for a capacity of maxa live nodes (LDS or HBM scratch)
constexpr int kQtMaxLds = 2048, kQtMaxGlobal = 32768;
// where the rimmed levels of the exported pyramid sit in the mapped host area (byte offsets, row strides: multiples of 64)
struct ExportLayout { unsigned long long offset[kMaxLevels]; int stride[kMaxLevels]; };
void launch_export_pyramid(const uint8_t* pyrFrame0, const KLevels& kl, const ExportLayout& lay, int maxRows, uint8_t* dst, hipStream_t s);
void launch_blur(const uint8_t* pyr, uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl, int stripsPerFrame, int batch, hipStream_t s);
// sel: per level lists {x | y<<16, level | resp<<8} at f*selPerFrame + selBase[l]; selCount[f][kMaxLevels]
void launch_describe(const uint8_t* pyr, const uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl,
                     const uint2* sel, const int* selCount, int selPerFrame, int cap, ssvio_keypoint* kpOut, uint8_t* desc,
                     int* kpCount, int* kpCountHost /* device view of pinned host memory */,
                     ssvio_keypoint* kpHost, uint8_t* descHost /* optional second copy of the outputs in mapped host memory */,
                     int batch, hipStream_t s);

}  // namespace ssvio
