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
    size_t slotBase;          // first slot entry of this level within a frame
    int quota;                // mnFeaturesPerLevel
    float scale;              // mvScaleFactor
    int kpSize;               // (int)(15*scale)
    int fastTileBase, fastTilesX, fastTilesY;   // k_fast_score tiles (64 x 32 px)
    int blurStripBase, blurStripsX, blurStripsY, blurIters;   // k_blur strips: 120 px wide, 7*blurIters-6 rows tall
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
    int fastTilesPerFrame = 0;
    int maxQuotaSum = 0;
};

// parameters handed to kernels by value
struct KLevels {
    int nlevels;
    int w[kMaxLevels], h[kMaxLevels], pitch[kMaxLevels];
    unsigned long long offset[kMaxLevels];
    int nCols[kMaxLevels], nRows[kMaxLevels], wCell[kMaxLevels], hCell[kMaxLevels];
    int cellBase[kMaxLevels + 1];
    int slotCap[kMaxLevels];
    unsigned long long slotBase[kMaxLevels];
    int blurStripBase[kMaxLevels + 1], blurStripsX[kMaxLevels], blurIters[kMaxLevels];
    int fastTileBase[kMaxLevels + 1], fastTilesX[kMaxLevels];
    float scale[kMaxLevels];
    int kpSize[kMaxLevels];
};

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
    ProfScope(KernelProfiler& p_, const char* name, hipStream_t s_) : p(p_), s(s_) { if (p.enabled) p.begin(name, s); }
    ~ProfScope() { if (p.enabled) p.end(s); }
};

// ---------------------------------------------------------------- kernel launchers (defined in *.hip)
// candidates are packed as x | y<<12 | score<<24 (x,y relative to minBorder)
void launch_rgb2gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                     uint8_t* pyr, size_t pyrFrameBytes, int pitch0, hipStream_t s);
void launch_copy_gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                      uint8_t* pyr, size_t pyrFrameBytes, int pitch0, hipStream_t s);
void launch_resize(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int level, const int16_t* tabs,
                   size_t xTabOff, size_t yTabOff, size_t quadTabOff, int mode /*0 generic, 1 quad, 2 area 2x2*/, int rows, int batch, hipStream_t s);
void launch_fast(const uint8_t* pyr, uint8_t* score, size_t pyrFrameBytes, const KLevels& kl, int tilesPerFrame, int cellsPerFrame,
                 size_t slotsPerFrame, int iniTh, int minTh, uint32_t* slots, int* cellCount, int batch, hipStream_t s);
void launch_compact(const KLevels& kl, int cellsPerFrame, size_t slotsPerFrame, const uint32_t* slots, const int* cellCount,
                    uint32_t* cand, int candCap, int* candCount /*[batch][kMaxLevels+1]*/, int batch, hipStream_t s);
void launch_blur(const uint8_t* pyr, uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl, int stripsPerFrame, int batch, hipStream_t s);
// kpIn: per keypoint {x | y<<16, level | resp<<8}; offsets per frame = f*cap
void launch_describe(const uint8_t* pyr, const uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl,
                     const uint2* kpIn, const int* kpCount, int cap, ssvio_keypoint* kpOut, uint8_t* desc, int batch, hipStream_t s);

}  // namespace ssvio
