// match.h -- argument blocks of the matching kernels.
#pragma once
#include "common.h"

namespace ssvio {

struct MatchArgs {
    const uint4* qdesc; const uint4* tdesc;     // [rows][2] uint4 = 32-byte descriptors
    int qSetStrideRows, tSetStrideRows;         // rows between consecutive sets
    const int* qcounts; const int* tcounts;     // rows per set (device)
    const int* pairs;                           // device [npairs][2] (query set, train set) or null = (0,0)
    int swap;                                   // 1: train set plays the query role (reverse pass of the cross-check)
    int nChunks, chunkRows;                     // the train set is cut into nChunks pieces of chunkRows rows
    uint2* partial;                             // [pair][chunk][maxRows] two smallest keys
    int maxRows;
};

struct FinalizeArgs {
    const uint2* partial; int nChunks;
    const uint2* partialRev; int nChunksRev;
    int maxRows;
    size_t pairStride, pairStrideRev;           // entries between consecutive pairs (filled by launch_finalize)
    const int* qcounts; const int* tcounts; const int* pairs;
    int mode; int threshold; float nnRatio;
    ssvio_dmatch* out; int* nout;               // [pair][maxRows], [pair]
    int32_t* raw;                               // mode 3: [pair][3][maxRows]
};

// MatchByProject (src/FeatureMatching.cpp:128-189)
struct ProjectArgs {
    const double* p3d; const uint4* qdesc; int nq;      // [nq][3], [nq][2] uint4
    const double* p2d; const uint4* tdesc; int nt;      // [nt][2], [nt][2] uint4
    double radian, K[9], R[9], t[3];
    int threshold; float nnRatio;
    ssvio_dmatch* cand; int* keep;                      // per query: the match it would emit, and whether it does
    ssvio_dmatch* out; int* nout;
};
void launch_project(const ProjectArgs& a, hipStream_t s);

// DBoW3 vocabulary tree on the device (MatchByDBoW, src/FeatureMatching.cpp:58-125, :192-208)
struct VocabDev {
    const uint4* cdesc;        // [nnodes-1][2] uint4: node descriptors in child-list order (the children of a node are contiguous)
    const int* childBegin;     // [nnodes+1] CSR offsets into child / cdesc
    const int* child;          // [nnodes-1] node ids, children of a node in ascending id
    const int* word;           // [nnodes] word id of a leaf
    const double* weight;      // [nnodes]
    int L, nnodes;
};
struct BowArgs {
    VocabDev v; int nidLevel;                  // L - levelsup
    const uint4* desc; int setStrideRows; const int* counts; int nsets;
    int maxRows;                               // counts[] are clamped to this (rows a set can hold)
    int32_t* word; int32_t* node; double* weight;   // per feature, same row indexing as desc
    int32_t* gnode;                            // node id, or -1 for a stopped word (weight <= 0): the grouping key
};
void launch_bow_transform(const BowArgs& a, int maxRows, hipStream_t s);

// grouped 2-NN of MatchByDBoW: candidates of query i = train rows with the same grouping key
struct GroupedArgs {
    const uint4* qdesc; const uint4* tdesc; const int32_t* qgnode; const int32_t* tgnode;
    int qSetStrideRows, tSetStrideRows; const int* qcounts; const int* tcounts; const int* pairs;
    int threshold; float nnRatio; int maxRows, maxT;   // maxRows: query rows per pair (output stride); maxT: train rows a set can hold
    ssvio_dmatch* cand; int32_t* keepNode;     // [pair][maxRows]: candidate match; node id if it is kept else -1
    ssvio_dmatch* out; int* nout;              // [pair][maxRows], [pair]
};
void launch_match_grouped(const GroupedArgs& a, int maxQ, int npairs, hipStream_t s);

// sorted form (sets of up to kBowSortMax rows): features of a set ordered by (grouping key, index), stopped words last
constexpr int kBowSortMax = 8192;
struct BowSortArgs {
    const int32_t* gnode; int setStrideRows; const int* counts;      // per set: gnode + set * setStrideRows, counts[set] rows
    int maxRows;                                                     // counts[] are clamped to this
    int32_t* sortedIdx; int32_t* sortedNode;                         // same indexing: rank r -> feature index / its key
};
void launch_bow_sort(const BowSortArgs& a, int nsets, hipStream_t s);
struct SortedMatchArgs {
    const uint4* qdesc; const uint4* tdesc;
    const int32_t* qSortedIdx; const int32_t* qSortedNode; const int32_t* tSortedIdx; const int32_t* tSortedNode;
    int qSetStrideRows, tSetStrideRows; const int* qcounts; const int* tcounts; const int* pairs;
    int threshold; float nnRatio; int maxRows, maxT;
    ssvio_dmatch* cand; int32_t* keep;         // [pair][maxRows], indexed by the query's rank
    ssvio_dmatch* out; int* nout;
};
void launch_match_sorted(const SortedMatchArgs& a, int maxQ, int maxT, int npairs, hipStream_t s);

// behind the matcher: Frame::Get3DPoint, correspondence assembly, match chaining
struct FrameCam { float cx, cy, invFx, invFy, depthScale; int w, h, stridePx; };
struct AssembleArgs {
    const ssvio_dmatch* matches; const int* nmatches; int maxRows;      // [pair][maxRows], [pair]
    const int* pairs;                                                   // (current set, last set) per pair; null = (0, 1)
    const ssvio_keypoint* kps; int kpStride; const int* kpCounts;       // set s: kps + s * kpStride, kpCounts[s] entries
    const uint16_t* depth; size_t depthFrameStride;                     // set s: depth + s * depthFrameStride (pixels)
    FrameCam cam;
    float* last3d; float* cur2d; float* cur3d; ssvio_dmatch* preGood;   // [pair][maxRows] each
    float* edgeLast3d; float* edgeCur3d; ssvio_dmatch* finalGood;
    int* counts;                                                        // [pair][2] = {n_pre, n_final}
};
void launch_get_3d_points(const ssvio_keypoint* kps, int n, const uint16_t* depth, const FrameCam& c, float* out, hipStream_t s);
void launch_assemble(const AssembleArgs& a, int npairs, hipStream_t s);
void launch_rgb_cloud(const uint16_t* depth, int depthStridePx, const uint8_t* bgr, int bgrStride, int w, int h,
                      float cx, float cy, float invFx, float invFy, double depthScale, void* out, hipStream_t s);
struct ManageXform { float m[16]; };                                    // Eigen::Matrix4f storage order (column-major)
void launch_first_index(const int32_t* rec, int n, int strideInts, int fieldOff, uint32_t* table, int tableSize, hipStream_t s);
void launch_manage(const ssvio_keypoint* kps, int nkp, const uint16_t* depth, const FrameCam& c, const ManageXform& T,
                   const ssvio_dmatch* last, const uint32_t* matchTable, const int32_t* obs, const uint32_t* obsTable, int obsTableSize,
                   int32_t* matchNum, int32_t* mapPoint, float* world, hipStream_t s);
constexpr int kTrackLdsTable = 12288;      // index maps up to this many keys are built inside k_track (48 KB of LDS)
void launch_track(const ssvio_dmatch* last, int nlast, const ssvio_dmatch* cur, int ncur, const uint32_t* table /* null: built in LDS */,
                  int tableSize, ssvio_dmatch* out, int* nout, hipStream_t s);

void launch_pair_summary(const ssvio_dmatch* out, const int* nout, int maxRows, const int* pairs, int npairs, int idMode, int idBase,
                         ssvio_pair_summary* rec, hipStream_t s);
void launch_knn2(const MatchArgs& a, int maxQ, int qBlocks, int npairs, bool needSecond, hipStream_t s);   // qBlocks = knn2_query_blocks(maxQ, slack)
bool knn2_small_applies(int maxQ, int maxT, int npairs);                 // small single pairs: chunks = the waves of one workgroup, no merge launch
void launch_knn2_small(const MatchArgs& a, int maxQ, int npairs, bool needSecond, hipStream_t s);   // writes partial[pair][q] (nChunks = 1, pair stride maxRows)
int  knn2_query_blocks(int maxQ, int slack); // grid.x of k_hamming_knn2 for sets of up to maxQ rows of which the last `slack` are rarely filled
int  knn2_queries_per_lane(int maxQ);       // R of k_hamming_knn2<R> for this many query rows (tuning knob: SSVIO_MATCH_R)
void launch_finalize(const FinalizeArgs& a, int npairs, int maxQ, int maxT, hipStream_t s);
// the same 2-NN on the matrix cores (kernels_match_mfma.hip): exact int8 contraction, selectable per handle
constexpr int kMfmaChunkRows = 8160;        // most train rows a chunk of k_hamming_knn2_mfma can hold (13-bit row field, 32-row tiles)
void launch_knn2_mfma(const MatchArgs& a, int maxQ, int qBlocks, int npairs, bool needSecond, hipStream_t s);   // needSecond = false: nearest neighbour only
int  knn2_mfma_query_blocks(int maxQ, int slack);      // grid.x of k_hamming_knn2_mfma for sets of up to maxQ rows

#if defined(__HIPCC__)
constexpr unsigned kKeyInit = 0xFFFFFFFFu;     // "no candidate yet" in the (dist << 23 | trainIdx) key order

// rows of a set as the kernels use them: the device count clamped to what the set's slice (and the per-pair output slice)
// can hold, so that a count above max_rows cannot push reads or writes into a neighbouring set / pair
__device__ __forceinline__ int clamp_rows(int n, int maxRows) { return min(max(n, 0), maxRows); }

// median of three unsigned values in one VALU instruction
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t hamming256(const uint4 alo, const uint4 ahi, const uint4 blo, const uint4 bhi)
{
    return __popc(alo.x ^ blo.x) + __popc(alo.y ^ blo.y) + __popc(alo.z ^ blo.z) + __popc(alo.w ^ blo.w) +
           __popc(ahi.x ^ bhi.x) + __popc(ahi.y ^ bhi.y) + __popc(ahi.z ^ bhi.z) + __popc(ahi.w ^ bhi.w);
}

// exclusive scan of v over an NT-thread workgroup (wsum: NT / 64 ints of LDS); total = sum over the workgroup
template <int NT = 256>
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int base = 0; total = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { if (i < wid) base += wsum[i]; total += wsum[i]; }
    return base + incl - v;
}
#endif

}  // namespace ssvio
