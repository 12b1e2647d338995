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

void launch_knn2(const MatchArgs& a, int maxQ, int npairs, hipStream_t s);
void launch_finalize(const FinalizeArgs& a, int npairs, int maxQ, int maxT, hipStream_t s);

}  // namespace ssvio
