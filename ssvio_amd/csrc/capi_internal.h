// capi_internal.h -- shared by the translation units of the C ABI (capi*.hip): the matcher handle and small host helpers.
#pragma once
#include "common.h"
#include "match.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ssvio {
int pick_device(int device);                       // SSVIO_OK, SSVIO_ERR_NOGPU or SSVIO_ERR_INVALID (capi.hip)
int prof_out(KernelProfiler& p, const char** names, double* total_ms, int32_t* launches, int cap, int* count);   // capi.hip
template <typename T> inline void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> inline void hfree(T*& p) { if (p) { (void)hipHostFree(p); p = nullptr; } }
// test hook: ssvio_test_fail_next_allocs(n) makes the next n growing device allocations of the handles fail (capi_match.hip)
bool alloc_fail_injected();
int noalloc_scratch_init(int device);              // capi_noalloc.hip: resolves the device address of the fixed scratch, once per device
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace ssvio

using namespace ssvio;

struct ssvio_matcher {
    int device = 0;
    int threshold = 50; float nnRatio = 0.6f;
    hipStream_t own = nullptr, stream = nullptr;
    uint8_t* d_q = nullptr; size_t qRows = 0;
    uint8_t* d_t = nullptr; size_t tRows = 0;
    int* d_counts2 = nullptr;            // {nq, nt} of the single-pair entry points
    // single-pair host entry points: counts and results live in pinned, device-mapped host memory that the kernels read / write
    // directly -- no copy commands, one stream wait per call (a small copy to or from pinned memory costs more than the kernels)
    int* h_counts2 = nullptr; int* d_counts2m = nullptr;                                  // {nq, nt} and its device address
    uint8_t* h_result = nullptr; uint8_t* d_resultm = nullptr; size_t resultCap = 0;      // {count, pad to 16 B, rows}
    uint8_t* h_descm = nullptr; uint8_t* d_descm = nullptr; size_t descmCap = 0;          // small sets: query rows, then train rows, packed by the host
    uint2* d_partial = nullptr; size_t partialCap = 0;
    uint2* d_partialRev = nullptr; size_t partialRevCap = 0;
    ssvio_dmatch* d_out = nullptr; size_t outCap = 0;
    int32_t* d_raw = nullptr; size_t rawCap = 0;
    int* d_nout = nullptr; size_t noutCap = 0;
    int* d_pairs = nullptr; size_t pairsCap = 0;
    double* d_p3d = nullptr; size_t p3dCap = 0; double* d_p2d = nullptr; size_t p2dCap = 0;     // MatchByProject inputs
    ssvio_dmatch* d_cand = nullptr; size_t candCap = 0; int* d_keep = nullptr; size_t keepCap = 0;
    // MatchByDBoW: per-feature transform results (query rows first, then train rows) and the grouped-match scratch
    int32_t* d_bowWord = nullptr; size_t bowWordCap = 0; int32_t* d_bowNode = nullptr; size_t bowNodeCap = 0;
    double* d_bowWeight = nullptr; size_t bowWeightCap = 0; int32_t* d_gnode = nullptr; size_t gnodeCap = 0;
    int32_t* d_keepNode = nullptr; size_t keepNodeCap = 0;
    int32_t* d_sortIdx = nullptr; size_t sortIdxCap = 0; int32_t* d_sortNode = nullptr; size_t sortNodeCap = 0;
    int lastBowQ = 0, lastBowT = 0; size_t lastBowTBase = 0;      // rows of the last host call's two sides; where the train side's results start
    // host entry points: k_bow_transform stores word / node / weight straight into mapped host memory ([rows] int32, [rows] int32,
    // [rows] double; query rows first, then train rows), from where ssvio_bow_last_transform serves them without touching the GPU
    uint8_t* h_bowm = nullptr; uint8_t* d_bowm = nullptr; size_t bowmCap = 0; size_t bowmRows = 0;
    // behind the matcher (f2, f3)
    uint8_t* d_frameIn = nullptr; size_t frameInCap = 0;      // uploaded matches / keypoints / depth of the host entry points
    uint8_t* d_frameOut = nullptr; size_t frameOutCap = 0;    // assembled lists
    uint32_t* d_table = nullptr; size_t tableCap = 0;
    // host entry points behind the matcher: small inputs (matches, keypoints, counts) and all outputs live in mapped host memory
    uint8_t* h_fin = nullptr; uint8_t* d_fin = nullptr; size_t finCap = 0;
    uint8_t* h_fout = nullptr; uint8_t* d_fout = nullptr; size_t foutCap = 0;
    // the point-cloud map of Map::MapPointCloudFusion (capi_voxel.hip): current map in d_mapA (mapCount points of 32 bytes)
    uint8_t* d_mapA = nullptr; size_t mapCapA = 0; uint8_t* d_mapB = nullptr; size_t mapCapB = 0; int mapCount = 0;
    uint32_t* d_voxKeys = nullptr; size_t voxKeysCap = 0; uint8_t* d_voxTmp = nullptr; size_t voxTmpCap = 0;
    void* d_voxState = nullptr; int* h_voxTotal = nullptr; int* d_voxTotal = nullptr;
    std::vector<uint8_t> stageQ, stageT;              // packed copies of strided descriptor rows
    std::vector<int32_t> h_pairs;                     // what d_pairs holds (an unchanged pair list is not uploaded again)
    KernelProfiler prof; std::string profOnly;
    int kernel = 0;                                   // SSVIO_KERNEL_VALU / SSVIO_KERNEL_MFMA: which 2-NN kernel the batched shapes run
    int lastPairs = 0, lastMaxRows = 0; bool lastHadPairs = false;    // the last batched call: d_out / d_nout hold lastPairs lists; d_pairs is its pair list
};

// grows a device array of the handle (never shrinks); waits for the stream first: queued work may still use the old block
template <typename T> inline int grow(T*& p, size_t& cap, size_t need, hipStream_t s)
{
    if (need <= cap) return SSVIO_OK;
    SSVIO_HIP(hipStreamSynchronize(s));
    dfree(p);
    cap = 0;                                  // a failed hipMalloc below must not leave the old capacity with a null block
    need += need / 4;
    if (alloc_fail_injected()) { set_error("hipMalloc: injected failure (SSVIO_TEST_FAIL_ALLOC)"); return SSVIO_ERR_HIP; }
    SSVIO_HIP(hipMalloc(&p, need * sizeof(T)));
    cap = need;
    return SSVIO_OK;
}

// descriptor-row buffers keep their capacity in rows (32 bytes each)
inline int grow_rows(uint8_t*& p, size_t& rows, size_t needRows, hipStream_t s)
{
    size_t cap = rows * 32;
    const int rc = grow(p, cap, needRows * 32, s);
    rows = cap / 32;                          // written back on the error path too (0 after a failed allocation)
    return rc;
}

// (re)allocates a pinned, device-mapped host block of at least `need` bytes; waits for the stream before the old one goes
inline int grow_mapped(uint8_t*& h, uint8_t*& d, size_t& cap, size_t need, hipStream_t s)
{
    if (need <= cap) return SSVIO_OK;
    SSVIO_HIP(hipStreamSynchronize(s));
    if (h) { (void)hipHostFree(h); h = nullptr; }
    d = nullptr; cap = 0;
    need += need / 4;
    SSVIO_HIP(hipHostMalloc((void**)&h, need, hipHostMallocMapped));
    SSVIO_HIP(hipHostGetDevicePointer((void**)&d, h, 0));
    cap = need;
    return SSVIO_OK;
}

inline int mt_check(ssvio_matcher* m) { if (!m) { set_error("null matcher handle"); return SSVIO_ERR_INVALID; } return SSVIO_OK; }
int upload_rows(uint8_t* dst, const uint8_t* src, int n, int stride, std::vector<uint8_t>& staging, hipStream_t s);   // capi_match.hip
int upload_pairs(ssvio_matcher* m, const int32_t* pairs, int npairs);                                                   // capi_match.hip
// descriptor rows of a single pair on the device: small sets are packed into mapped host memory that the kernels read directly,
// larger ones are copied into d_q / d_t (capi_match.hip)
int stage_pair(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride, const uint8_t** d_q, const uint8_t** d_t);
// the batched matcher in two steps, shared by the allocating entry points and the one that allocates nothing (capi_match.hip)
struct MatchPlan { int maxRows, nChunks, chunkRows, nChunksR, chunkRowsR, qBlocks, qBlocksR; bool small, smallRev, mfma; size_t partialEntries, partialRevEntries; };
int match_plan(int maxQ, int maxT, int npairs, int mode, int kernel, int slack, size_t limitEntries, MatchPlan& p);   // slack: rows at the end of a set that are rarely filled (32 for the extractor's sets, 0 for exact counts)
int match_launch(ssvio_matcher* m, const MatchPlan& p, const uint8_t* d_qdesc, int qStrideRows, const int* d_qcounts,
                 const uint8_t* d_tdesc, int tStrideRows, const int* d_tcounts, const int* d_pairs, int npairs, int maxQ, int maxT, int mode,
                 uint2* partial, uint2* partialRev, ssvio_dmatch* out, int* nout, int32_t* raw, bool profile);
int match_batch_on_device_pairs(ssvio_matcher* m, const uint8_t* d_desc, int strideRows, const int* d_counts, int npairs, int maxRows, int mode);   // capi_match.hip
// mapped host landing area for {count, pad to 16 B, rows} of at least rowBytes (capi_match.hip)
int ensure_result(ssvio_matcher* m, size_t rowBytes);
