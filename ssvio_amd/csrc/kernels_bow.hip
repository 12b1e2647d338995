// kernels_bow.hip -- MatchByDBoW on gfx950 (src/FeatureMatching.cpp:58-125, :192-208).
//
// k_bow_transform: DBoW3's Vocabulary::transform for every descriptor of every set (frame): descend the
// vocabulary tree, at each level to the child with the smallest Hamming distance (first child on ties), report the
// leaf's word id and weight and the node passed at level L - levelsup.  16 lanes share one descriptor: lane c scores
// child c (c + 16, ... for wider trees), a 4-step shuffle picks the minimum key (dist << 20 | child position).
// k_match_grouped: one wave per query feature; the lanes walk the train features, keep those with the query's node
// id and track the two smallest keys (dist << 23 | trainIdx) -- the 2-NN loop of :85-102 over the node's feature
// list, which is in ascending feature index.  k_grouped_compact: emits the kept matches in the reference's order,
// ascending node id (std::map iteration, :72-123) and ascending query index inside a node (:82).
#include "match.h"

namespace ssvio {

__global__ __launch_bounds__(256) void k_bow_transform(BowArgs a)
{
    const int set = blockIdx.y;
    const int n = min(a.counts[set], a.maxRows);
    const int g = threadIdx.x & 15;
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (f >= n) return;                                               // uniform over the 16 lanes of a feature
    const size_t row = (size_t)set * a.setStrideRows + f;
    const uint4 qlo = a.desc[2 * row], qhi = a.desc[2 * row + 1];
    int node = 0, level = 0, nid = 0;
    // every pass moves one level down; a tree of nnodes nodes is at most nnodes - 1 deep (exit for malformed tables)
    for (int guard = 0; guard < a.v.nnodes; ++guard) {
        const int b = a.v.childBegin[node], e = a.v.childBegin[node + 1];
        if (b == e) break;                                            // leaf
        ++level;
        uint32_t best = 0xFFFFFFFFu;
        for (int c = b + g; c < e; c += 16) {
            const uint32_t d = hamming256(qlo, qhi, a.v.cdesc[2 * (size_t)c], a.v.cdesc[2 * (size_t)c + 1]);
            best = min(best, (d << 20) | (uint32_t)(c - b));          // 'd < best_d': the first child wins ties
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
        node = a.v.child[b + (int)(best & 0xFFFFFu)];
        if (level == a.nidLevel) nid = node;
    }
    if (g == 0) {
        const double w = a.v.weight[node];
        a.word[row] = a.v.word[node]; a.node[row] = nid; a.weight[row] = w;
        a.gnode[row] = w > 0 ? nid : -1;                              // 'if (w > 0)': stopped words enter no vector
    }
}

void launch_bow_transform(const BowArgs& a, int maxRows, hipStream_t s)
{
    hipLaunchKernelGGL(k_bow_transform, dim3((maxRows + 15) / 16, a.nsets), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void k_match_grouped(GroupedArgs a)
{
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair] : 0, tset = a.pairs ? a.pairs[2 * pair + 1] : 0;
    const int nq = min(a.qcounts[qset], a.maxRows), nt = min(a.tcounts[tset], a.maxT);
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;                                              // wave-uniform
    const size_t qrow = (size_t)qset * a.qSetStrideRows + i, tbase = (size_t)tset * a.tSetStrideRows;
    const int node = a.qgnode[qrow];
    uint32_t k1 = kKeyInit, k2 = kKeyInit;
    if (node >= 0) {
        const uint4 qlo = a.qdesc[2 * qrow], qhi = a.qdesc[2 * qrow + 1];
        for (int j = lane; j < nt; j += 64) {
            if (a.tgnode[tbase + j] == node) {
                const uint32_t d = hamming256(qlo, qhi, a.tdesc[2 * (tbase + j)], a.tdesc[2 * (tbase + j) + 1]);
                const uint32_t key = (d << 23) | (uint32_t)j;
                k2 = umed3(k1, k2, key); k1 = min(k1, key);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
        const uint32_t lo1 = min(k1, o1), hi1 = max(k1, o1);
        k2 = min(hi1, min(k2, o2)); k1 = lo1;
    }
    if (lane == 0) {
        const bool any = k1 != kKeyInit;                              // the node exists in the train frame too (:78)
        const int d1 = min((int)(k1 >> 23), 256), d2 = min((int)(k2 >> 23), 256);
        const bool keep = any && d1 <= a.threshold && (float)d1 < a.nnRatio * (float)d2;     // :104-105
        ssvio_dmatch m; m.queryIdx = i; m.trainIdx = d1 < 256 ? (int)(k1 & 0x7FFFFF) : -1; m.imgIdx = -1; m.distance = (float)d1;
        const size_t o = (size_t)pair * a.maxRows + i;
        a.cand[o] = m; a.keepNode[o] = keep ? node : -1;
    }
}

// rank of a kept match = number of kept matches with a smaller (node id, query index)
__global__ __launch_bounds__(256) void k_grouped_compact(GroupedArgs a)
{
    __shared__ int32_t tile[1024];
    const int pair = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair] : 0;
    const int nq = min(a.qcounts[qset], a.maxRows);
    if ((int)(blockIdx.x * 256) >= max(nq, 1)) return;               // block-uniform
    const int32_t* keepNode = a.keepNode + (size_t)pair * a.maxRows;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int mine = i < nq ? keepNode[i] : -1;
    int rank = 0, total = 0;
    for (int j0 = 0; j0 < nq; j0 += 1024) {
        const int rows = min(1024, nq - j0);
        __syncthreads();
        for (int j = threadIdx.x; j < rows; j += 256) tile[j] = keepNode[j0 + j];
        __syncthreads();
        for (int j = 0; j < rows; ++j) {
            const int other = tile[j];                                // same address in every lane: LDS broadcast
            total += other >= 0;
            rank += (other >= 0) && (other < mine || (other == mine && j0 + j < i));
        }
    }
    if (mine >= 0) a.out[(size_t)pair * a.maxRows + rank] = a.cand[(size_t)pair * a.maxRows + i];
    if (i == 0) a.nout[pair] = total;
}

void launch_match_grouped(const GroupedArgs& a, int maxQ, int npairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_match_grouped, dim3((maxQ + 3) / 4, npairs), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_grouped_compact, dim3((maxQ + 255) / 256, npairs), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------
// Sorted form.  k_bow_sort: one workgroup per set, bitonic sort of (key << 32 | index) in LDS, so that every vocabulary
// node's features are one contiguous run in ascending feature index (a FeatureVector entry, laid out flat) and the
// runs follow each other in ascending node id (the std::map order); stopped words (key -1 = 0xFFFFFFFF) sort last.
__global__ __launch_bounds__(1024) void k_bow_sort(BowSortArgs a)
{
    __shared__ unsigned long long key[kBowSortMax];
    const int set = blockIdx.x, tid = threadIdx.x;
    const int n = min(min(a.counts[set], a.maxRows), kBowSortMax);
    const size_t base = (size_t)set * a.setStrideRows;
    int P = 1024;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += 1024)
        key[i] = i < n ? (((unsigned long long)(uint32_t)a.gnode[base + i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = key[i], y = key[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { key[i] = y; key[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 1024) {
        a.sortedIdx[base + i] = (int32_t)(uint32_t)key[i];
        a.sortedNode[base + i] = (int32_t)(uint32_t)(key[i] >> 32);
    }
}

void launch_bow_sort(const BowSortArgs& a, int nsets, hipStream_t s)
{
    hipLaunchKernelGGL(k_bow_sort, dim3(nsets), dim3(1024), 0, s, a);
}

// one thread per query feature, taken in sorted rank order: its node's run in the train set is found by binary search and
// walked in ascending train index -- literally the loop of src/FeatureMatching.cpp:82-102 over vIdx2.  The train set's
// sorted keys and indices are staged in LDS once per workgroup (2 x 4 bytes per row, <= 64 KB), so the searches and the
// run walk stay on chip and only the candidates' descriptors are fetched from HBM / L2.
__global__ __launch_bounds__(256) void k_match_sorted(SortedMatchArgs a)
{
    extern __shared__ int32_t lds[];
    const int pair = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair] : 0, tset = a.pairs ? a.pairs[2 * pair + 1] : 0;
    const int nq = min(a.qcounts[qset], a.maxRows), nt = min(min(a.tcounts[tset], a.maxT), kBowSortMax);
    if ((int)(blockIdx.x * 256) >= nq) return;                          // block-uniform
    const size_t qb = (size_t)qset * a.qSetStrideRows, tb = (size_t)tset * a.tSetStrideRows;
    int32_t* sNode = lds; int32_t* sIdx = lds + nt;
    for (int j = threadIdx.x; j < nt; j += 256) { sNode[j] = a.tSortedNode[tb + j]; sIdx[j] = a.tSortedIdx[tb + j]; }
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nq) return;
    const int node = a.qSortedNode[qb + r], i = a.qSortedIdx[qb + r];
    int bestDist1 = 256, bestDist2 = 256, bestIdx = -1;                 // :85-87
    int lo = 0, hi = 0;
    if (node >= 0) {
        int l = 0, h = nt;                                              // first rank whose key is >= node (keys compare unsigned)
        while (l < h) { const int mid = (l + h) >> 1; if ((uint32_t)sNode[mid] < (uint32_t)node) l = mid + 1; else h = mid; }
        lo = l; h = nt;                                                 // first rank whose key is > node
        while (l < h) { const int mid = (l + h) >> 1; if ((uint32_t)sNode[mid] <= (uint32_t)node) l = mid + 1; else h = mid; }
        hi = l;
    }
    if (lo < hi) {
        const uint4 qlo = a.qdesc[2 * (qb + i)], qhi = a.qdesc[2 * (qb + i) + 1];
#pragma unroll 4
        for (int p = lo; p < hi; ++p) {
            const int j = sIdx[p];
            const int dist = (int)hamming256(qlo, qhi, a.tdesc[2 * (tb + j)], a.tdesc[2 * (tb + j) + 1]);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }      // :94-101
            else if (dist < bestDist2) bestDist2 = dist;
        }
    }
    const bool keep = lo < hi && bestDist1 <= a.threshold && (float)bestDist1 < a.nnRatio * (float)bestDist2;   // :104-105
    ssvio_dmatch m; m.queryIdx = i; m.trainIdx = bestIdx; m.imgIdx = -1; m.distance = (float)bestDist1;
    const size_t o = (size_t)pair * a.maxRows + r;
    a.cand[o] = m; a.keep[o] = keep ? 1 : 0;
}

// kept matches in rank order = ascending (node id, query index), the reference's push_back order
__global__ __launch_bounds__(256) void k_sorted_compact(SortedMatchArgs a)
{
    __shared__ int wsum[4];
    const int pair = blockIdx.x;
    const int qset = a.pairs ? a.pairs[2 * pair] : 0, tset = a.pairs ? a.pairs[2 * pair + 1] : 0;
    const int nq = a.tcounts[tset] > 0 ? min(a.qcounts[qset], a.maxRows) : 0;   // :61-63: nothing with an empty train set
    const size_t ob = (size_t)pair * a.maxRows;
    int written = 0;
    for (int r0 = 0; r0 < nq; r0 += 256) {
        const int r = r0 + threadIdx.x;
        const bool keep = r < nq && a.keep[ob + r];
        int total;
        const int off = block_excl_scan(keep ? 1 : 0, wsum, total);
        if (keep) a.out[ob + written + off] = a.cand[ob + r];
        written += total;
    }
    if (threadIdx.x == 0) a.nout[pair] = written;
}

void launch_match_sorted(const SortedMatchArgs& a, int maxQ, int maxT, int npairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_match_sorted, dim3((maxQ + 255) / 256, npairs), dim3(256), (size_t)min(maxT, kBowSortMax) * 8, s, a);      // LDS: keys + indices of the largest train set
    hipLaunchKernelGGL(k_sorted_compact, dim3(npairs), dim3(256), 0, s, a);
}

}  // namespace ssvio
