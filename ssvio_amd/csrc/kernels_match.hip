// kernels_match.hip -- brute-force Hamming 2-NN on gfx950 and the match filters.
//
// k_hamming_knn2: each lane keeps R query descriptors (8 dwords each) in VGPRs; the train
// descriptors of the block's chunk stream through LDS in 4-KB tiles and are read back as
// wave-wide broadcasts (2 x ds_read_b128 per train row); per (query, train): 8 v_xor +
// 8 accumulating v_bcnt_u32_b32 + pack + v_med3/v_min on the key (dist << 23 | trainIdx),
// so the two smallest keys are exactly ComputeMatchingScore + the best/second-best loop of
// src/FeatureMatching.cpp:85-102,211-228 with lowest-index ties.  No MFMA: the path is
// integer xor/popcount, VALU-bound.
#include "match.h"
#include <algorithm>
#include <cstdlib>

namespace ssvio {

constexpr int kTT = 128;                 // train rows per LDS tile
// d += popcount(x): the accumulating form of v_bcnt_u32_b32 (one instruction per dword; left to itself the
// compiler splits the chain into bcnt(x, 0) + v_add3_u32, 3 more instructions per descriptor pair)
__device__ __forceinline__ void bcnt_acc(uint32_t& d, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
}

// d = popcount(x)
__device__ __forceinline__ uint32_t bcnt0(uint32_t x)
{
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x));
    return d;
}

// one train row against the lane's R queries: ComputeMatchingScore + the best/second-best update
// SECOND = false (MatchByBruteForce and the cross-checks never read the second key): one instruction less per pair
template <int R, bool SECOND = true>
__device__ __forceinline__ void row_update(const uint32_t (&q)[R][8], uint32_t (&k1)[R], uint32_t (&k2)[R],
                                           const uint4 lo, const uint4 hi, const uint32_t idx)
{
    // the R chains are interleaved dword by dword (volatile asm keeps the order): a v_bcnt that directly follows
    // the v_xor it depends on costs ~0.5 cycle more than one issued a few instructions later
    const uint32_t t[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t d[R], x[R];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[r]) : "v"(q[r][i]), "v"(t[i]));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (i == 0) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d[r]) : "v"(x[r]));
            else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[r]) : "v"(x[r]));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t key = (d[r] << 23) | idx;
        if (SECOND) k2[r] = umed3(k1[r], k2[r], key);     // second smallest of {k1,k2,key}, k1 <= k2
        k1[r] = min(k1[r], key);
    }
}

template <int R, bool SECOND>
__global__ __launch_bounds__(256) void k_hamming_knn2(MatchArgs a)
{
    __shared__ uint4 tile[2][kTT * 2 + 2];            // two buffers: the next tile lands while this one is consumed (+ a row of slack: the row prefetch)
    const int tid = threadIdx.x;
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair + a.swap] : 0;
    const int tset = a.pairs ? a.pairs[2 * pair + 1 - a.swap] : 0;
    const int nq = clamp_rows(a.qcounts[qset], a.maxRows), nt = clamp_rows(a.tcounts[tset], a.maxRows);
    const uint4* Q = a.qdesc + (size_t)qset * a.qSetStrideRows * 2;
    const uint4* T = a.tdesc + (size_t)tset * a.tSetStrideRows * 2;
    // the train rows a set actually holds are split evenly over the chunks (a.chunkRows is the host's upper bound, from the rows a
    // set CAN hold): no chunk is left empty or short when the sets are not full
    const int rowsPer = min(a.chunkRows, ((nt + a.nChunks - 1) / a.nChunks + 7) & ~7);
    const int cBegin = chunk * rowsPer, cEnd = min(nt, cBegin + rowsPer);
    uint2* out = a.partial + ((size_t)pair * a.nChunks + chunk) * a.maxRows;
    // The grid holds the query blocks the sets are EXPECTED to fill (launch_knn2: the extractor's sets come with up to 32 rows of
    // slack), so that no workgroup starts only to find itself empty; a block walks on to the (rare) query rows beyond the grid.
    for (int qtile = blockIdx.x * 256 * R; qtile < nq; qtile += gridDim.x * 256 * R) {      // block-uniform
        uint32_t q[R][8], k1[R], k2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = qtile + r * 256 + tid;
            uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
            if (qi < nq) { lo = Q[(size_t)qi * 2]; hi = Q[(size_t)qi * 2 + 1]; }
            q[r][0] = lo.x; q[r][1] = lo.y; q[r][2] = lo.z; q[r][3] = lo.w;
            q[r][4] = hi.x; q[r][5] = hi.y; q[r][6] = hi.z; q[r][7] = hi.w;
            k1[r] = kKeyInit; k2[r] = kKeyInit;
        }
        // train rows stream HBM/L2 -> registers -> LDS, one tile ahead of the compute
        int buf = 0;
        __syncthreads();                                          // (a second query tile: the previous one is done with the buffers)
        if (cBegin < cEnd) {
            const int rows = min(kTT, cEnd - cBegin);
            if (tid < rows * 2) tile[0][tid] = T[(size_t)cBegin * 2 + tid];
        }
        __syncthreads();
        for (int t0 = cBegin; t0 < cEnd; t0 += kTT) {
            const int rows = min(kTT, cEnd - t0);
            const int nextRows = min(kTT, cEnd - (t0 + kTT));
            uint4 pre = make_uint4(0, 0, 0, 0);
            if (tid < nextRows * 2) pre = T[(size_t)(t0 + kTT) * 2 + tid];
            const uint4* tl = tile[buf];
            // the next row's 32 bytes are read from LDS while this row is scored (the tile has one row of slack behind it)
            uint4 lo = tl[0], hi = tl[1];
            const int rows4 = rows & ~3;
            for (int j = 0; j < rows4; j += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 nlo = tl[2 * (j + u) + 2], nhi = tl[2 * (j + u) + 3];
                    row_update<R, SECOND>(q, k1, k2, lo, hi, (uint32_t)(t0 + j + u));
                    lo = nlo; hi = nhi;
                }
            }
            for (int j = rows4; j < rows; ++j) {
                const uint4 nlo = tl[2 * j + 2], nhi = tl[2 * j + 3];
                row_update<R, SECOND>(q, k1, k2, lo, hi, (uint32_t)(t0 + j));
                lo = nlo; hi = nhi;
            }
            if (tid < nextRows * 2) tile[buf ^ 1][tid] = pre;
            __syncthreads();
            buf ^= 1;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = qtile + r * 256 + tid;
            if (qi < nq) out[qi] = make_uint2(k1[r], k2[r]);
        }
    }
}

// Small single pairs (the per-frame call: ~1000 x 1000): the chunks of the train set are the 8 waves of one workgroup, which all
// hold the same 64 queries (one per lane) and merge their two smallest keys through LDS -- no k_match_merge launch behind it.
// Each wave stages 64 train rows at a time in its own LDS slice (no workgroup barrier in the loop) and reads them back as
// wave-wide broadcasts like the kernel above.  Output: partial[pair][0][q], i.e. nChunks = 1.
constexpr int kSmallWaves = 8;
template <bool SECOND>
__global__ __launch_bounds__(64 * kSmallWaves) void k_hamming_knn2_small(MatchArgs a)
{
    __shared__ uint4 rows[kSmallWaves][64 * 2];
    __shared__ uint2 keys[kSmallWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair + a.swap] : 0;
    const int tset = a.pairs ? a.pairs[2 * pair + 1 - a.swap] : 0;
    const int nq = clamp_rows(a.qcounts[qset], a.maxRows), nt = clamp_rows(a.tcounts[tset], a.maxRows);
    const int q0 = blockIdx.x * 64;
    if (q0 >= nq) return;                                      // block-uniform
    const uint4* Q = a.qdesc + (size_t)qset * a.qSetStrideRows * 2;
    const uint4* T = a.tdesc + (size_t)tset * a.tSetStrideRows * 2;
    const int qi = q0 + lane;
    uint32_t q[1][8], k1[1] = {kKeyInit}, k2[1] = {kKeyInit};
    {
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) { lo = Q[(size_t)qi * 2]; hi = Q[(size_t)qi * 2 + 1]; }
        q[0][0] = lo.x; q[0][1] = lo.y; q[0][2] = lo.z; q[0][3] = lo.w; q[0][4] = hi.x; q[0][5] = hi.y; q[0][6] = hi.z; q[0][7] = hi.w;
    }
    const int per = ((nt + kSmallWaves - 1) / kSmallWaves + 63) & ~63;            // rows per wave, a multiple of the 64-row tile
    const int tBeg = min(w * per, nt), tEnd = min(tBeg + per, nt);
    uint4* my = rows[w];
    for (int t0 = tBeg; t0 < tEnd; t0 += 64) {
        const int n = min(64, tEnd - t0);
        if (lane < n) { my[2 * lane] = T[(size_t)(t0 + lane) * 2]; my[2 * lane + 1] = T[(size_t)(t0 + lane) * 2 + 1]; }
        __builtin_amdgcn_wave_barrier();
        if (n == 64) {
#pragma unroll 4
            for (int j = 0; j < 64; ++j) row_update<1, SECOND>(q, k1, k2, my[2 * j], my[2 * j + 1], (uint32_t)(t0 + j));
        } else {
            for (int j = 0; j < n; ++j) row_update<1, SECOND>(q, k1, k2, my[2 * j], my[2 * j + 1], (uint32_t)(t0 + j));
        }
        __builtin_amdgcn_wave_barrier();
    }
    keys[w][lane] = make_uint2(k1[0], k2[0]);
    __syncthreads();
    if (w == 0 && qi < nq) {
        uint32_t m1 = kKeyInit, m2 = kKeyInit;
#pragma unroll
        for (int c = 0; c < kSmallWaves; ++c) {
            const uint2 p = keys[c][lane];
            m2 = umed3(m1, m2, p.x); m1 = min(m1, p.x);
            m2 = umed3(m1, m2, p.y); m1 = min(m1, p.y);
        }
        a.partial[(size_t)pair * a.maxRows + qi] = make_uint2(m1, m2);
    }
}

bool knn2_small_applies(int maxQ, int maxT, int npairs)
{
    static const bool off = getenv("SSVIO_NO_SMALL_MATCH") != nullptr;
    return !off && npairs <= 2 && maxQ <= 2048 && maxT <= 4096;
}

void launch_knn2_small(const MatchArgs& a, int maxQ, int npairs, bool needSecond, hipStream_t s)
{
    if (needSecond) hipLaunchKernelGGL(k_hamming_knn2_small<true>, dim3((maxQ + 63) / 64, npairs), dim3(64 * kSmallWaves), 0, s, a);
    else hipLaunchKernelGGL(k_hamming_knn2_small<false>, dim3((maxQ + 63) / 64, npairs), dim3(64 * kSmallWaves), 0, s, a);
}

// R = 4 keeps LDS reads at a quarter of the VALU time; small query sets use fewer so that more workgroups exist
int knn2_queries_per_lane(int maxQ)
{
    static const int forced = [] { const char* e = getenv("SSVIO_MATCH_R"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // one query per lane everywhere since the nearest-neighbour form of the kernel (round 3): with fewer registers per lane the extra
    // waves hide the LDS broadcasts better than R = 2 or 4 save them (tools/probe/match_r_sweep.py: 1006 / 2000 / 4000 / 8000-row sets,
    // R = 1 against the old choice: +3 %, +2 %, +1 %, +3 %; +7 % on the bench step's matcher with the other steps' kernels beside it)
    (void)maxQ;
    return 1;
}

// query blocks of a launch: those the sets are expected to fill.  A set handed over by the extractor can hold 32 rows more than it
// normally does (the quadtree's overshoot slack, 4 per level; slack = 32 from the batched entry points); the kernel walks on to rows
// beyond the grid, so this is only balance.  Single pairs from the host are exactly as long as they say (slack = 0).
int knn2_query_blocks(int maxQ, int slack)
{
    const int R = knn2_queries_per_lane(maxQ);
    return (std::max(maxQ - slack, 1) + 256 * R - 1) / (256 * R);
}

void launch_knn2(const MatchArgs& a, int maxQ, int qBlocks, int npairs, bool needSecond, hipStream_t s)
{
    const int R = knn2_queries_per_lane(maxQ);
    dim3 grid(qBlocks, a.nChunks, npairs);
    if (needSecond) {
        if (R == 4) hipLaunchKernelGGL((k_hamming_knn2<4, true>), grid, dim3(256), 0, s, a);
        else if (R == 2) hipLaunchKernelGGL((k_hamming_knn2<2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_hamming_knn2<1, true>), grid, dim3(256), 0, s, a);
    } else {
        if (R == 4) hipLaunchKernelGGL((k_hamming_knn2<4, false>), grid, dim3(256), 0, s, a);
        else if (R == 2) hipLaunchKernelGGL((k_hamming_knn2<2, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_hamming_knn2<1, false>), grid, dim3(256), 0, s, a);
    }
}

// ---------------------------------------------------------------------------
// merge the chunks' two smallest keys
__device__ __forceinline__ void merge_chunks(const uint2* __restrict__ partial, int nChunks, int maxRows, int i,
                                             uint32_t& k1, uint32_t& k2)
{
    k1 = kKeyInit; k2 = kKeyInit;
    auto take = [&](const uint2 p) {
        k2 = umed3(k1, k2, p.x); k1 = min(k1, p.x);
        k2 = umed3(k1, k2, p.y); k1 = min(k1, p.y);
    };
    const uint2* col = partial + i;
    int c = 0;
    for (; c + 4 <= nChunks; c += 4) {                   // four loads in flight
        const uint2 p0 = col[(size_t)c * maxRows], p1 = col[(size_t)(c + 1) * maxRows], p2 = col[(size_t)(c + 2) * maxRows], p3 = col[(size_t)(c + 3) * maxRows];
        take(p0); take(p1); take(p2); take(p3);
    }
    for (; c < nChunks; ++c) take(col[(size_t)c * maxRows]);
}

// fold the nChunks partial lists of every pair into chunk 0 (each thread owns one query row, so in place is safe);
// k_match_finalize then reads one uint2 per row instead of walking the chunks in a single workgroup
__global__ __launch_bounds__(256) void k_match_merge(uint2* partial, int nChunks, int maxRows, const int* counts,
                                                     const int* pairs, int side)
{
    const int pair = blockIdx.y;
    const int set = pairs ? pairs[2 * pair + side] : 0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= clamp_rows(counts[set], maxRows)) return;
    uint2* P = partial + (size_t)pair * nChunks * maxRows;
    uint32_t k1, k2;
    merge_chunks(P, nChunks, maxRows, i, k1, k2);
    P[i] = make_uint2(k1, k2);
}

// one workgroup per pair: filter and emit DMatch rows in ascending query order.
//   mode 0: MatchByBruteForce (src/FeatureMatching.cpp:35-54): keep d < max(2*min_d, 30)
//   mode 1: Threshold / NNRatio acceptance (src/FeatureMatching.cpp:104-111)
//   mode 2: mutual nearest neighbour
//   mode 3: raw (idx, d1, d2) triples, no filtering
//   mode 4: BFMatcher(crossCheck = true) the way OpenCV 3.x runs it (SURVEY App. A6 "OpenCV-compat"): only the reverse search
//           exists; every train row votes for its nearest query, a query keeps the closest of its voters (lowest train index on
//           ties).  The votes are scattered with atomicMin on the key (dist << 23 | trainIdx) into the pair's otherwise unused
//           forward list.
//
// NT = 256: the lists were folded by k_match_merge (nChunks = 1) or the set is large.  NT = 1024: per-frame sets (a row per thread):
// the thread merges its row's chunks itself, keeps the two keys in registers for both passes, and no merge launch runs before.
template <int NT>
__global__ __launch_bounds__(NT) void k_match_finalize(FinalizeArgs a)
{
    __shared__ int wsum[NT / 64];
    __shared__ int sMin;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int qset = a.pairs ? a.pairs[2 * pair] : 0;
    const int tset = a.pairs ? a.pairs[2 * pair + 1] : 0;
    const int nq = clamp_rows(a.qcounts[qset], a.maxRows), nt = clamp_rows(a.tcounts[tset], a.maxRows);
    const uint2* P = a.partial + (size_t)pair * a.pairStride;
    const uint2* PR = a.partialRev ? a.partialRev + (size_t)pair * a.pairStrideRev : nullptr;
    ssvio_dmatch* out = a.out ? a.out + (size_t)pair * a.maxRows : nullptr;
    if (nq <= 0 || nt <= 0) { if (tid == 0 && a.nout) a.nout[pair] = 0; return; }   // src/FeatureMatching.cpp:25-27

    if (a.mode == 3) {
        int32_t* raw = a.raw + (size_t)pair * a.maxRows * 3;
        for (int i = tid; i < nq; i += NT) {
            uint32_t k1, k2; merge_chunks(P, a.nChunks, a.maxRows, i, k1, k2);
            const int d1 = min((int)(k1 >> 23), 256), d2 = min((int)(k2 >> 23), 256);
            raw[i] = d1 < 256 ? (int)(k1 & 0x7FFFFF) : -1;       // 'dist < bestDist1' with bestDist1 = 256 never fires for 256
            raw[a.maxRows + i] = d1; raw[2 * a.maxRows + i] = d2;
        }
        return;
    }
    uint32_t* vote = (uint32_t*)const_cast<uint2*>(P);           // mode 4: [nq] best voter key per query
    if (a.mode == SSVIO_MODE_CROSSCHECK_CV) {
        for (int i = tid; i < nq; i += NT) __hip_atomic_store(&vote[i], kKeyInit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int t = tid; t < nt; t += NT) {
            uint32_t r1, r2; merge_chunks(PR, a.nChunksRev, a.maxRows, t, r1, r2);
            const uint32_t q = r1 & 0x7FFFFF;
            if (r1 != kKeyInit && q < (uint32_t)nq) atomicMin(&vote[q], (r1 & ~0x7FFFFFu) | (uint32_t)t);
        }
        __syncthreads();
    }
    // a row per thread: merged once, kept for both passes
    const bool held = nq <= NT && a.mode != SSVIO_MODE_CROSSCHECK_CV;
    uint32_t hk1 = kKeyInit, hk2 = kKeyInit;
    if (held && tid < nq) merge_chunks(P, a.nChunks, a.maxRows, tid, hk1, hk2);
    int minD = 1 << 20;
    if (a.mode == 0) {
        if (tid == 0) sMin = 1 << 20;
        __syncthreads();
        for (int i = tid; i < nq; i += NT) {
            uint32_t k1 = hk1, k2 = hk2; if (!held) merge_chunks(P, a.nChunks, a.maxRows, i, k1, k2);
            minD = min(minD, (int)(k1 >> 23));
        }
        atomicMin(&sMin, minD);
        __syncthreads();
        minD = sMin;
    }
    const int lim = max(2 * minD, 30);
    int written = 0;
    for (int i0 = 0; i0 < nq; i0 += NT) {
        const int i = i0 + tid;
        bool keep = false; uint32_t k1 = kKeyInit, k2 = kKeyInit;
        if (i < nq && a.mode == SSVIO_MODE_CROSSCHECK_CV) {
            k1 = __hip_atomic_load(&vote[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            keep = k1 != kKeyInit;
        } else if (i < nq) {
            if (held) { k1 = hk1; k2 = hk2; } else merge_chunks(P, a.nChunks, a.maxRows, i, k1, k2);
            const int d1 = (int)(k1 >> 23), d2 = min((int)(k2 >> 23), 256);
            if (a.mode == 0) keep = d1 < lim;
            else if (a.mode == 1) keep = d1 <= a.threshold && (float)d1 < a.nnRatio * (float)d2;
            else {
                uint32_t r1, r2; merge_chunks(PR, a.nChunksRev, a.maxRows, (int)(k1 & 0x7FFFFF), r1, r2);
                keep = (int)(r1 & 0x7FFFFF) == i;
            }
        }
        int total;
        const int off = block_excl_scan<NT>(keep ? 1 : 0, wsum, total);
        if (keep) {
            ssvio_dmatch m;
            const int d1 = (int)(k1 >> 23);
            m.queryIdx = i; m.imgIdx = a.mode == 1 ? -1 : 0; m.distance = (float)d1;
            // :94 'dist < bestDist1' never fires at 256, so the 2-NN loop leaves bestIdx = -1 there; BFMatcher keeps the row
            m.trainIdx = (a.mode == 1 && d1 >= 256) ? -1 : (int)(k1 & 0x7FFFFF);
            out[written + off] = m;
        }
        written += total;
    }
    if (tid == 0) a.nout[pair] = written;
}

// ---------------------------------------------------------------------------
// MatchByProject (src/FeatureMatching.cpp:128-189): one wave per reference point.  The projection is the reference's
// double sequence (no contraction: this file is compiled with -ffp-contract=off); the lanes walk the train keypoints,
// test the +-radian window, and keep the two smallest keys (dist << 23 | j) of those inside, which a wave-wide
// shuffle reduction merges: smallest key = first minimal distance in ascending j (:166-170), second = sec_min_dist.
__global__ __launch_bounds__(256) void k_match_project(ProjectArgs a)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.nq) return;                                               // wave-uniform
    const double x = a.p3d[3 * i], y = a.p3d[3 * i + 1], z = a.p3d[3 * i + 2];
    double px = a.R[0] * x + a.R[1] * y + a.R[2] * z + a.t[0];           // :137
    double py = a.R[3] * x + a.R[4] * y + a.R[5] * z + a.t[1];
    const double pz = a.R[6] * x + a.R[7] * y + a.R[8] * z + a.t[2];
    const double inv = 1. / pz;                                          // :138, Mat /= s multiplies by 1./s
    px = px * inv; py = py * inv;
    const double u = a.K[0] * px + a.K[2], v = a.K[4] * py + a.K[5];     // :139-140
    const bool inside = !(u < 0 || u > a.K[2] || v < 0 || v > a.K[5]);   // :141
    uint32_t k1 = kKeyInit, k2 = kKeyInit;
    if (inside) {
        const uint4 qlo = a.qdesc[2 * (size_t)i], qhi = a.qdesc[2 * (size_t)i + 1];
        const double ulo = u - a.radian, uhi = u + a.radian, vlo = v - a.radian, vhi = v + a.radian;
        for (int j = lane; j < a.nt; j += 64) {
            const double x2 = a.p2d[2 * j], y2 = a.p2d[2 * j + 1];
            if (ulo < x2 && uhi > x2 && vlo < y2 && vhi > y2) {          // :151-152
                const uint4 lo = a.tdesc[2 * (size_t)j], hi = a.tdesc[2 * (size_t)j + 1];
                const uint32_t d = __popc(qlo.x ^ lo.x) + __popc(qlo.y ^ lo.y) + __popc(qlo.z ^ lo.z) + __popc(qlo.w ^ lo.w) +
                                   __popc(qhi.x ^ hi.x) + __popc(qhi.y ^ hi.y) + __popc(qhi.z ^ hi.z) + __popc(qhi.w ^ hi.w);
                const uint32_t key = (d << 23) | (uint32_t)j;
                k2 = umed3(k1, k2, key); k1 = min(k1, key);
            }
        }
    }
    // wavefront-shuffle argmin: merge the lanes' (smallest, second smallest) pairs
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
        const uint32_t lo1 = min(k1, o1), hi1 = max(k1, o1);
        k2 = min(hi1, min(k2, o2)); k1 = lo1;
    }
    if (lane == 0) {
        const int minDist = min((int)(k1 >> 23), 256), secDist = min((int)(k2 >> 23), 256);
        const bool keep = inside && minDist < a.threshold && (float)minDist < a.nnRatio * (float)secDist;   // :177-178
        // :166 'dist < min_dist' never fires at 256 (nor with an empty window), so best_id stays -1 there; with Threshold > 256 such a
        // row is still emitted (:177)
        ssvio_dmatch m; m.queryIdx = i; m.trainIdx = minDist < 256 ? (int)(k1 & 0x7FFFFF) : -1; m.imgIdx = -1; m.distance = (float)minDist;
        a.cand[i] = m; a.keep[i] = keep ? 1 : 0;
    }
}

// matches in ascending reference-point order (the reference's push_back order)
__global__ __launch_bounds__(256) void k_project_compact(ProjectArgs a)
{
    __shared__ int wsum[4];
    int written = 0;
    for (int i0 = 0; i0 < a.nq; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < a.nq && a.keep[i];
        int total;
        const int off = block_excl_scan(keep ? 1 : 0, wsum, total);
        if (keep) a.out[written + off] = a.cand[i];
        written += total;
    }
    if (threadIdx.x == 0) *a.nout = written;
}

void launch_project(const ProjectArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_match_project, dim3((a.nq + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_project_compact, dim3(1), dim3(256), 0, s, a);
}

// per-pair summary record {pair id, number of matches, sum of their distances}: one wave per pair walks the pair's DMatch
// list (distances are integers 0..256 held in floats, so the integer sum is exact)
__global__ __launch_bounds__(256) void k_pair_summary(const ssvio_dmatch* __restrict__ out, const int* __restrict__ nout, int maxRows,
                                                      const int* __restrict__ pairs, int npairs, int idMode, int idBase,
                                                      ssvio_pair_summary* __restrict__ rec)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;                                   // wave-uniform
    const int n = clamp_rows(nout[p], maxRows);
    const ssvio_dmatch* M = out + (size_t)p * maxRows;
    long long sum = 0;
    for (int i = lane; i < n; i += 64) sum += (long long)(int)M[i].distance;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
        int id = idBase + p;
        if (idMode == SSVIO_PAIR_ID_TRIANGULAR) {
            const int a = pairs[2 * p], b = pairs[2 * p + 1];
            const long long hi = max(a, b), lo = min(a, b);
            id = (int)(hi * (hi - 1) / 2 + lo);
        }
        ssvio_pair_summary r; r.pair = id; r.n_matches = n; r.sum_dist = sum;
        rec[p] = r;
    }
}

void launch_pair_summary(const ssvio_dmatch* out, const int* nout, int maxRows, const int* pairs, int npairs, int idMode, int idBase,
                         ssvio_pair_summary* rec, hipStream_t s)
{
    hipLaunchKernelGGL(k_pair_summary, dim3((npairs + 3) / 4), dim3(256), 0, s, out, nout, maxRows, pairs, npairs, idMode, idBase, rec);
}

void launch_finalize(const FinalizeArgs& a0, int npairs, int maxQ, int maxT, hipStream_t s)
{
    FinalizeArgs a = a0;
    a.pairStride = (size_t)a.nChunks * a.maxRows; a.pairStrideRev = (size_t)a.nChunksRev * a.maxRows;
    // per-frame sets without a reverse list: one launch, a row per thread (see k_match_finalize)
    const bool fused = !a.partialRev && a.mode != 3 && maxQ <= 2048 && a.nChunks > 1 && a.nChunks <= 64;
    if (a.nChunks > 1 && !fused) {
        hipLaunchKernelGGL(k_match_merge, dim3((maxQ + 255) / 256, npairs), dim3(256), 0, s,
                           const_cast<uint2*>(a.partial), a.nChunks, a.maxRows, a.qcounts, a.pairs, 0);
        a.nChunks = 1;
    }
    if (a.partialRev && a.nChunksRev > 1) {
        hipLaunchKernelGGL(k_match_merge, dim3((maxT + 255) / 256, npairs), dim3(256), 0, s,
                           const_cast<uint2*>(a.partialRev), a.nChunksRev, a.maxRows, a.tcounts, a.pairs, 1);
        a.nChunksRev = 1;
    }
    if (fused) hipLaunchKernelGGL(k_match_finalize<1024>, dim3(npairs), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(k_match_finalize<256>, dim3(npairs), dim3(256), 0, s, a);
}

}  // namespace ssvio
