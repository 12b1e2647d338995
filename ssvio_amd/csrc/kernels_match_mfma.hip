// kernels_match_mfma.hip -- the brute-force Hamming 2-NN of kernels_match.hip on the matrix cores (selectable second kernel,
// SSVIO_MATCH_KERNEL=mfma / ssvio_matcher_set_kernel).  Same contract as k_hamming_knn2: per (pair, train chunk, query row) the
// two smallest keys (dist << 23 | trainIdx) -- ComputeMatchingScore + the best / second-best loop of
// src/FeatureMatching.cpp:85-102,211-228 with lowest-index ties -- into partial[pair][chunk][maxRows].
//
// A Hamming distance is an exact integer contraction.  With the query bits as a = +-64 (bit set / clear) and the train bits
// as t = 0 / -128 (clear / set), both int8,
//     sum_k t_k * a_k = -8192 * (n11 - n01),      hamming = n10 + n01 = |a| - (n11 - n01),
// so one chain of 8 v_mfma_i32_32x32x32_i8 (K = 8 x 32 = 256 bits) started on the accumulator C = F + row leaves
//     acc = 8192 * (hamming - |a|) + F + row
// in the lane that owns the query column: the distance sits above a 13-bit train-row field, i.e. the accumulator IS the
// sort key of the 2-NN search (up to the per-query constant 8192 * |a|, which does not change the order and is added back
// once at the end; the comparisons are signed).  The selection is then v_med3_i32 + v_min_i32 per descriptor pair, against
// 8 xor + 8 bcnt + 3 in k_hamming_knn2.  The row field counts relative to the tile being processed: after every 32-row tile
// the lane's two running keys are lowered by 32 instead of raising 16 accumulator constants, which bounds a train chunk to
// kMfmaChunkRows rows.  Every value is an integer far inside int32: the result is bit-identical to the popcount form.
//
// Layout: a workgroup of 4 waves holds 4 * G * 32 query rows as MFMA B fragments in registers (a lane owns the column
// lane & 31 of each of its G groups); the train chunk streams HBM -> registers -> (bits expanded to int8) -> LDS in stages of
// 128 rows, stored fragment-major (tile, k-step, lane) so that an A fragment is one conflict-free ds_read_b128 per wave; the
// k order inside a step is the same on both sides, which is all a dot product needs.
#include "match.h"
#include <algorithm>
#include <cstdlib>

namespace ssvio {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMStage = 128;                 // train rows per LDS stage (4 tiles of 32)
constexpr int kMFBase = 8160;                // row field of row 0 of the tile in flight (8160 + 31 < 8192)
constexpr int kMSent = 0x3FFFFFFF;           // "no candidate": stays above every real key through all the -32 steps

__device__ __forceinline__ int imed3(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 4 bits -> 4 bytes: 0x80 where the bit is set (bit i lands on bit 8i + 7; the shifted copies do not overlap, so no carries)
__device__ __forceinline__ uint32_t spread4(uint32_t nibble) { return (nibble * 0x10204080u) & 0x80808080u; }

// 16 descriptor bits -> 16 int8 of a train row: 0 / -128
__device__ __forceinline__ v4i expand_train(uint32_t bits16)
{
    v4i r;
    r.x = (int)spread4(bits16 & 15u); r.y = (int)spread4((bits16 >> 4) & 15u);
    r.z = (int)spread4((bits16 >> 8) & 15u); r.w = (int)spread4((bits16 >> 12) & 15u);
    return r;
}

// 16 descriptor bits -> 16 int8 of a query row: +64 where set, -64 where clear
__device__ __forceinline__ v4i expand_query(uint32_t bits16)
{
    v4i r = expand_train(bits16);
    r.x = (r.x ^ (int)0x80808080u) | 0x40404040; r.y = (r.y ^ (int)0x80808080u) | 0x40404040;
    r.z = (r.z ^ (int)0x80808080u) | 0x40404040; r.w = (r.w ^ (int)0x80808080u) | 0x40404040;
    return r;
}

// one thread's 16 bytes of a train row (dwords 4 * half .. 4 * half + 3) -> eight 16-byte fragment pieces in LDS
__device__ __forceinline__ void stage_write(v4i* frag, int row, int half, const uint4 raw)
{
    const int tile = row >> 5, r = row & 31;
    const uint32_t d[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = 4 * half + i;
        v4i* p = frag + (tile * 8 + s) * 64 + r;
        p[0] = expand_train(d[i] & 0xFFFFu);
        p[32] = expand_train(d[i] >> 16);
    }
}

template <int G, int MINW>
__global__ __launch_bounds__(256, MINW) void k_hamming_knn2_mfma(MatchArgs a)
{
    __shared__ v4i frag[2][(kMStage / 32) * 8 * 64];            // 2 x 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int qset = a.pairs ? a.pairs[2 * pair + a.swap] : 0;
    const int tset = a.pairs ? a.pairs[2 * pair + 1 - a.swap] : 0;
    const int nq = clamp_rows(a.qcounts[qset], a.maxRows), nt = clamp_rows(a.tcounts[tset], a.maxRows);
    const uint4* Q = a.qdesc + (size_t)qset * a.qSetStrideRows * 2;
    const uint4* T = a.tdesc + (size_t)tset * a.tSetStrideRows * 2;
    // the rows the set really holds, split evenly over the chunks (a.chunkRows: the host's upper bound, <= kMfmaChunkRows)
    const int rowsPer = min(a.chunkRows, ((nt + a.nChunks - 1) / a.nChunks + 31) & ~31);
    const int cBegin = min(nt, chunk * rowsPer), cEnd = min(nt, cBegin + rowsPer);
    uint2* out = a.partial + ((size_t)pair * a.nChunks + chunk) * a.maxRows;
    constexpr int QWG = 4 * G * 32;

    // accumulator start: the row field of this lane's 16 rows of a tile
    v16i C;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[r] = kMFBase + 8 * (r >> 2) + 4 * h + (r & 3);

    for (int qtile = blockIdx.x * QWG; qtile < nq; qtile += gridDim.x * QWG) {          // block-uniform
        v4i B[G][8];
        int m1[G], m2[G], na[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int qi = qtile + (wave * G + g) * 32 + col;
            uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
            if (qi < nq) { lo = Q[(size_t)qi * 2]; hi = Q[(size_t)qi * 2 + 1]; }
            const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            int n = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) { B[g][s] = expand_query(h ? q[s] >> 16 : q[s] & 0xFFFFu); n += __popc(q[s]); }
            na[g] = n; m1[g] = kMSent; m2[g] = kMSent;
        }
        int buf = 0;
        __syncthreads();                                          // (a second query tile: the previous one is done with the buffers)
        if (cBegin < cEnd) {
            const int rows = min(kMStage, cEnd - cBegin);
            if (tid < rows * 2) stage_write(frag[0], tid >> 1, tid & 1, T[(size_t)cBegin * 2 + tid]);
        }
        __syncthreads();
        for (int t0 = cBegin; t0 < cEnd; t0 += kMStage) {
            const int rows = min(kMStage, cEnd - t0);
            const int nextRows = min(kMStage, cEnd - (t0 + kMStage));
            uint4 pre = make_uint4(0, 0, 0, 0);
            if (tid < nextRows * 2) pre = T[(size_t)(t0 + kMStage) * 2 + tid];
            const int ntiles = (rows + 31) >> 5;
            for (int tile = 0; tile < ntiles; ++tile) {
                const v4i* fp = frag[buf] + tile * 8 * 64 + lane;
                v16i acc[G];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const v4i A = fp[s * 64];
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[g][s], s == 0 ? C : acc[g], 0, 0, 0);
                }
                const int left = cEnd - (t0 + tile * 32);         // valid rows of this tile, wave-uniform
                if (left >= 32) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { m2[g] = imed3(m1[g], m2[g], acc[g][r]); m1[g] = min(m1[g], acc[g][r]); }
                        m1[g] -= 32; m2[g] -= 32;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int v = (8 * (r >> 2) + 4 * h + (r & 3)) < left ? acc[g][r] : kMSent;
                            m2[g] = imed3(m1[g], m2[g], v); m1[g] = min(m1[g], v);
                        }
                        m1[g] -= 32; m2[g] -= 32;
                    }
                }
            }
            if (tid < nextRows * 2) stage_write(frag[buf ^ 1], tid >> 1, tid & 1, pre);
            __syncthreads();
            buf ^= 1;
        }
        // back to the (dist << 23 | trainIdx) keys: the row field now counts from the end of the last tile
        const int tEnd = cBegin + ((cEnd - cBegin + 31) & ~31);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t k[2];
            const int m[2] = {m1[g], m2[g]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int K = m[i] + 8192 * na[g];
                k[i] = m[i] >= 0x30000000 ? kKeyInit : ((uint32_t)(K >> 13) << 23) | (uint32_t)(tEnd + (K & 8191) - kMFBase);
            }
            // the other half-wave holds the same query's keys over the other 16 rows of every tile
            const uint32_t o1 = __shfl_xor(k[0], 32), o2 = __shfl_xor(k[1], 32);
            const uint32_t lo1 = min(k[0], o1), hi1 = max(k[0], o1);
            const uint32_t k2 = min(hi1, min(k[1], o2));
            const int qi = qtile + (wave * G + g) * 32 + col;
            if (h == 0 && qi < nq) out[qi] = make_uint2(lo1, k2);
        }
    }
}

// query groups of 32 per wave (tuning knob: SSVIO_MFMA_G).  3 groups = 384 query rows per workgroup in 244 registers: two
// workgroups per CU, one wave's selection under the other's matrix instructions
int knn2_mfma_groups(int maxQ)
{
    static const int forced = [] { const char* e = getenv("SSVIO_MFMA_G"); return e ? atoi(e) : 0; }();
    if (forced >= 1 && forced <= 4) return forced;
    return maxQ > 256 ? 3 : maxQ > 128 ? 2 : 1;
}

int knn2_mfma_query_blocks(int maxQ)
{
    const int qwg = 4 * knn2_mfma_groups(maxQ) * 32;
    return (std::max(maxQ - 32, 1) + qwg - 1) / qwg;
}

void launch_knn2_mfma(const MatchArgs& a, int maxQ, int npairs, hipStream_t s)
{
    const int G = knn2_mfma_groups(maxQ);
    dim3 grid(knn2_mfma_query_blocks(maxQ), a.nChunks, npairs);
    static const bool g4two = getenv("SSVIO_MFMA_G4_TWO") != nullptr;      // G = 4 squeezed into 256 registers (spills) for two waves per SIMD
    if (G == 4 && g4two) hipLaunchKernelGGL((k_hamming_knn2_mfma<4, 2>), grid, dim3(256), 0, s, a);
    else if (G == 4) hipLaunchKernelGGL((k_hamming_knn2_mfma<4, 1>), grid, dim3(256), 0, s, a);
    else if (G == 3) hipLaunchKernelGGL((k_hamming_knn2_mfma<3, 2>), grid, dim3(256), 0, s, a);
    else if (G == 2) hipLaunchKernelGGL((k_hamming_knn2_mfma<2, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_hamming_knn2_mfma<1, 2>), grid, dim3(256), 0, s, a);
}

}  // namespace ssvio
