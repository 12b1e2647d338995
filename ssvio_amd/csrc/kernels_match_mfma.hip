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
#include <type_traits>
#include <utility>

namespace ssvio {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMStage = 128;                 // train rows per LDS stage (4 tiles of 32)
constexpr int kMFBase = 8160;                // row field of row 0 of the tile in flight (8160 + 31 < 8192)
// The keys are kept as int32 values whose bit patterns are also positive normal floats (2.0 +- 2^21 ulps), so that the one
// three-input median the compiler exposes as a builtin, v_med3_f32, orders them exactly like the integer instructions do.
constexpr int kMBias = 0x40000000;
constexpr int kMSent = 0x7F000000;           // "no candidate": stays above every real key through all the -32 steps

__device__ __forceinline__ int med3k(int a, int b, int c)
{
    return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(a), __int_as_float(b), __int_as_float(c)));
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

// one dword of a thread's 16 bytes of a train row -> two 16-byte fragment pieces (lane halves 0 and 1 of k-step s)
__device__ __forceinline__ void stage_write_dword(v4i* frag, int row, int s, uint32_t d)
{
    v4i* p = frag + ((row >> 5) * 8 + s) * 64 + (row & 31);
    p[0] = expand_train(d & 0xFFFFu);
    p[32] = expand_train(d >> 16);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), expanded at compile time
template <class F, int... U>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }

// the last tile of a chunk, and the few tiles around it: the two smallest of {m1 <= m2} and a tile's 16 accumulators of one
// query column; rows at or beyond `left` do not exist
__device__ __forceinline__ void select16_ragged(int& m1, int& m2, const v16i acc, int h, int left)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = (8 * (r >> 2) + 4 * h + (r & 3)) < left ? acc[r] : kMSent;
        m2 = med3k(m1, m2, v); m1 = min(m1, v);
    }
    m1 -= 32; m2 -= 32;
}

// The inner loop is written slot by slot, a slot being one matrix instruction and the vector work that is issued in its
// shadow (32 cycles), with scheduling fences between the slots so that the order below is the order of the instruction stream:
//   - the matrix instruction of k-step s of the chain (tile, query group g);
//   - the read of the A fragment three slots ahead (a ring of four fragment registers);
//   - two values (four in slot 1, none in slot 0) of the selection out of the accumulators of the chain BEFORE this one (group g - 1 of this tile, or the last
//     group of the previous tile): the chains are skewed by one, so one accumulator set per group is enough and a wave keeps
//     the matrix pipe and the vector ALU busy at the same time.  Second smallest of {m1, m2, a, b} = min(m2, second smallest
//     of {m1, a, b}) because m1 <= m2: 1.5 instructions per value; 0.5 when only the nearest neighbour is asked for
//     (SECOND = false: the filters of modes 0, 2 and 4 never read the second key);
//   - every G-th slot, one nibble of the thread's share of the next stage's bit expansion.
// Two waves per SIMD that each alternate between a matrix phase and a vector phase fall into step and leave each pipe idle
// half of the time; this order does not.
template <int G, int MINW, bool SECOND>
__global__ __launch_bounds__(256, MINW) void k_hamming_knn2_mfma(MatchArgs a)
{
    static_assert(G >= 2, "the chain being selected and the chain being computed need different accumulators");
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
    const int nTiles = (cEnd - cBegin + 31) >> 5;
    uint2* out = a.partial + ((size_t)pair * a.nChunks + chunk) * a.maxRows;
    constexpr int QWG = 4 * G * 32;
    const int srow = tid & 127, shalf = tid >> 7;                // this thread's share of a stage: 16 bytes (half) of row srow
    const int wofs = ((srow >> 5) * 8 + 4 * shalf) * 64 + (srow & 31);   // where its first fragment piece goes (v4i units)

    // accumulator start: the row field of this lane's 16 rows of a tile
    v16i C;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[r] = kMBias + kMFBase + 8 * (r >> 2) + 4 * h + (r & 3);

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
        if (nTiles <= 0) {                                        // block-uniform: an empty chunk still owes its (empty) lists
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int qi = qtile + (wave * G + g) * 32 + col;
                if (h == 0 && qi < nq) out[qi] = make_uint2(kKeyInit, kKeyInit);
            }
            continue;
        }
        // a thread's 16 bytes of stage st (rows cBegin + 128 st ...).  Rows beyond the chunk are never selected (ragged last tile),
        // so the address is only kept inside the set
        auto load_stage = [&](int st) { return T[(size_t)min(cBegin + st * kMStage + srow, nt - 1) * 2 + shalf]; };
        __syncthreads();                                          // (a second query tile: the previous one is done with the buffers)
        {
            const uint4 v = load_stage(0);
            stage_write_dword(frag[0], srow, 4 * shalf + 0, v.x); stage_write_dword(frag[0], srow, 4 * shalf + 1, v.y);
            stage_write_dword(frag[0], srow, 4 * shalf + 2, v.z); stage_write_dword(frag[0], srow, 4 * shalf + 3, v.w);
        }
        uint4 pre = load_stage(1), pre2 = pre;                    // the stage being expanded; the one behind it, in flight
        __syncthreads();

        v16i acc[G];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[G - 1][r] = kMSent;      // "the chain before the first": selecting it changes nothing
        // stages all of whose tiles are full and followed by another tile
        const int nMain = (nTiles - 1) >> 2;
        for (int st = 0; st < nMain; ++st) {
            const v4i* fp = frag[st & 1] + lane;
            v4i* wp = frag[(st + 1) & 1] + wofs;
            pre2 = load_stage(st + 2);
            const uint32_t pd[4] = {pre.x, pre.y, pre.z, pre.w};
            v4i Ar[4];
            Ar[0] = fp[0]; Ar[1] = fp[64]; Ar[2] = fp[128];
            v4i piece;
            static_for([&](auto uc) {                             // slot u = ((tile I) * G + g) * 8 + s
                constexpr int u = decltype(uc)::value;
                constexpr int I = u / (8 * G), g = (u / 8) % G, s = u % 8, v = u % (8 * G);
                constexpr int p = (g + G - 1) % G;
                __builtin_amdgcn_sched_barrier(0);
                acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ar[u & 3], B[g][s], s == 0 ? C : acc[g], 0, 0, 0);
                if (u + 3 < 32 * G) {
                    constexpr int u3 = u + 3;
                    Ar[u3 & 3] = fp[(u3 / (8 * G)) * 8 * 64 + (u3 % 8) * 64];
                }
                // (slot 0 of a chain selects nothing: the previous chain's last matrix instruction is still on its way)
                constexpr int r0 = s == 0 ? 0 : s == 1 ? 0 : 2 * s, r1 = s == 0 ? 0 : 2 * s + 2;
#pragma unroll
                for (int r = r0; r < r1; r += 2) {
                    if (SECOND) m2[p] = min(m2[p], med3k(m1[p], acc[p][r], acc[p][r + 1]));
                    m1[p] = min(min(m1[p], acc[p][r]), acc[p][r + 1]);
                }
                if (s == 7) { m1[p] -= 32; if (SECOND) m2[p] -= 32; }
                if (v % G == 0) {                                 // nibble j of dword I of the thread's share of the next stage
                    constexpr int j = v / G;
                    piece[j & 3] = (int)spread4((pd[I] >> (4 * j)) & 15u);
                    if ((j & 3) == 3) wp[I * 64 + (j >> 2) * 32] = piece;
                }
            }, std::make_integer_sequence<int, 32 * G>{});
            __builtin_amdgcn_sched_barrier(0);
            pre = pre2;
            __syncthreads();
        }
        // the 1..4 tiles left, all in stage nMain; every tile before the last is full
        {
            // the chain still waiting for its selection (group G - 1 of the last tile above, or the sentinel chain)
            if (SECOND) { select16_ragged(m1[G - 1], m2[G - 1], acc[G - 1], h, 32); }
            else { int d = kMSent; select16_ragged(m1[G - 1], d, acc[G - 1], h, 32); }
            for (int tt = nMain * 4; tt < nTiles; ++tt) {
                const v4i* fp = frag[nMain & 1] + (tt & 3) * 8 * 64 + lane;
                const int left = min(32, cEnd - (cBegin + tt * 32));
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const v4i A = fp[s * 64];
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[g][s], s == 0 ? C : acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) select16_ragged(m1[g], m2[g], acc[g], h, left);
            }
        }
        // back to the (dist << 23 | trainIdx) keys: the row field now counts from the end of the last tile
        const int tEnd = cBegin + nTiles * 32;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t k[2];
            // group G - 1 was selected once more than the others (the sentinel chain before the first tile): one more -32 ...
            // which only ever touched sentinels, so its real keys carry the same offsets as everyone's
            const int m[2] = {m1[g], m2[g]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int K = m[i] - kMBias + 8192 * na[g];
                k[i] = m[i] >= 0x70000000 ? kKeyInit : ((uint32_t)(K >> 13) << 23) | (uint32_t)(tEnd + (K & 8191) - kMFBase);
            }
            // the other half-wave holds the same query's keys over the other 16 rows of every tile
            const uint32_t o1 = __shfl_xor(k[0], 32), o2 = __shfl_xor(k[1], 32);
            const uint32_t lo1 = min(k[0], o1), hi1 = max(k[0], o1);
            const uint32_t k2 = SECOND ? min(hi1, min(k[1], o2)) : kKeyInit;
            const int qi = qtile + (wave * G + g) * 32 + col;
            if (h == 0 && qi < nq) out[qi] = make_uint2(lo1, k2);
        }
    }
}

// query groups of 32 per wave (tuning knob: SSVIO_MFMA_G): 2 or 3 -- the skewed chains need at least two accumulator sets, and four
// groups of fragments do not fit the 256 registers that two workgroups per CU leave a wave
int knn2_mfma_groups(int maxQ)
{
    static const int forced = [] { const char* e = getenv("SSVIO_MFMA_G"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 3) return forced;
    (void)maxQ;
    return 2;
}

int knn2_mfma_query_blocks(int maxQ, int slack)
{
    const int qwg = 4 * knn2_mfma_groups(maxQ) * 32;
    return (std::max(maxQ - slack, 1) + qwg - 1) / qwg;
}

template <bool SECOND>
static void launch_mfma_g(const MatchArgs& a, int G, dim3 grid, hipStream_t s)
{
    if (G == 3) hipLaunchKernelGGL((k_hamming_knn2_mfma<3, 2, SECOND>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_hamming_knn2_mfma<2, 2, SECOND>), grid, dim3(256), 0, s, a);
}

// needSecond = false: only the nearest neighbour is searched (the second key of every partial entry is kKeyInit)
void launch_knn2_mfma(const MatchArgs& a, int maxQ, int qBlocks, int npairs, bool needSecond, hipStream_t s)
{
    const int G = knn2_mfma_groups(maxQ);
    dim3 grid(qBlocks, a.nChunks, npairs);
    if (needSecond) launch_mfma_g<true>(a, G, grid, s); else launch_mfma_g<false>(a, G, grid, s);
}

}  // namespace ssvio
