// kernels_quadtree.hip -- DistributeOctTree (src/Feature.cpp:570-798, DivideNode :17-81) on the GPU:
// one 1024-thread workgroup per (frame, pyramid level), all node state in LDS.
//
// The reference's linked list / per-node key vectors reduce to arrays (see geometry.cpp for the
// same formulation on the host, which is tested against the oracle's linked-list restatement):
//   * a node is a rectangle [x0,x1) x [y0,y1); the std::list is "reverse creation order", so list
//     position = array index and a round rebuilds the array as [children, newest first][survivors];
//   * keys keep their input order inside every node, so "first key with the largest response"
//     (:783-793) = largest response, then smallest input-order key;
//   * a round = count keys per child quadrant (LDS atomics), choose the nodes to split (all
//     expandable nodes, or -- once size + 3*expandable > N (:707) -- the prefix of the
//     (size desc, creation desc) order that reaches N (:716-765)), number the children by a prefix
//     sum, relabel the keys.
// The candidates arrive as an unordered append list (k_fast); every entry carries its position in
// the reference's append order (cell-row-major, raster inside a cell, :851-891) as a 30-bit key, so
// no sorting or compaction pass is needed.  The 20 -> 7 threshold fallback (:875) is resolved here:
// an entry of a cell whose iniTh list is non-empty is used iff it passed NMS at iniTh.
#include "common.h"
#include <cstdlib>

namespace ssvio {

constexpr int kQtThreads = 1024;
// keys per thread kept in registers (and, in the 256-node short form, in LDS); the rest stay in HBM scratch.  8 / 4 / 3 / 2: the short form's
// workgroup takes 36.6 / 28.4 / 26.4 / 24.3 KB of LDS and 118 / 90 / .. VGPRs, i.e. 4 / 5 / 6 / 6+ fit a CU beside the other steps' kernels:
// 311 / 318 / 319 / 317 k frames/s sustained on the bench line (768-frame steps), although the kernel alone is fastest at 2 (280 us against 475)
constexpr int kQtRegKeys = 4;

struct QtShared {
    int wsum[16];
    int n, nE, cut, nExpand, total;
};

__device__ __forceinline__ int qt_block_scan_excl(int v, int* wsum, int& total)
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
    for (int i = 0; i < 16; ++i) { if (i < wid) base += wsum[i]; total += wsum[i]; }
    return base + incl - v;
}

// MAXA = capacity of the live node list (>= quota + 3 and >= 4 * roots); IT = nodes per thread.
// GLOBAL: the node arrays live in an HBM scratch block of this workgroup instead of LDS (quotas beyond 2045 keypoints per
// level: 73 bytes per node do not fit 160 KB).  Same code: the arrays are only touched by this workgroup, whose waves share
// the CU's vector L1, so a workgroup barrier orders the accesses exactly as it does for LDS; atomics go to L2.
template <int MAXA, bool GLOBAL>
__global__ __launch_bounds__(kQtThreads) void k_quadtree(QtArgs a)
{
    constexpr int IT = MAXA >= kQtThreads ? MAXA / kQtThreads : 1;     // nodes per thread (MAXA = 256: threads 256.. hold no node)
    extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
    __shared__ QtShared shGlobal;
    unsigned char* smem = GLOBAL ? a.nodeScratch + (size_t)blockIdx.x * a.nodeScratchStride : dynLds;
    // ---- LDS carve (all offsets multiples of 16)
    unsigned long long* skeys = (unsigned long long*)smem;                 // MAXA   sort keys / final best
    uint2* rect[2]; int* cnt[2]; int* seq[2];
    unsigned char* p = smem + sizeof(unsigned long long) * MAXA;
    rect[0] = (uint2*)p; p += sizeof(uint2) * MAXA; rect[1] = (uint2*)p; p += sizeof(uint2) * MAXA;
    cnt[0] = (int*)p; p += 4 * MAXA; cnt[1] = (int*)p; p += 4 * MAXA;
    seq[0] = (int*)p; p += 4 * MAXA; seq[1] = (int*)p; p += 4 * MAXA;
    int* cc = (int*)p; p += 16 * MAXA;                                       // 4 child counts / child indices per node
    int* ord = (int*)p; p += 4 * MAXA;
    int* ord2 = (int*)p; p += 4 * MAXA;
    int* remap = (int*)p; p += 4 * MAXA;
    int* cpre = (int*)p; p += 4 * MAXA;
    unsigned char* split = p; p += MAXA;
    QtShared* sh = GLOBAL ? &shGlobal : (QtShared*)p;

    const int tid = threadIdx.x, lane = tid & 63;
    int frame, level;
    xcd_frame_block(blockIdx.x, a.kl.nlevels, a.batch, frame, level);      // frame f -> XCD f % 8, like the kernels around it
    const KLevel& L = a.kl.lv[level];
    const int N = L.quota;
    const int H = L.h - 2 * kMinBorder;                       // maxY - minY
    uint32_t* kval = a.kval + (size_t)frame * a.rawPerFrame + L.rawBase;
    uint32_t* kord = a.kord + (size_t)frame * a.rawPerFrame + L.rawBase;
    uint32_t* knode = a.knode + (size_t)frame * a.rawPerFrame + L.rawBase;
    const uint2* raw = a.raw + (size_t)frame * a.rawPerFrame + L.rawBase;
    const int nraw = min(a.rawCount[frame * kMaxLevels + level], (int)L.rawCap);
    const int* cellIni = a.cellIni + (size_t)frame * a.cellsPerFrame + L.cellBase;
    int* selCount = a.selCount + frame * kMaxLevels + level;
    uint2* sel = a.sel + (size_t)frame * a.selPerFrame + L.selBase;

    if (tid == 0) { sh->n = 0; sh->nExpand = 0; }
    __syncthreads();
    // ---- 0. resolve the per-cell threshold fallback, gather the used entries
    for (int i0 = 0; i0 < nraw; i0 += kQtThreads) {
        const int i = i0 + tid;
        bool use = false; uint2 e = make_uint2(0, 0);
        if (i < nraw) {
            e = raw[i];
            const int cell = (e.x >> 12) & 0x3FFFF, flags = e.x >> 30;
            use = cellIni[cell] ? (flags & 1) : (flags & 2);
        }
        const unsigned long long m = __ballot(use);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh->n, __popcll(m));
            base = __shfl(base, 0);
            if (use) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1));
                kval[pos] = e.y; kord[pos] = e.x & 0x3FFFFFFFu;
            }
        }
    }
    __syncthreads();
    const int n = sh->n;
    if (n == 0) { if (tid == 0) *selCount = 0; return; }
    // key k = tid + 1024 j: value and node id live in registers for j < kQtRegKeys, so that the rounds below touch
    // LDS only (with every key in HBM scratch a round is two chains of dependent global loads: ~10 us)
    uint32_t rv[kQtRegKeys]; int rn[kQtRegKeys];
#pragma unroll
    for (int j = 0; j < kQtRegKeys; ++j) { const int k = tid + j * kQtThreads; rv[j] = k < n ? kval[k] : 0u; rn[j] = 0; }
    const int nReg = min(n, kQtRegKeys * kQtThreads);            // keys [nReg, n) use kval / knode in HBM

    // ---- 1. roots (:574-619)
    const int nIni = L.nIni;
    const float hX = L.hX;
    int cur = 0;
    if (tid < MAXA) { cnt[0][tid] = 0; }
    if (IT > 1) for (int j = 1; j < IT; ++j) cnt[0][tid + j * kQtThreads] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kQtRegKeys; ++j) {
        if (tid + j * kQtThreads < n) {
            int b = (int)((float)(rv[j] & 0xFFF) / hX);          // vpInitNodes[kp.pt.x / hX], :599
            b = min(max(b, 0), nIni - 1);
            rn[j] = b;
            atomicAdd(&cnt[0][b], 1);
        }
    }
    for (int k = nReg + tid; k < n; k += kQtThreads) {
        const int x = kval[k] & 0xFFF;
        int b = (int)((float)x / hX);
        b = min(max(b, 0), nIni - 1);
        knode[k] = b;
        atomicAdd(&cnt[0][b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < nIni; ++i) {
            if (cnt[0][i] > 0) {                                  // empty roots are erased (:611-614)
                rect[0][m] = make_uint2((uint32_t)L.rootX[i] | 0u << 16, (uint32_t)L.rootX[i + 1] | (uint32_t)H << 16);
                const int c = cnt[0][i];
                remap[i] = m; cnt[0][m] = c; seq[0][m] = -i; ++m;
            } else remap[i] = 0;
        }
        sh->total = m;
    }
    __syncthreads();
    int Lsize = sh->total;
#pragma unroll
    for (int j = 0; j < kQtRegKeys; ++j) rn[j] = remap[rn[j]];
    for (int k = nReg + tid; k < n; k += kQtThreads) knode[k] = remap[knode[k]];
    __syncthreads();

    // ---- 2. rounds
    bool careful = false;
    int seqCounter = 0;
    for (int round = 0; round < 64; ++round) {
        const int prevSize = Lsize;
        uint2* R = rect[cur]; int* C = cnt[cur]; int* S = seq[cur];
        uint2* R2 = rect[cur ^ 1]; int* C2 = cnt[cur ^ 1]; int* S2 = seq[cur ^ 1];
        for (int i = tid; i < 4 * Lsize; i += kQtThreads) cc[i] = 0;
        if (tid == 0) { sh->cut = 1 << 30; sh->nExpand = 0; }
        __syncthreads();
        // (b) key counts per child quadrant (DivideNode :50-69)
        auto count_key = [&](int pn, uint32_t v) {
            if (C[pn] >= 2) {
                const int x = v & 0xFFF, y = (v >> 12) & 0xFFF;
                const uint2 r = R[pn];
                const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
                const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);   // ceil(d / 2), :21-22
                atomicAdd(&cc[4 * pn + (x < hx ? 0 : 1) + (y < hy ? 0 : 2)], 1);
            }
        };
#pragma unroll
        for (int j = 0; j < kQtRegKeys; ++j) if (tid + j * kQtThreads < n) count_key(rn[j], rv[j]);
        for (int k = nReg + tid; k < n; k += kQtThreads) count_key(knode[k], kval[k]);
        __syncthreads();
        // (c) expandable nodes in list order
        int myE[IT], myC[IT], nMine = 0;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int i = tid * IT + j;
            myE[j] = 0; myC[j] = 0;
            if (i < Lsize && C[i] >= 2) {
                myE[j] = 1;
                myC[j] = (cc[4 * i] > 0) + (cc[4 * i + 1] > 0) + (cc[4 * i + 2] > 0) + (cc[4 * i + 3] > 0);
                ++nMine;
            }
        }
        int nE;
        int eBase = qt_block_scan_excl(nMine, sh->wsum, nE);
#pragma unroll
        for (int j = 0; j < IT; ++j) if (myE[j]) ord[eBase++] = tid * IT + j;
        __syncthreads();
        if (nE == 0) break;                                       // nothing to divide: size stays, :703
        int* order = ord;
        if (careful) {
            // largest first, ties: latest created first (= ascending sort() walked from the back, :716-717)
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int e = tid * IT + j;
                if (e < nE) { const int nd = ord[e]; skeys[e] = ((unsigned long long)(uint32_t)C[nd] << 32) | (uint32_t)(S[nd] + (1 << 30)); }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int e = tid * IT + j;
                if (e < nE) {
                    const unsigned long long me = skeys[e];
                    int rank = 0;
                    for (int u = 0; u < nE; ++u) rank += skeys[u] > me ? 1 : 0;
                    ord2[rank] = ord[e];
                }
            }
            __syncthreads();
            order = ord2;
        }
        // (d) which prefix of `order` is split, children numbering
        int myD = 0, dj[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int t = tid * IT + j;
            dj[j] = 0;
            if (t < nE) {
                const int nd = order[t];
                dj[j] = (cc[4 * nd] > 0) + (cc[4 * nd + 1] > 0) + (cc[4 * nd + 2] > 0) + (cc[4 * nd + 3] > 0) - 1;
                myD += dj[j];
            }
        }
        int totD;
        int dBase = qt_block_scan_excl(myD, sh->wsum, totD);      // exclusive sum of (children - 1)
        {
            int run = dBase;
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int t = tid * IT + j;
                if (t < nE) {
                    cpre[t] = run + t;                            // children created before order[t]
                    run += dj[j];
                    if (careful && prevSize + run >= N) atomicMin(&sh->cut, t);     // :762
                }
            }
        }
        __syncthreads();
        const int nsplit = min(sh->cut, nE - 1) + 1;
        if (tid < MAXA) split[tid] = 0;
        if (IT > 1) for (int j = 1; j < IT; ++j) split[tid + j * kQtThreads] = 0;
        __syncthreads();
        // total children = cpre[nsplit-1] + children(order[nsplit-1])
        if (tid == 0) {
            const int nd = order[nsplit - 1];
            sh->total = cpre[nsplit - 1] + (cc[4 * nd] > 0) + (cc[4 * nd + 1] > 0) + (cc[4 * nd + 2] > 0) + (cc[4 * nd + 3] > 0);
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) { const int t = tid * IT + j; if (t < nsplit) split[order[t]] = 1; }
        __syncthreads();
        const int nc = sh->total;
        // (e) survivors keep their relative order behind the new children
        int mySurv = 0;
#pragma unroll
        for (int j = 0; j < IT; ++j) { const int i = tid * IT + j; if (i < Lsize && !split[i]) ++mySurv; }
        int nsurv;
        int sBase = qt_block_scan_excl(mySurv, sh->wsum, nsurv);
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int i = tid * IT + j;
            if (i < Lsize && !split[i]) {
                const int idx = nc + sBase++;
                R2[idx] = R[i]; C2[idx] = C[i]; S2[idx] = S[i];
                remap[i] = idx;
            }
        }
        // (f) children, pushed to the front one by one (n1..n4, :655-694) => reverse creation order
        int myExpand = 0;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int t = tid * IT + j;
            if (t < nsplit) {
                const int nd = order[t];
                const uint2 r = R[nd];
                const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
                const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);
                int m = cpre[t];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cc[4 * nd + q];
                    if (c > 0) {
                        const int idx = nc - 1 - m;
                        const int cx0 = (q & 1) ? hx : x0, cx1 = (q & 1) ? x1 : hx, cy0 = (q & 2) ? hy : y0, cy1 = (q & 2) ? y1 : hy;
                        R2[idx] = make_uint2((uint32_t)cx0 | (uint32_t)cy0 << 16, (uint32_t)cx1 | (uint32_t)cy1 << 16);
                        C2[idx] = c; S2[idx] = seqCounter + 1 + m;
                        cc[4 * nd + q] = idx;                     // child index for the key relabelling
                        if (c > 1) ++myExpand;
                        ++m;
                    }
                }
            }
        }
        if (myExpand) atomicAdd(&sh->nExpand, myExpand);
        __syncthreads();
        // (g) relabel the keys
        auto relabel = [&](int pn, uint32_t v) -> int {
            if (split[pn]) {
                const int x = v & 0xFFF, y = (v >> 12) & 0xFFF;
                const uint2 r = R[pn];
                const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
                const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);
                return cc[4 * pn + (x < hx ? 0 : 1) + (y < hy ? 0 : 2)];
            }
            return remap[pn];
        };
#pragma unroll
        for (int j = 0; j < kQtRegKeys; ++j) if (tid + j * kQtThreads < n) rn[j] = relabel(rn[j], rv[j]);
        for (int k = nReg + tid; k < n; k += kQtThreads) knode[k] = relabel(knode[k], kval[k]);
        const int nToExpand = sh->nExpand;
        __syncthreads();
        cur ^= 1;
        Lsize = nc + nsurv;
        seqCounter += nc;
        if (Lsize >= N || Lsize == prevSize) break;               // :703 / :769
        if (!careful && Lsize + 3 * nToExpand > N) careful = true;   // :707
    }

    // ---- 3. best key of every node (:778-796): largest response, then first in input order
    unsigned long long* best = skeys;
    for (int i = tid; i < Lsize; i += kQtThreads) best[i] = 0;
    __syncthreads();
    auto offer = [&](int node, uint32_t v, uint32_t o) {
        const unsigned long long key = ((unsigned long long)(v >> 24) << 56) | ((unsigned long long)(0x3FFFFFFFu - o) << 24) | (v & 0xFFFFFFu);
        atomicMax(&best[node], key);
    };
#pragma unroll
    for (int j = 0; j < kQtRegKeys; ++j) { const int k = tid + j * kQtThreads; if (k < n) offer(rn[j], rv[j], kord[k]); }
    for (int k = nReg + tid; k < n; k += kQtThreads) offer(knode[k], kval[k], kord[k]);
    __syncthreads();
    for (int i = tid; i < Lsize; i += kQtThreads) {
        const unsigned long long b = best[i];
        const uint32_t x = ((uint32_t)b & 0xFFF) + kMinBorder, y = (((uint32_t)b >> 12) & 0xFFF) + kMinBorder;   // :905-906
        sel[i] = make_uint2(x | (y << 16), (uint32_t)level | ((uint32_t)(b >> 56) << 8));
    }
    if (tid == 0) *selCount = Lsize;
}

// ---------------------------------------------------------------------------------------------------------------
// Short-round form for quotas up to 253 (MAXA = 256 list positions = 256 threads, one node per thread): the same array
// formulation with a round structure of 3 workgroup barriers instead of ~15.
//   * everything a round has to know about the NODES -- who is expandable, how many children come before a node, where the
//     survivors go, the new list size -- falls out of ONE packed block scan (expandable | children << 10 | survivor << 21:
//     wave scan in DPP, one barrier to exchange the four wave totals); every thread derives the round's decisions from the
//     scan totals itself, so no flags travel through LDS;
//   * everything about the KEYS is one fused, straight-line pass: move the key to its new node and count it into that node's
//     quadrants for the NEXT round, with the quadrant counters double-buffered (a thread's 8 keys overlap their LDS round trips);
//   * only the round in the largest-first regime (src/Feature.cpp:707-765, normally the last one) needs more: its order
//     (size descending, then list position ascending) is a rank count, and the split prefix a second scan.
// The creation sequence numbers of the long form are not needed: the list is always in reverse creation order (children
// newest first in front of the survivors), so "latest created first" among equal sizes IS "smaller list position first".
// The first 2048 keys of a level live in LDS / registers (value, order key); HBM scratch is only touched beyond that.
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);      // row_shr 1, 2, 4, 8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);     // row_bcast 15, 31
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
    return incl;
}

// exclusive scan over a workgroup of NW waves with ONE barrier; slot: NW ints not touched by the three scans before it
template <int NW>
__device__ __forceinline__ int block_excl_scan1(int v, int* slot, int& total)
{
    const int incl = wave_incl_scan_dpp(v);
    if ((threadIdx.x & 63) == 63) slot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int t = slot[w]; if (w < (int)(threadIdx.x >> 6)) base += t; total += t; }
    return base + incl - v;
}

#ifdef SSVIO_QT_TIMING
__device__ unsigned long long g_qtTimes[64];
#define QT_STAMP(K) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_qtTimes[K] = wall_clock64(); } while (0)
extern "C" int ssvio_debug_qt_times(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qtTimes), sizeof(g_qtTimes)); }
#else
#define QT_STAMP(K) do { } while (0)
#endif

// LDS of the short form, carved from one dynamic block (the 1024-node form needs 79 KB, above the static limit):
//   rec[2][MAXA] uint4 | cc[2][4 MAXA + 64] int | skeys[MAXA] u64 | ord2[MAXA] int | lval, lord [LDSKEYS ? 8 MAXA : 0] u32 | split[MAXA] u8
template <int MAXA, bool LDSKEYS>
constexpr size_t qt_short_lds_bytes()
{
    return (size_t)MAXA * (2 * 16 + 8 + 4 + 1) + 2 * (4 * MAXA + 64) * 4 + (LDSKEYS ? (size_t)2 * kQtRegKeys * MAXA * 4 : 0);
}

// MAXA = 256, LDSKEYS: quotas up to 253 (every level of 1000 features on VGA), the level's first 2048 keys staged in LDS.
// MAXA = 1024, !LDSKEYS: quotas up to 1021 (4000 features on 1080p); the keys are staged in the HBM scratch instead (one coalesced
// round trip before the rounds, one after), so that the workgroup's LDS stays at 79 KB.
template <int MAXA, bool LDSKEYS>
__global__ __launch_bounds__(MAXA) void k_quadtree_short(QtArgs a)
{
    constexpr int THREADS = MAXA, NW = THREADS / 64;
    constexpr int KPT = kQtRegKeys;                // keys per thread kept in registers (and LDS)
    constexpr int POSBITS = MAXA <= 256 ? 8 : 10;  // list position inside the largest-first round's sort key
    extern __shared__ __attribute__((aligned(16))) unsigned char dynLdsS[];
    uint4 (*rec)[MAXA] = (uint4 (*)[MAXA])dynLdsS;                        // node record: {x0 | y0 << 16, x1 | y1 << 16, keys in the node, -}
    int (*cc)[4 * MAXA + 64] = (int (*)[4 * MAXA + 64])(rec + 2);         // quadrant counts, then child list positions, per node; the last
                                                                          // 64 ints: where the counting atomics of keys that count nowhere land
    unsigned long long* skeys = (unsigned long long*)(cc + 2);            // the largest-first round's 32-bit sort keys; best key per node at the end
    int* ord2 = (int*)(skeys + MAXA);
    uint32_t* lval = (uint32_t*)(ord2 + MAXA);                            // the level's first keys: value, order key (LDSKEYS only)
    uint32_t* lord = lval + (LDSKEYS ? KPT * THREADS : 0);
    unsigned char* split = (unsigned char*)(lord + (LDSKEYS ? KPT * THREADS : 0));
    __shared__ int rootCnt0[kMaxRoots];
    __shared__ __attribute__((aligned(16))) uint4 rootTab[kMaxRoots];     // per root: {rectangle, keys, list position}
    __shared__ int slots[4][NW];
    __shared__ int shN, shExp[2], shNc;

    const int tid = threadIdx.x, lane = tid & 63;
    int frame, level;
    xcd_frame_block(blockIdx.x, a.kl.nlevels, a.batch, frame, level);
    const KLevel& L = a.kl.lv[level];
    const int N = L.quota;
    const int H = L.h - 2 * kMinBorder;
    uint32_t* kval = a.kval + (size_t)frame * a.rawPerFrame + L.rawBase;
    uint32_t* kord = a.kord + (size_t)frame * a.rawPerFrame + L.rawBase;
    uint32_t* knode = a.knode + (size_t)frame * a.rawPerFrame + L.rawBase;
    const uint2* raw = a.raw + (size_t)frame * a.rawPerFrame + L.rawBase;
    const int nraw = min(a.rawCount[frame * kMaxLevels + level], (int)L.rawCap);
    const int* cellIni = a.cellIni + (size_t)frame * a.cellsPerFrame + L.cellBase;
    int* selCount = a.selCount + frame * kMaxLevels + level;
    uint2* sel = a.sel + (size_t)frame * a.selPerFrame + L.selBase;

    QT_STAMP(0);
    if (tid == 0) { shN = 0; shExp[0] = 0; shExp[1] = 0; }
    if (tid < kMaxRoots) rootCnt0[tid] = 0;
    *(int4*)&cc[0][4 * tid] = make_int4(0, 0, 0, 0);
    __syncthreads();
    // ---- 0. resolve the per-cell threshold fallback, gather the used entries.  A thread's KPT entries (and then their cells' flags)
    // are requested together: one round trip to L2 per 2048 entries, and one LDS reservation per wave
    for (int i0 = 0; i0 < nraw; i0 += KPT * THREADS) {
        uint2 e[KPT]; int ini[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) { const int i = i0 + tid + j * THREADS; e[j] = i < nraw ? raw[i] : make_uint2(0, 0); }
#pragma unroll
        for (int j = 0; j < KPT; ++j) { const int i = i0 + tid + j * THREADS; ini[j] = i < nraw ? cellIni[(e[j].x >> 12) & 0x3FFFF] : 0; }
        unsigned useMask = 0;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = i0 + tid + j * THREADS;
            const int flags = e[j].x >> 30;
            if (i < nraw && (ini[j] ? (flags & 1) : (flags & 2))) useMask |= 1u << j;
        }
        const int mine = __popc(useMask);
        const int incl = wave_incl_scan_dpp(mine);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot) {
            int base = 0;
            if (lane == 63) base = atomicAdd(&shN, tot);
            int pos = __builtin_amdgcn_readlane(base, 63) + incl - mine;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                if (useMask & (1u << j)) {
                    if (LDSKEYS && pos < KPT * THREADS) { lval[pos] = e[j].y; lord[pos] = e[j].x & 0x3FFFFFFFu; }
                    else { kval[pos] = e[j].y; kord[pos] = e[j].x & 0x3FFFFFFFu; }
                    ++pos;
                }
            }
        }
    }
    __syncthreads();
    QT_STAMP(1);
    const int n = shN;
    if (n == 0) { if (tid == 0) *selCount = 0; return; }
    uint32_t rv[KPT]; int rn[KPT]; uint2 rr[KPT];            // per key: value, list position of its node, that node's rectangle
#pragma unroll
    for (int j = 0; j < KPT; ++j) { const int k = tid + j * THREADS; rv[j] = k < n ? (LDSKEYS ? lval[k] : kval[k]) : 0u; rn[j] = 0; }
    const int nReg = min(n, KPT * THREADS);                  // keys [nReg, n) use kval / kord / knode in HBM

    // ---- 1. roots (:574-619)
    const int nIni = L.nIni;
    const float hX = L.hX;
    if (nIni > 1) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int b = (int)((float)(rv[j] & 0xFFF) / hX);    // vpInitNodes[kp.pt.x / hX], :599
            rn[j] = tid + j * THREADS < n ? min(max(b, 0), nIni - 1) : -1;
        }
        for (int b = 0; b < nIni; ++b) {                         // one LDS add per wave and root
            int c = 0;
#pragma unroll
            for (int j = 0; j < KPT; ++j) c += rn[j] == b ? 1 : 0;
            const int w = wave_incl_scan_dpp(c);
            if (lane == 63 && w) atomicAdd(&rootCnt0[b], w);
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) rn[j] = max(rn[j], 0);
        for (int k = nReg + tid; k < n; k += THREADS) {
            int b = (int)((float)(kval[k] & 0xFFF) / hX);
            b = min(max(b, 0), nIni - 1);
            knode[k] = b;
            atomicAdd(&rootCnt0[b], 1);
        }
        __syncthreads();
    } else {                                                     // one root (every 4:3 image): all keys are its keys
        for (int k = nReg + tid; k < n; k += THREADS) knode[k] = 0;
    }
    // empty roots are erased (:611-614): thread i < nIni places root i (its list position = the number of non-empty roots before it)
    int Lsize = 0;
    {
        int myPos = 0, myCnt = 0;
#pragma unroll
        for (int i = 0; i < kMaxRoots; ++i) {
            const int c = __builtin_amdgcn_readfirstlane(nIni > 1 ? (i < nIni ? rootCnt0[i] : 0) : (i == 0 ? n : 0));     // uniform
            if (tid == i) { myPos = Lsize; myCnt = c; }
            if (c > 0) ++Lsize;
        }
        if (tid < nIni) {
            const uint4 r4 = make_uint4((uint32_t)L.rootX[tid] | 0u << 16, (uint32_t)L.rootX[tid + 1] | (uint32_t)H << 16, (uint32_t)myCnt, 0u);
            rootTab[tid] = make_uint4(r4.x, r4.y, r4.z, (uint32_t)myPos);
            if (myCnt > 0) rec[0][myPos] = r4;
        }
        if (nIni > 1) __syncthreads();                           // (with one root every thread already knows it all)
    }
    auto quadrant = [](const uint2 r, uint32_t v) -> int {
        const int x = v & 0xFFF, y = (v >> 12) & 0xFFF;
        const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
        const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);   // ceil(d / 2), :21-22
        return (x < hx ? 0 : 1) + (y < hy ? 0 : 2);
    };
    const int sink = 4 * MAXA + lane;                            // a cell of cc[] that belongs to no node
    // keys move to their root's list position and are counted into its quadrants: the first round's counts
    {
        const uint4 root0 = make_uint4((uint32_t)L.rootX[0] | 0u << 16, (uint32_t)L.rootX[1] | (uint32_t)H << 16, (uint32_t)n, 0u);
        auto first = [&](int root, uint32_t v, bool live, uint2& rct) -> int {
            const uint4 r4 = nIni > 1 ? rootTab[root] : root0;   // {rectangle, keys, list position}
            const int nn = (int)r4.w;
            rct = make_uint2(r4.x, r4.y);
            const bool counts = live && r4.z >= 2;
            atomicAdd(&cc[0][counts ? 4 * nn + quadrant(rct, v) : sink], 1);
            return nn;
        };
        {
            uint4 r8[KPT]; int dst8[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) r8[j] = nIni > 1 ? rootTab[rn[j]] : root0;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const bool live = tid + j * THREADS < n;
                rr[j] = make_uint2(r8[j].x, r8[j].y);
                dst8[j] = (live && r8[j].z >= 2) ? 4 * (int)r8[j].w + quadrant(rr[j], rv[j]) : sink;
                rn[j] = live ? (int)r8[j].w : 0;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) atomicAdd(&cc[0][dst8[j]], 1);
        }
        uint2 dummyRect;
        for (int k = nReg + tid; k < n; k += THREADS) knode[k] = first(knode[k], kval[k], true, dummyRect);
    }
    __syncthreads();

    // ---- 2. rounds
    QT_STAMP(2);
    int cur = 0, scanNo = 0;
    bool careful = false;
    for (int round = 0; round < 64; ++round) {
        QT_STAMP(3 + 2 * round);
        const uint4* REC = rec[cur]; int* CC = cc[cur];
        uint4* REC2 = rec[cur ^ 1]; int* CC2 = cc[cur ^ 1];
        const int i = tid;                                          // my list position
        const uint4 myRec = REC[i];
        const int myC = i < Lsize ? (int)myRec.z : 0;
        const bool e = myC >= 2;                                    // expandable
        const int4 q4 = *(const int4*)&CC[4 * i];
        const int q = e ? (q4.x > 0) + (q4.y > 0) + (q4.z > 0) + (q4.w > 0) : 0;
        if (tid == 0) shExp[(round + 1) & 1] = 0;
        int nE, nc, nsurv, sBase, dq = 0, node = -1;                // node: the node whose children I create (dq children come before them)
        int4 nq4 = q4;
        if (!careful) {
            // all expandable nodes are split, in list order
            int tot;
            const int ex = block_excl_scan1<NW>((e ? 1 : 0) | (q << 10) | ((i < Lsize && !e) ? 1 << 21 : 0), slots[scanNo++ & 3], tot);
            nE = tot & 1023; nc = (tot >> 10) & 2047; nsurv = tot >> 21;
            if (nE == 0) break;                                     // nothing to divide: the list stays as it is (:703)
            dq = (ex >> 10) & 2047; sBase = ex >> 21;
            split[i] = e ? 1 : 0;
            if (e) node = i;
        } else {
            int tot;
            const int eBase = block_excl_scan1<NW>(e ? 1 : 0, slots[scanNo++ & 3], nE);
            if (nE == 0) break;
            // largest first, ties: latest created (= smaller list position) first (= ascending sort() walked from the back, :716-717).
            // 32-bit keys: a node holds fewer than 2^24 keys (a level has at most 4128^2 / 4 candidates).
            uint32_t* sk = (uint32_t*)skeys;
            if (e) sk[eBase] = ((uint32_t)myC << POSBITS) | (uint32_t)(MAXA - 1 - i);
            if (tid >= nE && tid < ((nE + 3) & ~3)) sk[tid] = 0;     // pad the last group of four
            split[i] = 0;
            __syncthreads();
            if (tid < nE) {
                const uint32_t me = sk[tid];
                int rank = 0;
                for (int u = 0; u < nE; u += 4) {
                    const uint4 k4 = *(const uint4*)&sk[u];
                    rank += (k4.x > me) + (k4.y > me) + (k4.z > me) + (k4.w > me);
                }
                ord2[rank] = MAXA - 1 - (int)(me & ((1u << POSBITS) - 1u));
            }
            __syncthreads();
            int qd = 0;
            if (tid < nE) {
                node = ord2[tid];
                nq4 = *(const int4*)&CC[4 * node];
                qd = (nq4.x > 0) + (nq4.y > 0) + (nq4.z > 0) + (nq4.w > 0);
            }
            const int run = block_excl_scan1<NW>(tid < nE ? qd - 1 : 0, slots[scanNo++ & 3], tot);    // (children - 1) before me in the order
            dq = run + tid;                                          // children created before order[tid]
            // the split prefix ends with the first node that brings the list to N (:762).  Every node adds >= 0, so position tid is
            // split iff the nodes before it have not reached N yet, and it is the last one iff it does (or the order ends with it)
            const bool mineSplit = tid < nE && (tid == 0 || Lsize + run < N);
            if (mineSplit) { split[node] = 1; if (tid == nE - 1 || Lsize + run + qd - 1 >= N) shNc = dq + qd; }      // children in total
            else node = -1;
            __syncthreads();
            nc = shNc;
            sBase = block_excl_scan1<NW>((i < Lsize && !split[i]) ? 1 : 0, slots[scanNo++ & 3], nsurv);
        }
        // survivors keep their relative order behind the new children
        // (a survivor's four quadrant cells all name its new position: the key pass looks a key's next node up in one place)
        if (i < Lsize && !split[i]) { const int idx = nc + sBase; REC2[idx] = myRec; *(int4*)&CC[4 * i] = make_int4(idx, idx, idx, idx); }
        // children, pushed to the front one by one (n1..n4, :655-694) => reverse creation order
        int myExpand = 0;
        if (node >= 0) {
            const uint4 r = REC[node];
            const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
            const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);
            int m = dq;
            int qq[4] = {nq4.x, nq4.y, nq4.z, nq4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = qq[k];
                if (c > 0) {
                    const int idx = nc - 1 - m;
                    const int cx0 = (k & 1) ? hx : x0, cx1 = (k & 1) ? x1 : hx, cy0 = (k & 2) ? hy : y0, cy1 = (k & 2) ? y1 : hy;
                    REC2[idx] = make_uint4((uint32_t)cx0 | (uint32_t)cy0 << 16, (uint32_t)cx1 | (uint32_t)cy1 << 16, (uint32_t)c, 0u);
                    qq[k] = idx;                                  // child position for the key relabelling
                    if (c > 1) ++myExpand;
                    ++m;
                }
            }
            *(int4*)&CC[4 * node] = make_int4(qq[0], qq[1], qq[2], qq[3]);
        }
        {
            const int w = wave_incl_scan_dpp(myExpand);
            if (lane == 63 && w) atomicAdd(&shExp[round & 1], w);
        }
        const int newSize = nc + nsurv;
        if (i < newSize) *(int4*)&CC2[4 * i] = make_int4(0, 0, 0, 0);       // the next round's quadrant counters
        __syncthreads();
        QT_STAMP(4 + 2 * round);
        const int nToExpand = shExp[round & 1];
        const bool last = newSize >= N || newSize == Lsize;                 // :703 / :769
        // ---- key phase: move every key to its new node; unless this was the last round, count it there.  Straight-line code (keys
        // beyond n walk node 0 and count into a sink cell): the dependent LDS reads of a thread's 8 keys overlap instead of queueing
        {
            // a key knows its node's rectangle (registers): one read for the next node, one for that node's record, one count
            auto relabel = [&](int pn, uint32_t v, bool live, uint2& rct) -> int {
                const int nn = CC[4 * pn + quadrant(rct, v)];
                const uint4 r4 = REC2[nn];
                rct = make_uint2(r4.x, r4.y);
                const bool counts = live && !last && r4.z >= 2;
                atomicAdd(&CC2[counts ? 4 * nn + quadrant(rct, v) : sink], 1);
                return nn;
            };
            // the register keys in stages, so that the 8 look-ups of a stage are in flight together (written as one call per key the
            // compiler waits for every LDS answer before it asks the next question: 16 round trips in a row, 1.5 us per pass)
            int nn8[KPT]; uint4 r8[KPT]; int dst8[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) nn8[j] = CC[4 * rn[j] + quadrant(rr[j], rv[j])];
#pragma unroll
            for (int j = 0; j < KPT; ++j) r8[j] = REC2[nn8[j]];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const bool live = tid + j * THREADS < n;
                rr[j] = make_uint2(r8[j].x, r8[j].y);
                const int cell = 4 * nn8[j] + quadrant(rr[j], rv[j]);
                dst8[j] = (live && !last && r8[j].z >= 2) ? cell : sink;
                rn[j] = live ? nn8[j] : 0;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) atomicAdd(&CC2[dst8[j]], 1);
            for (int k = nReg + tid; k < n; k += THREADS) {
                const int pn = knode[k];
                const uint4 p4 = REC[pn];
                uint2 rct = make_uint2(p4.x, p4.y);
                knode[k] = relabel(pn, kval[k], true, rct);
            }
        }
        cur ^= 1;
        Lsize = newSize;
        if (!careful && newSize + 3 * nToExpand > N) careful = true;        // :707
        __syncthreads();
        if (last) break;
    }

    // ---- 3. best key of every node (:778-796): largest response, then first in input order
    QT_STAMP(40);
    unsigned long long* best = skeys;
    best[tid] = 0;
    __syncthreads();
    auto offer = [&](int nd, uint32_t v, uint32_t o) {
        const unsigned long long key = ((unsigned long long)(v >> 24) << 56) | ((unsigned long long)(0x3FFFFFFFu - o) << 24) | (v & 0xFFFFFFu);
        atomicMax(&best[nd], key);
    };
#pragma unroll
    for (int j = 0; j < KPT; ++j) { const int k = tid + j * THREADS; if (k < n) offer(rn[j], rv[j], LDSKEYS ? lord[k] : kord[k]); }
    for (int k = nReg + tid; k < n; k += THREADS) offer(knode[k], kval[k], kord[k]);
    __syncthreads();
    if (tid < Lsize) {
        const unsigned long long b = best[tid];
        const uint32_t x = ((uint32_t)b & 0xFFF) + kMinBorder, y = (((uint32_t)b >> 12) & 0xFFF) + kMinBorder;   // :905-906
        sel[tid] = make_uint2(x | (y << 16), (uint32_t)level | ((uint32_t)(b >> 56) << 8));
    }
    if (tid == 0) *selCount = Lsize;
    QT_STAMP(41);
}

size_t quadtree_lds_bytes(int maxa)
{
    return (size_t)maxa * (8 + 16 + 8 + 8 + 16 + 4 * 4 + 1) + sizeof(QtShared) + 64;
}

// raises the dynamic-LDS limit of the instantiations on the current device (once per device)
void quadtree_prepare()
{
    static bool attrSetDev[64] = {};                  // the attribute belongs to the device's copy of the kernel
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attrSet = attrSetDev[dev & 63];
    if (!attrSet) {
        (void)hipFuncSetAttribute((const void*)k_quadtree<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)quadtree_lds_bytes(256));
        (void)hipFuncSetAttribute((const void*)k_quadtree<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)quadtree_lds_bytes(1024));
        (void)hipFuncSetAttribute((const void*)k_quadtree<2048, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)quadtree_lds_bytes(2048));
        (void)hipFuncSetAttribute((const void*)k_quadtree_short<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(qt_short_lds_bytes<1024, false>()));
        attrSet = true;
    }
}

int launch_quadtree(const QtArgs& a, int maxAlive, int nlevels, int batch, hipStream_t s)
{
    quadtree_prepare();
    // the smallest node capacity that holds quota + 3: 28 KB of LDS at 256 nodes with the level's first 1024 keys (79 KB at 1024 nodes), so that the
    // workgroups of other kernels (k_fast, k_describe of the steps in flight beside this one) still fit on the CU
    static const bool longForm = getenv("SSVIO_QUADTREE_LONG") != nullptr;      // measuring knob: the round structure with ~15 barriers
    unsigned maxRawCap = 0;
    for (int l = 0; l < nlevels; ++l) maxRawCap = std::max(maxRawCap, a.kl.lv[l].rawCap);
    // (the largest-first round's sort key holds a node's key count above 10 bits of list position: levels of up to 2^22 candidates)
    constexpr size_t lds256 = qt_short_lds_bytes<256, true>(), lds1024 = qt_short_lds_bytes<1024, false>();
    if (!longForm && maxAlive <= 256) hipLaunchKernelGGL((k_quadtree_short<256, true>), dim3(nlevels * batch), dim3(256), lds256, s, a);
    else if (!longForm && maxAlive <= 1024 && maxRawCap < (1u << 22))
        hipLaunchKernelGGL((k_quadtree_short<1024, false>), dim3(nlevels * batch), dim3(1024), lds1024, s, a);
    else if (maxAlive <= 256) hipLaunchKernelGGL((k_quadtree<256, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(256), s, a);
    else if (maxAlive <= 1024) hipLaunchKernelGGL((k_quadtree<1024, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(1024), s, a);
    else if (maxAlive <= 2048) hipLaunchKernelGGL((k_quadtree<2048, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(2048), s, a);
    else if (maxAlive <= 8192 && a.nodeScratch) hipLaunchKernelGGL((k_quadtree<8192, true>), dim3(nlevels * batch), dim3(kQtThreads), 0, s, a);
    else if (maxAlive <= 32768 && a.nodeScratch) hipLaunchKernelGGL((k_quadtree<32768, true>), dim3(nlevels * batch), dim3(kQtThreads), 0, s, a);
    else return -1;
    return 0;
}

}  // namespace ssvio
