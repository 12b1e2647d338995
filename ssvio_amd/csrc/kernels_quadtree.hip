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
constexpr int kQtRegKeys = 8;            // keys per thread kept in registers (8192 per level); the rest stay in HBM scratch

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
// Short-round form (quotas up to MAXA - 3 with MAXA = 64 * NPL nodes): the same array formulation, but a round costs two
// workgroup barriers instead of ~15.  Everything about NODES (who is expandable, the split prefix, child numbering, the new
// node list) is done by wave 0 alone with wave-wide scans -- lane l owns list positions NPL*l .. NPL*l + NPL-1 --, and
// everything about KEYS (relabelling to the new node, counting into the new node's quadrants for the NEXT round) is one fused
// pass of all waves, with the quadrant counters double-buffered:
//     [all waves]  relabel keys (round r-1), count quadrants of the new nodes into cc[r & 1]          -- barrier
//     [wave 0]     node phase of round r: reads cc[r & 1], writes child indices into it, new list, clears cc[(r+1) & 1]
//                                                                                                     -- barrier
// Only a round in the largest-first regime (src/Feature.cpp:707-765, normally the last one) adds two barriers: its ordering
// (size descending, then list position ascending) is a rank count done by all threads.
// The creation sequence numbers of the long form are not needed: the list is always in reverse creation order (children
// newest first in front of the survivors), so "latest created first" among equal sizes IS "smaller list position first".
template <int NPL>
__device__ __forceinline__ int wave_excl_scan(int v, int& total)
{
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    total = __shfl(incl, 63);
    return incl - v;
}

struct QtsShared { int n, Lsize, nExpand, careful, done, nE, nsplit, nc; };

template <int MAXA, int THREADS>
__global__ __launch_bounds__(THREADS) void k_quadtree_short(QtArgs a)
{
    constexpr int NPL = MAXA / 64;                 // nodes per lane of wave 0
    constexpr int KPT = kQtRegKeys;                // keys per thread kept in registers
    __shared__ uint2 rect[2][MAXA];
    __shared__ int cnt[2][MAXA];
    __shared__ __attribute__((aligned(16))) int cc[2][4 * MAXA];          // quadrant counts, then child list positions, per node
    __shared__ int remap[MAXA];
    __shared__ unsigned long long skeys[MAXA];
    __shared__ int ord2[MAXA];
    __shared__ unsigned char split[MAXA];
    __shared__ unsigned long long best[MAXA];
    __shared__ QtsShared sh;

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

    if (tid == 0) { sh.n = 0; sh.nExpand = 0; sh.careful = 0; sh.done = 0; }
    for (int i = tid; i < MAXA; i += THREADS) { cnt[0][i] = 0; }
    for (int i = tid; i < 4 * MAXA; i += THREADS) { cc[0][i] = 0; }
    __syncthreads();
    // ---- 0. resolve the per-cell threshold fallback, gather the used entries (value to kval, order key to kord)
    for (int i0 = 0; i0 < nraw; i0 += THREADS) {
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
            if (lane == 0) base = atomicAdd(&sh.n, __popcll(m));
            base = __shfl(base, 0);
            if (use) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1));
                kval[pos] = e.y; kord[pos] = e.x & 0x3FFFFFFFu;
            }
        }
    }
    __syncthreads();
    const int n = sh.n;
    if (n == 0) { if (tid == 0) *selCount = 0; return; }
    uint32_t rv[KPT]; int rn[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) { const int k = tid + j * THREADS; rv[j] = k < n ? kval[k] : 0u; rn[j] = 0; }
    const int nReg = min(n, KPT * THREADS);                  // keys [nReg, n) use kval / knode in HBM

    // ---- 1. roots (:574-619)
    const int nIni = L.nIni;
    const float hX = L.hX;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        if (tid + j * THREADS < n) {
            int b = (int)((float)(rv[j] & 0xFFF) / hX);          // vpInitNodes[kp.pt.x / hX], :599
            b = min(max(b, 0), nIni - 1);
            rn[j] = b;
            atomicAdd(&cnt[0][b], 1);
        }
    }
    for (int k = nReg + tid; k < n; k += THREADS) {
        int b = (int)((float)(kval[k] & 0xFFF) / hX);
        b = min(max(b, 0), nIni - 1);
        knode[k] = b;
        atomicAdd(&cnt[0][b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < nIni; ++i) {
            if (cnt[0][i] > 0) {                                  // empty roots are erased (:611-614)
                const int c = cnt[0][i];
                rect[0][m] = make_uint2((uint32_t)L.rootX[i] | 0u << 16, (uint32_t)L.rootX[i + 1] | (uint32_t)H << 16);
                remap[i] = m; cnt[0][m] = c; ++m;
            } else remap[i] = 0;
        }
        sh.Lsize = m;
    }
    __syncthreads();
    // keys move to their root's list position and are counted into its quadrants: the first round's counts
    auto quadrant = [](const uint2 r, uint32_t v) -> int {
        const int x = v & 0xFFF, y = (v >> 12) & 0xFFF;
        const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
        const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);   // ceil(d / 2), :21-22
        return (x < hx ? 0 : 1) + (y < hy ? 0 : 2);
    };
    {
        auto first = [&](int root, uint32_t v) -> int {
            const int nn = remap[root];
            if (cnt[0][nn] >= 2) atomicAdd(&cc[0][4 * nn + quadrant(rect[0][nn], v)], 1);
            return nn;
        };
#pragma unroll
        for (int j = 0; j < KPT; ++j) if (tid + j * THREADS < n) rn[j] = first(rn[j], rv[j]);
        for (int k = nReg + tid; k < n; k += THREADS) knode[k] = first(knode[k], kval[k]);
    }
    __syncthreads();

    // ---- 2. rounds.  The list size and the regime of a round are read from LDS BEFORE the barrier that ends the previous round:
    // wave 0 rewrites them in its node phase, which no barrier separates from the top of the round.
    int cur = 0;
    int Lsize = sh.Lsize;
    bool careful = false;
    __syncthreads();
    for (int round = 0; round < 64; ++round) {
        uint2* R = rect[cur]; int* C = cnt[cur]; int* CC = cc[cur];
        uint2* R2 = rect[cur ^ 1]; int* C2 = cnt[cur ^ 1]; int* CC2 = cc[cur ^ 1];
        // ---- node phase, part 1 (wave 0): expandable nodes; in the largest-first regime their sort keys
        int myE[NPL], myQ[NPL];
        int eBase = 0, nE = 0;
        if (tid < 64) {
            int nMine = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int i = lane * NPL + j;
                myE[j] = 0; myQ[j] = 0;
                if (i < Lsize && C[i] >= 2) {
                    const int4 q = *(const int4*)&CC[4 * i];
                    myE[j] = 1; myQ[j] = (q.x > 0) + (q.y > 0) + (q.z > 0) + (q.w > 0);
                    ++nMine;
                }
            }
            eBase = wave_excl_scan<NPL>(nMine, nE);
            if (careful) {
                int t = eBase;
#pragma unroll
                for (int j = 0; j < NPL; ++j) if (myE[j]) { const int i = lane * NPL + j; skeys[t++] = ((unsigned long long)(uint32_t)C[i] << 12) | (unsigned long long)(MAXA - 1 - i); }
            }
            if (lane == 0) { sh.nE = nE; if (nE == 0) sh.done = 1; }
        }
        if (careful) {
            // largest first, ties: latest created (= smaller list position) first (= ascending sort() walked from the back, :716-717)
            __syncthreads();
            const int nEs = sh.nE;
            for (int e = tid; e < nEs; e += THREADS) {
                const unsigned long long me = skeys[e];
                int rank = 0;
                for (int u = 0; u < nEs; ++u) rank += skeys[u] > me ? 1 : 0;
                ord2[rank] = MAXA - 1 - (int)(me & 0xFFF);
            }
            __syncthreads();
        }
        // ---- node phase, part 2 (wave 0): the split prefix, child numbering, the new list
        if (tid < 64 && nE > 0) {
            int nd[NPL], dq[NPL];             // node at my positions of the split order, its number of children
            int nsplit = nE, nc;
            if (!careful) {
                // all expandable nodes, in list order: position t = eBase + (ordinal among my nodes)
                int myC = 0;
#pragma unroll
                for (int j = 0; j < NPL; ++j) myC += myQ[j];
                int totC;
                int cBase = wave_excl_scan<NPL>(myC, totC);
                nc = totC;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int i = lane * NPL + j;
                    nd[j] = myE[j] ? i : -1; dq[j] = cBase;          // dq = children created before this node
                    cBase += myQ[j];
                    if (i < MAXA) split[i] = (unsigned char)myE[j];
                }
            } else {
                int myD = 0, q4[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int t = lane * NPL + j;
                    nd[j] = -1; q4[j] = 0;
                    if (t < nE) {
                        nd[j] = ord2[t];
                        const int4 q = *(const int4*)&CC[4 * nd[j]];
                        q4[j] = (q.x > 0) + (q.y > 0) + (q.z > 0) + (q.w > 0);
                        myD += q4[j] - 1;
                    }
                }
                int totD;
                int run = wave_excl_scan<NPL>(myD, totD);          // exclusive sum of (children - 1) over the order
                int cut = 1 << 30;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int t = lane * NPL + j;
                    dq[j] = run + t;                                // children created before order[t]
                    if (t < nE) { run += q4[j] - 1; if (Lsize + run >= N) cut = min(cut, t); }     // :762
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cut = min(cut, __shfl_xor(cut, o));
                nsplit = min(cut, nE - 1) + 1;
                // children in total = those before order[nsplit-1] plus its own
                int last = 0;
#pragma unroll
                for (int j = 0; j < NPL; ++j) if (lane * NPL + j == nsplit - 1) last = dq[j] + q4[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
                nc = last;
#pragma unroll
                for (int j = 0; j < NPL; ++j) { const int i = lane * NPL + j; if (i < MAXA) split[i] = 0; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < NPL; ++j) { const int t = lane * NPL + j; if (t < nsplit) split[nd[j]] = 1; else nd[j] = -1; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            // survivors keep their relative order behind the new children
            int mySurv = 0, surv[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) { const int i = lane * NPL + j; surv[j] = (i < Lsize && !split[i]) ? 1 : 0; mySurv += surv[j]; }
            int nsurv;
            int sBase = wave_excl_scan<NPL>(mySurv, nsurv);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int i = lane * NPL + j;
                if (surv[j]) { const int idx = nc + sBase++; R2[idx] = R[i]; C2[idx] = C[i]; remap[i] = idx; }
            }
            // children, pushed to the front one by one (n1..n4, :655-694) => reverse creation order
            int myExpand = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (nd[j] >= 0) {
                    const int node = nd[j];
                    const uint2 r = R[node];
                    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
                    const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);
                    int m = dq[j];
                    int4 q = *(const int4*)&CC[4 * node];
                    int qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = qq[k];
                        if (c > 0) {
                            const int idx = nc - 1 - m;
                            const int cx0 = (k & 1) ? hx : x0, cx1 = (k & 1) ? x1 : hx, cy0 = (k & 2) ? hy : y0, cy1 = (k & 2) ? y1 : hy;
                            R2[idx] = make_uint2((uint32_t)cx0 | (uint32_t)cy0 << 16, (uint32_t)cx1 | (uint32_t)cy1 << 16);
                            C2[idx] = c;
                            qq[k] = idx;                              // child position for the key relabelling
                            if (c > 1) ++myExpand;
                            ++m;
                        }
                    }
                    *(int4*)&CC[4 * node] = make_int4(qq[0], qq[1], qq[2], qq[3]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) myExpand += __shfl_xor(myExpand, o);
            const int newSize = nc + nsurv;
            // the next round's quadrant counters
            for (int i = lane; i < newSize; i += 64) *(int4*)&CC2[4 * i] = make_int4(0, 0, 0, 0);
            if (lane == 0) {
                sh.Lsize = newSize;
                if (newSize >= N || newSize == Lsize) sh.done = 1;                       // :703 / :769
                if (!careful && newSize + 3 * myExpand > N) sh.careful = 1;              // :707
            }
        }
        __syncthreads();
        if (sh.nE == 0) break;                                     // nothing to divide: the list stays as it is
        // ---- key phase (all waves): move every key to its new node; unless this was the last round, count it there
        const bool last = sh.done != 0;
        Lsize = sh.Lsize; careful = sh.careful != 0;               // the next round's, stable until its node phase
        {
            auto relabel = [&](int pn, uint32_t v) -> int {
                const int nn = split[pn] ? CC[4 * pn + quadrant(R[pn], v)] : remap[pn];
                if (!last && C2[nn] >= 2) atomicAdd(&CC2[4 * nn + quadrant(R2[nn], v)], 1);
                return nn;
            };
#pragma unroll
            for (int j = 0; j < KPT; ++j) if (tid + j * THREADS < n) rn[j] = relabel(rn[j], rv[j]);
            for (int k = nReg + tid; k < n; k += THREADS) knode[k] = relabel(knode[k], kval[k]);
        }
        cur ^= 1;
        __syncthreads();
        if (last) break;
    }

    // ---- 3. best key of every node (:778-796): largest response, then first in input order
    const int Lfin = sh.Lsize;
    for (int i = tid; i < Lfin; i += THREADS) best[i] = 0;
    __syncthreads();
    auto offer = [&](int node, uint32_t v, uint32_t o) {
        const unsigned long long key = ((unsigned long long)(v >> 24) << 56) | ((unsigned long long)(0x3FFFFFFFu - o) << 24) | (v & 0xFFFFFFu);
        atomicMax(&best[node], key);
    };
#pragma unroll
    for (int j = 0; j < KPT; ++j) { const int k = tid + j * THREADS; if (k < n) offer(rn[j], rv[j], kord[k]); }
    for (int k = nReg + tid; k < n; k += THREADS) offer(knode[k], kval[k], kord[k]);
    __syncthreads();
    for (int i = tid; i < Lfin; i += THREADS) {
        const unsigned long long b = best[i];
        const uint32_t x = ((uint32_t)b & 0xFFF) + kMinBorder, y = (((uint32_t)b >> 12) & 0xFFF) + kMinBorder;   // :905-906
        sel[i] = make_uint2(x | (y << 16), (uint32_t)level | ((uint32_t)(b >> 56) << 8));
    }
    if (tid == 0) *selCount = Lfin;
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
        attrSet = true;
    }
}

int launch_quadtree(const QtArgs& a, int maxAlive, int nlevels, int batch, hipStream_t s)
{
    quadtree_prepare();
    // the smallest node capacity that holds quota + 3: 19 KB of LDS at 256 nodes (73 KB at 1024), so that the workgroups of
    // other kernels (k_fast, k_describe of the step in flight beside this one) still fit on the CU
    static const bool longForm = getenv("SSVIO_QUADTREE_LONG") != nullptr;      // measuring knob: the round structure with ~15 barriers
    if (!longForm && maxAlive <= 256) hipLaunchKernelGGL((k_quadtree_short<256, 256>), dim3(nlevels * batch), dim3(256), 0, s, a);
    else if (maxAlive <= 256) hipLaunchKernelGGL((k_quadtree<256, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(256), s, a);
    else if (maxAlive <= 1024) hipLaunchKernelGGL((k_quadtree<1024, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(1024), s, a);
    else if (maxAlive <= 2048) hipLaunchKernelGGL((k_quadtree<2048, false>), dim3(nlevels * batch), dim3(kQtThreads), quadtree_lds_bytes(2048), s, a);
    else if (maxAlive <= 8192 && a.nodeScratch) hipLaunchKernelGGL((k_quadtree<8192, true>), dim3(nlevels * batch), dim3(kQtThreads), 0, s, a);
    else if (maxAlive <= 32768 && a.nodeScratch) hipLaunchKernelGGL((k_quadtree<32768, true>), dim3(nlevels * batch), dim3(kQtThreads), 0, s, a);
    else return -1;
    return 0;
}

}  // namespace ssvio
