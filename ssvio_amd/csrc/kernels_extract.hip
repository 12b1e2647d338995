// kernels_extract.hip -- gfx950 kernels of the ORB extractor.
//
// HBM layout: one pyramid per frame, level l at byte offset KLevels::offset[l],
// rows of KLevels::pitch[l] bytes (64-B aligned, >= 64 B slack), NO border rim:
// nothing in Extract reads the 19-px rim of src/Feature.cpp:806-821 (FAST windows
// live in [16,w-16), IC_Angle reaches 15 px and rBRIEF 18 px from keypoints that
// are >= 19 px inside).  A second pyramid of the same layout holds the blurred levels.
//
// All arithmetic is integer except the angle / rotation in k_describe, which is
// written with explicit round-to-nearest intrinsics so that no FMA contraction
// can change a bit relative to the reference's separate float operations.
#include "common.h"

// hipcc defines __fmul_rn/__fadd_rn/... as plain operators, so contraction has to be
// switched off for the whole translation unit (and the build passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace ssvio {

// ---------------------------------------------------------------------------
// Every level row carries kPadX reflect-101 columns on both sides (x = -4..-1 and w..w+3), written
// by the kernel that produces the level, so that the streaming blur can load plain aligned dwords
// at the image edges.  (This is a 4-px slice of the reference's 19-px rim, used only internally.)
__device__ __forceinline__ int refl101(int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * n - 2 - p : p; }
__device__ __forceinline__ int refl_clamp(int p, int n) { return min(max(refl101(p, n), 0), n - 1); }

// level 0: copy / RGB2GRAY (src/Frame.cpp:7; SURVEY App. A1: 14-bit fixed point).  Thread = output quad
// x0 = 4*qi - 4 .. x0+3 of one row.
template <int CH>
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ src, int w, int h, int stride, size_t frameStride,
                                                uint8_t* __restrict__ pyr, size_t pyrFrameBytes, int pitch0, int aligned4)
{
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4 - kPadX;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= w + kPadX || y >= h) return;
    const uint8_t* s = src + (size_t)blockIdx.z * frameStride + (size_t)y * stride;
    uint8_t* d = pyr + (size_t)blockIdx.z * pyrFrameBytes + (size_t)y * pitch0 + x0;
    uint32_t v = 0;
    if (CH == 1 && aligned4 && x0 >= 0 && x0 + 3 < w) v = *(const uint32_t*)(s + x0);
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = refl_clamp(x0 + i, w);
            uint32_t g;
            if (CH == 1) g = s[c];
            else g = (s[3 * c] * 4899u + s[3 * c + 1] * 9617u + s[3 * c + 2] * 1868u + 8192u) >> 14;
            v |= g << (8 * i);
        }
    }
    *(uint32_t*)d = v;           // rows are 4-B aligned with >= 64 B of slack on both sides
}

void launch_copy_gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                      uint8_t* pyr, size_t pyrFrameBytes, int pitch0, hipStream_t s)
{
    dim3 grid((w + 2 * kPadX + 255) / 256, (h + 3) / 4, batch);
    const int aligned4 = ((uintptr_t)src % 4 == 0) && (stride % 4 == 0) && (frameStride % 4 == 0);
    hipLaunchKernelGGL(k_level0<1>, grid, dim3(256), 0, s, src, w, h, stride, frameStride, pyr, pyrFrameBytes, pitch0, aligned4);
}
void launch_rgb2gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                     uint8_t* pyr, size_t pyrFrameBytes, int pitch0, hipStream_t s)
{
    dim3 grid((w + 2 * kPadX + 255) / 256, (h + 3) / 4, batch);
    hipLaunchKernelGGL(k_level0<3>, grid, dim3(256), 0, s, src, w, h, stride, frameStride, pyr, pyrFrameBytes, pitch0, 0);
}

// ---------------------------------------------------------------------------
// pyramid level l from level l-1: cv::resize INTER_LINEAR 8UC1 (src/Feature.cpp:813, App. A2);
// coefficients come from host-built tables (xs: {sx0, sx1, a0, a1} per dx; ys: {sy0, sy1, b0, b1}
// per dy).  Generic form (any scale; also the exact-2:1 INTER_AREA case): one thread per output quad.
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, size_t pyrFrameBytes,
                                                unsigned long long srcOff, int srcPitch,
                                                unsigned long long dstOff, int dstPitch, int dw, int dh,
                                                const short4* __restrict__ xtab, const short4* __restrict__ ytab, int area2)
{
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4 - kPadX;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= dw + kPadX || y >= dh) return;
    const uint8_t* S = pyr + (size_t)blockIdx.z * pyrFrameBytes + srcOff;
    uint8_t* D = pyr + (size_t)blockIdx.z * pyrFrameBytes + dstOff + (size_t)y * dstPitch + x0;
    uint32_t v = 0;
    if (area2) {       // exact 2:1 shrink: cv::resize uses INTER_AREA, (a+b+c+d+2)>>2
        const uint8_t* A = S + (size_t)(2 * y) * srcPitch; const uint8_t* B = A + srcPitch;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = refl_clamp(x0 + i, dw);
            v |= (uint32_t)((A[2 * x] + A[2 * x + 1] + B[2 * x] + B[2 * x + 1] + 2) >> 2) << (8 * i);
        }
        *(uint32_t*)D = v;
        return;
    }
    const short4 ty = ytab[y];
    const uint8_t* S0 = S + (size_t)ty.x * srcPitch;
    const uint8_t* S1 = S + (size_t)ty.y * srcPitch;
    const int b0 = ty.z, b1 = ty.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const short4 tx = xtab[refl_clamp(x0 + i, dw)];
        const int t0 = S0[tx.x] * tx.z + S0[tx.y] * tx.w;
        const int t1 = S1[tx.x] * tx.z + S1[tx.y] * tx.w;
        const int r = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        v |= (uint32_t)(r & 255) << (8 * i);
    }
    *(uint32_t*)D = v;
}

// Streaming form, used whenever the 8 taps of an output quad fit an 8-byte source window (scale <= ~2).
// A wave owns 64 output quads (256 px) x `rows` output rows (<= 8).  All source rows the strip needs
// (<= kResizeSrcRows) are requested up front: three aligned dwords per lane and row, shifted to the lane's
// window by v_alignbyte, the 4+4 taps gathered by two v_perm_b32 with per-lane constant selectors (host
// quad table).  Each source row's horizontal sums are computed once and folded into every output row
// that uses it (wave-uniform tests on the y table).  3 dword loads per lane and source row instead of
// 20 byte/short loads per output row, and one round of memory latency per strip.
constexpr int kResizeSrcRows = 12;

__global__ __launch_bounds__(256) void k_resize_quad(uint8_t* __restrict__ pyr, size_t pyrFrameBytes,
                                                     unsigned long long srcOff, int srcPitch,
                                                     unsigned long long dstOff, int dstPitch, int nq, int dh, int rows,
                                                     const ResizeQuad* __restrict__ qtab, const short4* __restrict__ ytab)
{
    const int qi = blockIdx.x * 64 + (threadIdx.x & 63);
    const int strip = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const int yBeg = strip * rows;
    if (yBeg >= dh) return;
    const int yEnd = min(yBeg + rows, dh);
    const bool live = qi < nq;
    const ResizeQuad e = qtab[live ? qi : 0];
    const uint32_t b = e.bOff & 0xFFFFFF, off = e.bOff >> 24;
    const uint8_t* S = pyr + (size_t)blockIdx.z * pyrFrameBytes + srcOff;
    short4 ty[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ty[r] = ytab[min(yBeg + r, yEnd - 1)];       // wave-uniform
    const int sBeg = ty[0].x, sEnd = ytab[yEnd - 1].y;
    uint32_t A[kResizeSrcRows], B[kResizeSrcRows], C[kResizeSrcRows];
#pragma unroll
    for (int j = 0; j < kResizeSrcRows; ++j) {
        const uint32_t* r = (const uint32_t*)(S + (size_t)min(sBeg + j, sEnd) * srcPitch) + b;
        A[j] = r[0]; B[j] = r[1]; C[j] = r[2];
    }
    int acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = 2;
#pragma unroll
    for (int j = 0; j < kResizeSrcRows; ++j) {
        const uint32_t w0 = __builtin_amdgcn_alignbyte(B[j], A[j], off), w1 = __builtin_amdgcn_alignbyte(C[j], B[j], off);
        const uint32_t p0 = __builtin_amdgcn_perm(w1, w0, e.sel0), p1 = __builtin_amdgcn_perm(w1, w0, e.sel1);
        int T[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) T[i] = ((int)((p0 >> (8 * i)) & 255u) * e.a0[i] + (int)((p1 >> (8 * i)) & 255u) * e.a1[i]) >> 4;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (yBeg + r < yEnd) {
                if (ty[r].x - sBeg == j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[r][i] += (ty[r].z * T[i]) >> 16;
                }
                if (ty[r].y - sBeg == j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[r][i] += (ty[r].w * T[i]) >> 16;
                }
            }
        }
    }
    uint8_t* D = pyr + (size_t)blockIdx.z * pyrFrameBytes + dstOff + 4 * qi - kPadX;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (live && yBeg + r < yEnd) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) v |= (uint32_t)((acc[r][i] >> 2) & 255) << (8 * i);
            *(uint32_t*)(D + (size_t)(yBeg + r) * dstPitch) = v;
        }
    }
}

void launch_resize(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int level, const int16_t* tabs,
                   size_t xTabOff, size_t yTabOff, size_t quadTabOff, int mode, int rows, int batch, hipStream_t s)
{
    const int dw = kl.w[level], dh = kl.h[level];
    if (mode == 1) {
        const int nq = (dw + 2 * kPadX + 3) / 4;
        dim3 grid((nq + 63) / 64, (dh + 4 * rows - 1) / (4 * rows), batch);
        hipLaunchKernelGGL(k_resize_quad, grid, dim3(256), 0, s, pyr, pyrFrameBytes,
                           kl.offset[level - 1], kl.pitch[level - 1], kl.offset[level], kl.pitch[level], nq, dh, rows,
                           (const ResizeQuad*)(tabs + quadTabOff), (const short4*)(tabs + yTabOff));
        return;
    }
    dim3 grid((dw + 2 * kPadX + 255) / 256, (dh + 3) / 4, batch);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, pyr, pyrFrameBytes,
                       kl.offset[level - 1], kl.pitch[level - 1], kl.offset[level], kl.pitch[level], dw, dh,
                       (const short4*)(tabs + xTabOff), (const short4*)(tabs + yTabOff), mode == 2 ? 1 : 0);
}

// ---------------------------------------------------------------------------
// FAST-9/16 + score + cell-local NMS + ini/min threshold fallback; one workgroup per
// 30-px grid cell (src/Feature.cpp:851-892 with cv::FAST per SURVEY App. A3).
//
// cv::FAST's score of a corner is  s = max(maxmin_dark, maxmin_bright) - 1  where
// maxmin_dark = max over the 16 arcs of 9 ring pixels of min(v - ring), and it does
// not depend on the threshold; pixel p is a corner at threshold T  <=>  s(p) >= T.
// So one score map serves both thresholds: M_T(q) = s(q) >= T ? s(q) : 0, and p is
// kept iff s(p) >= T and s(p) > M_T(q) for its 8 neighbours inside the cell window's
// tested interior (everything else counts as 0).
constexpr int kTileW = 72;   // >= 65 (max window) rounded so rows are dword aligned
constexpr int kTileH = 66;
constexpr int kScW = 68;     // interior (<= 59) + 1-px zero rim each side, padded

__device__ __forceinline__ int arc9_maxmin(const int (&d)[16])
{
    int m2[16], m4[16], m8[16], r = -1024;
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m8[k] = min(m4[k], m4[(k + 4) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) r = max(r, min(m8[k], d[(k + 8) & 15]));
    return r;
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t pyrFrameBytes, KLevels kl,
                                                    int cellsPerFrame, size_t slotsPerFrame, int iniTh, int minTh,
                                                    uint32_t* __restrict__ slots, int* __restrict__ cellCount)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileH * kTileW];
    __shared__ __attribute__((aligned(16))) uint8_t sc[kScW * kScW];
    __shared__ uint16_t queue[kMaxCellDim * kMaxCellDim];
    __shared__ unsigned long long rowmask[kMaxCellDim];
    __shared__ int rowoff[kMaxCellDim + 1];
    __shared__ int qn, nIni;

    const int tid = threadIdx.x;
    const int cell = blockIdx.x, frame = blockIdx.y;
    int level = 0;
    while (level + 1 < kl.nlevels && cell >= kl.cellBase[level + 1]) ++level;
    const int lc = cell - kl.cellBase[level];
    const int ci = lc / kl.nCols[level], cj = lc - ci * kl.nCols[level];
    const int wCell = kl.wCell[level], hCell = kl.hCell[level];
    const int maxBX = kl.w[level] - kMinBorder, maxBY = kl.h[level] - kMinBorder;
    const int iniX = kMinBorder + cj * wCell, iniY = kMinBorder + ci * hCell;
    int* myCount = cellCount + (size_t)frame * cellsPerFrame + cell;
    // the reference skips these cells entirely (src/Feature.cpp:856, :866)
    if (iniY >= maxBY - 3 || iniX >= maxBX - 6) { if (tid == 0) *myCount = 0; return; }
    const int ww = min(iniX + wCell + 6, maxBX) - iniX, wh = min(iniY + hCell + 6, maxBY) - iniY;
    const int iw = ww - 6, ih = wh - 6;                 // tested interior (cv::FAST rows/cols 3..n-4)
    if (iw <= 0 || ih <= 0) { if (tid == 0) *myCount = 0; return; }

    const int pitch = kl.pitch[level];
    const uint8_t* img = pyr + (size_t)frame * pyrFrameBytes + kl.offset[level] + (size_t)iniY * pitch + iniX;
    for (int i = tid; i < wh * ww; i += 256) {
        const int y = i / ww, x = i - y * ww;
        tile[y * kTileW + x] = img[(size_t)y * pitch + x];
    }
    for (int i = tid; i < kScW * kScW / 4; i += 256) ((uint32_t*)sc)[i] = 0;
    if (tid < kMaxCellDim) rowmask[tid] = 0;
    if (tid == 0) { qn = 0; nIni = 0; }
    __syncthreads();

    const int tLow = min(iniTh, minTh);
    // phase 1: high-speed rejection on the 4 compass pixels; survivors are queued
    for (int p0 = 0; p0 < iw * ih; p0 += 256) {
        const int p = p0 + tid;
        bool pass = false;
        if (p < iw * ih) {
            const int iy = p / iw, ix = p - iy * iw;
            const uint8_t* c = tile + (iy + 3) * kTileW + ix + 3;
            const int v = c[0];
            const int lo = v - tLow, hi = v + tLow;
            const int r0 = c[3 * kTileW], r8 = c[-3 * kTileW], r4 = c[3], r12 = c[-3];
            const int c0 = (r0 < lo ? 1 : 0) | (r0 > hi ? 2 : 0), c8 = (r8 < lo ? 1 : 0) | (r8 > hi ? 2 : 0);
            const int c4 = (r4 < lo ? 1 : 0) | (r4 > hi ? 2 : 0), c12 = (r12 < lo ? 1 : 0) | (r12 > hi ? 2 : 0);
            pass = ((c0 | c8) & (c4 | c12)) != 0;
        }
        const unsigned long long m = __ballot(pass);
        if (m) {
            int base = 0;
            const int lane = tid & 63;
            if (lane == 0) base = atomicAdd(&qn, __popcll(m));
            base = __shfl(base, 0);
            if (pass) queue[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)p;
        }
    }
    __syncthreads();
    const int nq = qn;
    // phase 2: exact score for queued pixels
    for (int q = tid; q < nq; q += 256) {
        const int p = queue[q];
        const int iy = p / iw, ix = p - iy * iw;
        const uint8_t* c = tile + (iy + 3) * kTileW + ix + 3;
        const int v = c[0];
        int d[16];
        d[0] = v - c[3 * kTileW];       d[1] = v - c[3 * kTileW + 1];   d[2] = v - c[2 * kTileW + 2];   d[3] = v - c[kTileW + 3];
        d[4] = v - c[3];                d[5] = v - c[-kTileW + 3];      d[6] = v - c[-2 * kTileW + 2];  d[7] = v - c[-3 * kTileW + 1];
        d[8] = v - c[-3 * kTileW];      d[9] = v - c[-3 * kTileW - 1];  d[10] = v - c[-2 * kTileW - 2]; d[11] = v - c[-kTileW - 3];
        d[12] = v - c[-3];              d[13] = v - c[kTileW - 3];      d[14] = v - c[2 * kTileW - 2];  d[15] = v - c[3 * kTileW - 1];
        const int a = arc9_maxmin(d);
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = -d[k];
        const int b = arc9_maxmin(d);
        const int s = max(a, b) - 1;
        if (s >= tLow && s > 0) sc[(iy + 1) * kScW + ix + 1] = (uint8_t)s;
    }
    __syncthreads();
    // phase 3: NMS at both thresholds
    int cntIni = 0;
    for (int q = tid; q < nq; q += 256) {
        const int p = queue[q];
        const int iy = p / iw, ix = p - iy * iw;
        const uint8_t* c = sc + (iy + 1) * kScW + ix + 1;
        const int s = c[0];
        int flags = 0;
        if (s) {
            const int n0 = c[-kScW - 1], n1 = c[-kScW], n2 = c[-kScW + 1], n3 = c[-1], n4 = c[1],
                      n5 = c[kScW - 1], n6 = c[kScW], n7 = c[kScW + 1];
            const int mAll = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            // M_T(q) = n >= T ? n : 0, so  s > max_q M_T(q)  <=>  every neighbour is < T or < s
            auto nmsAt = [&](int T) {
                if (s < T) return false;
                if (s > mAll) return true;
                const int t1 = T - 1;   // neighbours <= t1 do not count
                return (n0 <= t1 || n0 < s) && (n1 <= t1 || n1 < s) && (n2 <= t1 || n2 < s) && (n3 <= t1 || n3 < s) &&
                       (n4 <= t1 || n4 < s) && (n5 <= t1 || n5 < s) && (n6 <= t1 || n6 < s) && (n7 <= t1 || n7 < s);
            };
            if (nmsAt(iniTh)) { flags |= 1; ++cntIni; }
            if (nmsAt(minTh)) flags |= 2;
        }
        queue[q] = (uint16_t)(p | (flags << 13));
    }
    if (cntIni) atomicAdd(&nIni, cntIni);
    __syncthreads();
    const int want = nIni > 0 ? 1 : 2;          // src/Feature.cpp:875: retry at minTh only if the first list is empty
    for (int q = tid; q < nq; q += 256) {
        const int e = queue[q];
        if ((e >> 13) & want) {
            const int p = e & 0x1FFF;
            const int iy = p / iw, ix = p - iy * iw;
            atomicOr(&rowmask[iy], 1ull << ix);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int y = 0; y < ih; ++y) { rowoff[y] = acc; acc += __popcll(rowmask[y]); }
        rowoff[ih] = acc;
        *myCount = acc;
    }
    __syncthreads();
    if (tid < ih) {
        unsigned long long m = rowmask[tid];
        uint32_t* out = slots + (size_t)frame * slotsPerFrame + kl.slotBase[level] + (size_t)lc * kl.slotCap[level] + rowoff[tid];
        while (m) {
            const int ix = __ffsll((long long)m) - 1;
            m &= m - 1;
            // window-relative FAST coordinates (ix+3, iy+3) moved to the level's border frame (src/Feature.cpp:886-887)
            const uint32_t x = ix + 3 + cj * wCell, y = tid + 3 + ci * hCell;
            *out++ = x | (y << 12) | ((uint32_t)sc[(tid + 1) * kScW + ix + 1] << 24);
        }
    }
}

void launch_fast(const uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int cellsPerFrame, size_t slotsPerFrame,
                 int iniTh, int minTh, uint32_t* slots, int* cellCount, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_fast_cells, dim3(cellsPerFrame, batch), dim3(256), 0, s, pyr, pyrFrameBytes, kl,
                       cellsPerFrame, slotsPerFrame, iniTh, minTh, slots, cellCount);
}

// ---------------------------------------------------------------------------
// per frame: concatenate the cells' keypoints level by level, cell-row-major, raster inside a
// cell -- the order in which the reference appends to vToDistributeKeys (src/Feature.cpp:851-891).
__global__ __launch_bounds__(256) void k_compact(KLevels kl, int cellsPerFrame, size_t slotsPerFrame,
                                                 const uint32_t* __restrict__ slots, const int* __restrict__ cellCount,
                                                 uint32_t* __restrict__ cand, int candCap, int* __restrict__ candCount)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int* cc = cellCount + (size_t)frame * cellsPerFrame;
    const uint32_t* sl = slots + (size_t)frame * slotsPerFrame;
    uint32_t* out = cand + (size_t)frame * candCap;
    int* cnt = candCount + frame * (kMaxLevels + 1);
    int total = 0;            // running offset of this frame (uniform)
    for (int l = 0; l < kl.nlevels; ++l) {
        const int ncell = kl.cellBase[l + 1] - kl.cellBase[l];
        int levelCount = 0;
        for (int c0 = 0; c0 < ncell; c0 += 256) {
            const int c = c0 + tid;
            const int n = c < ncell ? cc[kl.cellBase[l] + c] : 0;
            // block exclusive scan of n
            int incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            if (lane == 63) wsum[wid] = incl;
            __syncthreads();
            int wbase = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (i < wid) wbase += wsum[i]; tot += wsum[i]; }
            const int off = total + levelCount + wbase + incl - n;
            const uint32_t* src = sl + kl.slotBase[l] + (size_t)c * kl.slotCap[l];
            for (int i = 0; i < n; ++i) if (off + i < candCap) out[off + i] = src[i];
            levelCount += tot;
            __syncthreads();
        }
        if (tid == 0) cnt[l] = levelCount;
        total += levelCount;
    }
    if (tid == 0) { cnt[kMaxLevels] = total; (void)carry; }
}

void launch_compact(const KLevels& kl, int cellsPerFrame, size_t slotsPerFrame, const uint32_t* slots, const int* cellCount,
                    uint32_t* cand, int candCap, int* candCount, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact, dim3(batch), dim3(256), 0, s, kl, cellsPerFrame, slotsPerFrame, slots, cellCount, cand, candCap, candCount);
}

// ---------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101 on the isolated level (src/Feature.cpp:547-548,
// SURVEY App. A4): separable, 8-bit kernel {18,34,49,55,49,34,18}; row sums <= 65535, column pass
// (sum + 32768) >> 16 saturated.
//
// Streaming form, no LDS: a half-wave (32 lanes) owns a strip of 120 output pixels x R rows and
// walks down the rows.  Each lane loads ONE dword (4 pixels) per row; its left/right neighbours'
// dwords arrive by DPP wave shifts (lanes 0 and 31 of the strip are halo lanes that only feed
// their neighbours).  Row pass: 6 v_alignbyte + 8 v_dot4_u32_u8 for 4 pixels.  The last 7 rows
// of row sums live in registers (static indices through a 7-step unrolled body); column pass in
// 32-bit v_mad_u32_u24; one dword store per lane and row.  HBM traffic = the algorithmic
// 2 bytes per pixel plus a 6-row halo per strip and the two halo lanes.
// rows carry 4 reflect-101 pad columns on both sides, so every lane that feeds an output loads a plain
// dword; lanes further out (x > w+3) are clamped onto the last pad dword and feed nobody
__device__ __forceinline__ uint32_t blur_load4(const uint8_t* __restrict__ row, int x, int w)
{
    return *(const uint32_t*)(row + min(x, (w + 3) & ~3));
}

__device__ __forceinline__ void blur_rowsum(uint32_t d1, uint32_t (&r)[4])
{
    const uint32_t d0 = __builtin_amdgcn_update_dpp(0u, d1, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane-1's dword
    const uint32_t d2 = __builtin_amdgcn_update_dpp(0u, d1, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane+1's dword
    const uint32_t WLO = 18u | (34u << 8) | (49u << 16) | (55u << 24);                // x-3 .. x
    const uint32_t WHI = 49u | (34u << 8) | (18u << 16);                              // x+1 .. x+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = i == 3 ? d1 : __builtin_amdgcn_alignbyte(d1, d0, i + 1);
        const uint32_t hi = i == 3 ? d2 : __builtin_amdgcn_alignbyte(d2, d1, i + 1);
        r[i] = __builtin_amdgcn_udot4(lo, WLO, __builtin_amdgcn_udot4(hi, WHI, 0u, false), false);
    }
}

__device__ __forceinline__ uint32_t blur_col4(const uint32_t (&a)[4], const uint32_t (&b)[4], const uint32_t (&c)[4],
                                              const uint32_t (&d)[4], const uint32_t (&e)[4], const uint32_t (&f)[4],
                                              const uint32_t (&g)[4])
{
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t s = 18u * (a[i] + g[i]) + 32768u;
        s += 34u * (b[i] + f[i]);
        s += 49u * (c[i] + e[i]);
        s += 55u * d[i];
        out |= min(s >> 16, 255u) << (8 * i);
    }
    return out;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, size_t pyrFrameBytes, KLevels kl)
{
    const int lane = threadIdx.x & 31;
    const int strip = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int frame = blockIdx.y;
    const bool live = strip < kl.blurStripBase[kl.nlevels];
    int level = 0;
    while (level + 1 < kl.nlevels && strip >= kl.blurStripBase[level + 1]) ++level;
    const int t = live ? strip - kl.blurStripBase[level] : 0;
    const int sy = t / kl.blurStripsX[level], sx = t - sy * kl.blurStripsX[level];
    const int w = kl.w[level], h = kl.h[level], pitch = kl.pitch[level];
    const int iters = live ? kl.blurIters[level] : 0;
    const int rows = 7 * iters - 6;
    const uint8_t* src = pyr + (size_t)frame * pyrFrameBytes + kl.offset[level];
    uint8_t* dst = blur + (size_t)frame * pyrFrameBytes + kl.offset[level];
    const int x = sx * 120 - 4 + 4 * lane;                 // first of this lane's 4 pixels
    const int y0 = sy * rows;
    const bool writer = lane >= 1 && lane <= 30 && x < w;
    uint32_t r0[4], r1[4], r2[4], r3[4], r4[4], r5[4], r6[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r0[i] = r1[i] = r2[i] = r3[i] = r4[i] = r5[i] = r6[i] = 0;
    // step s consumes input row y0-3+s and (from s >= 6) emits output row y0+s-6.  The 7 loads of
    // the NEXT iteration are issued before the current one is processed (latency hiding).
#define BLUR_LOAD(S0, S) blur_load4(src + (size_t)min(max(refl101(y0 - 3 + (S0) + (S), h), 0), h - 1) * pitch, x, w)
#define BLUR_STEP(S, RN, A, B, C, D, E, F, G)                                                        \
    {                                                                                                \
        blur_rowsum(cur[S], RN);                                                                     \
        const int yout = y0 + s0 + (S) - 6;                                                          \
        if (s0 + (S) >= 6 && writer && yout < h)                                                     \
            *(uint32_t*)(dst + (size_t)yout * pitch + x) = blur_col4(A, B, C, D, E, F, G);           \
    }
    uint32_t cur[7], nxt[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { cur[i] = iters > 0 ? BLUR_LOAD(0, i) : 0u; nxt[i] = 0; }
    for (int it = 0; it < iters; ++it) {
        const int s0 = it * 7;
        if (it + 1 < iters) {
#pragma unroll
            for (int i = 0; i < 7; ++i) nxt[i] = BLUR_LOAD(s0 + 7, i);
        }
        BLUR_STEP(0, r0, r1, r2, r3, r4, r5, r6, r0)
        BLUR_STEP(1, r1, r2, r3, r4, r5, r6, r0, r1)
        BLUR_STEP(2, r2, r3, r4, r5, r6, r0, r1, r2)
        BLUR_STEP(3, r3, r4, r5, r6, r0, r1, r2, r3)
        BLUR_STEP(4, r4, r5, r6, r0, r1, r2, r3, r4)
        BLUR_STEP(5, r5, r6, r0, r1, r2, r3, r4, r5)
        BLUR_STEP(6, r6, r0, r1, r2, r3, r4, r5, r6)
#pragma unroll
        for (int i = 0; i < 7; ++i) cur[i] = nxt[i];
    }
#undef BLUR_LOAD
#undef BLUR_STEP
}

void launch_blur(const uint8_t* pyr, uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl, int stripsPerFrame, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_blur, dim3((stripsPerFrame + 7) / 8, batch), dim3(256), 0, s, pyr, blur, pyrFrameBytes, kl);
}

// ---------------------------------------------------------------------------
// IC_Angle (src/Feature.cpp:84-113) + steered BRIEF (src/Feature.cpp:446-482): one wave per keypoint.
__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};
// umax of src/Feature.cpp:425-443 for HALF_PATCH_SIZE 15
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// cv::fastAtan2: float ops, each rounded once, no contraction.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
                p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    const float eps = (float)2.2204460492503131e-16;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cosf/sinf of src/Feature.cpp:449 evaluated in double with a fixed IEEE sequence
// (Cody-Waite by pi/2, fdlibm kernel polynomials, explicit fma), rounded once to float.
__device__ __forceinline__ void sincos_canonical(float xf, float& s, float& c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double x = (double)xf;
    const double kd = rint(__dmul_rn(x, TWO_OVER_PI));
    const int k = (int)kd;
    double r = __fma_rn(-kd, PIO2_1, x);
    r = __fma_rn(-kd, PIO2_1T, r);
    const double z = __dmul_rn(r, r);
    double ps = __fma_rn(z, S6, S5); ps = __fma_rn(z, ps, S4); ps = __fma_rn(z, ps, S3); ps = __fma_rn(z, ps, S2); ps = __fma_rn(z, ps, S1);
    const double sn = __fma_rn(__dmul_rn(r, z), ps, r);
    double pc = __fma_rn(z, C6, C5); pc = __fma_rn(z, pc, C4); pc = __fma_rn(z, pc, C3); pc = __fma_rn(z, pc, C2); pc = __fma_rn(z, pc, C1);
    const double cs = __fma_rn(__dmul_rn(z, z), pc, __fma_rn(z, -0.5, 1.0));
    double so, co;
    switch (k & 3) {
        case 0: so = sn; co = cs; break;
        case 1: so = cs; co = -sn; break;
        case 2: so = -sn; co = -cs; break;
        default: so = -cs; co = sn; break;
    }
    s = (float)so; c = (float)co;
}

__global__ __launch_bounds__(256) void k_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                  size_t pyrFrameBytes, KLevels kl, const uint2* __restrict__ kpIn,
                                                  const int* __restrict__ kpCount, int cap,
                                                  ssvio_keypoint* __restrict__ kpOut, uint8_t* __restrict__ desc)
{
    const int frame = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= kpCount[frame]) return;              // wave-uniform
    const uint2 in = kpIn[(size_t)frame * cap + k];
    const int x = in.x & 0xFFFF, y = in.x >> 16, level = in.y & 0xFF, resp = (in.y >> 8) & 0xFF;
    const int pitch = kl.pitch[level];
    const uint8_t* center = pyr + (size_t)frame * pyrFrameBytes + kl.offset[level] + (size_t)y * pitch + x;
    // moments: lane -> row v = lane-14 (29 rows -14..14, row 0 spans u=-15..15, others u=-d..d-1); lanes 32..63 take the
    // upper half of each row's columns to balance
    int m01 = 0, m10 = 0;
    {
        const int row = lane & 31, half = lane >> 5;
        if (row < 29) {
            const int v = row - 14, av = v < 0 ? -v : v;
            const int d = c_umax[av];
            const int u0 = -d, u1 = (v == 0) ? d + 1 : d;          // [u0, u1)
            const int mid = (u0 + u1) >> 1;
            const int ub = half ? mid : u0, ue = half ? u1 : mid;
            const uint8_t* rp = center + v * pitch;
            int sv = 0, su = 0;
            for (int u = ub; u < ue; ++u) { const int val = rp[u]; sv += val; su += u * val; }
            m10 = su; m01 = v * sv;
        }
    }
    m01 = wave_sum(m01); m10 = wave_sum(m10);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // descriptor
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    float a, b;
    sincos_canonical(ang, b, a);
    const uint8_t* bc = blur + (size_t)frame * pyrFrameBytes + kl.offset[level] + (size_t)y * pitch + x;
    unsigned long long words[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int8_t* p = c_pattern + (size_t)(j * 64 + lane) * 4;
        const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[r0 * pitch + q0], t1 = bc[r1 * pitch + q1];
        words[j] = __ballot(t0 < t1);
    }
    const size_t o = (size_t)frame * cap + k;
    if (lane < 4) ((unsigned long long*)(desc + o * 32))[lane] = words[lane];
    if (lane == 0) {
        ssvio_keypoint kp;
        const float scale = kl.scale[level];
        kp.x = level ? __fmul_rn((float)x, scale) : (float)x;       // src/Feature.cpp:557-563
        kp.y = level ? __fmul_rn((float)y, scale) : (float)y;
        kp.size = (float)kl.kpSize[level];
        kp.angle = angle; kp.response = (float)resp; kp.octave = level; kp.class_id = -1;
        kpOut[o] = kp;
    }
}

void launch_describe(const uint8_t* pyr, const uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl,
                     const uint2* kpIn, const int* kpCount, int cap, ssvio_keypoint* kpOut, uint8_t* desc, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_describe, dim3((cap + 3) / 4, batch), dim3(256), 0, s, pyr, blur, pyrFrameBytes, kl, kpIn, kpCount, cap, kpOut, desc);
}

}  // namespace ssvio
