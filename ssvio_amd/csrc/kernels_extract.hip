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
                                                uint8_t* __restrict__ pyr, size_t pyrFrameBytes, int pitch0, int aligned4,
                                                int* __restrict__ zeroA, int nA, int* __restrict__ zeroB, int nB,
                                                int blocksPerFrame, int batch)
{
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    // first kernel of the pipeline: it also clears the FAST stage's counters (no separate memset launches)
    {
        const unsigned nthreads = gridDim.x * 256u;
        const unsigned flat = blockIdx.x * 256u + threadIdx.x;
        for (unsigned i = flat; i < (unsigned)nA; i += nthreads) zeroA[i] = 0;
        for (unsigned i = flat; i < (unsigned)nB; i += nthreads) zeroB[i] = 0;
    }
    // a wave takes one row: lane -> quads lane, lane + 64, ...
    const int y = blk * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const uint8_t* s = src + (size_t)frame * frameStride + (size_t)y * stride;
    uint8_t* d = pyr + (size_t)frame * pyrFrameBytes + (size_t)y * pitch0 - kPadX;
    const int nq = (w + 2 * kPadX + 3) >> 2;
    if (CH == 1 && (aligned4 & 2)) {
        // 16-byte rows (source pointer, strides and the level's pixel 0 are 16-byte aligned): the interior goes as dwordx4 per lane,
        // the pad quads and the row's tail as one more pass of single quads
        const int nv = w >> 4;
        const uint4* s4 = (const uint4*)s; uint4* d4 = (uint4*)(d + kPadX);
#pragma unroll 2
        for (int j = threadIdx.x & 63; j < nv; j += 64) d4[j] = s4[j];
        const int l = threadIdx.x & 63, q = l == 0 ? 0 : 4 * nv + l;        // quad 0 (left pad), then the quads behind the last 16-byte chunk
        if (q < nq) {
            const int x0 = 4 * q - kPadX;
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) v |= (uint32_t)s[refl_clamp(x0 + i, w)] << (8 * i);
            *(uint32_t*)(d + 4 * q) = v;
        }
        return;
    }
#pragma unroll 4
    for (int q = threadIdx.x & 63; q < nq; q += 64) {
        const int x0 = 4 * q - kPadX;
        uint32_t v = 0;
        if (CH == 1 && (aligned4 & 1) && x0 >= 0 && x0 + 3 < w) v = *(const uint32_t*)(s + x0);
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
        *(uint32_t*)(d + 4 * q) = v;           // rows are 4-B aligned with >= 64 B of slack on both sides
    }
}

// The common case of level 0 -- 8-bit gray frames whose rows, strides and pyramid slot are all 16-byte aligned -- as a plain block copy:
// a workgroup takes 16 rows and deals their 16-byte chunks over its 256 threads (a VGA row is only 40 chunks: a wave per row, as in
// k_level0, leaves 24 lanes idle and spends its instructions on set-up), every thread with two or three independent loads in flight,
// all addresses 32-bit offsets from two wave-uniform bases.  The pad quads and the tail of a row behind its last whole chunk are
// gathered byte-wise by the first threads, one quad each.
constexpr int kL0Rows = 16;
__global__ __launch_bounds__(256) void k_level0_rows(const uint8_t* __restrict__ src, int w, int h, int stride, size_t frameStride,
                                                     uint8_t* __restrict__ pyr, size_t pyrFrameBytes, int pitch0, unsigned nvMagic,
                                                     int* __restrict__ zeroA, int nA, int* __restrict__ zeroB, int nB,
                                                     int blocksPerFrame, int batch)
{
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    {   // first kernel of the pipeline: it also clears the FAST stage's counters (no separate memset launches)
        const unsigned nthreads = gridDim.x * 256u;
        const unsigned flat = blockIdx.x * 256u + threadIdx.x;
        for (unsigned i = flat; i < (unsigned)nA; i += nthreads) zeroA[i] = 0;
        for (unsigned i = flat; i < (unsigned)nB; i += nthreads) zeroB[i] = 0;
    }
    const int y0 = blk * kL0Rows;
    const int nrows = min(kL0Rows, h - y0);
    const uint8_t* s = src + (size_t)frame * frameStride + (size_t)y0 * stride;          // wave-uniform
    uint8_t* d = pyr + (size_t)frame * pyrFrameBytes + (size_t)y0 * pitch0;              // pixel 0 of row y0
    const unsigned nv = (unsigned)w >> 4, total = (unsigned)nrows * nv;
    for (unsigned c = threadIdx.x; c < total; c += 256u) {
        const unsigned r = (unsigned)(((unsigned long long)c * nvMagic) >> 22), col = c - r * nv;      // c / nv, exact for c < 16 * 4128 / 16 (host-built magic)
        *(uint4*)(d + (r * (unsigned)pitch0 + 16u * col)) = *(const uint4*)(s + (r * (unsigned)stride + 16u * col));
    }
    const int nq = (w + 2 * kPadX + 3) >> 2;                  // quads of a padded row; quad q covers x = 4 q - kPadX ..
    const int extra = nq - 4 * (int)nv;                       // quad 0 (left pad) + the quads behind the last whole chunk (<= 8)
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (r < nrows && j < extra) {
        const int q = j == 0 ? 0 : 4 * (int)nv + j;
        const int x0 = 4 * q - kPadX;
        const uint8_t* sr = s + (unsigned)r * (unsigned)stride;
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v |= (uint32_t)sr[refl_clamp(x0 + i, w)] << (8 * i);
        *(uint32_t*)(d - kPadX + ((unsigned)r * (unsigned)pitch0 + 4u * (unsigned)q)) = v;
    }
}

void launch_copy_gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                      uint8_t* pyr, size_t pyrFrameBytes, int pitch0, int* zeroA, int nA, int* zeroB, int nB, hipStream_t s)
{
    const int bpf = (h + 3) / 4;
    int aligned4 = ((uintptr_t)src % 4 == 0) && (stride % 4 == 0) && (frameStride % 4 == 0);             // bit 0: dword rows, bit 1: 16-byte rows
    if (((uintptr_t)src % 16 == 0) && (stride % 16 == 0) && (frameStride % 16 == 0) && (pitch0 % 16 == 0) && (pyrFrameBytes % 16 == 0) && ((uintptr_t)pyr % 16 == 0) && w >= 16) aligned4 |= 2;
    static const bool noRows = getenv("SSVIO_LEVEL0_WAVE_ROWS") != nullptr;       // the one form of rounds 1-2, kept for A/B runs
    if ((aligned4 & 2) && !noRows) {
        const int nv = w >> 4, bpfR = (h + kL0Rows - 1) / kL0Rows;
        const unsigned magic = (unsigned)(((1ull << 22) + nv - 1) / nv);
        hipLaunchKernelGGL(k_level0_rows, dim3(bpfR * batch), dim3(256), 0, s, src, w, h, stride, frameStride, pyr, pyrFrameBytes, pitch0, magic, zeroA, nA, zeroB, nB,
                           bpfR, batch);
        return;
    }
    hipLaunchKernelGGL(k_level0<1>, dim3(bpf * batch), dim3(256), 0, s, src, w, h, stride, frameStride, pyr, pyrFrameBytes, pitch0, aligned4, zeroA, nA, zeroB, nB,
                       bpf, batch);
}
void launch_rgb2gray(const uint8_t* src, int w, int h, int stride, size_t frameStride, int batch,
                     uint8_t* pyr, size_t pyrFrameBytes, int pitch0, int* zeroA, int nA, int* zeroB, int nB, hipStream_t s)
{
    const int bpf = (h + 3) / 4;
    hipLaunchKernelGGL(k_level0<3>, dim3(bpf * batch), dim3(256), 0, s, src, w, h, stride, frameStride, pyr, pyrFrameBytes, pitch0, 0, zeroA, nA, zeroB, nB,
                       bpf, batch);
}

// ---------------------------------------------------------------------------
// pyramid level l from level l-1: cv::resize INTER_LINEAR 8UC1 (src/Feature.cpp:813, App. A2);
// coefficients come from host-built tables (xs: {sx0, sx1, a0, a1} per dx; ys: {sy0, sy1, b0, b1}
// per dy).  Generic form (any scale; also the exact-2:1 INTER_AREA case): one thread per output quad.
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, size_t pyrFrameBytes,
                                                unsigned long long srcOff, int srcPitch,
                                                unsigned long long dstOff, int dstPitch, int dw, int dh,
                                                const short4* __restrict__ xtab, const short4* __restrict__ ytab, int area2,
                                                int gx, int blocksPerFrame, int batch)
{
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    const int by = blk / gx, bx = blk - by * gx;
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4 - kPadX;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= dw + kPadX || y >= dh) return;
    const uint8_t* S = pyr + (size_t)frame * pyrFrameBytes + srcOff;
    uint8_t* D = pyr + (size_t)frame * pyrFrameBytes + dstOff + (size_t)y * dstPitch + x0;
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
// quad table).  The source rows are then consumed in order; an output row is finished as soon as its
// second source row (always its first row or the one after it) has been reduced, so only the previous
// and the current row's horizontal sums are live and all control flow is wave-uniform and short.
constexpr int kResizeSrcRows = 12;
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

// NF = 2 or 4: the wave's halves (quarters) take the same 32 (16) quads of the same strip in consecutive frames -- control flow and
// tables are the strip's, so nothing diverges; only the lane offsets differ by a frame.  Rows are then dealt in chunks of 32 (16) quads
// instead of 64, which is what the upper pyramid levels need: a 257-px row is 67 quads - 2 x 64 lanes at 52 %, 5 x 16 at 84 %.
template <int NF>
__global__ __launch_bounds__(256) void k_resize_quad(uint8_t* __restrict__ pyr, size_t pyrFrameBytes,
                                                     unsigned long long srcOff, int srcPitch,
                                                     unsigned long long dstOff, int dstPitch, int nq, int dh, int rows,
                                                     const ResizeQuad* __restrict__ qtab, const short4* __restrict__ ytab,
                                                     int gx, int blocksPerFrame, int batch)
{
    constexpr int CH = 64 / NF;                       // quads per frame in a wave
    int frame, blk;                                   // NF > 1: `frame` is the first frame of the wave's group, `batch` the number of frames
    xcd_frame_block(blockIdx.x, blocksPerFrame, (batch + NF - 1) / NF, frame, blk);
    frame *= NF;
    const int by = blk / gx, bx = blk - by * gx;
    const int sub = (threadIdx.x & 63) / CH;          // which frame of the group
    const int qi = bx * CH + (threadIdx.x & (CH - 1));
    const int strip = __builtin_amdgcn_readfirstlane(by * 4 + (threadIdx.x >> 6));
    const int yBeg = strip * rows;
    if (yBeg >= dh) return;
    const int yEnd = min(yBeg + rows, dh);
    const bool frameOk = frame + sub < batch;                                  // the last group of a batch may be short
    const bool live = qi < nq && frameOk;
    const ResizeQuad e = qtab[qi < nq ? qi : 0];
    const uint32_t off = e.bOff >> 24;
    const uint32_t fo = frameOk ? (uint32_t)sub * (uint32_t)pyrFrameBytes : 0u;           // lane offset of the group's frame (host: NF frames < 2 GB)
    const uint32_t b = (e.bOff & 0xFFFFFF) + (fo >> 2);                      // in dwords, as the table's
    const uint8_t* S = pyr + (size_t)frame * pyrFrameBytes + srcOff;
    uint8_t* D = pyr + (size_t)frame * pyrFrameBytes + dstOff;               // wave-uniform; the lane's quad starts 4 qi - kPadX into a row
    const uint32_t dq = 4u * (uint32_t)qi + fo;                              // ... taken from D - kPadX, so that the offsets stay unsigned
    // wave-uniform, and told so: 16-bit loads come back in VGPRs, and with them the row addresses below would be 64-bit VALU arithmetic
    const int sBeg = __builtin_amdgcn_readfirstlane(ytab[yBeg].x), sEnd = __builtin_amdgcn_readfirstlane(ytab[yEnd - 1].y);
    // per pixel of the quad: the selector that lays its taps out as (tap0, 0, tap1, 0) and its coefficients (0 .. 2048) as a u16 pair
    uint32_t selp[4]; ushort2v wa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        selp[i] = ((e.sel0 >> (8 * i)) & 255u) | (((e.sel1 >> (8 * i)) & 255u) << 16) | 0x0c000c00u;
        wa[i] = ushort2v{(unsigned short)e.a0[i], (unsigned short)e.a1[i]};
    }
    uint32_t A[kResizeSrcRows], B[kResizeSrcRows], C[kResizeSrcRows];
#pragma unroll
    for (int j = 0; j < kResizeSrcRows; ++j) {
        // scalar base, 32-bit lane offset (a level is far below 4 GB): one add per row instead of a 64-bit multiply-add
        const uint32_t* r = (const uint32_t*)(S + ((uint32_t)(min(sBeg + j, sEnd) * srcPitch) + 4u * b));
        A[j] = r[0]; B[j] = r[1]; C[j] = r[2];
    }
    int y = yBeg;                                        // next output row to finish (uniform)
    short4 ty = ytab[y];
    int Tp[4] = {0, 0, 0, 0}, Tc[4];
#pragma unroll
    for (int j = 0; j < kResizeSrcRows; ++j) {
        const uint32_t w0 = __builtin_amdgcn_alignbyte(B[j], A[j], off), w1 = __builtin_amdgcn_alignbyte(C[j], B[j], off);
        // pixel i: its two taps as a u16 pair (one v_perm), times its two coefficients (one v_dot2_u32_u16)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            Tc[i] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(w1, w0, selp[i])), wa[i], 0u, false) >> 4);
        while (y < yEnd && ty.y - sBeg == j) {           // rows whose second source row is row j
            const bool same = ty.x == ty.y;              // clamped at the bottom edge: both taps on one row
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t0 = same ? Tc[i] : Tp[i];
                const int r = (((ty.z * t0) >> 16) + ((ty.w * Tc[i]) >> 16) + 2) >> 2;
                v |= (uint32_t)(r & 255) << (8 * i);
            }
            if (live) *(uint32_t*)((D - kPadX) + ((uint32_t)(y * dstPitch) + dq)) = v;
            ++y;
            ty = ytab[min(y, dh - 1)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) Tp[i] = Tc[i];
    }
}

// Two levels per launch (small batches, where the chain of dependent launches is the cost): a workgroup owns a 64 x 16 tile of
// level B = l + 1.  It first computes the slice of level A = l that the tile's taps touch (from level l - 1, same arithmetic as
// k_resize) into LDS, stores the part of that slice it owns to HBM (ownership cuts level A along the first taps of the tile
// borders, so every pixel of A is stored exactly once; the edge tiles add A's pad columns), then resizes its tile of B out of LDS.
// Valid while the taps of 64 (16) consecutive outputs span at most kR2W (kR2H) source pixels, i.e. scale factors up to ~1.9; the
// host checks that from the tables and falls back to one launch per level otherwise.
constexpr int kR2TW = 64, kR2TH = 16, kR2W = 144, kR2H = 40;

__device__ __forceinline__ int resize_px(const uint8_t* S0, const uint8_t* S1, const short4 tx, int b0, int b1)
{
    const int t0 = S0[tx.x] * tx.z + S0[tx.y] * tx.w;
    const int t1 = S1[tx.x] * tx.z + S1[tx.y] * tx.w;
    return ((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2) & 255;
}

__global__ __launch_bounds__(256) void k_resize2(uint8_t* __restrict__ pyr, size_t pyrFrameBytes,
                                                 unsigned long long offS, int pitchS,
                                                 unsigned long long offA, int pitchA, int wA, int hA,
                                                 unsigned long long offB, int pitchB, int wB, int hB,
                                                 const short4* __restrict__ xtA, const short4* __restrict__ ytA,
                                                 const short4* __restrict__ xtB, const short4* __restrict__ ytB,
                                                 int tilesX, int tilesY, int blocksPerFrame, int batch)
{
    __shared__ uint8_t lds[kR2H][kR2W];
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    const int ty = blk / tilesX, tx = blk - ty * tilesX;
    const bool firstX = tx == 0, lastX = tx == tilesX - 1, firstY = ty == 0, lastY = ty == tilesY - 1;
    const int bx0 = tx * kR2TW, bx1 = min(bx0 + kR2TW, wB), by0 = ty * kR2TH, by1 = min(by0 + kR2TH, hB);
    // slice of A: from the first tap of the tile's first output (pulled in for the pads' reflections at the right edge) to the
    // last tap of its last output; the edge tiles reach A's own edges so that they can own (and pad) them
    const int bxr = lastX ? max(0, min(bx0, wB - 1 - kPadX)) : bx0;
    int ax0 = firstX ? 0 : xtB[bxr].x, ax1 = lastX ? wA - 1 : xtB[bx1 - 1].y;
    if (lastX) ax0 = max(0, min(ax0, wA - 1 - kPadX));
    const int ay0 = firstY ? 0 : ytB[by0].x, ay1 = lastY ? hA - 1 : ytB[by1 - 1].y;
    const int RW = ax1 - ax0 + 1, RH = ay1 - ay0 + 1;
    const uint8_t* S = pyr + (size_t)frame * pyrFrameBytes + offS;
    uint8_t* A = pyr + (size_t)frame * pyrFrameBytes + offA;
    uint8_t* B = pyr + (size_t)frame * pyrFrameBytes + offB;
#pragma unroll 4
    for (int i = threadIdx.x; i < RW * RH; i += 256) {
        const int ry = i / RW, rx = i - ry * RW;
        const short4 tyA = ytA[ay0 + ry];
        lds[ry][rx] = (uint8_t)resize_px(S + (size_t)tyA.x * pitchS, S + (size_t)tyA.y * pitchS, xtA[ax0 + rx], tyA.z, tyA.w);
    }
    __syncthreads();
    {   // owned part of A, pad columns included on the edge tiles
        const int ox0 = firstX ? -kPadX : xtB[bx0].x, ox1 = lastX ? wA + kPadX : xtB[bx1].x;      // [ox0, ox1)
        const int oy0 = firstY ? 0 : ytB[by0].x, oy1 = lastY ? hA : ytB[by1].x;
        const int OW = ox1 - ox0, OH = oy1 - oy0;
#pragma unroll 4
        for (int i = threadIdx.x; i < OW * OH; i += 256) {
            const int ry = i / OW, rx = i - ry * OW;
            const int x = ox0 + rx, y = oy0 + ry;
            A[(size_t)y * pitchA + x] = lds[y - ay0][refl_clamp(x, wA) - ax0];
        }
    }
    {   // the tile of B out of LDS
        const int px0 = firstX ? -kPadX : bx0, px1 = lastX ? wB + kPadX : bx1;
        const int PW = px1 - px0, PH = by1 - by0;
#pragma unroll 4
        for (int i = threadIdx.x; i < PW * PH; i += 256) {
            const int ry = i / PW, rx = i - ry * PW;
            const int x = px0 + rx, y = by0 + ry;
            const short4 tyB = ytB[y];
            short4 txB = xtB[refl_clamp(x, wB)];
            txB.x -= (short)ax0; txB.y -= (short)ax0;
            B[(size_t)y * pitchB + x] = (uint8_t)resize_px(lds[tyB.x - ay0], lds[tyB.y - ay0], txB, tyB.z, tyB.w);
        }
    }
}

// host check of k_resize2's slice bounds for the level pair (A from S, B from A); tables as built by build_resize_tabs
bool resize2_fits(const int16_t* xsB, const int16_t* ysB, int wA, int hA, int wB, int hB)
{
    if (wA < 2 * kPadX + 2 || wB < 2 * kPadX + 2) return false;
    const int tilesX = (wB + kR2TW - 1) / kR2TW, tilesY = (hB + kR2TH - 1) / kR2TH;
    for (int tx = 0; tx < tilesX; ++tx) {
        const int bx0 = tx * kR2TW, bx1 = std::min(bx0 + kR2TW, wB);
        const bool lastX = tx == tilesX - 1;
        const int bxr = lastX ? std::max(0, std::min(bx0, wB - 1 - kPadX)) : bx0;
        int ax0 = tx == 0 ? 0 : xsB[4 * bxr], ax1 = lastX ? wA - 1 : xsB[4 * (bx1 - 1) + 1];
        if (lastX) ax0 = std::max(0, std::min(ax0, wA - 1 - kPadX));
        if (ax1 - ax0 + 1 > kR2W || ax1 < ax0) return false;
        if (!lastX && (xsB[4 * bx1] > ax1 + 1 || xsB[4 * bx1] < ax0)) return false;     // the owned range lies inside the slice
        for (int x = bxr; x < bx1; ++x) if (xsB[4 * x] < ax0 || xsB[4 * x + 1] > ax1) return false;
    }
    for (int ty = 0; ty < tilesY; ++ty) {
        const int by0 = ty * kR2TH, by1 = std::min(by0 + kR2TH, hB);
        const bool lastY = ty == tilesY - 1;
        const int ay0 = ty == 0 ? 0 : ysB[4 * by0], ay1 = lastY ? hA - 1 : ysB[4 * (by1 - 1) + 1];
        if (ay1 - ay0 + 1 > kR2H || ay1 < ay0) return false;
        if (!lastY && (ysB[4 * by1] > ay1 + 1 || ysB[4 * by1] < ay0)) return false;
        for (int y = by0; y < by1; ++y) if (ysB[4 * y] < ay0 || ysB[4 * y + 1] > ay1) return false;
    }
    return true;
}

void launch_resize2(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int levelA, const int16_t* tabs,
                    size_t xTabOffA, size_t yTabOffA, size_t xTabOffB, size_t yTabOffB, int batch, hipStream_t s)
{
    const KLevel& S = kl.lv[levelA - 1]; const KLevel& A = kl.lv[levelA]; const KLevel& B = kl.lv[levelA + 1];
    const int tilesX = (B.w + kR2TW - 1) / kR2TW, tilesY = (B.h + kR2TH - 1) / kR2TH;
    hipLaunchKernelGGL(k_resize2, dim3(tilesX * tilesY * batch), dim3(256), 0, s, pyr, pyrFrameBytes, S.offset, S.pitch,
                       A.offset, A.pitch, A.w, A.h, B.offset, B.pitch, B.w, B.h,
                       (const short4*)(tabs + xTabOffA), (const short4*)(tabs + yTabOffA), (const short4*)(tabs + xTabOffB), (const short4*)(tabs + yTabOffB),
                       tilesX, tilesY, tilesX * tilesY, batch);
}

void launch_resize(uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int level, const int16_t* tabs,
                   size_t xTabOff, size_t yTabOff, size_t quadTabOff, int mode, int rows, int batch, hipStream_t s)
{
    const int dw = kl.lv[level].w, dh = kl.lv[level].h;
    if (mode == 1) {
        const int nq = (dw + 2 * kPadX + 3) / 4;
        // two or four frames per wave where chunks of 32 / 16 quads waste fewer lanes than chunks of 64 (and the group stays inside 32-bit offsets)
        static const bool noGroup = getenv("SSVIO_RESIZE_NO_PAIR") != nullptr;
        auto lanes = [nq](int ch) { return (nq + ch - 1) / ch * ch; };
        const bool fits = !noGroup && pyrFrameBytes % 4 == 0 && 4 * pyrFrameBytes < (1ull << 31);
        // (four frames per wave, chunks of 16 quads, was measured too: fewer idle lanes still, but 64-byte row segments per frame --
        // the seven launches went from 66.0 back to 68.2 us on average per 768 frames; SSVIO_RESIZE_QUARTERS=1 selects it)
        static const bool quarters = getenv("SSVIO_RESIZE_QUARTERS") != nullptr;
        const int nf = (quarters && fits && batch >= 4 && lanes(16) < lanes(32)) ? 4 : (fits && batch >= 2 && lanes(32) < lanes(64)) ? 2 : 1;
        if (nf > 1) {
            const int ch = 64 / nf, gx = (nq + ch - 1) / ch, bpf = gx * ((dh + 4 * rows - 1) / (4 * rows));
            const dim3 grid(bpf * ((batch + nf - 1) / nf));
            if (nf == 4) hipLaunchKernelGGL(k_resize_quad<4>, grid, dim3(256), 0, s, pyr, pyrFrameBytes,
                               kl.lv[level - 1].offset, kl.lv[level - 1].pitch, kl.lv[level].offset, kl.lv[level].pitch, nq, dh, rows,
                               (const ResizeQuad*)(tabs + quadTabOff), (const short4*)(tabs + yTabOff), gx, bpf, batch);
            else hipLaunchKernelGGL(k_resize_quad<2>, grid, dim3(256), 0, s, pyr, pyrFrameBytes,
                               kl.lv[level - 1].offset, kl.lv[level - 1].pitch, kl.lv[level].offset, kl.lv[level].pitch, nq, dh, rows,
                               (const ResizeQuad*)(tabs + quadTabOff), (const short4*)(tabs + yTabOff), gx, bpf, batch);
            return;
        }
        const int gx = (nq + 63) / 64, bpf = gx * ((dh + 4 * rows - 1) / (4 * rows));
        hipLaunchKernelGGL(k_resize_quad<1>, dim3(bpf * batch), dim3(256), 0, s, pyr, pyrFrameBytes,
                           kl.lv[level - 1].offset, kl.lv[level - 1].pitch, kl.lv[level].offset, kl.lv[level].pitch, nq, dh, rows,
                           (const ResizeQuad*)(tabs + quadTabOff), (const short4*)(tabs + yTabOff), gx, bpf, batch);
        return;
    }
    const int gx = (dw + 2 * kPadX + 255) / 256, bpf = gx * ((dh + 3) / 4);
    hipLaunchKernelGGL(k_resize, dim3(bpf * batch), dim3(256), 0, s, pyr, pyrFrameBytes,
                       kl.lv[level - 1].offset, kl.lv[level - 1].pitch, kl.lv[level].offset, kl.lv[level].pitch, dw, dh,
                       (const short4*)(tabs + xTabOff), (const short4*)(tabs + yTabOff), mode == 2 ? 1 : 0, gx, bpf, batch);
}

// ---------------------------------------------------------------------------
// FAST-9/16 + score + cell-local NMS + ini/min threshold flags (src/Feature.cpp:851-892 with
// cv::FAST per SURVEY App. A3), one kernel, 64x32-pixel tiles.
//
// cv::FAST's score of a corner is  s = max(maxmin_dark, maxmin_bright) - 1  where maxmin_dark =
// max over the 16 arcs of 9 ring pixels of min(v - ring); it does not depend on the threshold, and
// pixel p is a corner at threshold T  <=>  s(p) >= T.  The score only needs the pixel's own radius-3
// ring, which always lies inside the window of the 30-px cell that tests the pixel, so scores are
// computed densely (tile + 1-px halo) and only the NMS looks at the cell grid: with
// M_T(q) = s(q) >= T ? s(q) : 0 for q in the SAME cell's tested interior (0 elsewhere, as cv::FAST's
// untested rim), p is kept at T iff s(p) >= T and s(p) > max_q M_T(q) over its 8 neighbours; since
// thresholding is monotone, max_q M_T(q) = M_T(max_q s(q)).
// Survivors are appended (unordered) to the level's raw list with both flags (iniTh / minTh) and their
// position in the reference's append order as key; cellIni[cell] is raised when a cell has an iniTh
// survivor, which is what decides the minTh retry of src/Feature.cpp:875 (resolved by the consumer).
constexpr int kFtW = 64;                     // output tile width; the height is the level's th <= FH (FH = 32 or 64, template parameter)
constexpr int kFtPitch = 72;                 // LDS row bytes: x0-4 .. x0+67
// The tile grid of a level starts at (kFastX0, kEdge) = (16, 19): FAST is only evaluated in [19, w-20] x [19, h-20] (src/Feature.cpp:835-873:
// windows inside [16, w-16) minus cv::FAST's 3-px rim), so no tile is spent on the untested border; x starts at 16 rather than 19 to
// keep the row loads dword-aligned.  The rows of the tested region are split evenly over the level's tile rows (th <= FH each).
// input rows y0-4 .. y0+th+3 (th + 8), score rows y0-1 .. y0+th (th + 2, same column layout as the input tile)

typedef short short2v __attribute__((ext_vector_type(2)));

// cv::FAST's corner score is max(dark, bright) - 1 with dark = max over the 16 circular arcs of 9 of min(v - ring) and bright the
// same on ring - v; it does not depend on the threshold.  Windows of 9 = three windows of 3 (v_min3).
// One polarity only: max over the 16 arcs of min(d) - 1 for d[k] = v - ring[k] (dark) or ring[k] - v (bright).  A pixel has a
// 9-arc in at most one polarity (two would need 18 ring pixels), and where it has one, the other polarity's value is negative
// (every other 9-arc contains at least two pixels of that arc), so for a corner this IS cv::FAST's score.  48 instructions.
__device__ __forceinline__ int arc9_score_one(const int (&d)[16])
{
    int mn3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) mn3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    int a = -512;
#pragma unroll
    for (int k = 0; k < 16; k += 2)
        a = max(max(a, min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15])), min(min(mn3[k + 1], mn3[(k + 4) & 15]), mn3[(k + 7) & 15]));
    return a - 1;
}

template <bool TALL>
__device__ __forceinline__ int find_fast_level(const KLevels& kl, int id)
{
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) level += (l < kl.nlevels && id >= (TALL ? kl.lv[l].fastTileBaseT : kl.lv[l].fastTileBase)) ? 1 : 0;
    return level;
}

// LDS budget: the two work queues are sized for what frames deliver (~5 % of a tile's pixels pass the compass test, ~1.5 % are
// corners), not for the worst case, so that 8 workgroups fit a CU (the kernel is bound by latency between its barriers, not by
// issue: more resident workgroups is what makes it faster).  A tile that overflows a queue takes a dense loop over the score domain
// for that phase instead -- same results, slower, rare (noise images).
constexpr int kFtQueueCap = 2048, kFtCornerCap = 1024, kFtOutCap = 256;

template <int FH>
__global__ __launch_bounds__(256) void k_fast(const uint8_t* __restrict__ pyr, size_t pyrFrameBytes, KLevels kl, int cellsPerFrame,
                                              size_t rawPerFrame, int iniTh, int minTh, uint2* __restrict__ raw,
                                              int* __restrict__ rawCount, int* __restrict__ cellIni, int tilesPerFrame, int batch)
{
    constexpr int kFtRows = FH + 8, kScRows = FH + 2;
    constexpr int kLoadPasses = (kFtRows + 13) / 14, kScanPasses = (kScRows + 13) / 14;      // rows r0, r0+14, ... per thread
    // rows of the half-value tile that phase 1 may address: score row r0 + 14 k (r0 <= 13) reads input rows r .. r + 6; never fewer than
    // the two phase-2/3 queues need, which live in the same memory once phase 1 is over
    constexpr int kHalfRows = 14 * (kScanPasses - 1) + 20 > 57 ? 14 * (kScanPasses - 1) + 20 : 57;
    __shared__ __attribute__((aligned(16))) uint32_t tile[kFtRows * kFtPitch / 4];
    __shared__ __attribute__((aligned(16))) uint32_t sc[kScRows * kFtPitch / 4];
    __shared__ __attribute__((aligned(16))) uint32_t half[kHalfRows * kFtPitch / 4];     // (pixel >> 1) & 0x7f, phase 1's operands
    __shared__ uint16_t queue[kFtQueueCap + 2];          // pixels that pass the compass test (+ a spare slot that takes what overflows)
    static_assert(sizeof(half) >= kFtCornerCap * sizeof(uint16_t) + kFtOutCap * sizeof(uint2), "the queues of phases 2 and 3 reuse the half-value tile");
    uint16_t* const cqueue = (uint16_t*)half;            // ... of those, the corners (score >= the lower threshold): the NMS walks only these
    uint2* const outbuf = (uint2*)(half + kFtCornerCap / 2);      // NMS survivors of one tile (beyond the capacity they are appended one by one)
    __shared__ int qn, cn, on, obase;
    const int tid = threadIdx.x;
    int frame, tileId;
    xcd_frame_block(blockIdx.x, tilesPerFrame, batch, frame, tileId);
    const int level = find_fast_level<FH == 64>(kl, tileId);
    const KLevel& L = kl.lv[level];
    const int t = tileId - (FH == 64 ? L.fastTileBaseT : L.fastTileBase);
    const int th = FH == 64 ? L.fastTileHT : L.fastTileH;              // rows of this level's tiles, <= FH
    const int tyi = t / L.fastTilesX, txi = t - tyi * L.fastTilesX;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int x0 = kFastX0 + txi * kFtW, y0 = kEdge + tyi * th;
    const int ftRows = th + 8, scRows = th + 2;
    const int tLow = min(iniTh, minTh);
    const uint8_t* img = pyr + (size_t)frame * pyrFrameBytes + L.offset;
    const int xMaxLoad = (w + 56) & ~3;                    // last dword that is still inside the row's slack
    // thread -> (row r0 = tid / 18, dword column c = tid % 18) once; rows r0, r0+14, r0+28, ... per pass (252 threads busy)
    const int r0 = (int)(__umul24(tid, 3641u) >> 16), c0 = tid - r0 * (kFtPitch / 4);
    const bool lane252 = tid < 252;
    constexpr uint32_t H = 0x80808080u, M7 = 0x7f7f7f7fu;
    {
        // rows r0, r0 + 14, ...: one base per array and fixed offsets (14 rows = 252 dwords) instead of an index product per pass; the source
        // offset is a 24-bit multiply-add (rows < 4128, pitch < 8192: a level is far below 4 GB) from the wave-uniform base
        uint32_t ld[kLoadPasses];
        const uint32_t xoff = (uint32_t)min(x0 - 4 + 4 * c0, xMaxLoad);
        constexpr int kPass = 14 * (kFtPitch / 4);
        uint32_t* const tp = tile + r0 * (kFtPitch / 4) + c0;
        uint32_t* const hp = half + r0 * (kFtPitch / 4) + c0;
        uint32_t* const sp = sc + r0 * (kFtPitch / 4) + c0;
        const int rLoad = lane252 ? ftRows - r0 : 0, rZero = lane252 ? scRows - r0 : 0;      // pass k loads / clears iff 14 k < rLoad / rZero
#pragma unroll
        for (int k = 0; k < kLoadPasses; ++k) {
            int yy;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(yy) : "v"(y0 - 4 + r0 + 14 * k), "v"(h - 1));
            ld[k] = 14 * k < rLoad ? *(const uint32_t*)(img + (__umul24((uint32_t)yy, (uint32_t)pitch) + xoff)) : 0u;
        }
#pragma unroll
        for (int k = 0; k < kLoadPasses; ++k) {
            if (14 * k < rLoad) { tp[kPass * k] = ld[k]; hp[kPass * k] = (ld[k] >> 1) & M7; }
            if (14 * k < rZero) sp[kPass * k] = 0u;
        }
    }
    if (tid == 0) { qn = 0; cn = 0; on = 0; }
    __syncthreads();
    const uint8_t* tb = (const uint8_t*)tile;
    // tested columns of this tile's score domain x0-1 .. x0+64, as positions in a score row (x = x0 - 4 + position)
    const int cLo = max(x0 - 1, kEdge) - (x0 - 4), cHi = min(x0 + kFtW, w - kEdge - 1) - (x0 - 4);
    // phase 1, bit-parallel: a 9-arc always contains two ADJACENT compass pixels (ring 0,4,8,12); "some adjacent compass pair in
    // one class" == (down|up) & (left|right) per class.  It is only a NECESSARY test (phase 2 scores exactly), so it runs on the
    // pixels' upper 7 bits, four pixels per instruction, with add / sub / and / or only (the fast issue class):
    //   v - X > T  =>  (v >> 1) - (X >> 1) >= ceil(T / 2) =: t,   X - v > T  =>  (X >> 1) - (v >> 1) >= t
    // and for 7-bit fields a, b in the bytes of a dword, bit 7 of ((a | 0x80) - b) says a >= b with no borrow between the bytes.
    // 3 % more pixels pass than with the full 8 bits (measured on the bench's frames); none fewer.
    // Domain: score rows y0-1 .. y0+th (r = 0..th+1), quad columns x0-4 .. x0+67 (pixels outside cLo .. cHi dropped).
    unsigned flagsD = 0, flagsB = 0;                        // bit (8i+k): pixel i of this thread's quad of pass k survives the darker / brighter test (kScanPasses <= 5)
    const uint32_t tq = (uint32_t)min((tLow + 1) >> 1, 128) * 0x01010101u;
    const int qc = c0;
    // One flag byte (0x80) per pixel.  A thread's four columns are the same in every pass: their mask is built once.
    uint32_t colMask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (lane252 && 4 * qc + i >= cLo && 4 * qc + i <= cHi) colMask |= 0x80u << (8 * i);
    // ... and so are its rows r0 + 14 k: score row r is tested iff r < scRows and kEdge <= y0 - 1 + r <= h - kEdge - 1, i.e. the passes
    // k < ceil((lim - r0) / 14), less pass 0 for the first score row of the level's first tile row.  Untested rows are scanned like the
    // others (whatever the half-value tile holds there) and dropped from the flags afterwards.
    const int lim = min(scRows, h - kEdge - y0 + 1) - r0;
    uint32_t rowBits = (1u << (lim > 0 ? min((int)(__umul24((uint32_t)(lim + 13), 4682u) >> 16), kScanPasses) : 0)) - 1u;     // x / 14 for x < 4096
    if (y0 - 1 + r0 < kEdge) rowBits &= ~1u;
    rowBits *= 0x01010101u;
    // the operands are the pixels' upper 7 bits (written beside the tile at load time); every pass is a fixed offset from these three
    const uint32_t* hC = half + ((lane252 ? r0 : 0) + 3) * (kFtPitch / 4) + qc;
    const uint32_t* hL = hC - (qc > 0 ? 1 : 0);
    const uint32_t* hR = hC + (qc < 17 ? 1 : 0);
#pragma unroll
    for (int k = 0; k < kScanPasses; ++k) {
        if (14 * k >= scRows) break;                        // block-uniform: short tiles need fewer passes
        constexpr int kRow = kFtPitch / 4;
        const int o = 14 * k * kRow;                        // score row r0 + 14 k <-> input row r + 3
        const uint32_t c = hC[o], Lf = hL[o], Rt = hR[o], u = hC[o - 3 * kRow], d = hC[o + 3 * kRow];
        const uint32_t l = __builtin_amdgcn_alignbyte(c, Lf, 1);      // bytes x-3 .. x
        const uint32_t rr = __builtin_amdgcn_alignbyte(Rt, c, 3);     // bytes x+3 .. x+6
        // darker: c - t >= X on one vertical AND one horizontal compass pixel (bit 7 of P: c >= t at all)
        const uint32_t P = (c | H) - tq, Ad = P | H;
        const uint32_t dk = (((Ad - u) | (Ad - d)) & ((Ad - l) | (Ad - rr))) & P;
        // brighter: X >= c + t (bit 7 of Q: c + t is beyond any 7-bit value)
        const uint32_t Q = c + tq, Bq = Q & M7;
        const uint32_t br = ((((u | H) - Bq) | ((d | H) - Bq)) & (((l | H) - Bq) | ((rr | H) - Bq))) & (Q ^ H);
        flagsD |= (dk & colMask) >> (7 - k);
        flagsB |= (br & colMask) >> (7 - k);
    }
    flagsD &= rowBits;
    flagsB &= rowBits;
    {   // one queue reservation per wave
        const int lane = tid & 63;
        const int cnt = __popc(flagsD) + __popc(flagsB);
        // inclusive scan over the wave in six DPP adds: shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 / lane 31 carried on
        int incl = cnt;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot) {
            int base = 0;
            if (lane == 63) base = atomicAdd(&qn, tot);
            base = __builtin_amdgcn_readlane(base, 63) + incl - cnt;
            // an entry = thread << 5 | flag bit (8 i + k), bit 13 = the brighter test (a pixel that passes both tests is queued twice; at
            // most one of the two is a corner); phase 2 turns it into the pixel's position, once per entry instead of inside this
            // divergent loop.  One darker and one brighter entry per round; entries beyond the capacity land on the spare slot.
            unsigned f = flagsD, g = flagsB;
            const unsigned eD = (unsigned)tid << 5, eB = eD | 0x2000u;
            int iD = base, iB = base + __popc(flagsD);
            while (f | g) {
                if (f) { queue[min(iD, kFtQueueCap)] = (uint16_t)(eD + (unsigned)(__ffs(f) - 1)); f &= f - 1; ++iD; }
                if (g) { queue[min(iB, kFtQueueCap)] = (uint16_t)(eB + (unsigned)(__ffs(g) - 1)); g &= g - 1; ++iB; }
            }
        }
    }
    __syncthreads();
    const int nq = qn;
    // phase 2: exact score of the queued pixels; the corners among them go on to a second, dense queue
    auto score = [&](int p, int bright) -> bool {
        const uint8_t* c = tb + p + 3 * kFtPitch;               // score row r <-> input row r + 3, same columns
        const int nsg = bright ? 1 : -1, sv = bright ? -(int)c[0] : (int)c[0];
        int d[16];
#define RING(K, OFF) d[K] = __mul24((int)c[OFF], nsg) + sv;
        RING(0, 3 * kFtPitch)      RING(1, 3 * kFtPitch + 1)   RING(2, 2 * kFtPitch + 2)   RING(3, kFtPitch + 3)
        RING(4, 3)                 RING(5, -kFtPitch + 3)      RING(6, -2 * kFtPitch + 2)  RING(7, -3 * kFtPitch + 1)
        RING(8, -3 * kFtPitch)     RING(9, -3 * kFtPitch - 1)  RING(10, -2 * kFtPitch - 2) RING(11, -kFtPitch - 3)
        RING(12, -3)               RING(13, kFtPitch - 3)      RING(14, 2 * kFtPitch - 2)  RING(15, 3 * kFtPitch - 1)
#undef RING
        const int s = arc9_score_one(d);
        const bool corner = s >= tLow && s > 0;
        if (corner) ((uint8_t*)sc)[p] = (uint8_t)s;
        return corner;
    };
    // wave-aggregated append to the corner queue (entries beyond its capacity are only counted: phase 3 then goes dense)
    auto push_corner = [&](bool corner, int p) {
        const unsigned long long vote = __ballot(corner);
        if (vote) {
            const int lane = tid & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&cn, __popcll(vote));
            base = __builtin_amdgcn_readfirstlane(base) + __popcll(vote & ((1ull << lane) - 1ull));
            if (corner && base < kFtCornerCap) cqueue[base] = (uint16_t)p;
        }
    };
    if (nq <= kFtQueueCap) {
        for (int q0 = 0; q0 < nq; q0 += 256) {                  // waves whose 64 entries lie beyond nq skip the body
            const int q = q0 + tid;
            int p = 0; bool corner = false;
            if (q < nq) {
                const int e = queue[q], t = (e >> 5) & 255, bit = e & 31;
                const int tr = (int)(__umul24((uint32_t)t, 3641u) >> 16), tc = t - tr * (kFtPitch / 4);
                p = tr * kFtPitch + 4 * tc + (bit & 7) * (14 * kFtPitch) + (bit >> 3);      // position in the score tile
                corner = score(p, e >> 13);
            }
            push_corner(corner, p);
        }
    } else {
        // the queue overflowed: every tested pixel of the score domain is scored (a superset of the queued ones; a pixel that fails
        // the compass test has no 9-arc, so its score stays below the threshold exactly as if it had not been looked at)
        for (int e0 = 0; e0 < scRows * kFtPitch; e0 += 256) {
            const int e = e0 + tid;
            const int r = (int)(__umul24(e, 14564u) >> 20), cidx = e - r * kFtPitch;
            const int y = y0 - 1 + r;
            bool corner = false;
            if (e < scRows * kFtPitch && cidx >= cLo && cidx <= cHi && y >= kEdge && y <= h - kEdge - 1) { corner = score(e, 0); if (!corner) corner = score(e, 1); }
            push_corner(corner, e);
        }
    }
    __syncthreads();
    // phase 3: NMS inside the pixel's own cell, for the tile's own 64 x th pixels
    const int nc = cn;
    const uint8_t* sb = (const uint8_t*)sc;
    auto nms = [&](int p) {
        const int s = sb[p];
        const int r = (int)(__umul24(p, 14564u) >> 20), cidx = p - r * kFtPitch;       // p / 72 for p < 33000
        const int lx = cidx - 4, ly = r - 1;
        if (lx < 0 || lx >= kFtW || ly < 0 || ly >= th) return;            // halo pixels belong to the neighbouring tile
        const int x = x0 + lx, y = y0 + ly;
        const int gx = x - kEdge, gy = y - kEdge;                         // position inside the tested region
        const int cj = (int)(__umul24(gx, L.wCellMagic) >> 20), ci = (int)(__umul24(gy, L.hCellMagic) >> 20);       // exact gx / wCell for gx < 4096 (host-built magic)
        const int rx = gx - cj * L.wCell, ry = gy - ci * L.hCell;        // position inside the cell's tested interior
        const bool okL = rx > 0, okR = rx < L.wCell - 1, okU = ry > 0, okD = ry < L.hCell - 1;
        const uint8_t* c = sb + p;
        const int nU = c[-kFtPitch], nUL = c[-kFtPitch - 1], nUR = c[-kFtPitch + 1], nD = c[kFtPitch], nDL = c[kFtPitch - 1],
                  nDR = c[kFtPitch + 1], nL = c[-1], nR = c[1];
        int mAll = max(max(okU ? nU : 0, okD ? nD : 0), max(okL ? nL : 0, okR ? nR : 0));
        mAll = max(mAll, max(max((okU && okL) ? nUL : 0, (okU && okR) ? nUR : 0), max((okD && okL) ? nDL : 0, (okD && okR) ? nDR : 0)));
        const int fIni = (s >= iniTh && (mAll < iniTh || s > mAll)) ? 1 : 0;
        const int fMin = (s >= minTh && (mAll < minTh || s > mAll)) ? 2 : 0;
        if (fIni | fMin) {
            const int cell = ci * L.nCols + cj;
            const uint2 rec = make_uint2(((uint32_t)(fIni | fMin) << 30) | ((uint32_t)cell << 12) | ((uint32_t)ry << 6) | (uint32_t)rx,
                                         (uint32_t)(x - kMinBorder) | ((uint32_t)(y - kMinBorder) << 12) | ((uint32_t)s << 24));
            const int slot = atomicAdd(&on, 1);
            if (slot < kFtOutCap) outbuf[slot] = rec;
            else {      // beyond the staging buffer: append directly
                const int g = atomicAdd(&rawCount[frame * kMaxLevels + level], 1);
                if ((unsigned)g < L.rawCap) raw[(size_t)frame * rawPerFrame + L.rawBase + g] = rec;
            }
            if (fIni) cellIni[(size_t)frame * cellsPerFrame + L.cellBase + cell] = 1;
        }
    };
    if (nc <= kFtCornerCap) {
        for (int q = tid; q < nc; q += 256) nms(cqueue[q]);
    } else {
        for (int e = tid; e < scRows * kFtPitch; e += 256) if (sb[e]) nms(e);
    }
    __syncthreads();
    const int no = min(on, kFtOutCap);
    if (no == 0) return;
    if (tid == 0) obase = atomicAdd(&rawCount[frame * kMaxLevels + level], no);
    __syncthreads();
    uint2* dst = raw + (size_t)frame * rawPerFrame + L.rawBase;
    for (int i = tid; i < no; i += 256) if ((unsigned)(obase + i) < L.rawCap) dst[obase + i] = outbuf[i];
}

void launch_fast(const uint8_t* pyr, size_t pyrFrameBytes, const KLevels& kl, int tilesPerFrame, bool tall, int cellsPerFrame, size_t rawPerFrame,
                 int iniTh, int minTh, uint2* raw, int* rawCount, int* cellIni, int batch, hipStream_t s)
{
    if (tall) hipLaunchKernelGGL(k_fast<64>, dim3(tilesPerFrame * batch), dim3(256), 0, s, pyr, pyrFrameBytes, kl, cellsPerFrame, rawPerFrame,
                                 iniTh, minTh, raw, rawCount, cellIni, tilesPerFrame, batch);
    else hipLaunchKernelGGL(k_fast<32>, dim3(tilesPerFrame * batch), dim3(256), 0, s, pyr, pyrFrameBytes, kl, cellsPerFrame, rawPerFrame,
                            iniTh, minTh, raw, rawCount, cellIni, tilesPerFrame, batch);
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
// dword; lanes further out (x > w+3) are clamped onto the last pad dword and feed nobody.
// Addresses: the frame's base is wave-uniform (SGPR pair), everything below it is a 32-bit lane offset (a frame's pyramid is
// far below 4 GB: levels are capped at 4128 px, geometry.cpp), one v_mad_u32_u24 per strip-iteration and an add per row.
__device__ __forceinline__ void blur_rowsum(uint32_t d1, uint32_t (&r)[4])
{
    // bound_ctrl: the wave's end lanes read 0 (they are halo lanes, their sums feed nobody) and no lane keeps an old value
    const uint32_t d0 = __builtin_amdgcn_update_dpp(0u, d1, 0x138, 0xf, 0xf, true);    // wave_shr:1 -> lane-1's dword
    const uint32_t d2 = __builtin_amdgcn_update_dpp(0u, d1, 0x130, 0xf, 0xf, true);    // wave_shl:1 -> lane+1's dword
    const uint32_t WLO = 18u | (34u << 8) | (49u << 16) | (55u << 24);                // x-3 .. x
    const uint32_t WHI = 49u | (34u << 8) | (18u << 16);                              // x+1 .. x+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = i == 3 ? d1 : __builtin_amdgcn_alignbyte(d1, d0, i + 1);
        const uint32_t hi = i == 3 ? d2 : __builtin_amdgcn_alignbyte(d2, d1, i + 1);
        r[i] = __builtin_amdgcn_udot4(lo, WLO, __builtin_amdgcn_udot4(hi, WHI, 0u, false), false);
    }
}

template <int K>      // a * K + acc, a < 2^24, K an inline constant
__device__ __forceinline__ uint32_t mad24k(uint32_t a, uint32_t acc)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(K), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t blur_col4(const uint32_t (&a)[4], const uint32_t (&b)[4], const uint32_t (&c)[4],
                                              const uint32_t (&d)[4], const uint32_t (&e)[4], const uint32_t (&f)[4],
                                              const uint32_t (&g)[4])
{
    uint32_t s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // row sums are <= 65535, pairs of them < 2^17: a chain of four v_mad_u32_u24 (written out: the compiler turns the
        // sum of products into four multiplies and two 3-operand adds, two more full-rate-class issue slots per pixel)
        s[i] = mad24k<18>(a[i] + g[i], 32768u);
        s[i] = mad24k<34>(b[i] + f[i], s[i]);
        s[i] = mad24k<49>(c[i] + e[i], s[i]);
        s[i] = mad24k<55>(d[i], s[i]);
    }
    // min(s >> 16, 255) of four pixels into one dword: the high halves as u16 pairs (v_perm), one packed min per pair (the sum of
    // the kernel is 257, so a saturated neighbourhood reaches 257), the low bytes of the four halves (v_perm)
    const ushort2v lim = ushort2v{255, 255};
    const ushort2v p01 = __builtin_elementwise_min(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(s[1], s[0], 0x07060302u)), lim);
    const ushort2v p23 = __builtin_elementwise_min(__builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(s[3], s[2], 0x07060302u)), lim);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, size_t pyrFrameBytes, KLevels kl,
                                              int blocksPerFrame, int batch)
{
    const int lane = threadIdx.x & 31;
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    const int strip = blk * 8 + (threadIdx.x >> 5);
    const bool live = strip < kl.lv[kl.nlevels].blurStripBase;
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) level += (l < kl.nlevels && strip >= kl.lv[l].blurStripBase) ? 1 : 0;
    const KLevel& L = kl.lv[level];
    const int t = live ? strip - L.blurStripBase : 0;
    const int sy = t / L.blurStripsX, sx = t - sy * L.blurStripsX;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const int iters = live ? L.blurIters : 0;
    const int rows = 7 * iters - 6;
    const uint8_t* fsrc = pyr + (size_t)frame * pyrFrameBytes;      // wave-uniform bases
    uint8_t* fdst = blur + (size_t)frame * pyrFrameBytes;
    const int x = sx * 120 - 4 + 4 * lane;                 // first of this lane's 4 pixels
    const int y0 = sy * rows;
    const bool writer = lane >= 1 && lane <= 30 && x < w;
    const uint32_t lbase = (uint32_t)L.offset + (uint32_t)min(x, (w + 3) & ~3);     // offset of this lane's dword in row 0 (x >= -4, offset >= 64)
    // the blurred level is written as 32-px-wide column panels (geometry.cpp): the lane's four pixels sit in one panel, a row further is 32 bytes on.
    // Store offset of step 0's output row (used from step 6 on; lanes that never store may compute anything)
    uint32_t sOff = (uint32_t)(L.offset - 64) + ((uint32_t)x >> 5) * (uint32_t)L.blurPanelStride + ((uint32_t)x & 31u) + (uint32_t)(y0 - 6) * 32u;
    uint32_t r0[4], r1[4], r2[4], r3[4], r4[4], r5[4], r6[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r0[i] = r1[i] = r2[i] = r3[i] = r4[i] = r5[i] = r6[i] = 0;
    // step s consumes input row y0-3+s and (from s >= 6) emits output row y0+s-6.  The 7 loads of
    // the NEXT iteration are issued before the current one is processed (latency hiding).
    // rows yb .. yb+6: consecutive rows for a strip-iteration inside the image, reflect-101 (and a clamp for the rows past the
    // last strip's end) at the top and bottom edges
#define BLUR_LOAD7(DST, YB)                                                                          \
    {                                                                                                \
        const int yb = (YB);                                                                         \
        if (yb >= 0 && yb + 6 < h) {                                                                 \
            uint32_t o = lbase + __umul24((uint32_t)yb, (uint32_t)pitch);                            \
            _Pragma("unroll") for (int i = 0; i < 7; ++i) { DST[i] = *(const uint32_t*)(fsrc + o); o += (uint32_t)pitch; } \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 7; ++i)                                            \
                DST[i] = *(const uint32_t*)(fsrc + (lbase + __umul24((uint32_t)refl_clamp(yb + i, h), (uint32_t)pitch))); \
        }                                                                                            \
    }
#define BLUR_STEP(S, RN, A, B, C, D, E, F, G)                                                        \
    {                                                                                                \
        blur_rowsum(cur[S], RN);                                                                     \
        const int yout = y0 + s0 + (S) - 6;                                                          \
        if (s0 + (S) >= 6 && writer && yout < h)                                                     \
            *(uint32_t*)(fdst + sOff) = blur_col4(A, B, C, D, E, F, G);                              \
        sOff += 32u;                                                                                 \
    }
    uint32_t cur[7], nxt[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) cur[i] = nxt[i] = 0;
    if (iters > 0) BLUR_LOAD7(cur, y0 - 3)
    for (int it = 0; it < iters; ++it) {
        const int s0 = it * 7;
        if (it + 1 < iters) BLUR_LOAD7(nxt, y0 - 3 + s0 + 7)
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
#undef BLUR_LOAD7
#undef BLUR_STEP
}

void launch_blur(const uint8_t* pyr, uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl, int stripsPerFrame, int batch, hipStream_t s)
{
    const int bpf = (stripsPerFrame + 7) / 8;
    hipLaunchKernelGGL(k_blur, dim3(bpf * batch), dim3(256), 0, s, pyr, blur, pyrFrameBytes, kl, bpf, batch);
}

// ---------------------------------------------------------------------------
// mvImagePyramid (public member, include/Feature.h:45; src/Feature.cpp:806-821): every level of frame 0 with its 19-px
// BORDER_REFLECT_101 rim, written straight into pinned, device-mapped host memory (the adapter's cv::Mat headers point into it).
// One wave per rimmed row, a dword (4 pixels) per lane and step.
__global__ __launch_bounds__(256) void k_export_pyramid(const uint8_t* __restrict__ pyr, KLevels kl, ExportLayout lay, uint8_t* __restrict__ dst)
{
    const int level = blockIdx.y;
    const KLevel& L = kl.lv[level];
    const int w = L.w, h = L.h, stride = lay.stride[level];
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h + 2 * kEdge) return;                               // wave-uniform
    const uint8_t* src = pyr + L.offset + (size_t)refl101(y - kEdge, h) * L.pitch;
    uint32_t* out = (uint32_t*)(dst + lay.offset[level] + (size_t)y * stride);
    for (int c = threadIdx.x & 63; 4 * c < w + 2 * kEdge; c += 64) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = 4 * c + i - kEdge;
            if (x < w + kEdge) v |= (uint32_t)src[refl101(x, w)] << (8 * i);
        }
        out[c] = v;
    }
}

void launch_export_pyramid(const uint8_t* pyrFrame0, const KLevels& kl, const ExportLayout& lay, int maxRows, uint8_t* dst, hipStream_t s)
{
    hipLaunchKernelGGL(k_export_pyramid, dim3((maxRows + 3) / 4, kl.nlevels), dim3(256), 0, s, pyrFrame0, kl, lay, dst);
}

// ---------------------------------------------------------------------------
// IC_Angle (src/Feature.cpp:84-113) + steered BRIEF (src/Feature.cpp:446-482): one wave per keypoint.
constexpr int8_t kPatternI8[1024] = {
#include "orb_pattern.inc"
};
// the same table as floats, one 16-byte {x0, y0, x1, y1} record per test (the kernel multiplies them by sin/cos)
struct PatternF { float v[1024]; };
constexpr PatternF make_pattern_f()
{
    PatternF t{};
    for (int i = 0; i < 1024; ++i) t.v[i] = (float)kPatternI8[i];
    return t;
}
__constant__ __attribute__((aligned(16))) PatternF c_patf = make_pattern_f();
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

// byte masks of the IC patch rows (src/Feature.cpp:84-113 variant of SURVEY App. A4): row v = r - 14 keeps the
// pixels u in [-d, d) with d = umax[|v|] (row 0: [-15, 16)); byte j of the 32-byte row window is pixel u = j - 16
constexpr int kDescThreads = 256;             // 4 waves = 8 keypoints per workgroup (512 / 256 / 128 threads: 303.2 / 307.9 / 305.0 k frames/s sustained: smaller workgroups find room beside k_fast's)
constexpr int kDescR = 18;                   // |rotated pattern coordinate| <= round(13 * sqrt 2) = 18
constexpr int kDescRows = 2 * kDescR + 1;     // 37 rows
constexpr int kDescDw = 10;                   // 37 columns + up to 3 bytes of alignment = 40 bytes per staged row
struct IcRowMasks { uint32_t m[29][8]; };
constexpr IcRowMasks make_ic_masks()
{
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    IcRowMasks t{};
    for (int r = 0; r < 29; ++r) {
        const int v = r - 14, av = v < 0 ? -v : v, d = umax[av];
        const int lo = 16 - d, hi = v == 0 ? 32 : 16 + d;
        for (int j = 0; j < 32; ++j)
            if (j >= lo && j < hi) t.m[r][j >> 2] |= 0xFFu << (8 * (j & 3));
    }
    return t;
}
__constant__ IcRowMasks c_icmask = make_ic_masks();

// sum over the 32 lanes of the lane's half-wave, in the VALU: an inclusive DPP scan inside the rows of 16, lane 15 carried into
// the next row (rows 1 and 3), which leaves the halves' sums in lanes 31 and 63 (five adds instead of five LDS-crossbar shuffles
// on the one serial chain of the wave)
__device__ __forceinline__ int half_sum(int v, int half)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return half ? hi : lo;
}

// orientation + descriptor + final KeyPoint: TWO keypoints per wave (one per 32-lane half), so that the per-keypoint
// scalar work (level lookup, fastAtan2, sin/cos) is issued once for both.
//   moments: lane r < 29 of the half takes patch row v = r - 14 as 8 realigned dwords, masked, summed with v_dot4
//   descriptor: lane takes tests sl, sl+32, ... (8 of the 256); the ballot's low / high word is the half's result
__global__ __launch_bounds__(kDescThreads) void k_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                  size_t pyrFrameBytes, KLevels kl, const uint2* __restrict__ sel,
                                                  const int* __restrict__ selCount, int selPerFrame, int cap,
                                                  ssvio_keypoint* __restrict__ kpOut, uint8_t* __restrict__ desc, int* __restrict__ kpCount,
                                                  int* __restrict__ kpCountHost, ssvio_keypoint* __restrict__ kpHost, uint8_t* __restrict__ descHost,
                                                  int blocksPerFrame, int batch)
{
    int frame, blk;
    xcd_frame_block(blockIdx.x, blocksPerFrame, batch, frame, blk);
    const int lane = threadIdx.x & 63, sl = lane & 31, half = lane >> 5;
    // A half-wave takes one SLOT of the frame's selected-keypoint list, not the k-th keypoint: the slot's level is known from the
    // host-built level bases alone, so the record can be fetched at once and the patch loads behind it -- two dependent memory round
    // trips per wave instead of three (the level counts are only needed for where the results go, and travel beside the record).
    // Slots beyond a level's count hold a valid position of that level (the list is initialised to (kEdge, kEdge) when it is
    // allocated and only ever overwritten by the quadtree with keypoints of the same level), so their loads are harmless; they are
    // dropped once the counts are in.
    const int slotRaw = (blk * (kDescThreads / 64) + (threadIdx.x >> 6)) * 2 + half;
    const bool inList = slotRaw < selPerFrame;                    // the grid is rounded up to whole workgroups
    const int slot = inList ? slotRaw : 0;
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) level += (l < kl.nlevels && slot >= kl.lv[l].selBase) ? 1 : 0;
    const KLevel& L = kl.lv[level];
    const int idx = slot - L.selBase;                             // position in the level's list
    const uint2 in = sel[(size_t)frame * selPerFrame + slot];
    // the level counts of the frame: lane l < 16 of each half holds the inclusive prefix (consumed below, after the patch loads are out)
    int incl = (sl & 15) < kl.nlevels ? selCount[frame * kMaxLevels + (sl & 15)] : 0;
    // the 4-KB test pattern goes to LDS once per workgroup; its load travels with the others, its barrier comes after them
    __shared__ float4 spat[256];
    constexpr int kPatPer = (256 + kDescThreads - 1) / kDescThreads;       // records of the pattern per thread
    float4 patv[kPatPer];
#pragma unroll
    for (int i = 0; i < kPatPer; ++i) {
        patv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x + i * kDescThreads < 256) patv[i] = ((const float4*)c_patf.v)[threadIdx.x + i * kDescThreads];
    }
    const int x = in.x & 0xFFFF, y = in.x >> 16, resp = (in.y >> 8) & 0xFF;
    const int pitch = L.pitch;
    const uint8_t* pyrF = pyr + (size_t)frame * pyrFrameBytes;           // wave-uniform bases, 32-bit lane offsets
    const uint8_t* blurF = blur + (size_t)frame * pyrFrameBytes;
    const uint32_t centerOff = (uint32_t)L.offset + (uint32_t)(__mul24(y, pitch) + x);
    // Both patches are fetched with lane -> (row, dword) so that one load instruction touches a handful of cache lines
    // (a lane-per-row or per-sample fetch touches up to 64 and the vector L1 serialises on lines), then pass through LDS:
    // the intensity patch to turn into one row per lane, the blurred patch to be sampled at the 512 rotated positions.
    __shared__ uint32_t patch[kDescThreads / 32][kDescRows * kDescDw];                      // one region per half-wave (keypoint)
    uint32_t* lp = patch[(threadIdx.x >> 6) * 2 + half];
    // level offsets and pitches are multiples of 64 and the buffers are 256-byte aligned: alignment shifts come from x alone
    uint32_t ip = centerOff - 14 * pitch - 16;                              // intensity window: rows -14..14, pixels u = -16..
    const unsigned sh = ip & 3u; ip -= sh;
    // blurred window: rows / columns -18 .. 18, out of the panel layout k_blur writes (32-px-wide column panels, a 128-byte line = 4 rows of a
    // panel: the window touches ~23 lines instead of ~48).  A lane's dword never straddles two panels.
    const unsigned shb = (unsigned)(x - kDescR) & 3u;
    // three rows per pass: lane -> (row ri of the pass, dword ii) once, then one add per pass (27 lanes of the half busy on the 9-dword
    // intensity rows, 30 on the 10-dword blurred rows; the last pass reads one or two rows below the window - inside the level's
    // buffer, the keypoint being kEdge = 19 rows from the bottom and the buffer one row taller than the level - and drops them)
    uint32_t wi[10], wb[13];
    const unsigned si = min((unsigned)sl, 26u), ri = (si * 57u) >> 9, ii = si - 9u * ri;            // si / 9, si % 9
    const unsigned sb = min((unsigned)sl, 29u), rb = (sb * 205u) >> 11, ib = sb - 10u * rb;         // sb / 10, sb % 10
    const uint32_t step3 = 3u * (unsigned)pitch;
    {
        uint32_t o = ip + __umul24(ri, (unsigned)pitch) + 4 * ii;
#pragma unroll
        for (int t = 0; t < 10; ++t) { wi[t] = *(const uint32_t*)(pyrF + o); o += step3; }
        const unsigned xa = ((unsigned)(x - kDescR) & ~3u) + 4 * ib;
        o = (uint32_t)(L.offset - 64) + (xa >> 5) * (unsigned)L.blurPanelStride + (xa & 31u) + 32u * (unsigned)(y - kDescR + (int)rb);
#pragma unroll
        for (int t = 0; t < 13; ++t) { wb[t] = *(const uint32_t*)(blurF + o); o += 96u; }
    }
#pragma unroll
    for (int i = 0; i < kPatPer; ++i) if (threadIdx.x + i * kDescThreads < 256) spat[threadIdx.x + i * kDescThreads] = patv[i];
    __syncthreads();
    // keypoints come out in level-major order (src/Feature.cpp:537-567): this slot's keypoint is number before + idx of the frame
    int total, before;
    {
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);      // row_shr 1, 2, 4, 8: inclusive scan inside each row of 16
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        total = __builtin_amdgcn_readlane(incl, 15);
        before = __shfl(incl, max(level - 1, 0), 16);                 // lanes 0..15 of the same half
        if (level == 0) before = 0;
    }
    const int count = __shfl(incl, level, 16) - before;               // this level's count
    if (blk == 0 && threadIdx.x == 0) { kpCount[frame] = total; kpCountHost[frame] = total; }   // second copy lands in pinned host memory
    const bool live = inList && idx < count;
    if (!__any(live)) return;
    const int k = before + idx;
#pragma unroll
    for (int t = 0; t < 10; ++t) if (sl < 27 && (t < 9 || ri < 2)) lp[sl + 27 * t] = wi[t];
    int m01 = 0, m10 = 0;
    {
        const int r = min(sl, 28), v = r - 14;
        uint32_t w[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) w[i] = lp[r * 9 + i];                   // stride 9 dwords: conflict-free
        const uint4 ma = ((const uint4*)c_icmask.m[r])[0], mb = ((const uint4*)c_icmask.m[r])[1];
        const uint32_t mk[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
        unsigned sv = 0, sj = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t px = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh) & mk[i];      // pixels u = 4i-16 .. 4i-13
            sv = __builtin_amdgcn_udot4(px, 0x01010101u, sv, false);
            sj = __builtin_amdgcn_udot4(px, 0x03020100u + 0x04040404u * i, sj, false);       // weights j = u + 16
        }
        if (sl < 29) { m10 = (int)sj - 16 * (int)sv; m01 = v * (int)sv; }
    }
    m01 = half_sum(m01, half); m10 = half_sum(m10, half);
    // the intensity rows have been consumed (LDS operations of a wave complete in order): reuse the region
#pragma unroll
    for (int t = 0; t < 13; ++t) if (sl < 30 && (t < 12 || rb == 0)) lp[sl + 30 * t] = wb[t];
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // descriptor
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    float a, b;
    sincos_canonical(ang, b, a);
    const uint8_t* bc = (const uint8_t*)lp + kDescR * (kDescDw * 4) + kDescR + shb;           // the keypoint inside the staged window
    uint32_t myword = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 pt = spat[j * 32 + sl];
        const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[__mul24(r0, kDescDw * 4) + q0], t1 = bc[__mul24(r1, kDescDw * 4) + q1];
        const unsigned long long bal = __ballot(t0 < t1);
        const uint32_t mine = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;       // tests 32j .. 32j+31 of this half's keypoint
        if (sl == j) myword = mine;
    }
    if (!live) return;
    const size_t o = (size_t)frame * cap + k;
    if (sl < 8) ((uint32_t*)(desc + o * 32))[sl] = myword;
    if (descHost && sl < 8) ((uint32_t*)(descHost + o * 32))[sl] = myword;      // host entry points: a second copy straight into mapped host memory
    if (sl == 0) {
        ssvio_keypoint kp;
        const float scale = L.scale;
        kp.x = level ? __fmul_rn((float)x, scale) : (float)x;       // src/Feature.cpp:557-563
        kp.y = level ? __fmul_rn((float)y, scale) : (float)y;
        kp.size = (float)L.kpSize;
        kp.angle = angle; kp.response = (float)resp; kp.octave = level; kp.class_id = -1;
        kpOut[o] = kp;
        if (kpHost) kpHost[o] = kp;
    }
}

void launch_describe(const uint8_t* pyr, const uint8_t* blur, size_t pyrFrameBytes, const KLevels& kl,
                     const uint2* sel, const int* selCount, int selPerFrame, int cap, ssvio_keypoint* kpOut, uint8_t* desc,
                     int* kpCount, int* kpCountHost, ssvio_keypoint* kpHost, uint8_t* descHost, int batch, hipStream_t s)
{
    const int bpf = (cap + kDescThreads / 32 - 1) / (kDescThreads / 32);
    hipLaunchKernelGGL(k_describe, dim3(bpf * batch), dim3(kDescThreads), 0, s, pyr, blur, pyrFrameBytes, kl, sel, selCount, selPerFrame,
                       cap, kpOut, desc, kpCount, kpCountHost, kpHost, descHost, bpf, batch);
}

}  // namespace ssvio
