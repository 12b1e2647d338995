// kernels_frame.hip -- the steps right behind the matcher, on gfx950.
//
//   k_get_3d_points   Frame::Get3DPoint (src/Frame.cpp:93-108) for a list of keypoints: depth gather + pinhole inverse
//   k_assemble        Odometry::OptimizeTransform's two filter loops (src/Odometry.cpp:44-64, :94-108): the PnP input
//                     lists and the ICP edge lists, both in match order (ordered compaction by workgroup scans)
//   k_first_index     index map "key -> first position in a DMatch list" (atomicMin), which replaces the linear
//                     searches of Map::InMatchVec (src/Map.cpp:151-162) and Map::TrackMapPoints (:100-123)
//   k_track           TrackMapPoints' join through that map, emitted in last_match_vec order
//   k_manage          Map::ManageMapPoints' per-keypoint decisions (src/Map.cpp:125-149): InMatchVec + Frame::GetMapPointID
//                     (src/Frame.cpp:138-147) through two index maps, new points back-projected into the world frame
//   k_rgb_cloud       the per-pixel cloud of XtionCamera::GrabImage (driver/Xtion_Driver.cpp:138-162, :219-229; f4, first half)
// Float code is single precision, left to right, no contraction (this library is built with -ffp-contract=off).
#include "match.h"
#include <cfloat>

namespace ssvio {

// Frame::Get3DPoint.  Mat::at<ushort>(pos_y, pos_x) with float arguments truncates both toward zero (:101); the
// reference does not check the pixel against the image, the clamp only keeps a corrupt keypoint from faulting.
__device__ __forceinline__ void get_3d_point(const ssvio_keypoint& kp, const uint16_t* depth, const FrameCam& c, float out[3])
{
    const float pos_x = kp.x, pos_y = kp.y;
    const int px = min(max((int)pos_x, 0), c.w - 1), py = min(max((int)pos_y, 0), c.h - 1);
    const float d = (float)depth[(size_t)py * c.stridePx + px] * c.depthScale;
    out[0] = (pos_x - c.cx) * d * c.invFx;                            // :103
    out[1] = (pos_y - c.cy) * d * c.invFy;                            // :104
    out[2] = d;                                                       // :105
}

__global__ __launch_bounds__(256) void k_get_3d_points(const ssvio_keypoint* kps, int n, const uint16_t* depth, FrameCam c, float* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float p[3];
    get_3d_point(kps[i], depth, c, p);
    out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
}

void launch_get_3d_points(const ssvio_keypoint* kps, int n, const uint16_t* depth, const FrameCam& c, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_get_3d_points, dim3((n + 255) / 256), dim3(256), 0, s, kps, n, depth, c, out);
}

// XtionCamera::GrabImage's cloud loop (driver/Xtion_Driver.cpp:219-229): point u * rows + v from pixel (u, v).  A 16 x 16
// pixel tile per workgroup: threads read along rows (coalesced depth / colour loads), transpose through LDS and write
// along columns (16 consecutive 32-byte points per tile column).  depth * _DepthScale is a double product (Xtion_Driver.h:47).
__global__ __launch_bounds__(256) void k_rgb_cloud(const uint16_t* depth, int depthStridePx, const uint8_t* bgr, int bgrStride,
                                                   int w, int h, float cx, float cy, float invFx, float invFy, double depthScale, uint4* out)
{
    __shared__ uint4 lo[16][17], hi[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int u = blockIdx.x * 16 + tx, v = blockIdx.y * 16 + ty;
    if (u < w && v < h) {
        const float d = (float)((double)depth[(size_t)v * depthStridePx + u] * depthScale);          // :146
        const float x = ((float)u - cx) * d * invFx, y = ((float)v - cy) * d * invFy;                 // :149-150
        const uint8_t* c = bgr + (size_t)v * bgrStride + 3 * u;
        lo[ty][tx] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(d), __float_as_uint(1.0f));
        hi[ty][tx] = make_uint4((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | 0xFF000000u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int ou = blockIdx.x * 16 + ty, ov = blockIdx.y * 16 + tx;       // transposed: consecutive threads -> consecutive v
    if (ou < w && ov < h) {
        uint4* o = out + 2 * ((size_t)ou * h + ov);
        o[0] = lo[tx][ty]; o[1] = hi[tx][ty];
    }
}

void launch_rgb_cloud(const uint16_t* depth, int depthStridePx, const uint8_t* bgr, int bgrStride, int w, int h,
                      float cx, float cy, float invFx, float invFy, double depthScale, void* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_rgb_cloud, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, s, depth, depthStridePx, bgr, bgrStride, w, h,
                       cx, cy, invFx, invFy, depthScale, (uint4*)out);
}

// one workgroup per frame pair; match i of the pair is handled by thread i % 256 in pass i / 256
__global__ __launch_bounds__(256) void k_assemble(AssembleArgs a)
{
    __shared__ int wsum[4];
    const int pair = blockIdx.x;
    const int curSet = a.pairs ? a.pairs[2 * pair] : 0, lastSet = a.pairs ? a.pairs[2 * pair + 1] : 1;
    const int nm = min(a.nmatches[pair], a.maxRows);
    const ssvio_dmatch* M = a.matches + (size_t)pair * a.maxRows;
    const ssvio_keypoint* lastK = a.kps + (size_t)lastSet * a.kpStride;
    const ssvio_keypoint* curK = a.kps + (size_t)curSet * a.kpStride;
    const uint16_t* lastD = a.depth + (size_t)lastSet * a.depthFrameStride;
    const uint16_t* curD = a.depth + (size_t)curSet * a.depthFrameStride;
    const int nLast = min(a.kpCounts[lastSet], a.kpStride), nCur = min(a.kpCounts[curSet], a.kpStride);
    const size_t ob = (size_t)pair * a.maxRows;
    int nPre = 0, nFin = 0;
    for (int i0 = 0; i0 < nm; i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool pre = false, fin = false;
        ssvio_dmatch m = {0, 0, 0, 0.f};
        float lp[3] = {0.f, 0.f, 0.f}, cp[3] = {0.f, 0.f, 0.f}, c2x = 0.f, c2y = 0.f;
        if (i < nm) {
            m = M[i];
            if (m.trainIdx >= 0 && m.trainIdx < nLast && m.queryIdx >= 0 && m.queryIdx < nCur) {   // indices a matcher emits
                get_3d_point(lastK[m.trainIdx], lastD, a.cam, lp);                     // src/Odometry.cpp:50
                pre = lp[2] >= a.cam.depthScale;                                       // :51
                const ssvio_keypoint ck = curK[m.queryIdx];
                c2x = ck.x; c2y = ck.y;                                                // :54-55
                get_3d_point(ck, curD, a.cam, cp);                                     // :59
                fin = pre && cp[2] >= a.cam.depthScale;                                // :102
            }
        }
        int totPre, totFin;
        const int oPre = block_excl_scan(pre ? 1 : 0, wsum, totPre);
        const int oFin = block_excl_scan(fin ? 1 : 0, wsum, totFin);
        if (pre) {
            const size_t o = ob + nPre + oPre;
            a.last3d[3 * o] = lp[0]; a.last3d[3 * o + 1] = lp[1]; a.last3d[3 * o + 2] = lp[2];
            a.cur2d[2 * o] = c2x; a.cur2d[2 * o + 1] = c2y;
            a.cur3d[3 * o] = cp[0]; a.cur3d[3 * o + 1] = cp[1]; a.cur3d[3 * o + 2] = cp[2];
            a.preGood[o] = m;
        }
        if (fin) {
            const size_t o = ob + nFin + oFin;
            a.edgeLast3d[3 * o] = lp[0]; a.edgeLast3d[3 * o + 1] = lp[1]; a.edgeLast3d[3 * o + 2] = lp[2];
            a.edgeCur3d[3 * o] = cp[0]; a.edgeCur3d[3 * o + 1] = cp[1]; a.edgeCur3d[3 * o + 2] = cp[2];
            a.finalGood[o] = m;
        }
        nPre += totPre; nFin += totFin;
    }
    if (threadIdx.x == 0) { a.counts[2 * pair] = nPre; a.counts[2 * pair + 1] = nFin; }
}

void launch_assemble(const AssembleArgs& a, int npairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_assemble, dim3(npairs), dim3(256), 0, s, a);
}

// table[key] = min position j with rec[j].field == key; table starts at 0xFFFFFFFF (= -1 as int32: "not found").
// Records are `strideInts` int32 wide, the key sits at int `fieldOff` (DMatch: 4 wide, queryIdx 0 / trainIdx 1;
// observation pair {map_point_id, point_id}: 2 wide, point_id 1).
__global__ __launch_bounds__(256) void k_first_index(const int32_t* rec, int n, int strideInts, int fieldOff, uint32_t* table, int tableSize)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int key = rec[(size_t)j * strideInts + fieldOff];
    if (key >= 0 && key < tableSize) atomicMin(&table[key], (uint32_t)j);
}

// Map::ManageMapPoints (src/Map.cpp:125-149), one thread per keypoint.  matchTable: first position in last_match_vec per
// queryIdx (InMatchVec, :130); obsTable: first position in the previous key frame's observation pairs per point id
// (Frame::GetMapPointID, src/Frame.cpp:138-147).  T: GetAbsPose().inverse() in Eigen's column-major order.
__global__ __launch_bounds__(256) void k_manage(const ssvio_keypoint* kps, int nkp, const uint16_t* depth, FrameCam c, ManageXform T,
                                                const ssvio_dmatch* last, const uint32_t* matchTable, const int32_t* obs,
                                                const uint32_t* obsTable, int obsTableSize, int32_t* matchNum, int32_t* mapPoint, float* world)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nkp) return;
    const uint32_t j = matchTable[i];
    int mp = -1;
    float o[3] = {0.f, 0.f, 0.f};
    if (j == 0xFFFFFFFFu) {
        float p[3];
        get_3d_point(kps[i], depth, c, p);                                                // :133
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a0 = T.m[r] * p[0], a1 = T.m[r + 4] * p[1], a2 = T.m[r + 8] * p[2];
            const float tail = a1 + a2;
            const float dot = a0 + tail;                                                  // Eigen's 3-term reduction: e0 + (e1 + e2)
            o[r] = dot + T.m[r + 12];                                                     // :135-136
        }
    } else {
        const int pointId = last[j].queryIdx;                                             // :143 (== i by construction of the table)
        if (pointId >= 0 && pointId < obsTableSize) {
            const uint32_t k = obsTable[pointId];
            if (k != 0xFFFFFFFFu) mp = obs[2 * (size_t)k];
        }
    }
    matchNum[i] = (int32_t)j; mapPoint[i] = mp;
    world[3 * i] = o[0]; world[3 * i + 1] = o[1]; world[3 * i + 2] = o[2];
}

// Map::TrackMapPoints (src/Map.cpp:100-123): one workgroup, output in last_match_vec order.  With ldsTable the index map over
// this_match_vec (first position per trainIdx) is built here in LDS first -- one launch for the whole call (tableSize <= kTrackLdsTable).
__global__ __launch_bounds__(256) void k_track(const ssvio_dmatch* last, int nlast, const ssvio_dmatch* cur, int ncur, const uint32_t* table,
                                               int tableSize, int ldsTable, ssvio_dmatch* out, int* nout)
{
    extern __shared__ uint32_t ltab[];
    __shared__ int wsum[4];
    if (ldsTable) {
        for (int k = threadIdx.x; k < tableSize; k += 256) ltab[k] = 0xFFFFFFFFu;
        __syncthreads();
        for (int j = threadIdx.x; j < ncur; j += 256) {
            const int key = cur[j].trainIdx;
            if (key >= 0 && key < tableSize) atomicMin(&ltab[key], (uint32_t)j);
        }
        __syncthreads();
        table = ltab;
    }
    int written = 0;
    for (int i0 = 0; i0 < nlast; i0 += 256) {
        const int i = i0 + threadIdx.x;
        uint32_t j = 0xFFFFFFFFu; int train = 0;
        if (i < nlast) {
            const ssvio_dmatch l = last[i];
            train = l.trainIdx;
            if (l.queryIdx >= 0 && l.queryIdx < tableSize) j = table[l.queryIdx];        // :108, first j
        }
        const bool keep = j != 0xFFFFFFFFu;
        int total;
        const int off = block_excl_scan(keep ? 1 : 0, wsum, total);
        if (keep) {
            ssvio_dmatch m; m.queryIdx = cur[j].queryIdx; m.trainIdx = train; m.imgIdx = -1; m.distance = FLT_MAX;   // cv::DMatch() defaults
            out[written + off] = m;
        }
        written += total;
    }
    if (threadIdx.x == 0) *nout = written;
}

void launch_first_index(const int32_t* rec, int n, int strideInts, int fieldOff, uint32_t* table, int tableSize, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_first_index, dim3((n + 255) / 256), dim3(256), 0, s, rec, n, strideInts, fieldOff, table, tableSize);
}

void launch_manage(const ssvio_keypoint* kps, int nkp, const uint16_t* depth, const FrameCam& c, const ManageXform& T,
                   const ssvio_dmatch* last, const uint32_t* matchTable, const int32_t* obs, const uint32_t* obsTable, int obsTableSize,
                   int32_t* matchNum, int32_t* mapPoint, float* world, hipStream_t s)
{
    hipLaunchKernelGGL(k_manage, dim3((nkp + 255) / 256), dim3(256), 0, s, kps, nkp, depth, c, T, last, matchTable, obs, obsTable, obsTableSize,
                       matchNum, mapPoint, world);
}

void launch_track(const ssvio_dmatch* last, int nlast, const ssvio_dmatch* cur, int ncur, const uint32_t* table, int tableSize,
                  ssvio_dmatch* out, int* nout, hipStream_t s)
{
    const int ldsTable = table == nullptr;                  // no global map given: build it in LDS (the caller checked the size)
    hipLaunchKernelGGL(k_track, dim3(1), dim3(256), ldsTable ? (size_t)tableSize * 4 : 0, s, last, nlast, cur, ncur, table, tableSize, ldsTable, out, nout);
}

}  // namespace ssvio
