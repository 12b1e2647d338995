/* frame_oracle.c -- CPU restatement of the steps right behind the matcher (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED (see ssvio_oracle.h).  Follows /root/reference:
 *   src/Frame.cpp:93-108        Frame::Get3DPoint                      -> orc_get_3d_point
 *   src/Odometry.cpp:44-64      correspondence assembly (PnP inputs)   -> orc_assemble_correspondences, first list
 *   src/Odometry.cpp:94-108     ICP edge list / _final_good_match      -> orc_assemble_correspondences, second list
 *   src/Map.cpp:100-123         Map::TrackMapPoints                    -> orc_track_map_points
 *   src/Map.cpp:151-162         Map::InMatchVec (for every keypoint)   -> orc_in_match_vec
 *   src/Map.cpp:125-149         Map::ManageMapPoints (decisions + values) with Frame::GetMapPointID, src/Frame.cpp:138-147
 *                                                                      -> orc_manage_map_points
 *   driver/Xtion_Driver.cpp:138-162, :219-229  XtionCamera::GetRGB3DPoint over the frame -> orc_rgb_cloud (f4, first half)
 *   src/Map.cpp:164-197         Map::MapPointCloudFusion, one key frame -> orc_voxel_fuse (f4, second half; PCL restated)
 *   All float arithmetic is single precision, evaluated left to right, no contraction.
 */
#include "ssvio_oracle.h"
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* cam: InnerCx, InnerCy, InnerFx, InnerFy, DepthScale (= yaml DepthScale / 65535.0 narrowed to float, include/Init.h:14-20) */
void orc_get_3d_point(const orc_keypoint* kps, int index, const uint16_t* depth, int depth_stride_px,
                      const float cam[5], float out[3])
{
    const float inv_fx = 1 / cam[2], inv_fy = 1 / cam[3];           /* include/Init.h:19-20 */
    const float pos_x = kps[index].x, pos_y = kps[index].y;         /* :98-99 */
    /* :101 Mat::at<ushort>(int, int) called with floats: both truncate toward zero */
    const float d = (float)depth[(size_t)(int)pos_y * depth_stride_px + (int)pos_x] * cam[4];
    out[0] = (pos_x - cam[0]) * d * inv_fx;                         /* :103 */
    out[1] = (pos_y - cam[1]) * d * inv_fy;                         /* :104 */
    out[2] = d;                                                     /* :105 */
}

/* src/Odometry.cpp:44-64 then :94-108.  match.trainIdx indexes the last frame, match.queryIdx the current one.
 * pre lists (one entry per match whose last-frame point has z >= DepthScale): last3d, cur2d, cur3d, pre_good;
 * final lists (those whose current-frame point has z >= DepthScale too): edge_last3d, edge_cur3d, final_good. */
void orc_assemble_correspondences(const orc_dmatch* matches, int nm,
                                  const orc_keypoint* last_kps, const uint16_t* last_depth,
                                  const orc_keypoint* cur_kps, const uint16_t* cur_depth, int depth_stride_px,
                                  const float cam[5],
                                  float* last3d, float* cur2d, float* cur3d, orc_dmatch* pre_good, int* n_pre,
                                  float* edge_last3d, float* edge_cur3d, orc_dmatch* final_good, int* n_final)
{
    int np = 0, nf = 0;
    for (int i = 0; i < nm; ++i) {
        float lp[3];
        orc_get_3d_point(last_kps, matches[i].trainIdx, last_depth, depth_stride_px, cam, lp);   /* :50 */
        if (lp[2] >= cam[4]) {                                                                   /* :51 */
            last3d[3 * np] = lp[0]; last3d[3 * np + 1] = lp[1]; last3d[3 * np + 2] = lp[2];
            cur2d[2 * np] = cur_kps[matches[i].queryIdx].x; cur2d[2 * np + 1] = cur_kps[matches[i].queryIdx].y;   /* :54-56 */
            orc_get_3d_point(cur_kps, matches[i].queryIdx, cur_depth, depth_stride_px, cam, cur3d + 3 * np);     /* :59 */
            pre_good[np] = matches[i];
            ++np;
        }
    }
    for (int i = 0; i < np; ++i) {                                                               /* :94-108 */
        if (cur3d[3 * i + 2] >= cam[4]) {
            for (int c = 0; c < 3; ++c) { edge_last3d[3 * nf + c] = last3d[3 * i + c]; edge_cur3d[3 * nf + c] = cur3d[3 * i + c]; }
            final_good[nf] = pre_good[i];
            ++nf;
        }
    }
    *n_pre = np; *n_final = nf;
}

/* Map::TrackMapPoints, src/Map.cpp:100-123: for every last match, in order, the FIRST this-match whose trainIdx equals
 * its queryIdx; emits DMatch{queryIdx = this.queryIdx, trainIdx = last.trainIdx} with cv::DMatch()'s defaults in the
 * other two fields (imgIdx -1, distance FLT_MAX). */
int orc_track_map_points(const orc_dmatch* last, int nlast, const orc_dmatch* cur, int ncur, orc_dmatch* out)
{
    int n = 0;
    for (int i = 0; i < nlast; ++i) {
        for (int j = 0; j < ncur; ++j) {
            if (last[i].queryIdx == cur[j].trainIdx) {
                orc_dmatch m; m.queryIdx = cur[j].queryIdx; m.trainIdx = last[i].trainIdx; m.imgIdx = -1; m.distance = FLT_MAX;
                out[n++] = m;
                break;
            }
        }
    }
    return n;
}

/* Map::InMatchVec(i, vec), src/Map.cpp:151-162, for i = 0..nkp-1 (the loop of ManageMapPoints, :128-131):
 * index of the first match whose queryIdx is i, or -1 */
void orc_in_match_vec(const orc_dmatch* vec, int n, int nkp, int32_t* out)
{
    for (int i = 0; i < nkp; ++i) {
        out[i] = -1;
        for (int j = 0; j < n; ++j) if (vec[j].queryIdx == i) { out[i] = j; break; }
    }
}

/* Map::ManageMapPoints, src/Map.cpp:125-149, for every keypoint i of the key frame -- the decisions and values of
 * the loop; creating MapPoint objects and the SetObserve bookkeeping stay with the caller.
 *   match_num[i] = InMatchVec(i, last_match_vec)                                                         (:130)
 *   match_num[i] == -1 (a new map point, :131-140): world[i] = R * Get3DPoint(i) + t with R, t the upper 3x4 of
 *       world_trans = GetAbsPose().inverse() (:127, :133-136), map_point[i] = -1
 *   otherwise (:141-147): map_point[i] = GetKeyFrames(-1)->GetMapPointID(last_match_vec[match_num].queryIdx)
 *       = .first of the FIRST observation pair whose .second is that id (src/Frame.cpp:138-147).  The reference runs
 *       off the end of a non-void function when no pair has it (undefined there): -1 here.  world[i] = 0.
 * world_trans: 16 floats in Eigen::Matrix4f's own (column-major) order, element (r, c) at [r + 4 c].
 * Canonical float order (Eigen's fixed-size 3-term reduction splits 1 + 2): out_r = (R(r,0) x + (R(r,1) y + R(r,2) z)) + t_r,
 * single precision, no contraction (the reference is built -O3 for baseline x86-64: no FMA).
 * obs: the key frame's _observed_mappid_pid_vec as int32 pairs {map_point_id, point_id} (include/Frame.h). */
void orc_manage_map_points(const orc_keypoint* kps, int nkp, const uint16_t* depth, int depth_stride_px, const float cam[5],
                           const float world_trans[16], const orc_dmatch* last_match_vec, int nm,
                           const int32_t* obs, int nobs, int32_t* match_num, int32_t* map_point, float* world)
{
    orc_in_match_vec(last_match_vec, nm, nkp, match_num);
    for (int i = 0; i < nkp; ++i) {
        map_point[i] = -1;
        world[3 * i] = world[3 * i + 1] = world[3 * i + 2] = 0.f;
        if (match_num[i] == -1) {
            float p[3];
            orc_get_3d_point(kps, i, depth, depth_stride_px, cam, p);                          /* :133 */
            for (int r = 0; r < 3; ++r) {
                const float a0 = world_trans[r] * p[0], a1 = world_trans[r + 4] * p[1], a2 = world_trans[r + 8] * p[2];
                const float tail = a1 + a2;
                const float dot = a0 + tail;
                world[3 * i + r] = dot + world_trans[r + 12];                                  /* :135-136 */
            }
        } else {
            const int point_id = last_match_vec[match_num[i]].queryIdx;                        /* :143 */
            for (int k = 0; k < nobs; ++k)                                                     /* src/Frame.cpp:140-146 */
                if (obs[2 * k + 1] == point_id) { map_point[i] = obs[2 * k]; break; }
        }
    }
}

/* XtionCamera::GrabImage's cloud loop (driver/Xtion_Driver.cpp:219-229) over GetRGB3DPoint (:138-162): point
 * i = u * rows + v (column-major) from pixel (x = u, y = v).  The driver's _DepthScale is a DOUBLE
 * (driver/Xtion_Driver.h:47, = (float)yaml / 65535.0 in double, .cpp:14): the product with the raw depth is formed in
 * double and narrowed once.  bgr: 8UC3 image whose bytes are B, G, R (the driver converted, :215).
 * out: pcl::PointXYZRGB records of 32 bytes {x, y, z, 1.0f, b, g, r, a = 255, 12 bytes 0}. */
void orc_rgb_cloud(const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride, int w, int h,
                   float cx, float cy, float fx, float fy, double depth_scale, uint8_t* out)
{
    const float inv_fx = 1 / fx, inv_fy = 1 / fy;                      /* .cpp:15-16 */
    for (int u = 0; u < w; ++u) {
        for (int v = 0; v < h; ++v) {
            const float d = (float)((double)depth[(size_t)v * depth_stride_px + u] * depth_scale);    /* :146 */
            float p[4];
            p[0] = ((float)u - cx) * d * inv_fx;                        /* :149 */
            p[1] = ((float)v - cy) * d * inv_fy;
            p[2] = d; p[3] = 1.0f;
            uint8_t* o = out + ((size_t)u * h + v) * 32;
            for (int k = 0; k < 16; ++k) o[k] = ((const uint8_t*)p)[k];
            const uint8_t* c = bgr + (size_t)v * bgr_stride + 3 * u;
            o[16] = c[0]; o[17] = c[1]; o[18] = c[2]; o[19] = 255;       /* :157-159; a = 255 is PointXYZRGB's default */
            for (int k = 20; k < 32; ++k) o[k] = 0;
        }
    }
}

/* ---- Map::MapPointCloudFusion, src/Map.cpp:164-197 (second half): ONE iteration of its key-frame loop.
 *
 * PARITY UNPINNED: the arithmetic lives in PCL (pcl::transformPointCloud, pcl::VoxelGrid<PointXYZRGB>, un-vendored, not in
 * this image; CMakeLists.txt:31 `find_package(PCL REQUIRED)` pins no version) and in Eigen (Affine3f).  Restated from the
 * published PCL 1.8/1.9 sources:
 *   transform (src/Map.cpp:175-181): Affine3f::Identity().rotate(R).translate(t) has linear part R and translation R*t, i.e.
 *     p' = R*p + R*t (NOT R*p + t).  R*t row r in Eigen's fixed-size order R(r,0)*t0 + (R(r,1)*t1 + R(r,2)*t2);
 *     transformPointCloud (dense cloud): x' = ((m00*x + m01*y) + m02*z) + m03, all float, colour and data[3] copied.
 *   concatenation (:183): the transformed points follow the points already in the map.
 *   VoxelGrid::applyFilter (:185-190), leaf = _VoxelSize on all axes, downsample_all_data = true, min_points_per_voxel = 0:
 *     inverse_leaf = 1.f / leaf; getMinMax3D over all points; d{x,y,z} = (int64)((max - min) * inverse_leaf) + 1 and, when
 *     dx*dy*dz > INT32_MAX, the filter warns and returns the input unchanged; min_b = (int)floor(min * inverse_leaf),
 *     div_b = max_b - min_b + 1; ijk = (int)(floor(p * inverse_leaf) - (float)min_b); idx = i + j*div_x + k*div_x*div_y;
 *     sort by idx; one output point per run of equal idx, in ascending idx: CentroidPoint -- xyz = (sum of xyz) / (float)n,
 *     rgba = (uint32)(sum a / n) << 24 | (uint32)(sum r / n) << 16 | (uint32)(sum g / n) << 8 | (uint32)(sum b / n) with float
 *     sums; data[3] = 1 (PointXYZRGB's constructor).
 * Where PCL is not a function of its inputs: std::sort is unstable, so the order in which a voxel's points are summed (and
 * with it the last bits of the float sums) depends on the library build.  Canonical here, as for the quadtree:
 * stable by position in the concatenated cloud.
 * this_trans: 16 floats, Eigen::Matrix4f::data() order (column-major) of GetAbsPose().inverse() (:174).
 * points: 32-byte records {x, y, z, data[3], b, g, r, a, 12 bytes}.  Returns the number of output points (<= cap assumed). */
typedef struct { uint32_t idx; uint32_t pos; } orc_vox_key;

static int orc_vox_cmp(const void* a, const void* b)
{
    const orc_vox_key* x = (const orc_vox_key*)a; const orc_vox_key* y = (const orc_vox_key*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

int orc_voxel_fuse(const uint8_t* final_cloud, int n_final, const uint8_t* frame_cloud, int n_frame,
                   const float* this_trans, float voxel_size, uint8_t* out)
{
    const int n = n_final + n_frame;
    if (n <= 0) return 0;
    uint8_t* all = (uint8_t*)malloc((size_t)n * 32);
    memcpy(all, final_cloud, (size_t)n_final * 32);
    /* :175-181 */
    float m[3][4];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[r][c] = this_trans[c * 4 + r];
        const float t0 = this_trans[12], t1 = this_trans[13], t2 = this_trans[14];
        m[r][3] = m[r][0] * t0 + (m[r][1] * t1 + m[r][2] * t2);
    }
    for (int i = 0; i < n_frame; ++i) {
        const uint8_t* s = frame_cloud + (size_t)i * 32; uint8_t* d = all + (size_t)(n_final + i) * 32;
        float p[3]; memcpy(p, s, 12);
        float q[3];
        for (int r = 0; r < 3; ++r) q[r] = ((m[r][0] * p[0] + m[r][1] * p[1]) + m[r][2] * p[2]) + m[r][3];
        memcpy(d, s, 32); memcpy(d, q, 12);
    }
    /* getMinMax3D */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, all + (size_t)i * 32, 12);
        for (int k = 0; k < 3; ++k) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; }
    }
    const float inv = 1.0f / voxel_size;
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {          /* "Leaf size is too small for the input dataset": output = input */
        memcpy(out, all, (size_t)n * 32); free(all); return n;
    }
    int minb[3], divb[3];
    for (int k = 0; k < 3; ++k) { minb[k] = (int)floorf(mn[k] * inv); divb[k] = (int)floorf(mx[k] * inv) - minb[k] + 1; }
    orc_vox_key* keys = (orc_vox_key*)malloc((size_t)n * sizeof(orc_vox_key));
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, all + (size_t)i * 32, 12);
        const int i0 = (int)(floorf(p[0] * inv) - (float)minb[0]), i1 = (int)(floorf(p[1] * inv) - (float)minb[1]), i2 = (int)(floorf(p[2] * inv) - (float)minb[2]);
        keys[i].idx = (uint32_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]); keys[i].pos = (uint32_t)i;
    }
    qsort(keys, (size_t)n, sizeof(orc_vox_key), orc_vox_cmp);
    int total = 0;
    for (int a = 0; a < n;) {
        int b = a + 1;
        while (b < n && keys[b].idx == keys[a].idx) ++b;
        float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0, sa = 0;
        for (int j = a; j < b; ++j) {
            const uint8_t* s = all + (size_t)keys[j].pos * 32;
            float p[3]; memcpy(p, s, 12);
            sx += p[0]; sy += p[1]; sz += p[2];
            sb += (float)s[16]; sg += (float)s[17]; sr += (float)s[18]; sa += (float)s[19];
        }
        const float cnt = (float)(b - a);
        uint8_t* o = out + (size_t)total * 32;
        const float q[4] = {sx / cnt, sy / cnt, sz / cnt, 1.0f};
        memcpy(o, q, 16);
        o[16] = (uint8_t)(uint32_t)(sb / cnt); o[17] = (uint8_t)(uint32_t)(sg / cnt); o[18] = (uint8_t)(uint32_t)(sr / cnt); o[19] = (uint8_t)(uint32_t)(sa / cnt);
        memset(o + 20, 0, 12);
        ++total; a = b;
    }
    free(keys); free(all);
    return total;
}
