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
 *   All float arithmetic is single precision, evaluated left to right, no contraction.
 */
#include "ssvio_oracle.h"
#include <float.h>
#include <stddef.h>

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
