// FrameOps.h -- the data-parallel steps right behind the matcher, on the gfx950 library (SSVIO §8 rows f2, f3):
//   Frame::Get3DPoint            include/Frame.h:38, src/Frame.cpp:93-108
//   Odometry::OptimizeTransform  the two filter loops, src/Odometry.cpp:44-64 and :94-108
//   Map::TrackMapPoints          include/Map.h, src/Map.cpp:100-123
//   Map::InMatchVec              src/Map.cpp:151-162
//   Map::ManageMapPoints         the loop's decisions and values, src/Map.cpp:125-149 (with Frame::GetMapPointID, src/Frame.cpp:138-147)
// Free functions over a FeatureMatching adapter (they share its handle and stream); INTEGRATION.md shows where
// the reference's member functions would call them.
#pragma once
#include <utility>
#include <vector>
#include "FeatureMatching.h"

namespace ssvio {

#ifndef SSVIO_HAVE_OPENCV
struct Point3f { float x, y, z; };
typedef cv::Point2f Point2f;
#else
typedef cv::Point3f Point3f;
typedef cv::Point2f Point2f;
#endif

// Frame's statics (include/Frame.h:46-52) as Init.h:14-20 computes them
inline ssvio_camera MakeCamera(float InnerCx, float InnerCy, float InnerFx, float InnerFy, float DepthScaleYaml)
{
    ssvio_camera c; c.cx = InnerCx; c.cy = InnerCy; c.fx = InnerFx; c.fy = InnerFy;
    c.depth_scale = (float)(DepthScaleYaml / 65535.0);               // include/Init.h:18
    return c;
}

// every keypoint's Frame::Get3DPoint(index); depth is the frame's 16U image
inline bool Get3DPoints(FeatureMatching& fm, const std::vector<cv::KeyPoint>& kps, const cv::Mat& depth, const ssvio_camera& cam,
                        std::vector<Point3f>& out)
{
    static_assert(sizeof(Point3f) == 12, "Point3f layout");
    out.resize(kps.size());
    if (kps.empty()) return true;
    return ssvio_get_3d_points(fm.handle(), (const ssvio_keypoint*)kps.data(), (int)kps.size(), (const uint16_t*)depth.ptr(0),
                               depth.cols, depth.rows, (int)(depth.step / 2), &cam, &out[0].x) == SSVIO_OK;
}

struct Correspondences {
    std::vector<Point3f> last_frame_3d_point_set, cur_frame_3d_point_set;    // src/Odometry.cpp:37-39
    std::vector<Point2f> cur_frame_2d_point_set;
    std::vector<cv::DMatch> pre_good_match;                                   // :40
    std::vector<Point3f> edge_last_frame_point, edge_cur_frame_point;         // the AddCVMeasure pairs, :104
    std::vector<cv::DMatch> final_good_match;                                 // _final_good_match, :105
};

// src/Odometry.cpp:44-64 and :94-108 over _bow_match
inline bool AssembleCorrespondences(FeatureMatching& fm, const std::vector<cv::DMatch>& matches,
                                    const std::vector<cv::KeyPoint>& last_kps, const cv::Mat& last_depth,
                                    const std::vector<cv::KeyPoint>& cur_kps, const cv::Mat& cur_depth,
                                    const ssvio_camera& cam, Correspondences& c)
{
    const size_t nm = matches.size();
    c.last_frame_3d_point_set.resize(nm); c.cur_frame_3d_point_set.resize(nm); c.cur_frame_2d_point_set.resize(nm);
    c.pre_good_match.resize(nm); c.edge_last_frame_point.resize(nm); c.edge_cur_frame_point.resize(nm); c.final_good_match.resize(nm);
    int np = 0, nf = 0, rc = SSVIO_OK;
    if (nm > 0) {
        if (last_depth.cols != cur_depth.cols || last_depth.rows != cur_depth.rows || last_depth.step != cur_depth.step) return false;
        rc = ssvio_assemble_correspondences(fm.handle(), (const ssvio_dmatch*)matches.data(), (int)nm,
                                            (const ssvio_keypoint*)last_kps.data(), (int)last_kps.size(), (const uint16_t*)last_depth.ptr(0),
                                            (const ssvio_keypoint*)cur_kps.data(), (int)cur_kps.size(), (const uint16_t*)cur_depth.ptr(0),
                                            last_depth.cols, last_depth.rows, (int)(last_depth.step / 2), &cam,
                                            &c.last_frame_3d_point_set[0].x, &c.cur_frame_2d_point_set[0].x, &c.cur_frame_3d_point_set[0].x,
                                            (ssvio_dmatch*)c.pre_good_match.data(), &np,
                                            &c.edge_last_frame_point[0].x, &c.edge_cur_frame_point[0].x, (ssvio_dmatch*)c.final_good_match.data(), &nf);
        if (rc != SSVIO_OK) np = nf = 0;
    }
    c.last_frame_3d_point_set.resize(np); c.cur_frame_3d_point_set.resize(np); c.cur_frame_2d_point_set.resize(np); c.pre_good_match.resize(np);
    c.edge_last_frame_point.resize(nf); c.edge_cur_frame_point.resize(nf); c.final_good_match.resize(nf);
    return rc == SSVIO_OK;
}

// Map::TrackMapPoints(last_match_vec, this_match_vec): last_match_vec is replaced, as in the reference (:121-122)
inline bool TrackMapPoints(FeatureMatching& fm, std::vector<cv::DMatch>& last_match_vec, const std::vector<cv::DMatch>& this_match_vec)
{
    std::vector<cv::DMatch> out(last_match_vec.size());
    int n = 0, rc = SSVIO_OK;
    if (!last_match_vec.empty() && !this_match_vec.empty())
        rc = ssvio_track_map_points(fm.handle(), (const ssvio_dmatch*)last_match_vec.data(), (int)last_match_vec.size(),
                                    (const ssvio_dmatch*)this_match_vec.data(), (int)this_match_vec.size(), (ssvio_dmatch*)out.data(), (int)out.size(), &n);
    if (rc != SSVIO_OK) return false;
    out.resize(n);
    last_match_vec.swap(out);
    return true;
}

// Map::InMatchVec(i, last_match_vec) for i = 0..nkp-1 in one call (the loop of Map::ManageMapPoints, :128-131)
inline bool InMatchVec(FeatureMatching& fm, const std::vector<cv::DMatch>& last_match_vec, int nkp, std::vector<int>& match_num)
{
    match_num.assign((size_t)(nkp > 0 ? nkp : 0), -1);
    if (nkp <= 0) return true;
    return ssvio_in_match_vec(fm.handle(), (const ssvio_dmatch*)last_match_vec.data(), (int)last_match_vec.size(), nkp, match_num.data()) == SSVIO_OK;
}

// The loop of Map::ManageMapPoints(key_frame, last_match_vec), src/Map.cpp:128-148, in one call.  world_trans points at the
// 16 floats of key_frame->GetAbsPose().inverse() (Eigen::Matrix4f::data()); observed is GetKeyFrames(-1)->GetObserve().
// Afterwards, for i in order: match_num[i] == -1 -> make_shared<MapPoint>(new_point[i]) (:135); else -> _map_point_vec[map_point[i]] (:143).
inline bool ManageMapPoints(FeatureMatching& fm, const std::vector<cv::KeyPoint>& kps, const cv::Mat& depth, const ssvio_camera& cam,
                            const float* world_trans, const std::vector<cv::DMatch>& last_match_vec,
                            const std::vector<std::pair<int, int>>& observed,
                            std::vector<int>& match_num, std::vector<int>& map_point, std::vector<Point3f>& new_point)
{
    static_assert(sizeof(std::pair<int, int>) == 8, "std::pair<int, int> layout");
    const size_t n = kps.size();
    match_num.assign(n, -1); map_point.assign(n, -1); new_point.assign(n, Point3f{0.f, 0.f, 0.f});
    if (n == 0) return true;
    return ssvio_manage_map_points(fm.handle(), (const ssvio_keypoint*)kps.data(), (int)n, (const uint16_t*)depth.ptr(0),
                                   depth.cols, depth.rows, (int)(depth.step / 2), &cam, world_trans,
                                   (const ssvio_dmatch*)last_match_vec.data(), (int)last_match_vec.size(),
                                   (const int32_t*)observed.data(), (int)observed.size(),
                                   match_num.data(), map_point.data(), &new_point[0].x) == SSVIO_OK;
}

// ---- Map::MapPointCloudFusion, src/Map.cpp:164-197: the map (_final_cloud) stays on the GPU inside the matcher handle.
// One MapFuse* call = one iteration of the reference's key-frame loop (transform the frame's cloud by this_trans exactly as
// written, append, VoxelGrid(voxel_size)); MapFetch = the cloud the reference writes to map.pcd.
struct PointXYZRGB { float x, y, z, w; unsigned char b, g, r, a; unsigned int pad[3]; };     // pcl::PointXYZRGB, 32 bytes
static_assert(sizeof(PointXYZRGB) == 32, "pcl::PointXYZRGB layout");

inline bool MapClear(FeatureMatching& fm) { return ssvio_map_set(fm.handle(), nullptr, 0) == SSVIO_OK; }
// this_trans: the 16 floats of the key frame's Eigen::Matrix4f (data(), column-major)
inline bool MapFuseCloud(FeatureMatching& fm, const PointXYZRGB* cloud, size_t n, const float* this_trans, float voxel_size)
{ return ssvio_map_fuse(fm.handle(), cloud, (int)n, this_trans, voxel_size) == SSVIO_OK; }
// the frame's cloud built on the GPU from its depth (CV_16UC1) and colour (CV_8UC3, BGR) images as XtionCamera::GrabImage does
// (driver/Xtion_Driver.cpp:219-229), then fused; depth_scale_yaml is the driver's _DepthScale before its division by 65535
inline bool MapFuseFrame(FeatureMatching& fm, const cv::Mat& depth, const cv::Mat& bgr, const ssvio_camera& cam, float depth_scale_yaml,
                         const float* this_trans, float voxel_size)
{
    return ssvio_map_fuse_frame(fm.handle(), (const uint16_t*)depth.ptr(0), (int)(depth.step / 2), (const uint8_t*)bgr.ptr(0), (int)bgr.step,
                                depth.cols, depth.rows, &cam, (double)depth_scale_yaml / 65535.0, this_trans, voxel_size) == SSVIO_OK;
}
inline bool MapFetch(FeatureMatching& fm, std::vector<PointXYZRGB>& out)
{
    int n = 0;
    if (ssvio_map_fetch(fm.handle(), nullptr, 0, &n) != SSVIO_OK) return false;
    out.resize((size_t)n);
    return n == 0 || ssvio_map_fetch(fm.handle(), out.data(), n, &n) == SSVIO_OK;
}

}  // namespace ssvio
