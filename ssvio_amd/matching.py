"""FeatureMatching -- host-side mirror of SSVIO's class (include/FeatureMatching.h:13-54)
on the gfx950 library.  Descriptor matrices are Nx32 uint8 numpy arrays (cv::Mat CV_8U),
match vectors are structured arrays with cv::DMatch's layout."""
import ctypes as C
import numpy as np
from . import _lib as L
from .vocabulary import Vocabulary


class FeatureMatching:
    def __init__(self, voc_dataset=None, config=None, device=0):
        """include/FeatureMatching.h:18, src/FeatureMatching.cpp:6-12.  `config` maps
        "Threshold"/"NNRatio" (config/default_conf.yaml:26-27).  voc_dataset: a Vocabulary, the path of a text
        vocabulary (vocabulary.py), or None (then MatchByDBoW raises)."""
        config = config or {}
        self.mTh_ = int(config.get("Threshold", 50))
        self.mNNRatio_ = float(config.get("NNRatio", 0.6))
        self.device = device
        self.mVoc_ = Vocabulary.load_txt(voc_dataset) if isinstance(voc_dataset, (str, bytes)) else voc_dataset
        self.mBowVec1_ = self.mBowVec2_ = self.mFeatVec1_ = self.mFeatVec2_ = None
        self._h = C.c_void_p()
        L.check(L.lib().ssvio_matcher_create(self.mTh_, self.mNNRatio_, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value and L is not None and L.lib is not None:
            L.lib().ssvio_matcher_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    @staticmethod
    def _rows(d):
        if d is None:
            return None, 0, 32
        d = np.asarray(d)
        if d.size == 0:
            return None, 0, 32
        if d.dtype != np.uint8 or d.ndim != 2 or d.shape[1] != 32 or d.strides[1] != 1:
            raise ValueError("descriptors must be an Nx32 uint8 matrix")
        return d, d.shape[0], d.strides[0]

    def _call(self, fn, desp1, desp2):
        q, nq, qs = self._rows(desp1); t, nt, ts = self._rows(desp2)
        out = np.zeros(max(nq, 1), L.DM_DTYPE); n = C.c_int(0)
        L.check(fn(self._h, L.ptr(q), nq, qs, L.ptr(t), nt, ts, L.ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def MatchByBruteForce(self, desp1, desp2):
        """src/FeatureMatching.cpp:20-55 -> (bool, matches).  matches is cleared first and the
        call returns False when either matrix has no rows."""
        m = self._call(L.lib().ssvio_match_bruteforce, desp1, desp2)
        return len(m) > 0, m

    def MatchByKnnRatio(self, desp1, desp2):
        """2-NN + Threshold/NNRatio of src/FeatureMatching.cpp:85-112 over all rows"""
        m = self._call(L.lib().ssvio_match_knn2_ratio, desp1, desp2)
        return len(m) > 0, m

    def MatchByCrossCheck(self, desp1, desp2):
        m = self._call(L.lib().ssvio_match_crosscheck, desp1, desp2)
        return len(m) > 0, m

    def MatchByCrossCheckCV(self, desp1, desp2):
        """BFMatcher(NORM_HAMMING, crossCheck=True) the way OpenCV 3.x computes it (OpenCV-compat mode)"""
        m = self._call(L.lib().ssvio_match_crosscheck_cv, desp1, desp2)
        return len(m) > 0, m

    def MatchByProject(self, vp3d1, desp1, vp2d2, desp2, radian, K, R, t):
        """include/FeatureMatching.h:30-32, src/FeatureMatching.cpp:128-189: vp3d1 [n1][3] reference points with
        descriptors desp1, vp2d2 [n2][2] keypoints with descriptors desp2, K / R 3x3, t 3 (float64)"""
        q, nq, qs = self._rows(desp1); tt, nt, ts = self._rows(desp2)
        p3 = np.ascontiguousarray(vp3d1, np.float64).reshape(-1, 3); p2 = np.ascontiguousarray(vp2d2, np.float64).reshape(-1, 2)
        assert len(p3) == nq and len(p2) == nt, "one point per descriptor row"
        K = np.ascontiguousarray(K, np.float64).reshape(9); R = np.ascontiguousarray(R, np.float64).reshape(9)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        out = np.zeros(max(nq, 1), L.DM_DTYPE); n = C.c_int(0)
        L.check(L.lib().ssvio_match_project(self._h, L.ptr(p3), L.ptr(q), nq, qs, L.ptr(p2), L.ptr(tt), nt, ts, float(radian),
                                            L.ptr(K), L.ptr(R), L.ptr(t), L.ptr(out), max(nq, 1), C.byref(n)))
        m = out[:n.value].copy()
        return len(m) > 0, m

    def _voc(self):
        if self.mVoc_ is None:
            raise ValueError("MatchByDBoW / ComputeBoWVector need a vocabulary (FeatureMatching(voc_dataset=...))")
        return self.mVoc_.handle(self.device)

    def ComputeBoWVector(self, desp, levelsup=4):
        """src/FeatureMatching.cpp:192-208 -> (bowVec, featVec, (word, node, weight) per feature)"""
        d, n, st = self._rows(desp)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        L.check(L.lib().ssvio_bow_transform(self._h, self._voc(), L.ptr(d), n, st, levelsup, L.ptr(word), L.ptr(node), L.ptr(weight)))
        return self.mVoc_.bow_vector(word, weight), Vocabulary.feature_vector(node, weight), (word, node, weight)

    def MatchByDBoW(self, desp1, desp2, matches=None, levelsup=4):
        """include/FeatureMatching.h:27, src/FeatureMatching.cpp:58-125 -> (bool, matches).  With either matrix empty the
        reference returns false BEFORE clearing `matches` (:61-65): the list passed in comes back unchanged."""
        q, nq, qs = self._rows(desp1); t, nt, ts = self._rows(desp2)
        if not (nq > 0 and nt > 0):
            return False, (matches if matches is not None else np.zeros(0, L.DM_DTYPE))
        voc = self._voc()
        out = np.zeros(nq, L.DM_DTYPE); n = C.c_int(0)
        L.check(L.lib().ssvio_match_dbow(self._h, voc, L.ptr(q), nq, qs, L.ptr(t), nt, ts, levelsup, L.ptr(out), nq, C.byref(n)))
        sides = []
        for side, rows in ((0, nq), (1, nt)):                       # mBowVec*_ / mFeatVec*_ (include/FeatureMatching.h:46-49)
            word = np.zeros(rows, np.int32); node = np.zeros(rows, np.int32); weight = np.zeros(rows, np.float64); got = C.c_int(0)
            L.check(L.lib().ssvio_bow_last_transform(self._h, side, L.ptr(word), L.ptr(node), L.ptr(weight), rows, C.byref(got)))
            sides.append((self.mVoc_.bow_vector(word, weight), Vocabulary.feature_vector(node, weight)))
        (self.mBowVec1_, self.mFeatVec1_), (self.mBowVec2_, self.mFeatVec2_) = sides
        m = out[:n.value].copy()
        return len(m) > 0, m

    def match_dbow_batch_dev(self, d_desc, set_stride_rows, max_rows, d_counts, nsets, pairs, levelsup=4):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        dm, dn = C.c_void_p(), C.c_void_p()
        L.check(L.lib().ssvio_match_dbow_batch_dev(self._h, self._voc(), L.ptr(d_desc), set_stride_rows, max_rows, L.ptr(d_counts), nsets,
                                                   L.ptr(pairs), len(pairs), levelsup, C.byref(dm), C.byref(dn)))
        self._last = (len(pairs), max_rows)
        return dm.value, dn.value

    # ---- behind the matcher
    @staticmethod
    def _cam(cam):
        c = np.ascontiguousarray(cam, np.float32).reshape(5)        # cx, cy, fx, fy, DepthScale/65535
        return c

    def Get3DPoints(self, kps, depth, cam):
        """Frame::Get3DPoint (src/Frame.cpp:93-108) for every keypoint -> [n][3] float32"""
        kps = np.ascontiguousarray(kps, L.KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16); c = self._cam(cam)
        out = np.zeros((len(kps), 3), np.float32)
        L.check(L.lib().ssvio_get_3d_points(self._h, L.ptr(kps), len(kps), L.ptr(depth), depth.shape[1], depth.shape[0], depth.shape[1],
                                            L.ptr(c), L.ptr(out)))
        return out

    def AssembleCorrespondences(self, matches, last_kps, last_depth, cur_kps, cur_depth, cam):
        """the two filter loops of Odometry::OptimizeTransform (src/Odometry.cpp:44-64, :94-108)"""
        m = np.ascontiguousarray(matches, L.DM_DTYPE); nm = len(m)
        lk = np.ascontiguousarray(last_kps, L.KP_DTYPE); ck = np.ascontiguousarray(cur_kps, L.KP_DTYPE)
        ld = np.ascontiguousarray(last_depth, np.uint16); cd = np.ascontiguousarray(cur_depth, np.uint16); c = self._cam(cam)
        if ld.shape != cd.shape:
            raise ValueError("depth images differ in size")
        cap = max(nm, 1)
        last3d = np.zeros((cap, 3), np.float32); cur2d = np.zeros((cap, 2), np.float32); cur3d = np.zeros((cap, 3), np.float32)
        pre = np.zeros(cap, L.DM_DTYPE); el = np.zeros((cap, 3), np.float32); ec = np.zeros((cap, 3), np.float32); fin = np.zeros(cap, L.DM_DTYPE)
        a, b = C.c_int(0), C.c_int(0)
        L.check(L.lib().ssvio_assemble_correspondences(self._h, L.ptr(m), nm, L.ptr(lk), len(lk), L.ptr(ld), L.ptr(ck), len(ck), L.ptr(cd),
                                                       ld.shape[1], ld.shape[0], ld.shape[1], L.ptr(c),
                                                       L.ptr(last3d), L.ptr(cur2d), L.ptr(cur3d), L.ptr(pre), C.byref(a),
                                                       L.ptr(el), L.ptr(ec), L.ptr(fin), C.byref(b)))
        a, b = a.value, b.value
        return {"last3d": last3d[:a].copy(), "cur2d": cur2d[:a].copy(), "cur3d": cur3d[:a].copy(), "pre_good": pre[:a].copy(),
                "edge_last3d": el[:b].copy(), "edge_cur3d": ec[:b].copy(), "final_good": fin[:b].copy()}

    def assemble_batch_dev(self, d_matches, d_nmatches, max_rows, pairs, d_kps, kp_stride, d_kpcounts, d_depth, depth_frame_stride,
                           w, h, stride_px, cam):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2); c = self._cam(cam)
        out = L.CorrespondencesDev()
        L.check(L.lib().ssvio_assemble_batch_dev(self._h, L.ptr(d_matches), L.ptr(d_nmatches), max_rows, L.ptr(pairs), len(pairs),
                                                 L.ptr(d_kps), kp_stride, L.ptr(d_kpcounts), L.ptr(d_depth), depth_frame_stride,
                                                 w, h, stride_px, L.ptr(c), C.byref(out)))
        return out

    def RGBCloud(self, depth, bgr, cam, depth_scale_yaml):
        """the cloud loop of XtionCamera::GrabImage (driver/Xtion_Driver.cpp:219-229): w*h pcl::PointXYZRGB records, index u*h + v"""
        depth = np.ascontiguousarray(depth, np.uint16); bgr = np.ascontiguousarray(bgr, np.uint8); c = self._cam(cam)
        h, w = depth.shape
        if bgr.shape != (h, w, 3):
            raise ValueError("bgr must be h x w x 3")
        out = np.zeros(w * h, L.CLOUD_DTYPE)
        L.check(L.lib().ssvio_rgb_cloud(self._h, L.ptr(depth), w, L.ptr(bgr), bgr.strides[0], w, h, L.ptr(c),
                                        float(np.float32(depth_scale_yaml)) / 65535.0, L.ptr(out)))
        return out

    # ---- Map::MapPointCloudFusion (src/Map.cpp:164-197): the map stays in HBM, one call per key frame
    @staticmethod
    def _trans(this_trans):
        return np.ascontiguousarray(np.asarray(this_trans, np.float32).reshape(4, 4).T)       # Eigen::Matrix4f storage order

    def MapSet(self, points=None):
        pts = np.zeros(0, L.CLOUD_DTYPE) if points is None else np.ascontiguousarray(points, L.CLOUD_DTYPE)
        L.check(L.lib().ssvio_map_set(self._h, L.ptr(pts) if len(pts) else None, len(pts)))

    def MapFuse(self, cloud, this_trans, voxel_size):
        """one key frame of the loop: transform (p' = R*p + R*t as written, :175-181), append, VoxelGrid(voxel_size)"""
        T = self._trans(this_trans)
        if hasattr(cloud, "data_ptr"):                                # a device cloud (torch tensor of 32-byte records)
            n = cloud.numel() * cloud.element_size() // 32
            L.check(L.lib().ssvio_map_fuse_dev(self._h, L.ptr(cloud), n, L.ptr(T), float(voxel_size)))
        else:
            c = np.ascontiguousarray(cloud, L.CLOUD_DTYPE)
            L.check(L.lib().ssvio_map_fuse(self._h, L.ptr(c) if len(c) else None, len(c), L.ptr(T), float(voxel_size)))

    def MapFuseFrame(self, depth, bgr, cam, depth_scale_yaml, this_trans, voxel_size):
        depth = np.ascontiguousarray(depth, np.uint16); bgr = np.ascontiguousarray(bgr, np.uint8); c = self._cam(cam); T = self._trans(this_trans)
        h, w = depth.shape
        L.check(L.lib().ssvio_map_fuse_frame(self._h, L.ptr(depth), w, L.ptr(bgr), bgr.strides[0], w, h, L.ptr(c),
                                             float(np.float32(depth_scale_yaml)) / 65535.0, L.ptr(T), float(voxel_size)))

    def MapFetch(self):
        n = C.c_int(0)
        L.check(L.lib().ssvio_map_fetch(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, L.CLOUD_DTYPE)
        if n.value:
            L.check(L.lib().ssvio_map_fetch(self._h, L.ptr(out), n.value, C.byref(n)))
        return out

    def TrackMapPoints(self, last_match_vec, this_match_vec):
        """Map::TrackMapPoints (src/Map.cpp:100-123) -> the new last_match_vec"""
        last = np.ascontiguousarray(last_match_vec, L.DM_DTYPE); cur = np.ascontiguousarray(this_match_vec, L.DM_DTYPE)
        out = np.zeros(max(len(last), 1), L.DM_DTYPE); n = C.c_int(0)
        L.check(L.lib().ssvio_track_map_points(self._h, L.ptr(last), len(last), L.ptr(cur), len(cur), L.ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def InMatchVec(self, match_vec, nkp):
        """Map::InMatchVec (src/Map.cpp:151-162) for keypoints 0..nkp-1"""
        v = np.ascontiguousarray(match_vec, L.DM_DTYPE)
        out = np.zeros(nkp, np.int32)
        L.check(L.lib().ssvio_in_match_vec(self._h, L.ptr(v), len(v), nkp, L.ptr(out)))
        return out

    def ManageMapPoints(self, kps, depth, cam, world_trans, last_match_vec, obs):
        """the loop of Map::ManageMapPoints (src/Map.cpp:125-149) for every keypoint of the key frame:
        (match_num, map_point, world).  world_trans: 4x4 GetAbsPose().inverse(); obs: the previous key frame's
        (map_point_id, point_id) observation pairs (Frame::GetMapPointID, src/Frame.cpp:138-147)"""
        kps = np.ascontiguousarray(kps, L.KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16); c = self._cam(cam)
        T = np.ascontiguousarray(np.asarray(world_trans, np.float32).reshape(4, 4).T)       # Eigen::Matrix4f storage order
        v = np.ascontiguousarray(last_match_vec, L.DM_DTYPE); obs = np.ascontiguousarray(obs, np.int32).reshape(-1, 2)
        n = len(kps)
        match_num = np.zeros(n, np.int32); map_point = np.zeros(n, np.int32); world = np.zeros((n, 3), np.float32)
        L.check(L.lib().ssvio_manage_map_points(self._h, L.ptr(kps), n, L.ptr(depth), depth.shape[1], depth.shape[0], depth.shape[1],
                                                L.ptr(c), L.ptr(T), L.ptr(v), len(v), L.ptr(obs), len(obs),
                                                L.ptr(match_num), L.ptr(map_point), L.ptr(world)))
        return match_num, map_point, world

    def knn2_raw(self, desp1, desp2):
        q, nq, qs = self._rows(desp1); t, nt, ts = self._rows(desp2)
        idx = np.zeros(nq, np.int32); d1 = np.zeros(nq, np.int32); d2 = np.zeros(nq, np.int32)
        L.check(L.lib().ssvio_match_knn2_raw(self._h, L.ptr(q), nq, qs, L.ptr(t), nt, ts, L.ptr(idx), L.ptr(d1), L.ptr(d2)))
        return idx, d1, d2

    # ---- batched, device-resident
    def match_batch_dev(self, d_desc, set_stride_rows, max_rows, d_counts, pairs, mode=L.MODE_BRUTEFORCE):
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        dm, dn = C.c_void_p(), C.c_void_p()
        L.check(L.lib().ssvio_match_batch_dev(self._h, L.ptr(d_desc), set_stride_rows, max_rows, L.ptr(d_counts),
                                              L.ptr(pairs), len(pairs), mode, C.byref(dm), C.byref(dn)))
        self._last = (len(pairs), max_rows)
        return dm.value, dn.value

    def batch_summary_dev(self, d_records, id_mode=L.PAIR_ID_SEQUENTIAL, id_base=0):
        """16-byte {pair, n_matches, sum_dist} records of the last batched call, reduced on the device into d_records"""
        L.check(L.lib().ssvio_match_batch_summary_dev(self._h, id_mode, id_base, L.ptr(d_records)))

    @staticmethod
    def loop_candidates_plan(nsets, sb, rank, world):
        """-> (records of this rank, largest number of records on any rank, super-blocks of this rank); host arithmetic only"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int(0)
        L.check(L.lib().ssvio_loop_candidates_plan(nsets, sb, rank, world, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def loop_candidates_dev(self, d_desc, set_stride_rows, max_rows, d_counts, nsets, sb, rank, world, mode, d_records, records_cap):
        """ssvio_loop_candidates_dev: this rank's super-block walk, records into d_records (device); -> number of records"""
        n = C.c_int64(0)
        L.check(L.lib().ssvio_loop_candidates_dev(self._h, L.ptr(d_desc), set_stride_rows, max_rows, L.ptr(d_counts), nsets, sb, rank, world,
                                                  mode, L.ptr(d_records), records_cap, C.byref(n)))
        return n.value

    def match_batch_dev_noalloc(self, d_desc, set_stride_rows, max_rows, d_counts, d_pairs, npairs, mode, d_matches, d_nmatches):
        """ssvio_match_batch_dev_noalloc: every pointer is the caller's device memory (torch tensors or raw addresses)"""
        L.check(L.lib().ssvio_match_batch_dev_noalloc(self._h, L.ptr(d_desc), set_stride_rows, max_rows, L.ptr(d_counts), L.ptr(d_pairs),
                                                      npairs, mode, L.ptr(d_matches), L.ptr(d_nmatches)))

    def sync(self): L.check(L.lib().ssvio_matcher_sync(self._h))

    def fetch_batch(self, counts_only=False):
        npairs, max_rows = self._last
        n = np.zeros(npairs, np.int32)
        out = None if counts_only else np.zeros((npairs, max_rows), L.DM_DTYPE)
        L.check(L.lib().ssvio_match_batch_fetch(self._h, npairs, max_rows, L.ptr(out), L.ptr(n)))
        return out, n

    def set_kernel(self, kernel):
        """which 2-NN kernel the batched shapes run: "valu" (k_hamming_knn2, default) or "mfma" (k_hamming_knn2_mfma); "mfma_all":
        the latter for small single pairs too.  Same results."""
        k = {"valu": L.KERNEL_VALU, "mfma": L.KERNEL_MFMA, "mfma_all": L.KERNEL_MFMA_ALL}.get(kernel, kernel)
        L.check(L.lib().ssvio_matcher_set_kernel(self._h, int(k)))

    def get_kernel(self):
        k = C.c_int(0)
        L.check(L.lib().ssvio_matcher_get_kernel(self._h, C.byref(k)))
        return {L.KERNEL_MFMA: "mfma", L.KERNEL_MFMA_ALL: "mfma_all"}.get(k.value, "valu")

    def set_profiling(self, on): L.check(L.lib().ssvio_matcher_set_profiling(self._h, int(on)))
    def kernel_times(self): return L.kernel_times(L.lib().ssvio_matcher_kernel_times, self._h)
    def profile_only(self, kernel): L.check(L.lib().ssvio_matcher_profile_only(self._h, kernel.encode() if kernel else None))
    def set_stream(self, stream): L.check(L.lib().ssvio_matcher_set_stream(self._h, L.ptr(stream)))
