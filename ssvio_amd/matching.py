"""FeatureMatching -- host-side mirror of SSVIO's class (include/FeatureMatching.h:13-54)
on the gfx950 library.  Descriptor matrices are Nx32 uint8 numpy arrays (cv::Mat CV_8U),
match vectors are structured arrays with cv::DMatch's layout."""
import ctypes as C
import numpy as np
from . import _lib as L


class FeatureMatching:
    def __init__(self, voc_dataset=None, config=None, device=0):
        """include/FeatureMatching.h:18, src/FeatureMatching.cpp:6-12.  `config` maps
        "Threshold"/"NNRatio" (config/default_conf.yaml:26-27).  The DBoW3 vocabulary named by
        voc_dataset belongs to MatchByDBoW, which is outside this library's scope."""
        config = config or {}
        self.mTh_ = int(config.get("Threshold", 50))
        self.mNNRatio_ = float(config.get("NNRatio", 0.6))
        self.voc_dataset = voc_dataset
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

    def MatchByDBoW(self, desp1, desp2):
        """include/FeatureMatching.h:27: needs the DBoW3 vocabulary (un-vendored library, blob
        missing from the reference tree) -- out of scope, never silently emulated."""
        raise NotImplementedError("MatchByDBoW needs DBoW3 + its vocabulary; out of scope for this library")

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

    def sync(self): L.check(L.lib().ssvio_matcher_sync(self._h))

    def fetch_batch(self, counts_only=False):
        npairs, max_rows = self._last
        n = np.zeros(npairs, np.int32)
        out = None if counts_only else np.zeros((npairs, max_rows), L.DM_DTYPE)
        L.check(L.lib().ssvio_match_batch_fetch(self._h, npairs, max_rows, L.ptr(out), L.ptr(n)))
        return out, n

    def set_profiling(self, on): L.check(L.lib().ssvio_matcher_set_profiling(self._h, int(on)))
    def kernel_times(self): return L.kernel_times(L.lib().ssvio_matcher_kernel_times, self._h)
    def profile_only(self, kernel): L.check(L.lib().ssvio_matcher_profile_only(self._h, kernel.encode() if kernel else None))
    def set_stream(self, stream): L.check(L.lib().ssvio_matcher_set_stream(self._h, L.ptr(stream)))
