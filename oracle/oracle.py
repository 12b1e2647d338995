"""ctypes bindings for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by ssvio_amd/.  PARITY UNPINNED: see
oracle/ssvio_oracle.h.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KP_DTYPE.itemsize == 28 and DM_DTYPE.itemsize == 16


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.c", "match_oracle.c", "bow_oracle.c", "frame_oracle.c",
                                                 "ssvio_oracle.h", "orb_pattern.inc")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class _Grid(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nCols", "nRows", "wCell", "hCell", "minBX", "minBY", "maxBX", "maxBY")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
        L.orc_cvround.restype = i32; L.orc_cvround.argtypes = [C.c_double]
        L.orc_rgb2gray.argtypes = [vp, i32, i32, i32, vp, i32]
        L.orc_resize_linear.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32]
        L.orc_border_reflect101.argtypes = [vp, i32, i32, i32, i32]
        L.orc_fast9.restype = i32; L.orc_fast9.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, i32]
        L.orc_fast_score_ref.restype = i32; L.orc_fast_score_ref.argtypes = [vp, i32, i32]
        L.orc_gaussian7.argtypes = [vp, i32, i32, i32, vp, i32]
        L.orc_fast_atan2.restype = f32; L.orc_fast_atan2.argtypes = [f32, f32]
        L.orc_sincosf.argtypes = [f32, vp, vp]
        L.orc_gauss_kernel_fixed.argtypes = [vp]
        L.orc_create.restype = vp; L.orc_create.argtypes = [i32, f32, i32, i32, i32]
        L.orc_destroy.argtypes = [vp]
        for n in ("orc_scale_factors", "orc_inv_scale_factors", "orc_level_sigma2", "orc_inv_level_sigma2",
                  "orc_features_per_level", "orc_umax"):
            getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp]
        L.orc_extract.restype = i32; L.orc_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32]
        L.orc_extract_rgb.restype = i32; L.orc_extract_rgb.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32]
        L.orc_level_size.restype = i32; L.orc_level_size.argtypes = [vp, i32, vp, vp]
        L.orc_level_roi.restype = vp; L.orc_level_roi.argtypes = [vp, i32, vp]
        L.orc_level_blur.restype = vp; L.orc_level_blur.argtypes = [vp, i32, vp]
        L.orc_level_candidates.restype = i32; L.orc_level_candidates.argtypes = [vp, i32, vp, vp, vp]
        L.orc_level_keypoints.restype = i32; L.orc_level_keypoints.argtypes = [vp, i32, vp]
        L.orc_level_tie_at_cut.restype = i32; L.orc_level_tie_at_cut.argtypes = [vp, i32]
        L.orc_cell_grid.restype = i32; L.orc_cell_grid.argtypes = [i32, i32, vp]
        L.orc_distribute_octtree.restype = i32
        L.orc_distribute_octtree.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]
        L.orc_ic_angle.restype = f32; L.orc_ic_angle.argtypes = [vp, i32, i32, i32, vp, vp]
        L.orc_orb_descriptor.argtypes = [vp, i32, i32, i32, f32, vp]
        L.orc_orb_descriptor_libm.argtypes = [vp, i32, i32, i32, f32, vp]
        L.orc_hamming256.restype = i32; L.orc_hamming256.argtypes = [vp, vp]
        L.orc_knn2_raw.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, vp]
        L.orc_bf_nn.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp]
        L.orc_match_bruteforce.restype = i32; L.orc_match_bruteforce.argtypes = [vp, i32, i32, vp, i32, i32, vp]
        L.orc_match_knn2_ratio.restype = i32; L.orc_match_knn2_ratio.argtypes = [vp, i32, i32, vp, i32, i32, i32, f32, vp]
        L.orc_match_crosscheck.restype = i32; L.orc_match_crosscheck.argtypes = [vp, i32, i32, vp, i32, i32, vp]
        L.orc_vocab_create.restype = vp; L.orc_vocab_create.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32, i32]
        L.orc_vocab_destroy.argtypes = [vp]
        L.orc_bow_transform.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.orc_bow_vector.restype = i32; L.orc_bow_vector.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.orc_match_dbow.restype = i32; L.orc_match_dbow.argtypes = [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, i32, f32, vp]
        L.orc_get_3d_point.argtypes = [vp, i32, vp, i32, vp, vp]
        L.orc_assemble_correspondences.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp] + [vp] * 9
        L.orc_track_map_points.restype = i32; L.orc_track_map_points.argtypes = [vp, i32, vp, i32, vp]
        L.orc_in_match_vec.argtypes = [vp, i32, i32, vp]
        L.orc_manage_map_points.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp]
        L.orc_rgb_cloud.argtypes = [vp, i32, vp, i32, i32, i32, f32, f32, f32, f32, C.c_double, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


# ---------------------------------------------------------------- primitives
def cvround(v): return lib().orc_cvround(float(v))


def rgb2gray(rgb):
    rgb = _u8(rgb); h, w, _ = rgb.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_rgb2gray(_p(rgb), w, h, rgb.strides[0], _p(out), w)
    return out


def resize_linear(src, dw, dh):
    src = _u8(src); sh, sw = src.shape
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear(_p(src), sw, sh, src.strides[0], _p(out), dw, dh, dw)
    return out


def border_reflect101(img, pad):
    img = _u8(img); h, w = img.shape
    buf = np.zeros((h + 2 * pad, w + 2 * pad), np.uint8)
    buf[pad:pad + h, pad:pad + w] = img
    lib().orc_border_reflect101(_p(buf), w, h, buf.strides[0], pad)
    return buf


def fast9(img, threshold):
    img = _u8(img); h, w = img.shape
    cap = max(16, w * h)
    xs = np.empty(cap, np.int32); ys = np.empty(cap, np.int32); sc = np.empty(cap, np.int32)
    n = lib().orc_fast9(_p(img), w, h, img.strides[0], threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_score_ref(img, x, y, threshold):
    img = _u8(img)
    return lib().orc_fast_score_ref(C.c_void_p(img.ctypes.data + y * img.strides[0] + x), img.strides[0], threshold)


def gaussian7(img):
    img = _u8(img); h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gaussian7(_p(img), w, h, img.strides[0], _p(out), w)
    return out


def gauss_kernel_fixed():
    k = np.zeros(7, np.int32); lib().orc_gauss_kernel_fixed(_p(k)); return k


def fast_atan2(y, x): return lib().orc_fast_atan2(float(y), float(x))


def sincosf(x):
    s = C.c_float(); c = C.c_float()
    lib().orc_sincosf(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def cell_grid(w, h):
    g = _Grid()
    rc = lib().orc_cell_grid(w, h, C.byref(g))
    return None if rc else {n: getattr(g, n) for n, _ in _Grid._fields_}


def distribute_octtree(xs, ys, resp, minX, maxX, minY, maxY, N):
    xs = np.ascontiguousarray(xs, np.float32); ys = np.ascontiguousarray(ys, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    n = len(xs); cap = n + 8
    out = np.empty(cap, np.int32); tie = C.c_int(0)
    m = lib().orc_distribute_octtree(_p(xs), _p(ys), _p(resp), n, minX, maxX, minY, maxY, N, _p(out), cap, C.byref(tie))
    if m < 0:
        raise ValueError("degenerate quadtree geometry")
    return out[:m].copy(), bool(tie.value)


def ic_angle(img, x, y):
    img = _u8(img)
    m01 = C.c_int(); m10 = C.c_int()
    a = lib().orc_ic_angle(_p(img), img.strides[0], x, y, C.byref(m01), C.byref(m10))
    return a, m01.value, m10.value


def orb_descriptor(blurred, x, y, angle_deg, libm=False):
    blurred = _u8(blurred)
    d = np.empty(32, np.uint8)
    fn = lib().orc_orb_descriptor_libm if libm else lib().orc_orb_descriptor
    fn(_p(blurred), blurred.strides[0], x, y, float(angle_deg), _p(d))
    return d


# ---------------------------------------------------------------- extractor
class Extractor:
    """Mirror of ORBextractor (include/Feature.h:32-103) on the oracle."""

    def __init__(self, nfeatures=1200, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._h = lib().orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self._h:
            raise ValueError("bad extractor parameters")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h); self._h = None

    def _arr(self, fn, ctype, n):
        p = getattr(lib(), fn)(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n,)).copy()

    def scale_factors(self): return self._arr("orc_scale_factors", C.c_float, self.nlevels)
    def inv_scale_factors(self): return self._arr("orc_inv_scale_factors", C.c_float, self.nlevels)
    def level_sigma2(self): return self._arr("orc_level_sigma2", C.c_float, self.nlevels)
    def inv_level_sigma2(self): return self._arr("orc_inv_level_sigma2", C.c_float, self.nlevels)
    def features_per_level(self): return self._arr("orc_features_per_level", C.c_int, self.nlevels)
    def umax(self): return self._arr("orc_umax", C.c_int, 16)

    def extract(self, gray):
        gray = _u8(gray); h, w = gray.shape
        cap = self.nfeatures + 40 * self.nlevels + 64      # a level's first quadtree round yields up to 4 nodes per root whatever its quota
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_extract(self._h, _p(gray), w, h, gray.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise ValueError("image too small / degenerate for this pyramid")
        return kps[:n].copy(), desc[:n].copy()

    def extract_rgb(self, rgb):
        rgb = _u8(rgb); h, w, _ = rgb.shape
        cap = self.nfeatures + 40 * self.nlevels + 64      # a level's first quadtree round yields up to 4 nodes per root whatever its quota
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_extract_rgb(self._h, _p(rgb), w, h, rgb.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise ValueError("image too small / degenerate for this pyramid")
        return kps[:n].copy(), desc[:n].copy()

    # intermediates of the last extract()
    def level_size(self, l):
        w = C.c_int(); h = C.c_int()
        if lib().orc_level_size(self._h, l, C.byref(w), C.byref(h)):
            raise IndexError(l)
        return w.value, h.value

    def _img(self, fn, l):
        w, h = self.level_size(l)
        st = C.c_int()
        p = getattr(lib(), fn)(self._h, l, C.byref(st))
        if not p:
            return None
        buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=((h - 1) * st.value + w,))
        return np.lib.stride_tricks.as_strided(buf, (h, w), (st.value, 1)).copy()

    def level_image(self, l): return self._img("orc_level_roi", l)
    def level_blur(self, l): return self._img("orc_level_blur", l)

    def level_candidates(self, l):
        xs = C.c_void_p(); ys = C.c_void_p(); rs = C.c_void_p()
        n = lib().orc_level_candidates(self._h, l, C.byref(xs), C.byref(ys), C.byref(rs))
        f = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).copy() if n else np.zeros(0, np.float32)
        return f(xs), f(ys), f(rs)

    def level_keypoints(self, l):
        p = C.c_void_p()
        n = lib().orc_level_keypoints(self._h, l, C.byref(p))
        if n == 0:
            return np.zeros(0, KP_DTYPE)
        raw = C.string_at(p, n * 28)
        return np.frombuffer(raw, KP_DTYPE).copy()

    def level_tie_at_cut(self, l): return bool(lib().orc_level_tie_at_cut(self._h, l))


# ---------------------------------------------------------------- matching
def _desc(a):
    a = _u8(a)
    assert a.ndim == 2 and a.shape[1] == 32
    return a


def hamming256(a, b):
    a = _u8(a); b = _u8(b)
    return lib().orc_hamming256(_p(a), _p(b))


def knn2_raw(q, t):
    q = _desc(q); t = _desc(t)
    idx = np.empty(len(q), np.int32); d1 = np.empty(len(q), np.int32); d2 = np.empty(len(q), np.int32)
    lib().orc_knn2_raw(_p(q), len(q), 32, _p(t), len(t), 32, _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


def bf_nn(q, t):
    q = _desc(q); t = _desc(t)
    idx = np.empty(len(q), np.int32); d = np.empty(len(q), np.int32)
    lib().orc_bf_nn(_p(q), len(q), 32, _p(t), len(t), 32, _p(idx), _p(d))
    return idx, d


def match_bruteforce(q, t):
    q = _desc(q); t = _desc(t)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    n = lib().orc_match_bruteforce(_p(q), len(q), 32, _p(t), len(t), 32, _p(out))
    return out[:n].copy()


def match_knn2_ratio(q, t, threshold=50, nn_ratio=0.6):
    q = _desc(q); t = _desc(t)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    n = lib().orc_match_knn2_ratio(_p(q), len(q), 32, _p(t), len(t), 32, threshold, nn_ratio, _p(out))
    return out[:n].copy()


def match_project(p3d, q, p2d, t, radian, K, R, tr, threshold=50, nn_ratio=0.6):
    """src/FeatureMatching.cpp:128-189"""
    q = _desc(q); t = _desc(t)
    p3d = np.ascontiguousarray(p3d, np.float64).reshape(-1, 3); p2d = np.ascontiguousarray(p2d, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9); R = np.ascontiguousarray(R, np.float64).reshape(9)
    tr = np.ascontiguousarray(tr, np.float64).reshape(3)
    assert len(p3d) == len(q) and len(p2d) == len(t)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    f = lib().orc_match_project
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    n = f(_p(p3d), _p(q), len(q), 32, _p(p2d), _p(t), len(t), 32, float(radian), _p(K), _p(R), _p(tr), threshold, nn_ratio, _p(out))
    return out[:n].copy()


def match_crosscheck(q, t):
    q = _desc(q); t = _desc(t)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    n = lib().orc_match_crosscheck(_p(q), len(q), 32, _p(t), len(t), 32, _p(out))
    return out[:n].copy()


def match_crosscheck_cv(q, t):
    """BFMatcher(NORM_HAMMING, crossCheck=True) the way OpenCV 3.x runs it (reverse search only)"""
    q = _desc(q); t = _desc(t)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    n = lib().orc_match_crosscheck_cv(_p(q), len(q), 32, _p(t), len(t), 32, _p(out))
    return out[:n].copy()


# ---------------------------------------------------------------- MatchByDBoW (src/FeatureMatching.cpp:58-125, :192-208)
class Vocabulary:
    """DBoW3::Vocabulary as a node table (bow_oracle.c): parent[nnodes], desc[nnodes][32], weight[nnodes], word[nnodes]"""

    def __init__(self, k, L, parent, desc, weight, word, scoring=0, weighting=0):
        self.k, self.L, self.scoring, self.weighting = k, L, scoring, weighting
        self.parent = np.ascontiguousarray(parent, np.int32); self.desc = _desc(desc)
        self.weight = np.ascontiguousarray(weight, np.float64); self.word = np.ascontiguousarray(word, np.int32)
        n = len(self.parent)
        assert len(self.desc) == n and len(self.weight) == n and len(self.word) == n
        self._h = lib().orc_vocab_create(k, L, n, _p(self.parent), _p(self.desc), _p(self.weight), _p(self.word), scoring, weighting)
        if not self._h:
            raise ValueError("bad vocabulary")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_vocab_destroy(self._h); self._h = None

    def transform(self, desc, levelsup=4):
        """per feature: (word id, node id at level L - levelsup, weight)"""
        desc = _desc(desc); n = len(desc)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        lib().orc_bow_transform(self._h, _p(desc), n, 32, levelsup, _p(word), _p(node), _p(weight))
        return word, node, weight

    def bow_vector(self, word, weight):
        word = np.ascontiguousarray(word, np.int32); weight = np.ascontiguousarray(weight, np.float64)
        ow = np.zeros(max(len(word), 1), np.int32); ov = np.zeros(max(len(word), 1), np.float64)
        m = lib().orc_bow_vector(_p(word), _p(weight), len(word), self.scoring, self.weighting, _p(ow), _p(ov))
        return ow[:m].copy(), ov[:m].copy()


def match_dbow(voc, q, t, threshold=50, nn_ratio=0.6, levelsup=4):
    q = _desc(q); t = _desc(t)
    if not (len(q) > 0 and len(t) > 0):
        return np.zeros(0, DM_DTYPE)
    _, qn, qw = voc.transform(q, levelsup); _, tn, tw = voc.transform(t, levelsup)
    out = np.zeros(max(len(q), 1), DM_DTYPE)
    n = lib().orc_match_dbow(_p(q), len(q), 32, _p(qn), _p(qw), _p(t), len(t), 32, _p(tn), _p(tw), threshold, nn_ratio, _p(out))
    return out[:n].copy()


# ---------------------------------------------------------------- behind the matcher
def camera(cx, cy, fx, fy, depth_scale_yaml):
    """{InnerCx, InnerCy, InnerFx, InnerFy, DepthScale/65535.0 narrowed to float} (include/Init.h:14-20)"""
    return np.array([cx, cy, fx, fy, np.float32(np.float64(np.float32(depth_scale_yaml)) / 65535.0)], np.float32)


def get_3d_points(kps, depth, cam):
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16)
    cam = np.ascontiguousarray(cam, np.float32)
    out = np.zeros((len(kps), 3), np.float32)
    for i in range(len(kps)):
        lib().orc_get_3d_point(_p(kps), i, _p(depth), depth.shape[1], _p(cam), C.c_void_p(out.ctypes.data + 12 * i))
    return out


def assemble_correspondences(matches, last_kps, last_depth, cur_kps, cur_depth, cam):
    m = np.ascontiguousarray(matches, DM_DTYPE); nm = len(m)
    lk = np.ascontiguousarray(last_kps, KP_DTYPE); ck = np.ascontiguousarray(cur_kps, KP_DTYPE)
    ld = np.ascontiguousarray(last_depth, np.uint16); cd = np.ascontiguousarray(cur_depth, np.uint16)
    assert ld.shape == cd.shape
    cam = np.ascontiguousarray(cam, np.float32)
    cap = max(nm, 1)
    last3d = np.zeros((cap, 3), np.float32); cur2d = np.zeros((cap, 2), np.float32); cur3d = np.zeros((cap, 3), np.float32)
    pre = np.zeros(cap, DM_DTYPE); el = np.zeros((cap, 3), np.float32); ec = np.zeros((cap, 3), np.float32); fin = np.zeros(cap, DM_DTYPE)
    np_ = C.c_int(0); nf = C.c_int(0)
    lib().orc_assemble_correspondences(_p(m), nm, _p(lk), _p(ld), _p(ck), _p(cd), ld.shape[1], _p(cam),
                                       _p(last3d), _p(cur2d), _p(cur3d), _p(pre), C.byref(np_),
                                       _p(el), _p(ec), _p(fin), C.byref(nf))
    a, b = np_.value, nf.value
    return {"last3d": last3d[:a].copy(), "cur2d": cur2d[:a].copy(), "cur3d": cur3d[:a].copy(), "pre_good": pre[:a].copy(),
            "edge_last3d": el[:b].copy(), "edge_cur3d": ec[:b].copy(), "final_good": fin[:b].copy()}


def track_map_points(last, cur):
    last = np.ascontiguousarray(last, DM_DTYPE); cur = np.ascontiguousarray(cur, DM_DTYPE)
    out = np.zeros(max(len(last), 1), DM_DTYPE)
    n = lib().orc_track_map_points(_p(last), len(last), _p(cur), len(cur), _p(out))
    return out[:n].copy()


def in_match_vec(vec, nkp):
    vec = np.ascontiguousarray(vec, DM_DTYPE)
    out = np.zeros(nkp, np.int32)
    lib().orc_in_match_vec(_p(vec), len(vec), nkp, _p(out))
    return out


def manage_map_points(kps, depth, cam, world_trans, last_match_vec, obs):
    """Map::ManageMapPoints (src/Map.cpp:125-149): (match_num, map_point, world) per keypoint; world_trans is the 4x4
    GetAbsPose().inverse() (any array-like, row = matrix row), obs the previous key frame's (map_point_id, point_id) pairs"""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16); cam = np.ascontiguousarray(cam, np.float32)
    T = np.ascontiguousarray(np.asarray(world_trans, np.float32).reshape(4, 4).T)           # Eigen's column-major storage
    v = np.ascontiguousarray(last_match_vec, DM_DTYPE); obs = np.ascontiguousarray(obs, np.int32).reshape(-1, 2)
    n = len(kps)
    match_num = np.zeros(n, np.int32); map_point = np.zeros(n, np.int32); world = np.zeros((n, 3), np.float32)
    lib().orc_manage_map_points(_p(kps), n, _p(depth), depth.shape[1], _p(cam), _p(T), _p(v), len(v), _p(obs), len(obs),
                                _p(match_num), _p(map_point), _p(world))
    return match_num, map_point, world


CLOUD_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
                        ("pad", "u1", 12)])        # pcl::PointXYZRGB, 32 bytes


def rgb_cloud(depth, bgr, cx, cy, fx, fy, depth_scale_yaml):
    """driver/Xtion_Driver.cpp:219-229: w*h points, index u*rows + v"""
    depth = np.ascontiguousarray(depth, np.uint16); bgr = _u8(bgr)
    h, w = depth.shape
    out = np.zeros(w * h, CLOUD_DTYPE)
    lib().orc_rgb_cloud(_p(depth), w, _p(bgr), bgr.strides[0], w, h, cx, cy, fx, fy, float(np.float32(depth_scale_yaml)) / 65535.0, _p(out))
    return out


def voxel_fuse(final_cloud, frame_cloud, this_trans, voxel_size):
    """one key frame of Map::MapPointCloudFusion (src/Map.cpp:164-197): transform the frame's cloud by
    Affine3f::Identity().rotate(R).translate(t), append it to the map, VoxelGrid-filter the whole -> the new map.
    this_trans: 4x4 (row-major numpy) GetAbsPose().inverse()"""
    fin = np.ascontiguousarray(final_cloud, CLOUD_DTYPE); frm = np.ascontiguousarray(frame_cloud, CLOUD_DTYPE)
    T = np.ascontiguousarray(np.asarray(this_trans, np.float32).reshape(4, 4).T)        # Eigen::Matrix4f storage order
    out = np.zeros(len(fin) + len(frm), CLOUD_DTYPE)
    f = lib().orc_voxel_fuse
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
    n = f(_p(fin), len(fin), _p(frm), len(frm), _p(T), float(voxel_size), _p(out))
    return out[:n].copy()
