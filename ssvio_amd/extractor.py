"""ORBextractor -- host-side mirror of SSVIO's class (include/Feature.h:32-103) on the
gfx950 library.  Same constructor arguments, getters and Extract semantics; numpy arrays
stand in for cv::Mat, structured arrays with cv::KeyPoint's memory layout for the
keypoint vector."""
import ctypes as C
import numpy as np
from . import _lib as L


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1          # include/Feature.h:36 (only FAST_SCORE is ever used)

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        """include/Feature.h:38, src/Feature.cpp:379-444"""
        self._h = C.c_void_p()
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        p = L.OrbParams(self.nfeatures, self.scaleFactor, self.nlevels, self.iniThFAST, self.minThFAST)
        L.check(L.lib().ssvio_orb_create(C.byref(p), device, C.byref(self._h)))
        self._cap = L.lib().ssvio_orb_max_keypoints(self._h)
        n = self.nlevels
        self._tabs = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32)]
        L.check(L.lib().ssvio_orb_get_tables(self._h, *[L.ptr(a) for a in self._tabs]))
        self._last = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value and L is not None and L.lib is not None:
            L.lib().ssvio_orb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # ---- getters, include/Feature.h:47-75
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return np.float32(self.scaleFactor)
    def GetScaleFactors(self): return self._tabs[0].copy()
    def GetInvScaleFactor(self): return self._tabs[1].copy()
    def GetLevelSigmaSquares(self): return self._tabs[2].copy()
    def GetInvLevelSigmaSquares(self): return self._tabs[3].copy()
    def features_per_level(self): return self._tabs[4].copy()
    @property
    def max_keypoints(self): return self._cap

    def level_geometry(self, width, height, level):
        v = [C.c_int32() for _ in range(6)]
        L.check(L.lib().ssvio_orb_level_geometry(self._h, width, height, level, *[C.byref(x) for x in v]))
        return dict(zip(("w", "h", "nCols", "nRows", "wCell", "hCell"), [x.value for x in v]))

    # ---- Extract, include/Feature.h:42-43 / src/Feature.cpp:504-568
    def Extract(self, image, mask=None):
        """image: HxW uint8 (gray) -> (keypoints[KP_DTYPE], descriptors Nx32 uint8).
        `mask` is accepted and ignored, as in the reference.  An empty image returns empty
        outputs (the reference returns silently, :507-508); a non-8UC1 image raises like the
        reference's assert (:511)."""
        if image is None or image.size == 0:
            return np.zeros(0, L.KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")
        kps, desc, n = self.extract_batch(image[None])
        return kps[0, :n[0]].copy(), desc[0, :n[0]].copy()

    def ExtractRGB(self, rgb):
        """Frame::Frame front half (src/Frame.cpp:3-9): cvtColor(CV_RGB2GRAY) + Extract"""
        if rgb is None or rgb.size == 0:
            return np.zeros(0, L.KP_DTYPE), np.zeros((0, 32), np.uint8)
        if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
            raise AssertionError("rgb must be HxWx3 uint8")
        kps, desc, n = self.extract_batch(rgb[None])
        return kps[0, :n[0]].copy(), desc[0, :n[0]].copy()

    def extract_batch(self, imgs):
        """imgs: BxHxW (gray) or BxHxWx3 uint8 host array -> (kps[B,cap], desc[B,cap,32], n[B])"""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        ch = 3 if imgs.ndim == 4 else 1
        B, H, W = imgs.shape[:3]
        kps = np.zeros((B, self._cap), L.KP_DTYPE); desc = np.zeros((B, self._cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        L.check(L.lib().ssvio_orb_extract_batch(self._h, L.ptr(imgs), B, ch, W, H, W * ch, H * W * ch,
                                                L.ptr(kps), L.ptr(desc), self._cap, L.ptr(n)))
        self._last = (B, W, H)
        return kps, desc, n

    def extract_batch_dev(self, d_imgs, batch, width, height, channels=1, stride=None, frame_stride=None, wait=True):
        """frames already in HBM (torch uint8 tensor or raw device pointer); results stay on the device.
        wait=False only enqueues the work on the handle's stream (pipelined form, see sync())"""
        stride = stride or width * channels
        frame_stride = frame_stride or stride * height
        fn = L.lib().ssvio_orb_extract_batch_dev if wait else L.lib().ssvio_orb_extract_batch_dev_async
        L.check(fn(self._h, L.ptr(d_imgs), batch, channels, width, height, stride, frame_stride))
        self._last = (batch, width, height)

    def sync(self): L.check(L.lib().ssvio_orb_sync(self._h))
    def profile_only(self, kernel): L.check(L.lib().ssvio_orb_profile_only(self._h, kernel.encode() if kernel else None))

    def get_stream(self):
        s = C.c_void_p()
        L.check(L.lib().ssvio_orb_get_stream(self._h, C.byref(s)))
        return s.value

    def device_results(self):
        kp, de, n, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        L.check(L.lib().ssvio_orb_device_results(self._h, C.byref(kp), C.byref(de), C.byref(n), C.byref(cap)))
        return kp.value, de.value, n.value, cap.value

    def last_counts(self, batch):
        n = np.zeros(batch, np.int32)
        L.check(L.lib().ssvio_orb_last_counts(self._h, batch, L.ptr(n)))
        return n

    def fetch_results(self, batch):
        kps = np.zeros((batch, self._cap), L.KP_DTYPE); desc = np.zeros((batch, self._cap, 32), np.uint8)
        n = np.zeros(batch, np.int32)
        L.check(L.lib().ssvio_orb_fetch_results(self._h, batch, L.ptr(kps), L.ptr(desc), self._cap, L.ptr(n)))
        return kps, desc, n

    # ---- mvImagePyramid, include/Feature.h:45
    def image_pyramid(self, frame=0):
        """list of level images with the 19-px reflect-101 rim, as the reference's padded Mats"""
        out = []
        for l in range(self.nlevels):
            w, h = C.c_int(), C.c_int()
            L.check(L.lib().ssvio_orb_pyramid_level(self._h, frame, l, None, 0, C.byref(w), C.byref(h)))
            buf = np.zeros((h.value + 38, w.value + 38), np.uint8)
            L.check(L.lib().ssvio_orb_pyramid_level(self._h, frame, l, L.ptr(buf), buf.strides[0], C.byref(w), C.byref(h)))
            out.append(buf)
        return out

    # ---- stage taps (tests)
    def set_graph(self, on):
        """replay this handle's launch sequence as a hipGraph (ssvio_orb_set_graph); same results"""
        L.check(L.lib().ssvio_orb_set_graph(self._h, int(on)))

    def set_export_pyramid(self, on):
        """every single-frame Extract / ExtractRGB also stores the rimmed pyramid into pinned host memory (mvImagePyramid, include/Feature.h:45)"""
        L.check(L.lib().ssvio_orb_set_export_pyramid(self._h, int(on)))

    def exported_pyramid(self):
        """the rimmed levels of the last single-frame call as copies of the handle's pinned area: (h + 38) x (w + 38) each"""
        out = []
        for l in range(self.nlevels):
            base = C.c_void_p(); stride, w, h = C.c_int(0), C.c_int(0), C.c_int(0)
            L.check(L.lib().ssvio_orb_exported_pyramid_level(self._h, l, C.byref(base), C.byref(stride), C.byref(w), C.byref(h)))
            rows = h.value + 38
            a = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(rows * stride.value,)).reshape(rows, stride.value)
            out.append(a[:, :w.value + 38].copy())
        return out

    def debug_level_image(self, level, blurred=False, frame=0):
        g = self.level_geometry(self._last[1], self._last[2], level)
        buf = np.zeros((g["h"], g["w"]), np.uint8)
        L.check(L.lib().ssvio_orb_debug_level_image(self._h, frame, level, int(blurred), L.ptr(buf), buf.strides[0]))
        return buf

    def debug_candidates(self, level, frame=0):
        cap = 1 << 20
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); rs = np.zeros(cap, np.int32); n = C.c_int()
        L.check(L.lib().ssvio_orb_debug_candidates(self._h, frame, level, L.ptr(xs), L.ptr(ys), L.ptr(rs), cap, C.byref(n)))
        return xs[:n.value].copy(), ys[:n.value].copy(), rs[:n.value].copy()

    def set_profiling(self, on): L.check(L.lib().ssvio_orb_set_profiling(self._h, int(on)))
    def kernel_times(self): return L.kernel_times(L.lib().ssvio_orb_kernel_times, self._h)
    def set_stream(self, stream): L.check(L.lib().ssvio_orb_set_stream(self._h, L.ptr(stream)))


def distribute_octtree(xs, ys, resp, minX, maxX, minY, maxY, N):
    """DistributeOctTree (src/Feature.cpp:570-798): indices of the kept keys in list order"""
    xs = np.ascontiguousarray(xs, np.float32); ys = np.ascontiguousarray(ys, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    cap = len(xs) + 8
    out = np.zeros(cap, np.int32); n = C.c_int()
    L.check(L.lib().ssvio_distribute_octtree(L.ptr(xs), L.ptr(ys), L.ptr(resp), len(xs), minX, maxX, minY, maxY, N,
                                             L.ptr(out), cap, C.byref(n)))
    return out[:n.value].copy()
