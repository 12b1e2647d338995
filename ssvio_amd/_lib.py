"""ctypes loader for libssvio_gpu.so (C ABI: include/ssvio_gpu.h).

There is no CPU fallback: if the HIP library is missing the import fails loudly,
and without a GPU every create() raises SsvioError(SSVIO_ERR_NOGPU).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssvio_gpu.so")

OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_GEOMETRY, ERR_NOGPU = 0, -1, -2, -3, -4, -5
MAX_LEVELS = 16
MODE_BRUTEFORCE, MODE_KNN2_RATIO, MODE_CROSSCHECK, MODE_CROSSCHECK_CV = 0, 1, 2, 4
NOALLOC_SCRATCH_BYTES = 32 << 20
PAIR_ID_SEQUENTIAL, PAIR_ID_TRIANGULAR = 0, 1
KERNEL_VALU, KERNEL_MFMA, KERNEL_MFMA_ALL = 0, 1, 2
REC_DTYPE = np.dtype([("pair", "<i4"), ("n_matches", "<i4"), ("sum_dist", "<i8")])   # ssvio_pair_summary

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])   # cv::KeyPoint
DM_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])  # cv::DMatch


CLOUD_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
                        ("pad", "u1", 12)])        # pcl::PointXYZRGB, 32 bytes


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class CorrespondencesDev(C.Structure):
    """ssvio_correspondences_dev: device pointers of ssvio_assemble_batch_dev's outputs"""
    _fields_ = [(n, C.c_void_p) for n in ("last3d", "cur2d", "cur3d", "pre_good", "edge_last3d", "edge_cur3d", "final_good", "counts")]


class SsvioError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ssvio error %d: %s" % (code, msg))
        self.code = code


# every symbol include/ssvio_gpu.h declares: name -> (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "ssvio_last_error": (C.c_char_p, []),
    "ssvio_version": (_i, []),
    "ssvio_device_count": (_i, [_vp]),
    "ssvio_test_fail_next_allocs": (None, [_i]),
    "ssvio_orb_create": (_i, [_vp, _i, _vp]),
    "ssvio_orb_destroy": (None, [_vp]),
    "ssvio_orb_get_tables": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ssvio_orb_max_keypoints": (_i, [_vp]),
    "ssvio_orb_level_geometry": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssvio_orb_extract": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ssvio_orb_extract_rgb": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ssvio_orb_extract_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _sz, _vp, _vp, _i, _vp]),
    "ssvio_orb_extract_batch_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _sz]),
    "ssvio_orb_extract_batch_dev_async": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _sz]),
    "ssvio_orb_sync": (_i, [_vp]),
    "ssvio_orb_get_stream": (_i, [_vp, _vp]),
    "ssvio_orb_device_results": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ssvio_orb_fetch_results": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "ssvio_orb_last_counts": (_i, [_vp, _i, _vp]),
    "ssvio_orb_pyramid_level": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "ssvio_orb_set_export_pyramid": (_i, [_vp, _i]),
    "ssvio_orb_set_graph": (_i, [_vp, _i]),
    "ssvio_orb_exported_pyramid_level": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ssvio_orb_debug_level_image": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "ssvio_orb_debug_candidates": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ssvio_distribute_octtree": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "ssvio_orb_set_profiling": (_i, [_vp, _i]),
    "ssvio_orb_kernel_times": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "ssvio_match_project": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, C.c_double, _vp, _vp, _vp, _vp, _i, _vp]),
    "ssvio_orb_profile_only": (_i, [_vp, C.c_char_p]),
    "ssvio_matcher_profile_only": (_i, [_vp, C.c_char_p]),
    "ssvio_orb_set_stream": (_i, [_vp, _vp]),
    "ssvio_matcher_create": (_i, [_i, _f, _i, _vp]),
    "ssvio_matcher_destroy": (None, [_vp]),
    "ssvio_matcher_set_stream": (_i, [_vp, _vp]),
    "ssvio_matcher_set_kernel": (_i, [_vp, _i]),
    "ssvio_matcher_get_kernel": (_i, [_vp, _vp]),
    "ssvio_matcher_set_profiling": (_i, [_vp, _i]),
    "ssvio_matcher_kernel_times": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "ssvio_match_bruteforce": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "ssvio_match_knn2_ratio": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "ssvio_match_crosscheck": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "ssvio_match_crosscheck_cv": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "ssvio_match_knn2_raw": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ssvio_match_batch_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "ssvio_match_batch_summary_dev": (_i, [_vp, _i, _i, _vp]),
    "ssvio_loop_candidates_plan": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "ssvio_loop_candidates_dev": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, C.c_int64, _vp]),
    "ssvio_match_batch_dev_noalloc": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "ssvio_matcher_sync": (_i, [_vp]),
    "ssvio_match_batch_fetch": (_i, [_vp, _i, _i, _vp, _vp]),
    "ssvio_vocab_create": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ssvio_vocab_destroy": (None, [_vp]),
    "ssvio_vocab_info": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ssvio_bow_transform": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ssvio_match_dbow": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "ssvio_bow_last_transform": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ssvio_match_dbow_batch_dev": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "ssvio_get_3d_points": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "ssvio_assemble_correspondences": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp] + [_vp] * 9),
    "ssvio_assemble_batch_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _vp, _vp]),
    "ssvio_track_map_points": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "ssvio_in_match_vec": (_i, [_vp, _vp, _i, _i, _vp]),
    "ssvio_manage_map_points": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "ssvio_rgb_cloud": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, C.c_double, _vp]),
    "ssvio_map_set": (_i, [_vp, _vp, _i]),
    "ssvio_map_fuse": (_i, [_vp, _vp, _i, _vp, _f]),
    "ssvio_map_fuse_dev": (_i, [_vp, _vp, _i, _vp, _f]),
    "ssvio_map_fuse_frame": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, C.c_double, _vp, _f]),
    "ssvio_map_fetch": (_i, [_vp, _vp, _i, _vp]),
    "ssvio_map_device": (_i, [_vp, _vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libssvio_gpu.so is not built (%s). Build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C ssvio_amd/csrc`. "
                              "There is no CPU fallback." % LIB_PATH)
        # Load order matters when PyTorch shares the process: its wheel bundles its own HIP + ROCr runtimes, and a process in
        # which the system ROCr (pulled in by libssvio_gpu.so) was loaded first leaves torch with "No HIP GPUs are available"
        # (measured: tools/probe/torch_after_lib.py).  With torch's libraries loaded first both runtimes work.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise SsvioError(rc, (lib().ssvio_last_error() or b"").decode())


def ptr(a):
    """void* of a numpy array or of anything with data_ptr() (torch tensor) or a raw int."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def device_count():
    n = C.c_int(0)
    check(lib().ssvio_device_count(C.byref(n)))
    return n.value


def kernel_times(fn, handle):
    names = (C.c_char_p * 32)(); ms = (C.c_double * 32)(); launches = (C.c_int32 * 32)(); cnt = C.c_int(0)
    check(fn(handle, names, ms, launches, 32, C.byref(cnt)))
    return {names[i].decode(): (ms[i], launches[i]) for i in range(min(cnt.value, 32))}
