"""CPU: the C-ABI library loads and exports exactly what include/ssvio_gpu.h declares, fails loudly
without a GPU, and its host-side logic (quadtree stage) matches the oracle."""
import ctypes
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ssvio_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ssvio_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from ssvio_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert sorted(_lib.SYMBOLS) == names, "python binding table and header disagree"
    assert _lib.lib().ssvio_version() >= 100


def test_struct_layouts_match_opencv_types():
    from ssvio_amd import _lib
    assert _lib.KP_DTYPE.itemsize == 28 and _lib.DM_DTYPE.itemsize == 16     # cv::KeyPoint / cv::DMatch
    assert ctypes.sizeof(_lib.OrbParams) == 20


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ssvio_amd
    assert ssvio_amd.device_count() == 0
    with pytest.raises(ssvio_amd.SsvioError) as e:
        ssvio_amd.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)
    with pytest.raises(ssvio_amd.SsvioError):
        ssvio_amd.FeatureMatching(None, {})


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ssvio_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "ssvio_oracle.h" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read()


@pytest.mark.parametrize("seed,n,N,W,H", [(1, 50, 10, 608, 448), (2, 2000, 217, 608, 448), (3, 5000, 1000, 608, 448),
                                          (4, 3000, 500, 1888, 1048), (5, 1, 10, 100, 100), (6, 300, 400, 300, 300),
                                          (7, 4000, 60, 147, 102), (8, 800, 100, 300, 900)])
def test_host_quadtree_equals_oracle(oracle, seed, n, N, W, H):
    import ssvio_amd
    rng = np.random.default_rng(seed)
    flat = rng.choice(W * H, size=min(n, W * H), replace=False)       # distinct pixels, as FAST+NMS delivers
    flat.sort()
    xs = (flat % W).astype(np.float32); ys = (flat // W).astype(np.float32)
    resp = rng.integers(7, 120, len(flat)).astype(np.float32)
    if W / H < 0.5:
        with pytest.raises(ssvio_amd.SsvioError):
            ssvio_amd.distribute_octtree(xs, ys, resp, 0, W, 0, H, N)
        return
    want, _ = oracle.distribute_octtree(xs, ys, resp, 0, W, 0, H, N)
    got = ssvio_amd.distribute_octtree(xs, ys, resp, 0, W, 0, H, N)
    assert np.array_equal(got, want)
    assert len(set(got.tolist())) == len(got)


def test_host_quadtree_on_clustered_points(oracle):
    """all keys in one corner: splits with a single non-empty child must not end the loop early"""
    import ssvio_amd
    rng = np.random.default_rng(11)
    pts = set()
    while len(pts) < 400:
        pts.add((int(rng.integers(0, 40)), int(rng.integers(0, 30))))
    pts = sorted(pts, key=lambda p: (p[1], p[0]))
    xs = np.array([p[0] for p in pts], np.float32); ys = np.array([p[1] for p in pts], np.float32)
    resp = rng.integers(7, 200, len(pts)).astype(np.float32)
    for N in (1, 5, 50, 300, 1000):
        want, _ = oracle.distribute_octtree(xs, ys, resp, 0, 608, 0, 448, N)
        got = ssvio_amd.distribute_octtree(xs, ys, resp, 0, 608, 0, 448, N)
        assert np.array_equal(got, want), N


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """the boundary is a C ABI: include/ssvio_gpu.h compiles as C99 (-pedantic) and a C program links against the library"""
    import subprocess
    src = tmp_path / "c_abi.c"
    src.write_text('#include "ssvio_gpu.h"\n'
                   'int main(void) { ssvio_orb_params p = {1000, 1.2f, 8, 20, 7}; ssvio_orb* h = 0;\n'
                   '  int rc = ssvio_orb_create(&p, 0, &h); if (rc == SSVIO_OK) ssvio_orb_destroy(h);\n'
                   '  return (rc == SSVIO_OK || rc == SSVIO_ERR_NOGPU) && ssvio_last_error() != 0 ? 0 : 1; }\n')
    lib = os.path.join(ROOT, "ssvio_amd")
    exe = tmp_path / "c_abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", lib, "-lssvio_gpu", "-Wl,-rpath," + lib])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr           # without a GPU: SSVIO_ERR_NOGPU and a message, never a crash or a fallback


def test_native_super_block_plan_equals_the_python_deal():
    """ssvio_loop_candidates_plan (host arithmetic of the C ABI, no GPU) against ssvio_amd.dist.deal_super_blocks: every rank's
    number of records, the largest one and the number of super-blocks"""
    import ssvio_amd
    from ssvio_amd import dist as D, _lib as L
    for nsets, sb, world in [(12, 4, 1), (12, 5, 2), (64, 8, 8), (4096, 32, 8), (100, 7, 3), (1, 4, 2), (0, 4, 1), (33, 32, 4), (4096, 64, 8)]:
        plan, load = D.deal_super_blocks(nsets, sb, world)
        for r in range(world):
            assert ssvio_amd.FeatureMatching.loop_candidates_plan(nsets, sb, r, world) == (load[r], max(load), len(plan[r]))
    with pytest.raises(ssvio_amd.SsvioError):
        ssvio_amd.FeatureMatching.loop_candidates_plan(10, 0, 0, 1)
