"""GPU-box probe: matcher throughput on BASELINE configs 2/3/4 (descriptor pairs per second, VALU fraction)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth, _lib as L
PEAK = 256 * 4 * 32 * 2.4e9
fm = ssvio_amd.FeatureMatching(None, {})
def run(name, nsets, rows, pairs, mode, iters=20):
    desc = np.concatenate([synth.descriptor_sets(7 + s, rows, rows)[0] for s in range(nsets)])
    d = torch.from_numpy(desc).cuda(); cnt = torch.full((nsets,), rows, dtype=torch.int32).cuda()
    pairs = np.asarray(pairs, np.int32)
    for _ in range(3): fm.match_batch_dev(d, rows, rows, cnt, pairs, mode); fm.sync()
    fm.set_profiling(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fm.match_batch_dev(d, rows, rows, cnt, pairs, mode)
    fm.sync(); dt = (time.perf_counter() - t) / iters
    kt = fm.kernel_times(); fm.set_profiling(False)
    npairs = len(pairs) * rows * rows * (2 if mode == 2 else 1)
    ms, n = kt["k_hamming_knn2"]
    print("%-34s wall %8.1f us  knn2 %8.1f us/launch  %.3e desc-pairs/s (kernel)  valu frac %.3f  finalize %.1f us" % (
        name, dt * 1e6, ms * 1e3 / n, rows * rows * len(pairs) / (ms * 1e-3 / n), 19 * rows * rows * len(pairs) / (ms * 1e-3 / n) / PEAK,
        kt["k_match_finalize"][0] * 1e3 / kt["k_match_finalize"][1]))
run("cfg2 64 pairs 1000x1000 bf", 64, 1000, [[f, (f - 1) % 64] for f in range(64)], 0)
run("cfg3 1 pair 4000x4000 knn2", 2, 4000, [[0, 1]], 1)
run("cfg4 64 pairs 4000x4000 crosscheck", 65, 4000, [[0, k + 1] for k in range(64)], 2)
run("cfg5-like 256 pairs 8000x8000 bf", 24, 8000, [[i, j] for i in range(24) for j in range(i)][:256], 0, iters=5)
# MatchByProject (host buffers in, matches out): kernel time and end-to-end call time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_match import _project_scene
for n1, n2, rad in ((1000, 1000, 15.0), (4000, 4000, 25.0)):
    P3, dq, p2, dt_, K, R, t = _project_scene(3, n1, n2, synth)
    for _ in range(3): fm.MatchByProject(P3, dq, p2, dt_, rad, K, R, t)
    fm.set_profiling(True); t0 = time.perf_counter()
    for _ in range(20): ok, m = fm.MatchByProject(P3, dq, p2, dt_, rad, K, R, t)
    wall = (time.perf_counter() - t0) / 20
    kt = fm.kernel_times(); fm.set_profiling(False)
    print("MatchByProject %dx%d radian %g: %d matches, kernels %.1f us, call (H2D + kernels + D2H) %.1f us" % (n1, n2, rad, len(m), kt["k_match_project"][0] * 1e3 / kt["k_match_project"][1], wall * 1e6))
# host-buffer entry points, as the C++ adapters call them (PCIe-inclusive)
ex = ssvio_amd.ORBextractor(1000, 1.2, 8, 20, 7)
fr = synth.frame_stream(9, 2, 480, 640)
for _ in range(3): ka, da = ex.Extract(fr[0]); kb, db = ex.Extract(fr[1]); fm.MatchByBruteForce(db, da)
t0 = time.perf_counter()
for _ in range(50): ka, da = ex.Extract(fr[0])
t1 = time.perf_counter()
for _ in range(50): fm.MatchByBruteForce(db, da)
t2 = time.perf_counter()
print("host path: Extract(VGA, 1000) %.1f us per call, MatchByBruteForce(%dx%d) %.1f us per call" % ((t1 - t0) / 50 * 1e6, len(db), len(da), (t2 - t1) / 50 * 1e6))
from ssvio_amd.vocabulary import Vocabulary
par, vdesc, vw = synth.vocabulary_tables_full(3, 10, 5)
voc = Vocabulary(10, 5, par, vdesc, vw)
fd = ssvio_amd.FeatureMatching(voc, {"Threshold": 50, "NNRatio": 0.6})
for _ in range(3): fd.MatchByDBoW(db, da)
t0 = time.perf_counter()
for _ in range(50): ok, md = fd.MatchByDBoW(db, da)
t1 = time.perf_counter()
print("host path: MatchByDBoW(%dx%d, vocabulary k=10 L=5, fills mBowVec / mFeatVec) %.1f us per call, %d matches" % (len(db), len(da), (t1 - t0) / 50 * 1e6, len(md)))
import ctypes as C
out = np.zeros(len(db), L.DM_DTYPE); n = C.c_int(0)
call = lambda: L.lib().ssvio_match_dbow(fd._h, voc.handle(0), L.ptr(db), len(db), 32, L.ptr(da), len(da), 32, 4, L.ptr(out), len(out), C.byref(n))
for _ in range(3): call()
t0 = time.perf_counter()
for _ in range(50): call()
t1 = time.perf_counter()
print("host path: ssvio_match_dbow through the C ABI alone %.1f us per call (%d matches)" % ((t1 - t0) / 50 * 1e6, n.value))
# the steps behind the matcher, host buffers in and out (python wrapper overhead included)
from oracle import oracle as O
depth_a, depth_b = synth.depth_image(12, 480, 640), synth.depth_image(13, 480, 640)
cam = O.camera(320.5, 240.5, 500.0, 500.0, 10.0)
ok, mb = fm.MatchByBruteForce(db, da)
def timeit(f, n=30):
    for _ in range(3): f()
    t = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t) / n * 1e6, r
t_as, c = timeit(lambda: fm.AssembleCorrespondences(mb, ka, depth_a, kb, depth_b, cam))
t_tr, trk = timeit(lambda: fm.TrackMapPoints(mb, c["final_good"]))
obs = np.stack([np.arange(len(ka)), np.arange(len(ka))], 1).astype(np.int32)
t_mg, _ = timeit(lambda: fm.ManageMapPoints(kb, depth_b, cam, np.eye(4), trk, obs))
print("host path: AssembleCorrespondences(%d matches, two 640x480 depth planes) %.1f us, TrackMapPoints %.1f us, ManageMapPoints(%d keypoints) %.1f us per call" % (len(mb), t_as, t_tr, len(kb), t_mg))
fd.close(); voc.close()
ex.close()
fm.close()
