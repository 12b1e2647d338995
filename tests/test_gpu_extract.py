"""GPU parity: ORB extraction through the C ABI vs the CPU oracle, stage by stage and end to end.
Bar: bit-exact for every integer output (pyramid bytes, FAST candidates, blurred bytes, keypoint
coordinates/levels/responses, descriptors) and -- because both sides run the same IEEE sequence --
bit-exact float angles as well (the north-star tolerance for angles is 1e-4 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import ssvio_amd
    from ssvio_amd import synth
    from oracle import oracle as O
    return ssvio_amd, synth, O


def _cmp_frame(O, ex_gpu, ex_cpu, img, kps, desc):
    rk, rd = ex_cpu.extract(img)
    assert len(kps) == len(rk)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], rk[f]), f
    assert np.array_equal(kps["angle"].view(np.uint32), rk["angle"].view(np.uint32)), "angle bits"
    assert np.array_equal(desc, rd)
    return rk, rd


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (640, 480, 500), (320, 240, 1200), (601, 397, 700)])
def test_stages_and_end_to_end(mods, w, h, nf):
    S, synth, O = mods
    img = synth.texture(7 + w, h, w)
    g = S.ORBextractor(nf, 1.2, 8, 20, 7)
    c = O.Extractor(nf, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    assert len(kps) > nf // 2
    for l in range(8):
        assert np.array_equal(g.debug_level_image(l), c.level_image(l)), "pyramid level %d" % l
        xs, ys, rs = g.debug_candidates(l)
        cx, cy, cr = c.level_candidates(l)
        assert np.array_equal(xs, cx.astype(np.int32)) and np.array_equal(ys, cy.astype(np.int32)), "FAST corners level %d" % l
        assert np.array_equal(rs, cr.astype(np.int32)), "FAST scores level %d" % l
        b = c.level_blur(l)
        if b is not None:
            assert np.array_equal(g.debug_level_image(l, blurred=True), b), "blur level %d" % l
    g.close()


@pytest.mark.parametrize("sf,nl,w,h", [(1.5, 5, 640, 480), (2.6, 3, 800, 600), (1.1, 12, 512, 384), (2.0, 3, 640, 480)])
def test_other_scale_factors(mods, sf, nl, w, h):
    """scale factors outside 1.2: the quad resize kernel up to ~2, the generic one beyond"""
    S, synth, O = mods
    img = synth.texture(55, h, w)
    g = S.ORBextractor(800, sf, nl, 20, 7); c = O.Extractor(800, sf, nl, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    for l in range(nl):
        assert np.array_equal(g.debug_level_image(l), c.level_image(l)), "pyramid level %d" % l
    g.close()


def test_tables_match_oracle(mods):
    S, synth, O = mods
    for nf, sf, nl in [(1200, 1.2, 8), (500, 1.2, 8), (4000, 1.2, 8), (1000, 1.5, 5), (777, 1.1, 12)]:
        g = S.ORBextractor(nf, sf, nl, 20, 7); c = O.Extractor(nf, sf, nl, 20, 7)
        assert np.array_equal(g.GetScaleFactors(), c.scale_factors())
        assert np.array_equal(g.GetInvScaleFactor(), c.inv_scale_factors())
        assert np.array_equal(g.GetLevelSigmaSquares(), c.level_sigma2())
        assert np.array_equal(g.GetInvLevelSigmaSquares(), c.inv_level_sigma2())
        assert np.array_equal(g.features_per_level(), c.features_per_level())
        assert g.GetLevels() == nl
        g.close()


def test_rgb_frame_ctor(mods):
    S, synth, O = mods
    gray = synth.texture(21, 240, 320)
    rgb = synth.to_rgb(gray, 3)
    g = S.ORBextractor(1200, 1.2, 8, 20, 7); c = O.Extractor(1200, 1.2, 8, 20, 7)
    kps, desc = g.ExtractRGB(rgb)
    rk, rd = c.extract_rgb(rgb)
    assert len(kps) == len(rk) and np.array_equal(desc, rd)
    assert np.array_equal(g.debug_level_image(0), O.rgb2gray(rgb))
    assert np.array_equal(kps.view(np.uint8), rk.view(np.uint8))
    g.close()


def test_batch_equals_singles_and_stream(mods):
    S, synth, O = mods
    frames = synth.frame_stream(5, 6)
    g = S.ORBextractor(1000, 1.2, 8, 20, 7); c = O.Extractor(1000, 1.2, 8, 20, 7)
    kps, desc, n = g.extract_batch(frames)
    for f in range(len(frames)):
        rk, rd = c.extract(frames[f])
        assert n[f] == len(rk)
        assert np.array_equal(kps[f, :n[f]].view(np.uint8), rk.view(np.uint8)), "frame %d keypoints" % f
        assert np.array_equal(desc[f, :n[f]], rd), "frame %d descriptors" % f
    # a second call with a smaller batch and the single-frame entry point reuse the same handle
    k1, d1 = g.Extract(frames[2])
    assert np.array_equal(d1, desc[2, :n[2]])
    g.close()


def test_batch_of_eleven_mixes_both_workgroup_orders(mods):
    """frames 0..7 take the frame -> XCD workgroup order, frames 8..10 the plain one (xcd_frame_block); small frames so
    that the short FAST tiles are used, then a batch large enough for the tall ones"""
    S, synth, O = mods
    for (w, h, nb, nf) in ((320, 240, 11, 500), (640, 480, 19, 1000)):
        frames = synth.frame_stream(77, nb, h, w)
        g = S.ORBextractor(nf, 1.2, 8, 20, 7); c = O.Extractor(nf, 1.2, 8, 20, 7)
        kps, desc, n = g.extract_batch(frames)
        for f in range(nb):
            rk, rd = c.extract(frames[f])
            assert n[f] == len(rk), f
            assert np.array_equal(kps[f, :n[f]].view(np.uint8), rk.view(np.uint8)), "frame %d keypoints" % f
            assert np.array_equal(desc[f, :n[f]], rd), "frame %d descriptors" % f
        g.close()


def test_pipelined_extract_and_match_on_one_stream(mods):
    """ssvio_orb_extract_batch_dev_async + a matcher bound to the extractor's stream: several steps are enqueued with no
    host wait in between; what is fetched afterwards equals the waiting calls' results and the oracle's matches"""
    import torch
    from ssvio_amd import _lib as L
    S, synth, O = mods
    B = 4
    streams = [synth.frame_stream(40 + i, B) for i in range(3)]
    g = S.ORBextractor(1000, 1.2, 8, 20, 7); fm = S.FeatureMatching(None, {})
    fm.set_stream(g.get_stream())
    pairs = np.array([[f, (f - 1) % B] for f in range(B)], np.int32)
    d = [torch.from_numpy(fr).cuda() for fr in streams]
    for i in range(3):                                   # three steps in flight; the last one's results stay
        g.extract_batch_dev(d[i], B, 640, 480, wait=False)
        _, d_desc, d_n, cap = g.device_results()
        fm.match_batch_dev(d_desc, cap, cap, d_n, pairs, L.MODE_BRUTEFORCE)
    kps, desc, n = g.fetch_results(B)                    # waits by itself
    out, nm = fm.fetch_batch()
    c = O.Extractor(1000, 1.2, 8, 20, 7)
    ref = [c.extract(streams[2][f]) for f in range(B)]
    for f in range(B):
        assert n[f] == len(ref[f][0])
        assert np.array_equal(kps[f, :n[f]].view(np.uint8), ref[f][0].view(np.uint8))
        assert np.array_equal(desc[f, :n[f]], ref[f][1])
    for p, (a, b) in enumerate(pairs):
        want = O.match_bruteforce(ref[a][1], ref[b][1])
        assert nm[p] == len(want)
        assert np.array_equal(out[p, :nm[p]].view(np.uint8), want.view(np.uint8))
    # the waiting form on the same handle afterwards
    g.extract_batch_dev(d[0], B, 640, 480)
    assert np.array_equal(g.last_counts(B), [len(c.extract(streams[0][f])[0]) for f in range(B)])
    g.close(); fm.close()


def test_low_texture_min_threshold_fallback(mods):
    """cells whose FAST(20) list is empty are retried at 7 (src/Feature.cpp:875-879)"""
    S, synth, O = mods
    img = (synth.texture(33, 480, 640).astype(np.float32) * 0.12 + 100).astype(np.uint8)   # low contrast
    g = S.ORBextractor(1000, 1.2, 8, 20, 7); c = O.Extractor(1000, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    rk, rd = _cmp_frame(O, g, c, img, kps, desc)
    assert (rk["response"] < 20).any(), "fixture must exercise the minThFAST path"
    g.close()


def test_nms_tie_then_min_threshold_retry_known_answer(mods):
    """hand-derived case (tests/test_oracle_kat.py): two adjacent equal-score corners suppress each other, the cell is empty
    at iniTh, the retry at minTh finds the weak corner; with one strong corner there is no retry"""
    S, synth, O = mods
    def spikes(pair):
        img = np.full((100, 100), 100, np.uint8); img[30, 30] = 200; img[40, 40] = 110
        if pair: img[30, 31] = 200
        return img
    g = S.ORBextractor(500, 1.2, 1, 20, 7); c = O.Extractor(500, 1.2, 1, 20, 7)
    for pair, want in ((True, (40.0, 40.0, 9.0)), (False, (30.0, 30.0, 99.0))):
        kps, desc = g.Extract(spikes(pair))
        assert [(float(k["x"]), float(k["y"]), float(k["response"])) for k in kps] == [want]
        rk, rd = c.extract(spikes(pair))
        assert np.array_equal(kps.view(np.uint8), rk.view(np.uint8)) and np.array_equal(desc, rd)
    g.close()


def test_flat_and_empty_inputs(mods):
    S, synth, O = mods
    g = S.ORBextractor(1000, 1.2, 8, 20, 7)
    kps, desc = g.Extract(np.full((480, 640), 77, np.uint8))        # no corners at all -> 0 keypoints, descriptors released
    assert len(kps) == 0 and desc.shape == (0, 32)
    kps, desc = g.Extract(np.zeros((0, 0), np.uint8))               # empty image: silent return (src/Feature.cpp:507)
    assert len(kps) == 0
    with pytest.raises(AssertionError):
        g.Extract(np.zeros((480, 640), np.float32))                 # assert(type == CV_8UC1) (src/Feature.cpp:511)
    with pytest.raises(S.SsvioError) as e:
        g.Extract(np.zeros((100, 100), np.uint8))                   # top levels smaller than one FAST cell
    assert e.value.code == -4
    g.close()


def test_unused_slots_of_the_keypoint_list_never_leak(mods):
    """k_describe takes slots of the per-level selected-keypoint list and fetches their patches before the level counts are in;
    slots beyond a level's count keep whatever an earlier, denser frame left there.  One handle: dense, sparse, flat, sparse batch
    with mixed frames, dense again -- every result against the oracle."""
    S, synth, O = mods
    h, w = 480, 640
    dense = synth.texture(3, h, w)
    sparse = np.full((h, w), 90, np.uint8)
    rng = np.random.default_rng(5)
    for _ in range(40):                                            # a few dozen isolated blobs: most levels stay far below their quota
        y, x = int(rng.integers(40, h - 48)), int(rng.integers(40, w - 48))
        sparse[y:y + 7, x:x + 9] = int(rng.integers(150, 255))
    flat = np.full((h, w), 31, np.uint8)
    g = S.ORBextractor(1000, 1.2, 8, 20, 7); c = O.Extractor(1000, 1.2, 8, 20, 7)
    n_dense = None
    for img in (dense, sparse, flat, sparse, dense):
        kps, desc = g.Extract(img)
        rk, _ = _cmp_frame(O, g, c, img, kps, desc)
        if n_dense is None: n_dense = len(rk)
    assert n_dense > 900
    frames = np.stack([dense, sparse, flat, sparse, dense, flat])
    kps, desc, n = g.extract_batch(frames)
    for f in range(len(frames)):
        rk, rd = c.extract(frames[f])
        assert n[f] == len(rk), "frame %d" % f
        assert np.array_equal(kps[f, :n[f]].view(np.uint8), rk.view(np.uint8)), "frame %d keypoints" % f
        assert np.array_equal(desc[f, :n[f]], rd), "frame %d descriptors" % f
    assert 0 < n[1] < 700 and n[2] == 0
    g.close()


def test_mvImagePyramid_has_reflect101_rim(mods):
    S, synth, O = mods
    img = synth.texture(9, 240, 320)
    g = S.ORBextractor(500, 1.2, 8, 20, 7); c = O.Extractor(500, 1.2, 8, 20, 7)
    g.Extract(img); c.extract(img)
    pyr = g.image_pyramid()
    for l in range(8):
        assert np.array_equal(pyr[l], O.border_reflect101(c.level_image(l), 19))
    g.close()


@pytest.mark.parametrize("shape", [(240, 320), (480, 640), (231, 333)])
def test_exported_pyramid_equals_the_fetched_one(mods, shape):
    """ssvio_orb_set_export_pyramid: the rimmed levels a kernel stores into pinned host memory during the call (what the C++ adapter
    points mvImagePyramid at) against the oracle's levels with their BORDER_REFLECT_101 rim; results of the call unchanged"""
    S, synth, O = mods
    img = synth.texture(19, *shape)
    g = S.ORBextractor(500, 1.2, 8, 20, 7); c = O.Extractor(500, 1.2, 8, 20, 7)
    k0, d0 = g.Extract(img)
    with pytest.raises(S.SsvioError):
        g.exported_pyramid()                                        # not enabled yet
    g.set_export_pyramid(True)
    for rgb in (False, True):
        k1, d1 = g.ExtractRGB(synth.to_rgb(img, 1)) if rgb else g.Extract(img)
        src = img
        if rgb:
            c.extract_rgb(synth.to_rgb(img, 1))
        else:
            c.extract(src)
            assert np.array_equal(k1.view(np.uint8), k0.view(np.uint8)) and np.array_equal(d1, d0)
        pyr = g.exported_pyramid()
        for l in range(8):
            assert np.array_equal(pyr[l], O.border_reflect101(c.level_image(l), 19)), (rgb, l)
    g.set_export_pyramid(False)
    g.Extract(img)
    with pytest.raises(S.SsvioError):
        g.exported_pyramid()
    g.close()


def test_1080p_4000_features(mods):
    S, synth, O = mods
    img = synth.texture(101, 1080, 1920)
    g = S.ORBextractor(4000, 1.2, 8, 20, 7); c = O.Extractor(4000, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    g.close()


@pytest.mark.parametrize("kind,w,h,nf", [("noise", 640, 480, 1000), ("noise", 640, 480, 8000), ("checker", 640, 480, 2000),
                                         ("texture", 1920, 1080, 8000), ("noise", 400, 300, 200), ("texture", 1280, 400, 1500)])
def test_gpu_quadtree_dense_and_large_quotas(mods, kind, w, h, nf):
    """the GPU quadtree (LDS capacities 1024 / 2048 live nodes) on very dense candidate sets, large quotas and a
    two-root aspect ratio; every case against the oracle"""
    S, synth, O = mods
    if kind == "noise":
        img = (synth.Rng(77).bytes(w * h).reshape(h, w) // 2 + synth.texture(3, h, w) // 2).astype(np.uint8)
    elif kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((((xx // 5) + (yy // 5)) % 2) * 140 + 40 + (synth.Rng(5).bytes(w * h).reshape(h, w) % 7)).astype(np.uint8)
    else:
        img = synth.texture(2024, h, w)
    g = S.ORBextractor(nf, 1.2, 8, 20, 7); c = O.Extractor(nf, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    for l in range(8):
        xs, ys, rs = g.debug_candidates(l)
        cx, cy, cr = c.level_candidates(l)
        assert np.array_equal(xs, cx.astype(np.int32)) and np.array_equal(ys, cy.astype(np.int32)) and np.array_equal(rs, cr.astype(np.int32))
    g.close()


def test_host_quadtree_mode_and_oversize_quotas(mods, monkeypatch, capfd):
    """SSVIO_QUADTREE=host selects the library's host stage; quotas beyond the LDS forms (> 2045 per level) run the GPU
    quadtree with its node arrays in HBM (8192 / 32768 nodes); only beyond 32765 per level the host stage takes over, and says so"""
    S, synth, O = mods
    img = synth.texture(808, 480, 640)
    monkeypatch.setenv("SSVIO_QUADTREE", "host")
    g = S.ORBextractor(1000, 1.2, 8, 20, 7); c = O.Extractor(1000, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    g.close()
    monkeypatch.delenv("SSVIO_QUADTREE")
    noise = (synth.Rng(9).bytes(640 * 480).reshape(480, 640) // 2 + img // 2).astype(np.uint8)
    for nf, nl in ((12000, 8), (40000, 8), (9000, 1), (30000, 2)):           # level quotas 2606, 8686, 9000, 16363: HBM node arrays
        g = S.ORBextractor(nf, 1.2, nl, 20, 7); c = O.Extractor(nf, 1.2, nl, 20, 7)
        capfd.readouterr()
        kps, desc = g.Extract(noise)
        assert "host stage" not in capfd.readouterr().err
        _cmp_frame(O, g, c, noise, kps, desc)
        assert len(kps) > 3000
        g.close()
    g = S.ORBextractor(160000, 1.2, 8, 20, 7); c = O.Extractor(160000, 1.2, 8, 20, 7)   # 34 745 at level 0
    kps, desc = g.Extract(noise)
    assert "host stage" in capfd.readouterr().err
    _cmp_frame(O, g, c, noise, kps, desc)
    g.close()


def test_two_threads_with_their_own_handles(mods):
    """handles are independent (SURVEY.md §8b threading): two host threads, each with its own extractor + matcher, run
    concurrently and both get the oracle's answers"""
    import threading
    S, synth, O = mods
    imgs = [synth.texture(900 + i, 240, 320) for i in range(4)]
    c = O.Extractor(800, 1.2, 8, 20, 7)
    ref = [c.extract(im) for im in imgs]
    refm = O.match_bruteforce(ref[1][1], ref[0][1])
    errs = []

    def worker(tid):
        try:
            g = S.ORBextractor(800, 1.2, 8, 20, 7); fm = S.FeatureMatching(None, {})
            for it in range(12):
                i = (tid + it) % 4
                k, d = g.Extract(imgs[i])
                assert np.array_equal(k.view(np.uint8), ref[i][0].view(np.uint8)) and np.array_equal(d, ref[i][1])
                ok, m = fm.MatchByBruteForce(ref[1][1], ref[0][1])
                assert np.array_equal(m.view(np.uint8), refm.view(np.uint8))
            g.close(); fm.close()
        except Exception as e:                                   # noqa: BLE001 -- reported by the main thread
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs


def test_4k_frame_and_size_limit(mods):
    """the largest supported level is 4128 px on a side (12-bit candidate coordinates); a 3840x2160 frame is inside, wider
    frames are refused with SSVIO_ERR_GEOMETRY instead of wrapping"""
    S, synth, O = mods
    rng = synth.Rng(4)
    base = synth.texture(31, 540, 960)
    img = np.ascontiguousarray(np.kron(base, np.ones((4, 4), np.uint8)) // 2 + rng.bytes(2160 * 3840).reshape(2160, 3840) // 2).astype(np.uint8)
    g = S.ORBextractor(5000, 1.2, 8, 20, 7); c = O.Extractor(5000, 1.2, 8, 20, 7)
    kps, desc = g.Extract(img)
    _cmp_frame(O, g, c, img, kps, desc)
    assert len(kps) > 4000
    with pytest.raises(S.SsvioError) as e:
        g.Extract(np.zeros((200, 4200), np.uint8))
    assert e.value.code == -4
    g.close()


def test_hipgraph_replay_gives_the_same_results():
    """SSVIO_GRAPH=1 (read once per process, hence the child process): captured launch sequence, several batch sizes and a
    size change on one handle, against the oracle"""
    import os, subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import ssvio_amd as S
from ssvio_amd import synth
from oracle import oracle as O
g = S.ORBextractor(700, 1.2, 8, 20, 7); c = O.Extractor(700, 1.2, 8, 20, 7)
for (h, w, reps) in ((240, 320, 3), (300, 400, 2), (240, 320, 2)):
    imgs = [synth.texture(50 + i, h, w) for i in range(3)]
    for r in range(reps):
        for im in imgs:
            k, d = g.Extract(im); rk, rd = c.extract(im)
            assert np.array_equal(k.view(np.uint8), rk.view(np.uint8)) and np.array_equal(d, rd)
    ks, ds, ns = g.extract_batch(np.stack(imgs))
    for f, im in enumerate(imgs):
        rk, rd = c.extract(im)
        assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8))
print("graph ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SSVIO_GRAPH="1"), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "graph ok" in r.stdout, r.stderr[-2000:]


def test_graph_replay_switched_on_per_handle(mods):
    """ssvio_orb_set_graph: the same replay for one handle of a process; batches of two sizes, a size change, and back off again"""
    S, synth, O = mods
    g = S.ORBextractor(700, 1.2, 8, 20, 7); c = O.Extractor(700, 1.2, 8, 20, 7)
    g.set_graph(True)
    for (h, w) in ((240, 320), (300, 400), (240, 320)):
        imgs = np.stack([synth.texture(60 + i, h, w) for i in range(4)])
        for n in (4, 2, 4):
            ks, ds, ns = g.extract_batch(imgs[:n])
            for f in range(n):
                rk, rd = c.extract(imgs[f])
                assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd)
    g.set_graph(False)
    k, d = g.Extract(imgs[0]); rk, rd = c.extract(imgs[0])
    assert np.array_equal(k.view(np.uint8), rk.view(np.uint8)) and np.array_equal(d, rd)
    g.close()


def test_copy_path_of_the_host_entry_points_gives_the_same_results():
    """Small host inputs are read from device-mapped memory by default; SSVIO_NO_MAPPED_INPUT=1 (read once per process, hence
    the child) sends the same calls down the copy path that large inputs take.  Both against the oracle, strided ROI included."""
    import os, subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import ssvio_amd as S
from ssvio_amd import synth
from oracle import oracle as O
g = S.ORBextractor(700, 1.2, 8, 20, 7); c = O.Extractor(700, 1.2, 8, 20, 7)
a, b = synth.frame_pair(77, 240, 320)
wide = np.zeros((240, 400), np.uint8); wide[:, 40:360] = a
ka, da = g.Extract(wide[:, 40:360]); rk, rd = c.extract(a)
assert np.array_equal(ka.view(np.uint8), rk.view(np.uint8)) and np.array_equal(da, rd)
kb, db = g.Extract(b)
fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
u8 = lambda x: np.ascontiguousarray(x).view(np.uint8)
assert np.array_equal(u8(fm.MatchByBruteForce(db, da)[1]), u8(O.match_bruteforce(db, da)))
assert np.array_equal(u8(fm.MatchByKnnRatio(db, da)[1]), u8(O.match_knn2_ratio(db, da, 50, 0.6)))
assert np.array_equal(u8(fm.MatchByCrossCheck(db, da)[1]), u8(O.match_crosscheck(db, da)))
wd = np.zeros((len(db), 48), np.uint8); wd[:, :32] = db
assert np.array_equal(u8(fm.MatchByBruteForce(wd[:, :32], da)[1]), u8(O.match_bruteforce(db, da)))
par, desc, w = synth.vocabulary_tables(5, 10, 4, stop_p=0.02)
v = S.Vocabulary(10, 4, par, desc, w); ov = O.Vocabulary(10, 4, par, desc, w, v.word)
fd = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
assert np.array_equal(u8(fd.MatchByDBoW(db, da)[1]), u8(O.match_dbow(ov, db, da, 50, 0.6, 4)))
rw, rn, rwt = ov.transform(db, 4)
assert fd.mFeatVec1_ == S.Vocabulary.feature_vector(rn, rwt)
print("copy path ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SSVIO_NO_MAPPED_INPUT="1"), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "copy path ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("w,h,sf,nl", [(640, 480, 1.2, 8), (333, 257, 1.5, 4), (500, 300, 1.9, 3), (400, 400, 2.0, 3), (317, 203, 1.1, 9)])
def test_two_levels_per_launch_for_small_batches(mods, w, h, sf, nl):
    """small launches build two pyramid levels at once (k_resize2) where the tables allow, larger ones (more frames, bigger levels)
    one level per launch: every level's bytes and the end results equal the oracle's either way (2.0 is the INTER_AREA case,
    never fused)"""
    S, synth, O = mods
    imgs = np.stack([synth.texture(300 + i, h, w) for i in range(3)])
    g = S.ORBextractor(600, sf, nl, 20, 7); c = O.Extractor(600, sf, nl, 20, 7)
    for batch in (1, 2, 3):
        ks, ds, ns = g.extract_batch(imgs[:batch])
        for f in range(batch):
            rk, rd = c.extract(imgs[f])
            assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd)
            if f == batch - 1:
                for l in range(nl):
                    assert np.array_equal(g.debug_level_image(l, frame=f), c.level_image(l)), (batch, l)
    g.close()


@pytest.mark.parametrize("kind,w,h,batch", [("uniform", 320, 240, 1), ("uniform", 640, 480, 9), ("dots", 333, 270, 1), ("checker", 400, 300, 1)])
def test_fast_queue_overflow_paths(mods, kind, w, h, batch):
    """k_fast sizes its LDS work queues for ordinary frames (2048 compass candidates, 1024 corners per 64 x 64 tile); tiles that
    overflow them take the dense loops.  Uniform noise overflows both (about a third of its pixels are corners), a 3-px checker
    overflows the candidate queue only; every FAST candidate, score and the final keypoints must still equal the oracle's.
    Small (one 64 x 32 tile form) and batched (64 x 64 form) launches."""
    import torch
    S, synth, O = mods
    rng = np.random.default_rng(w + h)
    if kind == "uniform":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "dots":
        img = np.full((h, w), 20, np.uint8); img[::4, ::4] = 250; img[2::8, 2::8] = 0
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.where(((xx // 3) + (yy // 3)) % 2 == 0, 230, 25).astype(np.uint8)
        img[rng.random((h, w)) < 0.02] = 128
    g = S.ORBextractor(1500, 1.2, 8, 20, 7); c = O.Extractor(1500, 1.2, 8, 20, 7)
    if batch == 1:
        kps, desc = g.Extract(img)
        _cmp_frame(O, g, c, img, kps, desc)
    else:
        frames = np.stack([np.roll(img, 7 * f, axis=1) for f in range(batch)])
        g.extract_batch_dev(torch.from_numpy(frames).cuda(), batch, w, h)
        kps, desc, n = g.fetch_results(batch)
        for f in (0, batch - 1):
            _cmp_frame(O, g, c, frames[f], kps[f, :n[f]], desc[f, :n[f]])
        img = frames[0]; c.extract(img)
        g.Extract(img)
    for l in range(8):
        xs, ys, rs = g.debug_candidates(l)
        cx, cy, cr = c.level_candidates(l)
        assert np.array_equal(xs, cx.astype(np.int32)) and np.array_equal(ys, cy.astype(np.int32)) and np.array_equal(rs, cr.astype(np.int32)), l
    if kind == "uniform":
        assert len(c.level_candidates(0)[0]) > 0.02 * w * h          # the image really is corner-dense
    g.close()


@pytest.mark.parametrize("w,stride,off", [(333, 336, 0), (333, 352, 16), (640, 640, 0), (640, 656, 4), (97, 112, 0), (100, 112, 16)])
def test_level0_row_copy_forms(mods, w, stride, off):
    """k_level0 copies 16-byte chunks when pointer, strides and pitch allow, dwords when only 4-byte alignment holds, bytes otherwise;
    rows that do not end on a chunk and the reflect-101 pad quads take the single-quad pass.  Device frames with a row stride and a
    start offset, two frames per call: level 0 bytes and the end results equal the oracle's."""
    import torch
    S, synth, O = mods
    h = 200 if w > 100 else 97
    imgs = [synth.texture(900 + w + i, h, w) for i in range(2)]
    buf = np.full((2, h, stride + 32), 0xAB, np.uint8)               # off + w <= stride + 32
    for f in range(2):
        buf[f, :, off:off + w] = imgs[f]
    d = torch.from_numpy(buf).cuda()
    nl = 2 if w <= 100 else 4
    g = S.ORBextractor(300, 1.2, nl, 20, 7); c = O.Extractor(300, 1.2, nl, 20, 7)
    base = d.data_ptr() + off
    g.extract_batch_dev(base, 2, w, h, stride=stride + 32, frame_stride=h * (stride + 32))
    ks, ds, ns = g.fetch_results(2)
    for f in range(2):
        rk, rd = c.extract(imgs[f])
        assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd)
    assert np.array_equal(g.debug_level_image(0, frame=1), c.level_image(0))
    g.close()


def test_matcher_outlives_the_extractor_whose_stream_it_borrowed(mods):
    """ssvio_matcher_set_stream borrows the extractor's stream; destroying the extractor first must leave a matcher that can still
    be destroyed (its destroy waits for the device instead of touching the stream that is gone).  Run a few times: the failure mode
    was a hang in one of four process ends of bench.py."""
    import torch
    S, synth, O = mods
    from ssvio_amd import _lib as L
    frames = torch.from_numpy(synth.frame_stream(5, 4, 240, 320)).cuda()
    pairs = np.array([[f, (f - 1) % 4] for f in range(4)], np.int32)
    for _ in range(6):
        g = S.ORBextractor(500, 1.2, 6, 20, 7); fm = S.FeatureMatching(None, {})
        fm.set_stream(g.get_stream())
        g.extract_batch_dev(frames, 4, 320, 240, wait=False)
        _, d_desc, d_n, cap = g.device_results()
        fm.match_batch_dev(d_desc, cap, cap, d_n, pairs, L.MODE_BRUTEFORCE)
        fm.sync()
        _, n = fm.fetch_batch(counts_only=True)
        assert (n > 50).all()
        g.close()                                                # the stream's owner goes first
        fm.close()
