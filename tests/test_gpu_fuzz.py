"""GPU parity on randomly drawn configurations (fixed seeds): image sizes that put tile, strip and cell borders in
odd places, scale factors on both sides of the streaming resize kernel's range, pyramid depths, quotas and FAST
thresholds.  Bit-exact end to end against the oracle, like test_gpu_extract.py; configurations whose upper levels
are too small for a 30-px cell must be refused on both sides."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("SSVIO_FUZZ_EXTRA", "0"))          # more random cases for a one-off soak (seeded by the count)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        w = int(rng.integers(97, 900)); h = int(rng.integers(97, 700))
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.25, 1.33, 1.5, 1.7, 2.0, 2.3])))
        nl = int(rng.integers(1, 10))
        nf = int(rng.choice([50, 300, 1000, 1200, 2500]))
        ini = int(rng.choice([20, 30, 12])); mn = int(rng.choice([7, 5, 10]))
        out.append((w, h, sf, nl, nf, ini, min(mn, ini), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("w,h,sf,nl,nf,ini,mn,seed", _cases(24, 20261110) + _cases(EXTRA, 777 + EXTRA) + [(880, 130, 1.2, 3, 600, 20, 7, 5), (900, 100, 1.2, 2, 300, 20, 7, 6),
                                                          (130, 420, 1.2, 2, 300, 20, 7, 7),
                                                          # tiny quotas: the first quadtree round yields 4 nodes per root whatever the quota
                                                          (640, 260, 1.2, 8, 12, 20, 7, 8), (640, 200, 1.2, 6, 5, 20, 7, 9), (640, 480, 1.2, 8, 3, 20, 7, 10),
                                                          (800, 150, 1.2, 4, 1, 20, 7, 11)])
def test_random_configuration_equals_oracle(w, h, sf, nl, nf, ini, mn, seed):
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    kind = seed % 3
    img = synth.texture(seed, h, w) if kind else (np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8))
    if kind == 2:
        img = (img // 3 + 80).astype(np.uint8)                      # low contrast: the minTh retry fires in many cells
    g = S.ORBextractor(nf, sf, nl, ini, mn); c = O.Extractor(nf, sf, nl, ini, mn)
    try:
        want = c.extract(img)
    except ValueError:
        with pytest.raises(S.SsvioError):
            g.Extract(img)
        g.close()
        return
    # documented limit: at most 8 quadtree roots on any level (nIni = round(width / height) of the level's tested region,
    # src/Feature.cpp:574; the ratio drifts from level to level with the rounding of the level sizes)
    shapes = [c.level_image(l).shape for l in range(nl)]
    if any(int(np.floor(np.float32(lw - 32) / np.float32(lh - 32) + np.float32(0.5))) > 8 for lh, lw in shapes):
        with pytest.raises(S.SsvioError, match="quadtree roots"):
            g.Extract(img)
        g.close()
        return
    kps, desc = g.Extract(img)
    rk, rd = want
    assert len(kps) == len(rk)
    assert np.array_equal(kps.view(np.uint8), rk.view(np.uint8)), "keypoints"
    assert np.array_equal(desc, rd), "descriptors"
    # twice on the same handle, then as a batch of 3 with a different frame between: same answer (no state carried over)
    other = np.ascontiguousarray(img[::-1])
    ks, ds, ns = g.extract_batch(np.stack([img, other, img]))
    for f in (0, 2):
        assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd)
    g.close()


def _match_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 3000)), int(rng.integers(1, 3000)), int(rng.integers(0, 1 << 30))) for _ in range(n)]


@pytest.mark.parametrize("nq,nt,seed", _match_cases(16, 99) + _match_cases(EXTRA, 555 + EXTRA))
def test_random_match_shapes_equal_oracle(nq, nt, seed):
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    q, t = synth.descriptor_sets(seed, nq, nt, flip_p=0.05 + (seed % 7) * 0.02)
    if seed % 3 == 0 and nt > 4:
        t[nt // 2:] = t[:nt - nt // 2]                               # duplicated rows: lowest-index ties
    th, ratio = [(50, 0.6), (30, 0.8), (80, 0.95)][seed % 3]
    fm = S.FeatureMatching(None, {"Threshold": th, "NNRatio": ratio})
    fm.set_kernel("mfma_all" if seed % 2 else "valu")               # both 2-NN kernels see half of the cases
    b = lambda a: np.ascontiguousarray(a).view(np.uint8)
    idx, d1, d2 = fm.knn2_raw(q, t)
    ri, r1, r2 = O.knn2_raw(q, t)
    assert np.array_equal(idx, ri) and np.array_equal(d1, r1) and np.array_equal(d2, r2)
    assert np.array_equal(b(fm.MatchByBruteForce(q, t)[1]), b(O.match_bruteforce(q, t)))
    assert np.array_equal(b(fm.MatchByKnnRatio(q, t)[1]), b(O.match_knn2_ratio(q, t, th, ratio)))
    assert np.array_equal(b(fm.MatchByCrossCheck(q, t)[1]), b(O.match_crosscheck(q, t)))
    fm.close()


@pytest.mark.parametrize("kernel", ["valu", "mfma"])
@pytest.mark.parametrize("seed,mode", [(1, 0), (2, 1), (3, 2), (4, 0), (5, 4)])
def test_ragged_batches_equal_oracle(seed, mode, kernel):
    """device-resident batch with sets of very different sizes (including empty ones) and repeated / self pairs"""
    import torch
    import ssvio_amd as S
    from ssvio_amd import synth, _lib as L
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    nsets, stride = 9, 1300
    counts = [int(c) for c in rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 700, 1024, 1300], nsets)]
    base = synth.descriptor_sets(seed + 10, 8, 1300)[1]
    sets = []
    for s, c in enumerate(counts):
        d = base[rng.permutation(1300)[:c]] ^ np.packbits(rng.random((c, 256)) < 0.04, axis=1, bitorder="little")
        sets.append(np.ascontiguousarray(d))
    buf = np.zeros((nsets, stride, 32), np.uint8)
    for s, d in enumerate(sets):
        buf[s, :len(d)] = d
    pairs = np.array([[rng.integers(0, nsets), rng.integers(0, nsets)] for _ in range(14)] + [[0, 0], [3, 3]], np.int32)
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    fm.set_kernel(kernel)
    d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    fm.match_batch_dev(d_desc, stride, stride, d_cnt, pairs, mode)
    fm.sync()
    out, n = fm.fetch_batch()
    ref = {0: O.match_bruteforce, 1: lambda a, c: O.match_knn2_ratio(a, c, 50, 0.6), 2: O.match_crosscheck, 4: O.match_crosscheck_cv}[mode]
    for p, (a, c) in enumerate(pairs):
        want = ref(sets[a], sets[c])
        assert n[p] == len(want) and np.array_equal(out[p, :n[p]].view(np.uint8), want.view(np.uint8)), (p, counts[a], counts[c])
    fm.close()


def _dbow_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([2, 3, 5, 10, 17, 40])); L = int(rng.integers(1, 6 if k <= 10 else 3))
        out.append((k, L, int(rng.integers(0, L + 3)), float(rng.choice([0.0, 0.2, 0.6])), int(rng.integers(1, 2500)), int(rng.integers(1, 2500)),
                    int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("k,L,levelsup,early,nq,nt,seed", _dbow_cases(12, 4242) + _dbow_cases(EXTRA // 3, 4243 + EXTRA))
def test_random_dbow_configurations_equal_oracle(k, L, levelsup, early, nq, nt, seed):
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    par, desc, w = synth.vocabulary_tables(seed % 100000, k, L, early_leaf_p=early, stop_p=[0.0, 0.05, 0.5][seed % 3])
    if len(par) > 200000:
        pytest.skip("vocabulary too large for the python generator")
    v = S.Vocabulary(k, L, par, desc, w); ov = O.Vocabulary(k, L, par, desc, w, v.word)
    rng = np.random.default_rng(seed)
    t = synth.descriptors_near_leaves(seed + 1, nt, par, desc, 0.03)
    q = t[rng.integers(0, nt, nq)] ^ np.packbits(rng.random((nq, 256)) < 0.03, axis=1, bitorder="little")
    th, ratio = [(50, 0.6), (30, 0.8), (100, 0.99)][seed % 3]
    fm = S.FeatureMatching(v, {"Threshold": th, "NNRatio": ratio})
    word, node, weight = fm.ComputeBoWVector(q, levelsup)[2]
    rw, rn, rwt = ov.transform(q, levelsup)
    assert np.array_equal(word, rw) and np.array_equal(node, rn) and np.array_equal(weight, rwt)
    ok, m = fm.MatchByDBoW(q, t, levelsup=levelsup)
    want = O.match_dbow(ov, q, t, th, ratio, levelsup)
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8))
    # behind the matcher, on the same match list with made-up keypoints and depth
    kq = np.zeros(nq, S.KP_DTYPE); kt = np.zeros(nt, S.KP_DTYPE)
    kq["x"] = rng.random(nq) * 319; kq["y"] = rng.random(nq) * 239; kt["x"] = rng.random(nt) * 319; kt["y"] = rng.random(nt) * 239
    dq, dt = synth.depth_image(seed % 1000, 240, 320, hole_p=0.3), synth.depth_image(seed % 1000 + 1, 240, 320, hole_p=0.3)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    got = fm.AssembleCorrespondences(m, kt, dt, kq, dq, cam)
    ref = O.assemble_correspondences(m, kt, dt, kq, dq, cam)
    for key in ref:
        assert np.array_equal(np.ascontiguousarray(got[key]).view(np.uint8), np.ascontiguousarray(ref[key]).view(np.uint8)), key
    last = np.zeros(min(nq, 700), S.DM_DTYPE); last["queryIdx"] = rng.integers(0, nt, len(last)); last["trainIdx"] = rng.integers(0, nt, len(last))
    assert np.array_equal(fm.TrackMapPoints(last, m).view(np.uint8), O.track_map_points(last, m).view(np.uint8))
    assert np.array_equal(fm.InMatchVec(m, nq), O.in_match_vec(m, nq))
    trk = O.track_map_points(last, m)
    obs = np.stack([rng.integers(0, 9999, nt // 2 + 1), rng.integers(0, nq + 3, nt // 2 + 1)], axis=1).astype(np.int32)
    T = np.eye(4); T[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]; T[:3, 3] = rng.normal(size=3)
    for g_, w_ in zip(fm.ManageMapPoints(kq, dq, cam, T, trk, obs), O.manage_map_points(kq, dq, cam, T, trk, obs)):
        assert np.array_equal(np.ascontiguousarray(g_).view(np.uint8), np.ascontiguousarray(w_).view(np.uint8))
    fm.close(); v.close()


def test_one_matcher_handle_across_changing_shapes_and_modes():
    """A single FeatureMatching handle reused for 40 batches of changing set sizes, pair counts, strides and modes (buffers grow, are
    reused with other chunkings, hold the previous call's lists): every result equals the oracle's -- nothing read in one call may be
    a leftover of an earlier one.  Single-pair host calls are mixed in (they share the handle's buffers)."""
    import torch
    import ssvio_amd as S
    from ssvio_amd import synth, _lib as L
    from oracle import oracle as O
    rng = np.random.default_rng(2024)
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    base = synth.descriptor_sets(77, 8, 3000)[1]
    refs = {0: O.match_bruteforce, 1: lambda a, c: O.match_knn2_ratio(a, c, 50, 0.6), 2: O.match_crosscheck, 4: O.match_crosscheck_cv}
    for it in range(40):
        mode = int(rng.choice([0, 1, 2, 4]))
        fm.set_kernel(["valu", "mfma", "mfma_all"][it % 3])      # the kernels alternate on the same handle and buffers
        nsets = int(rng.integers(2, 7))
        stride = int(rng.choice([64, 300, 1032, 2100, 3000]))
        counts = [int(min(stride, c)) for c in rng.choice([0, 1, 5, 63, 64, 65, 200, 513, 1006, 2049, 3000], nsets)]
        sets = [np.ascontiguousarray(base[rng.permutation(3000)[:c]] ^ np.packbits(rng.random((c, 256)) < 0.04, axis=1, bitorder="little")) for c in counts]
        buf = np.zeros((nsets, stride, 32), np.uint8)
        for s, d in enumerate(sets):
            buf[s, :len(d)] = d
        pairs = np.array([[rng.integers(0, nsets), rng.integers(0, nsets)] for _ in range(int(rng.integers(1, 9)))], np.int32)
        fm.match_batch_dev(torch.from_numpy(buf).cuda(), stride, stride, torch.tensor(counts, dtype=torch.int32).cuda(), pairs, mode)
        fm.sync()
        out, n = fm.fetch_batch()
        for p, (a, c) in enumerate(pairs):
            want = refs[mode](sets[a], sets[c])
            assert n[p] == len(want) and np.array_equal(out[p, :n[p]].view(np.uint8), want.view(np.uint8)), (it, mode, p, counts[a], counts[c])
        if it % 5 == 0:                                           # a host single-pair call in between
            a, c = sets[0], sets[-1]
            got = fm.MatchByBruteForce(a, c)[1] if len(a) and len(c) else None
            if got is not None:
                assert np.array_equal(np.ascontiguousarray(got).view(np.uint8), O.match_bruteforce(a, c).view(np.uint8)), it
    fm.close()


def test_one_extractor_handle_across_changing_sizes_and_batches():
    """A single ORBextractor handle reused for frames of changing size and batch count (geometry tables rebuilt, buffers regrown and
    reused): every frame equals the oracle's result."""
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(99)
    g = S.ORBextractor(600, 1.2, 6, 20, 7); c = O.Extractor(600, 1.2, 6, 20, 7)
    sizes = [(320, 240), (640, 480), (203, 151), (640, 480), (800, 600), (256, 256), (320, 240), (1024, 300)]
    for it, (w, h) in enumerate(sizes):
        batch = int(rng.integers(1, 5))
        imgs = np.stack([synth.texture(500 + 10 * it + f, h, w) for f in range(batch)])
        if (w, h) == (203, 151):                                  # too small for six levels: an error, and the handle stays usable
            with pytest.raises(S.SsvioError):
                g.extract_batch(imgs)
            continue
        ks, ds, ns = g.extract_batch(imgs)
        for f in range(batch):
            rk, rd = c.extract(imgs[f])
            assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd), (it, w, h, f)
    g.close()


def test_one_dbow_handle_across_changing_shapes():
    """One vocabulary-bound FeatureMatching handle reused for 14 MatchByDBoW calls of changing set sizes, with brute-force calls and
    the steps behind the matcher mixed in on the same handle (the sorted views, candidate and staging buffers are reused)."""
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    par, desc, w = synth.vocabulary_tables(31, 10, 4, stop_p=0.02)
    v = S.Vocabulary(10, 4, par, desc, w); ov = O.Vocabulary(10, 4, par, desc, w, v.word)
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    rng = np.random.default_rng(7)
    u8 = lambda x: np.ascontiguousarray(x).view(np.uint8)
    for it, (nq, nt) in enumerate([(900, 1000), (30, 2000), (2400, 50), (1, 1), (1000, 1000), (9000, 300), (64, 64), (500, 8300), (1000, 999), (3, 700), (2049, 2047), (100, 100), (1200, 1), (777, 1555)]):
        t = synth.descriptors_near_leaves(100 + it, nt, par, desc, 0.03)
        q = t[rng.integers(0, nt, nq)] ^ np.packbits(rng.random((nq, 256)) < 0.03, axis=1, bitorder="little")
        ok, m = fm.MatchByDBoW(q, t, levelsup=2 + it % 3)
        assert np.array_equal(u8(m), u8(O.match_dbow(ov, q, t, 50, 0.6, 2 + it % 3))), (it, nq, nt)
        if it % 3 == 0:
            assert np.array_equal(u8(fm.MatchByBruteForce(q, t)[1]), u8(O.match_bruteforce(q, t))), it
        if it % 4 == 1:
            last = np.zeros(min(nq, 300), S.DM_DTYPE); last["queryIdx"] = rng.integers(0, nt, len(last)); last["trainIdx"] = rng.integers(0, nt, len(last))
            assert np.array_equal(u8(fm.TrackMapPoints(last, m)), u8(O.track_map_points(last, m))), it
    fm.close(); v.close()
