"""GPU parity: brute-force Hamming matching through the C ABI vs the CPU oracle.  Integer
work: bit-exact indices, distances and DMatch rows, lowest-index tie-breaking."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import ssvio_amd
    from ssvio_amd import synth
    from oracle import oracle as O
    return ssvio_amd, synth, O


@pytest.fixture(scope="module")
def fm(mods):
    S, _, _ = mods
    m = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    yield m
    m.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 300), (300, 1), (255, 257), (500, 500), (1000, 1010), (4000, 4000), (2049, 777)])
def test_all_modes_match_oracle(mods, fm, nq, nt):
    S, synth, O = mods
    q, t = synth.descriptor_sets(nq * 31 + nt, nq, nt)
    idx, d1, d2 = fm.knn2_raw(q, t)
    ri, r1, r2 = O.knn2_raw(q, t)
    assert np.array_equal(idx, ri) and np.array_equal(d1, r1) and np.array_equal(d2, r2)
    ok, m = fm.MatchByBruteForce(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_bruteforce(q, t).view(np.uint8)) and ok == (len(m) > 0)
    ok, m = fm.MatchByKnnRatio(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_knn2_ratio(q, t, 50, 0.6).view(np.uint8))
    ok, m = fm.MatchByCrossCheck(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_crosscheck(q, t).view(np.uint8))
    ok, m = fm.MatchByCrossCheckCV(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_crosscheck_cv(q, t).view(np.uint8)) and ok == (len(m) > 0)


def test_crosscheck_cv_differs_from_mutual_nn_where_it_should(mods, fm):
    """OpenCV's cross-check keeps, per query, the best train among the trains that voted for it"""
    S, synth, O = mods
    def row(nbits):
        bits = np.zeros(256, np.uint8); bits[:nbits] = 1
        return np.packbits(bits, bitorder="little")
    # queries q0 = 0 bits, q1 = 10 bits; trains t0 = 4 bits (nearest to q0: d 4; to q1: d 6), t1 = 30 bits (d 30 / 20 -> votes q1)
    q = np.stack([row(0), row(10)]); t = np.stack([row(4), row(30)])
    ok, cv = fm.MatchByCrossCheckCV(q, t)
    ok, mu = fm.MatchByCrossCheck(q, t)
    assert [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in cv] == [(0, 0, 4.0), (1, 1, 20.0)]     # q1 keeps its only voter
    assert [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in mu] == [(0, 0, 4.0)]                   # q1's nearest (t0) prefers q0
    assert np.array_equal(cv.view(np.uint8), O.match_crosscheck_cv(q, t).view(np.uint8))
    # ties: two identical trains vote for the same query -> the lower train index stays; distance 256 still votes
    q = np.stack([row(0)]); t = np.stack([row(256), row(256), row(256)])
    ok, cv = fm.MatchByCrossCheckCV(q, t)
    assert [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in cv] == [(0, 0, 256.0)]
    assert np.array_equal(cv.view(np.uint8), O.match_crosscheck_cv(q, t).view(np.uint8))


def test_ties_resolve_to_lowest_index(mods, fm):
    S, synth, O = mods
    q, t = synth.descriptor_sets(5, 64, 40)
    t = np.concatenate([t, t, t[:7]])          # every train row appears 2-3 times
    q[:10] = t[5:15]                           # exact matches, duplicated later
    idx, d1, d2 = fm.knn2_raw(q, t)
    ri, r1, r2 = O.knn2_raw(q, t)
    assert np.array_equal(idx, ri) and np.array_equal(d1, r1) and np.array_equal(d2, r2)
    assert (d1[:10] == 0).all() and (d2[:10] == 0).all() and (idx[:10] == np.arange(5, 15)).all()
    for a, b in ((fm.MatchByBruteForce(q, t)[1], O.match_bruteforce(q, t)), (fm.MatchByCrossCheck(q, t)[1], O.match_crosscheck(q, t))):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_distance_256_and_filter_boundaries(mods, fm):
    S, synth, O = mods
    q = np.zeros((4, 32), np.uint8); t = np.full((3, 32), 255, np.uint8)     # every pair at distance 256
    idx, d1, d2 = fm.knn2_raw(q, t)
    assert (idx == -1).all() and (d1 == 256).all() and (d2 == 256).all()      # 'dist < 256' never fires (:94)
    ok, m = fm.MatchByBruteForce(q, t)                                         # BFMatcher still reports index 0
    r = O.match_bruteforce(q, t)
    assert np.array_equal(m.view(np.uint8), r.view(np.uint8)) and (m["trainIdx"] == 0).all() and len(m) == 4
    # filter boundary of src/FeatureMatching.cpp:49: keep dist < max(2*min, 30), strict
    t = np.zeros((1, 32), np.uint8)
    q = np.zeros((4, 32), np.uint8)
    def setbits(row, n):
        bits = np.zeros(256, np.uint8); bits[:n] = 1
        q[row] = np.packbits(bits, bitorder="little")
    setbits(0, 10); setbits(1, 29); setbits(2, 30); setbits(3, 31)             # min = 10 -> limit 30
    ok, m = fm.MatchByBruteForce(q, t)
    assert list(m["queryIdx"]) == [0, 1] and list(m["distance"]) == [10.0, 29.0]
    setbits(0, 16)                                                             # min = 16 -> limit 32
    ok, m = fm.MatchByBruteForce(q, t)
    assert list(m["queryIdx"]) == [0, 1, 2, 3]
    assert np.array_equal(m.view(np.uint8), O.match_bruteforce(q, t).view(np.uint8))


def test_empty_sets_and_strided_rows(mods, fm):
    S, synth, O = mods
    q, t = synth.descriptor_sets(9, 100, 120)
    e = np.zeros((0, 32), np.uint8)
    for a, b in ((e, t), (q, e), (e, e)):
        ok, m = fm.MatchByBruteForce(a, b)
        assert ok is False and len(m) == 0                                      # src/FeatureMatching.cpp:25-27
    big = np.zeros((100, 64), np.uint8); big[:, :32] = q                       # non-continuous Mat rows
    ok, m = fm.MatchByBruteForce(big[:, :32], t)
    assert np.array_equal(m.view(np.uint8), O.match_bruteforce(q, t).view(np.uint8))


def test_threshold_and_ratio_parameters(mods):
    S, synth, O = mods
    q, t = synth.descriptor_sets(77, 600, 650)
    for th, ratio in ((50, 0.6), (30, 0.9), (256, 1.5), (0, 0.5)):
        m = S.FeatureMatching(None, {"Threshold": th, "NNRatio": ratio})
        ok, got = m.MatchByKnnRatio(q, t)
        assert np.array_equal(got.view(np.uint8), O.match_knn2_ratio(q, t, th, ratio).view(np.uint8)), (th, ratio)
        m.close()


def test_match_real_descriptors_frame_pair(mods, fm):
    """BASELINE config 1: frame pair, 500 features, MatchByBruteForce(descB, descA)"""
    S, synth, O = mods
    a, b = synth.frame_pair(3)
    g = S.ORBextractor(500, 1.2, 8, 20, 7)
    ka, da = g.Extract(a); kb, db = g.Extract(b)
    ok, m = fm.MatchByBruteForce(db, da)
    assert ok and np.array_equal(m.view(np.uint8), O.match_bruteforce(db, da).view(np.uint8))
    assert len(m) > 50
    g.close()


def _project_scene(seed, n1, n2, synth):
    rng = np.random.default_rng(seed)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1.0]])
    p2 = np.stack([rng.uniform(0, 640, n2), rng.uniform(0, 480, n2)], 1)
    src = rng.integers(0, n2, n1)
    depth = rng.uniform(0.5, 6.0, n1)
    a, b = 0.02, -0.015
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1.0, 0], [-np.sin(b), 0, np.cos(b)]])
    R = Rz @ Ry; t = np.array([0.03, -0.01, 0.05])
    cam = np.stack([(p2[src, 0] - 319.5) / 525 * depth, (p2[src, 1] - 239.5) / 525 * depth, depth], 1)
    cam += rng.normal(0, 0.004, cam.shape) * depth[:, None]
    P3 = (cam - t) @ R
    dq, dt = synth.descriptor_sets(seed, n1, n2)
    dq = dt[src].copy(); flips = rng.random((n1, 256)) < 0.06
    dq ^= np.packbits(flips, axis=1, bitorder="little")
    return P3, dq, p2, dt, K, R, t


@pytest.mark.parametrize("n1,n2,radian", [(1, 1, 5.0), (300, 500, 8.0), (1000, 1000, 15.0), (4000, 4000, 25.0), (257, 3, 1000.0)])
def test_match_by_project_equals_oracle(mods, fm, n1, n2, radian):
    """MatchByProject (src/FeatureMatching.cpp:128-189): the projection runs the same IEEE double sequence on both sides,
    so the candidate windows, indices and distances are bit-exact"""
    S, synth, O = mods
    P3, dq, p2, dt, K, R, t = _project_scene(n1 * 7 + n2, n1, n2, synth)
    ok, m = fm.MatchByProject(P3, dq, p2, dt, radian, K, R, t)
    want = O.match_project(P3, dq, p2, dt, radian, K, R, t, 50, 0.6)
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8)) and ok == (len(want) > 0)
    if n1 >= 300:
        assert len(want) > n1 // 20                     # the scene is not degenerate


def test_match_by_project_empty_inputs(mods, fm):
    S, synth, O = mods
    P3, dq, p2, dt, K, R, t = _project_scene(5, 10, 20, synth)
    ok, m = fm.MatchByProject(np.zeros((0, 3)), None, p2, dt, 5.0, K, R, t)
    assert not ok and len(m) == 0
    ok, m = fm.MatchByProject(P3, dq, np.zeros((0, 2)), None, 5.0, K, R, t)
    assert not ok and len(m) == 0


def test_full_size_blocks_by_properties(mods, fm):
    """BASELINE config 5 block size (8000 x 8000 descriptors): the oracle checks a sample of query rows, the rest is held
    by size-independent properties (self-match is the identity at distance 0, best <= second, mutual-NN lists are
    symmetric, the BF filter is the stated function of the raw 1-NN)"""
    S, synth, O = mods
    q, t = synth.descriptor_sets(77, 8000, 8000)
    idx, d1, d2 = fm.knn2_raw(q, t)
    pick = np.arange(0, 8000, 125)
    ri, r1, r2 = O.knn2_raw(q[pick], t)
    assert np.array_equal(idx[pick], ri) and np.array_equal(d1[pick], r1) and np.array_equal(d2[pick], r2)
    assert (d1 <= d2).all() and (idx >= 0).all() and (idx < 8000).all()
    si, s1, _ = fm.knn2_raw(t, t)
    assert (s1 == 0).all()
    dup = si != np.arange(8000)                                     # only exact duplicate rows may point to an earlier copy
    assert (si[dup] < np.arange(8000)[dup]).all() and all((t[a] == t[b]).all() for a, b in zip(np.flatnonzero(dup), si[dup]))
    ok, m = fm.MatchByBruteForce(q, t)
    lim = max(2 * int(d1.min()), 30)
    keep = d1 < lim
    assert np.array_equal(m["queryIdx"], np.flatnonzero(keep)) and np.array_equal(m["trainIdx"], idx[keep])
    ok, c = fm.MatchByCrossCheck(q, t)
    ok, cr = fm.MatchByCrossCheck(t, q)
    fwd = set(zip(c["queryIdx"].tolist(), c["trainIdx"].tolist())); rev = set(zip(cr["trainIdx"].tolist(), cr["queryIdx"].tolist()))
    assert fwd == rev and len(fwd) > 4000


def test_count_above_max_rows_is_clamped_and_neighbours_stay_exact(mods):
    """a device count above max_rows must not push reads / writes into the neighbouring set or pair slices:
    the kernels clamp it to max_rows, the other pairs' lists stay bit-identical to the oracle"""
    import torch
    from ssvio_amd import _lib as L
    S, synth, O = mods
    nsets, stride, max_rows = 5, 640, 600
    sets = [synth.descriptor_sets(300 + s, 8, 600)[1][:c] for s, c in enumerate([600, 431, 600, 77, 512])]
    buf = np.zeros((nsets, stride, 32), np.uint8)
    for s, d in enumerate(sets):
        buf[s, :len(d)] = d
    buf[2, 600:] = synth.descriptor_sets(999, 8, 40)[1]            # rows beyond max_rows of set 2: must never be matched
    counts = [len(d) for d in sets]; counts[2] = 5000             # the bad count
    pairs = np.array([[0, 1], [2, 3], [1, 2], [4, 0], [3, 4], [2, 2]], np.int32)
    for mode, ref in ((L.MODE_BRUTEFORCE, O.match_bruteforce), (L.MODE_KNN2_RATIO, lambda a, c: O.match_knn2_ratio(a, c, 50, 0.6)),
                      (L.MODE_CROSSCHECK, O.match_crosscheck), (L.MODE_CROSSCHECK_CV, O.match_crosscheck_cv)):
        fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
        d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor(counts, dtype=torch.int32).cuda()
        fm.match_batch_dev(d_desc, stride, max_rows, d_cnt, pairs, mode)
        fm.sync()
        out, n = fm.fetch_batch()
        for p, (a, c) in enumerate(pairs):
            want = ref(sets[a], sets[c])                          # set 2 counts as its first max_rows rows
            assert n[p] == len(want) and np.array_equal(out[p, :n[p]].view(np.uint8), want.view(np.uint8)), (mode, p)
        fm.close()


def test_failed_allocation_is_reported_and_the_handle_recovers(mods):
    """a failed hipMalloc while a handle grows must not leave the old capacity with a null block"""
    from ssvio_amd import _lib as L
    S, synth, O = mods
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    q, t = synth.descriptor_sets(41, 9000, 9500)                   # > 4096 rows: the copied (not mapped) input path, d_q / d_t grow
    ok, m = fm.MatchByBruteForce(q[:4200], t[:4300])
    want = O.match_bruteforce(q[:4200], t[:4300])
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8))
    for nfail in (1, 2, 3):
        L.lib().ssvio_test_fail_next_allocs(nfail)
        with pytest.raises(S.SsvioError):
            fm.MatchByBruteForce(q, t)                             # larger than before: every buffer has to grow
        L.lib().ssvio_test_fail_next_allocs(0)
        ok, m = fm.MatchByBruteForce(q[:4200], t[:4300])           # fits the OLD capacity: must reallocate, not trust it
        assert np.array_equal(m.view(np.uint8), want.view(np.uint8))
    ok, m = fm.MatchByBruteForce(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_bruteforce(q, t).view(np.uint8))
    fm.close()


def test_failed_pair_list_upload_does_not_leave_a_stale_list_looking_current(mods):
    """list X ok, then a larger list Y whose device copy cannot be allocated, then X again -- the cached host copy of X
    must not make the call skip the upload (the device list is gone); the batch must equal the oracle"""
    import torch
    from ssvio_amd import _lib as L
    S, synth, O = mods
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    q, t = synth.descriptor_sets(43, 600, 600)
    buf = np.stack([q[:300], q[300:], t[:300], t[300:]])
    d = torch.from_numpy(buf).cuda(); c = torch.tensor([300, 299, 300, 17], dtype=torch.int32).cuda()
    X = np.array([[0, 2], [1, 3], [3, 0]], np.int32)
    Y = np.tile(np.array([[0, 1], [2, 3]], np.int32), (4000, 1))      # 8000 pairs: d_pairs has to grow (and so do the result buffers)

    def check():
        fm.match_batch_dev(d, 300, 300, c, X, L.MODE_BRUTEFORCE); fm.sync()
        got, n = fm.fetch_batch()
        cnt = [300, 299, 300, 17]
        for k, (a, b) in enumerate(X):
            w = O.match_bruteforce(buf[a, :cnt[a]], buf[b, :cnt[b]])
            assert n[k] == len(w) and np.array_equal(got[k, :n[k]].view(np.uint8), w.view(np.uint8))
    check()
    L.lib().ssvio_test_fail_next_allocs(1)                            # the first growing allocation of the next call is d_pairs
    with pytest.raises(S.SsvioError):
        fm.match_batch_dev(d, 300, 300, c, Y, L.MODE_BRUTEFORCE)
    L.lib().ssvio_test_fail_next_allocs(0)
    check()
    fm.close()


def test_match_by_project_empty_window_keeps_index_minus_one(mods):
    """a reference point that projects inside the image with no train keypoint in its window keeps best_id = -1 and
    min_dist = 256 (src/FeatureMatching.cpp:160-170); with Threshold > 256 the reference still emits it (:177)"""
    S, synth, O = mods
    P3, dq, p2, dt, K, R, t = _project_scene(77, 200, 300, synth)
    fm = S.FeatureMatching(None, {"Threshold": 300, "NNRatio": 1.5})
    ok, m = fm.MatchByProject(P3, dq, p2, dt, 0.5, K, R, t)          # a 1-px window: most points find nothing
    want = O.match_project(P3, dq, p2, dt, 0.5, K, R, t, 300, 1.5)
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8))
    assert (want["trainIdx"] == -1).any() and (want["trainIdx"] >= 0).any()
    far = np.full_like(dt, 255); near = np.zeros_like(dq)            # every candidate at distance exactly 256
    ok, m = fm.MatchByProject(P3, near, p2, far, 40.0, K, R, t)
    want = O.match_project(P3, near, p2, far, 40.0, K, R, t, 300, 1.5)
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8)) and len(want) > 0 and (want["trainIdx"] == -1).all()
    fm.close()
