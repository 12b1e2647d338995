"""GPU parity through the C ABI vs the CPU oracle for the rows SURVEY.md §8 marks "next":
f1 MatchByDBoW (vocabulary transform + grouped 2-NN, src/FeatureMatching.cpp:58-125, :192-208),
f2 Get3DPoint + correspondence assembly (src/Frame.cpp:93-108, src/Odometry.cpp:44-64, :94-108),
f3 match chaining (src/Map.cpp:100-123, :151-162).
Integer work is bit-exact; the float outputs are compared bit for bit too (same IEEE sequence on both sides)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import ssvio_amd
    from ssvio_amd import synth, vocabulary, _lib
    from oracle import oracle as O
    return ssvio_amd, synth, O, vocabulary, _lib


def _pair(mods, seed, k, L, early=0.0, stop=0.05):
    S, synth, O, V, _ = mods
    par, desc, w = synth.vocabulary_tables(seed, k, L, early_leaf_p=early, stop_p=stop)
    v = V.Vocabulary(k, L, par, desc, w)
    return v, O.Vocabulary(k, L, par, desc, w, v.word), par, desc


def _bytes(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.parametrize("seed,k,L,levelsup,early,n", [(1, 10, 3, 1, 0.0, 1000), (2, 4, 5, 2, 0.3, 700), (3, 10, 4, 4, 0.1, 1200),
                                                        (4, 20, 2, 1, 0.0, 333), (5, 37, 2, 0, 0.0, 64), (6, 2, 9, 4, 0.2, 500),
                                                        (7, 10, 6, 4, 0.85, 400)])
def test_bow_transform_equals_oracle(mods, seed, k, L, levelsup, early, n):
    """word, node at level L - levelsup, weight per feature; k > 16 takes more than one child per lane; early leaves
    leave the node id at 0 (canonical choice for DBoW3's unassigned nid)"""
    S, synth, O, V, _ = mods
    v, ov, par, desc = _pair(mods, seed, k, L, early)
    d = np.concatenate([synth.descriptors_near_leaves(seed + 50, n // 2, par, desc, 0.04), synth.Rng(seed).bytes((n - n // 2) * 32).reshape(-1, 32)])
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    (bw, bv), fv, (word, node, weight) = fm.ComputeBoWVector(d, levelsup)
    rw, rn, rwt = ov.transform(d, levelsup)
    assert np.array_equal(word, rw) and np.array_equal(node, rn) and np.array_equal(weight, rwt)
    ow, ovv = ov.bow_vector(rw, rwt)
    assert np.array_equal(bw, ow) and np.array_equal(bv, ovv)
    # strided rows (a cv::Mat ROI) give the same answer
    wide = np.zeros((n, 48), np.uint8); wide[:, :32] = d
    assert np.array_equal(fm.ComputeBoWVector(wide[:, :32], levelsup)[2][1], rn)
    fm.close(); v.close()


def test_transform_ties_pick_the_first_child(mods):
    S, synth, O, V, _ = mods
    # every child of every node carries the same descriptor: all distances tie, the first child must win at each level
    k, L = 5, 3
    par = [-1]; lvl = [0]
    i = 0
    while i < len(par):
        if lvl[i] < L:
            for _ in range(k):
                par.append(i); lvl.append(lvl[i] + 1)
        i += 1
    par = np.array(par, np.int32)
    desc = np.zeros((len(par), 32), np.uint8); w = np.ones(len(par))
    v = V.Vocabulary(k, L, par, desc, w); ov = O.Vocabulary(k, L, par, desc, w, v.word)
    d = synth.Rng(3).bytes(40 * 32).reshape(40, 32)
    fm = S.FeatureMatching(v, {})
    word, node, weight = fm.ComputeBoWVector(d, 1)[2]
    rw, rn, _ = ov.transform(d, 1)
    assert np.array_equal(word, rw) and np.array_equal(node, rn) and (word == 0).all()
    fm.close(); v.close()


@pytest.mark.parametrize("seed,k,L,levelsup,nq,nt", [(11, 10, 3, 1, 1000, 1010), (12, 10, 4, 2, 500, 300), (13, 4, 4, 4, 257, 255),
                                                      (14, 10, 6, 4, 1200, 1200), (15, 3, 3, 1, 1, 1), (16, 10, 3, 0, 2000, 1500)])
def test_match_dbow_equals_oracle(mods, seed, k, L, levelsup, nq, nt):
    S, synth, O, V, _ = mods
    v, ov, par, desc = _pair(mods, seed, k, L, early=0.15 if L > 3 else 0.0)
    t = synth.descriptors_near_leaves(seed + 1, nt, par, desc, 0.03)
    rng = np.random.default_rng(seed)
    q = t[rng.integers(0, nt, nq)].copy()
    q ^= np.packbits(rng.random((nq, 256)) < 0.02, axis=1, bitorder="little")
    q[::7] = synth.Rng(seed).bytes(len(q[::7]) * 32).reshape(-1, 32)          # outliers
    for th, ratio in ((50, 0.6), (20, 0.9), (256, 1.5)):
        fm = S.FeatureMatching(v, {"Threshold": th, "NNRatio": ratio})
        ok, m = fm.MatchByDBoW(q, t, levelsup=levelsup)
        want = O.match_dbow(ov, q, t, th, ratio, levelsup)
        assert np.array_equal(_bytes(m), _bytes(want)) and ok == (len(want) > 0)
        # members the reference fills on the way (include/FeatureMatching.h:46-49)
        rw, rn, rwt = ov.transform(q, levelsup)
        ow, ovv = ov.bow_vector(rw, rwt)
        assert np.array_equal(fm.mBowVec1_[0], ow) and np.array_equal(fm.mBowVec1_[1], ovv)
        assert fm.mFeatVec1_ == V.Vocabulary.feature_vector(rn, rwt)
        fm.close()
    if nq > 100:
        assert len(O.match_dbow(ov, q, t, 50, 0.6, levelsup)) > nq // 20
    v.close()


def test_match_dbow_on_extracted_descriptors(mods):
    """ORB descriptors of a frame pair through the whole chain: Extract -> MatchByDBoW (levelsup = 4 like :206)"""
    S, synth, O, V, _ = mods
    a, b = synth.frame_pair(21, 240, 320)
    ex = S.ORBextractor(1200, 1.2, 8, 20, 7)
    _, da = ex.Extract(a); _, db = ex.Extract(b)
    v, ov, par, desc = _pair(mods, 22, 10, 5, stop=0.02)
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    ok, m = fm.MatchByDBoW(db, da)
    want = O.match_dbow(ov, db, da, 50, 0.6, 4)
    assert np.array_equal(_bytes(m), _bytes(want)) and len(m) > 10
    ex.close(); fm.close(); v.close()


def test_match_dbow_edge_cases(mods):
    S, synth, O, V, L = mods
    v, ov, par, desc = _pair(mods, 31, 4, 3)
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    d = synth.descriptors_near_leaves(1, 20, par, desc)
    keep = np.zeros(3, L.DM_DTYPE); keep["queryIdx"] = [5, 6, 7]
    # either matrix empty: false, and `matches` comes back as it went in (:61-65 returns before the clear)
    for a, b in ((d[:0], d), (d, d[:0]), (None, d)):
        ok, m = fm.MatchByDBoW(a, b, matches=keep)
        assert ok is False and m is keep
    # every word stopped: nothing can match
    v0 = V.Vocabulary(4, 3, par, desc, np.zeros(len(par)))
    fm0 = S.FeatureMatching(v0, {})
    ok, m = fm0.MatchByDBoW(d, d)
    assert ok is False and len(m) == 0
    # identical sets: every non-stopped feature finds itself at distance 0
    ok, m = fm.MatchByDBoW(d, d)
    want = O.match_dbow(ov, d, d, 50, 0.6, 4)
    assert np.array_equal(_bytes(m), _bytes(want))
    # without a vocabulary the call fails loudly; a bad table is rejected by the library
    fmn = S.FeatureMatching(None, {})
    with pytest.raises(ValueError):
        fmn.MatchByDBoW(d, d)
    bad = par.copy(); bad[5] = 9
    with pytest.raises(S.SsvioError):
        V.Vocabulary(4, 3, bad, desc, np.ones(len(par))).handle(0)
    n = C.c_int(0); out = np.zeros(20, L.DM_DTYPE)
    assert L.lib().ssvio_match_dbow(fm._h, None, L.ptr(d), 20, 32, L.ptr(d), 20, 32, 4, L.ptr(out), 20, C.byref(n)) == L.ERR_INVALID
    assert L.lib().ssvio_match_dbow(fm._h, v.handle(0), L.ptr(d), 20, 32, L.ptr(d), 20, 32, 4, L.ptr(out), 10, C.byref(n)) == L.ERR_CAPACITY
    for h in (fm, fm0, fmn): h.close()
    v.close(); v0.close()


def test_match_dbow_batch_dev_equals_single_calls(mods):
    import torch
    S, synth, O, V, L = mods
    v, ov, par, desc = _pair(mods, 41, 10, 4)
    nsets, stride = 5, 700
    counts = [700, 650, 0, 512, 699]
    sets = [synth.descriptors_near_leaves(100 + s, max(c, 1), par, desc, 0.03)[:c] for s, c in enumerate(counts)]
    sets[1][:300] = sets[0][100:400]; sets[4][:400] = sets[3][:400] ^ 1
    buf = np.zeros((nsets, stride, 32), np.uint8)
    for s, d in enumerate(sets): buf[s, :len(d)] = d
    d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    pairs = np.array([[1, 0], [4, 3], [0, 2], [2, 1], [3, 3], [0, 4]], np.int32)
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    fm.match_dbow_batch_dev(d_desc, stride, stride, d_cnt, nsets, pairs)
    fm.sync()
    out, n = fm.fetch_batch()
    for p, (a, b) in enumerate(pairs):
        want = O.match_dbow(ov, sets[a], sets[b], 50, 0.6, 4)
        assert n[p] == len(want) and np.array_equal(_bytes(out[p, :n[p]]), _bytes(want)), p
    assert n[0] > 100 and n[2] == 0 and n[3] == 0
    fm.close(); v.close()


# ---------------------------------------------------------------- f2, f3
def _frame(mods, seed, h=240, w=320):
    S, synth, O, V, L = mods
    a, b = synth.frame_pair(seed, h, w)
    ex = S.ORBextractor(1200, 1.2, 8, 20, 7)
    ka, da = ex.Extract(a); kb, db = ex.Extract(b)
    ex.close()
    return ka, da, kb, db, synth.depth_image(seed + 1, h, w), synth.depth_image(seed + 2, h, w)


def test_get_3d_points_equals_oracle(mods):
    S, synth, O, V, L = mods
    ka, da, kb, db, depth_a, depth_b = _frame(mods, 51)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)            # config/default_conf.yaml:9-13
    fm = S.FeatureMatching(None, {})
    got = fm.Get3DPoints(ka, depth_a, cam)
    want = O.get_3d_points(ka, depth_a, cam)
    assert got.shape == want.shape and np.array_equal(_bytes(got), _bytes(want))
    assert (want[:, 2] == 0).any() and (want[:, 2] > 0).any()               # holes and valid depth both occur
    assert len(fm.Get3DPoints(ka[:0], depth_a, cam)) == 0
    fm.close()


def test_assemble_correspondences_equals_oracle(mods):
    S, synth, O, V, L = mods
    ka, da, kb, db, depth_a, depth_b = _frame(mods, 61)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    ok, m = fm.MatchByBruteForce(db, da)                                     # query = current (b), train = last (a)
    assert len(m) > 300                                                      # more than one 256-match pass
    got = fm.AssembleCorrespondences(m, ka, depth_a, kb, depth_b, cam)
    want = O.assemble_correspondences(m, ka, depth_a, kb, depth_b, cam)
    for key in want:
        assert got[key].shape == want[key].shape and np.array_equal(_bytes(got[key]), _bytes(want[key])), key
    assert 0 < len(want["final_good"]) < len(want["pre_good"]) < len(m)
    empty = fm.AssembleCorrespondences(m[:0], ka, depth_a, kb, depth_b, cam)
    assert all(len(x) == 0 for x in empty.values())
    bad = m.copy(); bad["trainIdx"][3] = len(ka)
    with pytest.raises(S.SsvioError):
        fm.AssembleCorrespondences(bad, ka, depth_a, kb, depth_b, cam)
    fm.close()


def test_extract_match_assemble_stays_on_the_device(mods):
    """extract_batch_dev -> match_batch_dev -> assemble_batch_dev without a host round trip, vs the oracle per pair"""
    import torch
    S, synth, O, V, L = mods
    B, H, W = 4, 240, 320
    frames = synth.frame_stream(71, B, H, W)
    depth = np.stack([synth.depth_image(80 + f, H, W) for f in range(B)])
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    ex = S.ORBextractor(1200, 1.2, 8, 20, 7); fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    fm.set_stream(ex.get_stream())
    d_frames = torch.from_numpy(frames).cuda(); d_depth = torch.from_numpy(np.ascontiguousarray(depth)).cuda()
    assert d_depth.is_contiguous()
    ex.extract_batch_dev(d_frames, B, W, H, wait=False)
    d_kp, d_desc, d_n, cap = ex.device_results()
    pairs = np.array([[f, f - 1] for f in range(1, B)], np.int32)
    dm, dn = fm.match_batch_dev(d_desc, cap, cap, d_n, pairs, L.MODE_BRUTEFORCE)
    r = fm.assemble_batch_dev(dm, dn, cap, pairs, d_kp, cap, d_n, d_depth, H * W, W, H, W, cam)
    fm.sync()
    kps, descs, nk = ex.fetch_results(B)
    kps = [kps[f, :nk[f]] for f in range(B)]; descs = [descs[f, :nk[f]] for f in range(B)]
    rows = len(pairs) * cap

    class _Dev:                                                           # a device buffer of the library, seen by torch (no copy)
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def dev(ptr, nbytes, dtype):
        return torch.as_tensor(_Dev(ptr, nbytes), device="cuda").cpu().numpy().view(dtype).copy()

    counts = dev(r.counts, len(pairs) * 8, np.int32).reshape(-1, 2)
    last3d = dev(r.last3d, rows * 12, np.float32).reshape(len(pairs), cap, 3)
    cur2d = dev(r.cur2d, rows * 8, np.float32).reshape(len(pairs), cap, 2)
    edge_cur = dev(r.edge_cur3d, rows * 12, np.float32).reshape(len(pairs), cap, 3)
    final = dev(r.final_good, rows * 16, L.DM_DTYPE).reshape(len(pairs), cap)
    for p, (c, l) in enumerate(pairs):
        m = O.match_bruteforce(descs[c], descs[l])
        want = O.assemble_correspondences(m, kps[l], depth[l], kps[c], depth[c], cam)
        assert counts[p].tolist() == [len(want["pre_good"]), len(want["final_good"])]
        assert np.array_equal(_bytes(last3d[p, :counts[p, 0]]), _bytes(want["last3d"]))
        assert np.array_equal(_bytes(cur2d[p, :counts[p, 0]]), _bytes(want["cur2d"]))
        assert np.array_equal(_bytes(edge_cur[p, :counts[p, 1]]), _bytes(want["edge_cur3d"]))
        assert np.array_equal(_bytes(final[p, :counts[p, 1]]), _bytes(want["final_good"]))
        assert counts[p, 1] > 20
    ex.close(); fm.close()


@pytest.mark.parametrize("seed,nlast,ncur,nkp", [(1, 600, 700, 1000), (2, 1, 1, 5), (3, 300, 5, 400), (4, 1000, 1000, 300),
                                                  (5, 3000, 3000, 20000), (6, 9000, 9000, 4000)])     # beyond the in-LDS index map: global table
def test_track_map_points_and_in_match_vec_equal_oracle(mods, seed, nlast, ncur, nkp):
    S, synth, O, V, L = mods
    rng = np.random.default_rng(seed)
    last = np.zeros(nlast, L.DM_DTYPE); cur = np.zeros(ncur, L.DM_DTYPE)
    last["queryIdx"] = rng.integers(0, nkp, nlast); last["trainIdx"] = rng.integers(0, nkp, nlast)      # duplicates on purpose
    cur["queryIdx"] = rng.integers(0, nkp, ncur); cur["trainIdx"] = rng.integers(0, nkp, ncur)
    fm = S.FeatureMatching(None, {})
    got = fm.TrackMapPoints(last, cur)
    want = O.track_map_points(last, cur)
    assert np.array_equal(_bytes(got), _bytes(want))
    assert np.array_equal(fm.InMatchVec(last, nkp), O.in_match_vec(last, nkp))
    assert len(fm.TrackMapPoints(last[:0], cur)) == 0 and len(fm.TrackMapPoints(last, cur[:0])) == 0
    assert np.array_equal(fm.InMatchVec(last[:0], 7), np.full(7, -1, np.int32))
    fm.close()


@pytest.mark.parametrize("seed,nm,nobs", [(1, 400, 900), (2, 0, 50), (3, 700, 0), (4, 1500, 3000)])
def test_manage_map_points_equals_oracle(mods, seed, nm, nobs):
    """the loop of Map::ManageMapPoints (src/Map.cpp:125-149): InMatchVec + GetMapPointID joins and the world points of the
    new map points, bit for bit (float order stated in include/ssvio_gpu.h)"""
    S, synth, O, V, L = mods
    ka, da, kb, db, depth_a, depth_b = _frame(mods, 61)
    nkp = len(kb)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    rng = np.random.default_rng(seed)
    v = np.zeros(nm, L.DM_DTYPE); v["queryIdx"] = rng.integers(0, nkp + 5, nm); v["trainIdx"] = rng.integers(0, nkp, nm)   # duplicates, some ids past the list
    obs = np.stack([rng.integers(0, 5000, nobs), rng.integers(0, nkp + 5, nobs)], axis=1).astype(np.int32)
    a = rng.normal(size=3); a /= np.linalg.norm(a); th = 0.7
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; T[:3, 3] = rng.normal(size=3) * 2
    fm = S.FeatureMatching(None, {})
    got = fm.ManageMapPoints(kb, depth_b, cam, T, v, obs)
    want = O.manage_map_points(kb, depth_b, cam, T, v, obs)
    for g, w_ in zip(got, want):
        assert g.shape == w_.shape and np.array_equal(_bytes(g), _bytes(w_))
    assert np.array_equal(got[0], O.in_match_vec(v, nkp))
    if nm and nobs:
        assert (got[1] >= 0).sum() > 10 and ((got[0] >= 0) & (got[1] < 0)).any() and (got[0] < 0).any()
    assert all(len(x) == 0 for x in fm.ManageMapPoints(kb[:0], depth_b, cam, T, v, obs))
    fm.close()


@pytest.mark.parametrize("h,w", [(240, 320), (480, 640), (37, 53), (1, 1)])
def test_rgb_cloud_equals_oracle(mods, h, w):
    S, synth, O, V, L = mods
    depth = synth.depth_image(91, h, w) if h > 8 else np.array([[4711]], np.uint16)
    bgr = synth.to_rgb(synth.texture(92, max(h, 16), max(w, 16))[:h, :w], 3)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    fm = S.FeatureMatching(None, {})
    got = fm.RGBCloud(depth, bgr, cam, 10.0)
    want = O.rgb_cloud(depth, bgr, 160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    fm.close()


def test_match_dbow_unsorted_path_and_large_sets(mods, monkeypatch):
    """sets beyond the LDS sort (> 8192 rows) take the wave-per-query kernels; both paths give the oracle's list"""
    S, synth, O, V, L = mods
    v, ov, par, desc = _pair(mods, 17, 10, 3)
    t = synth.descriptors_near_leaves(18, 8300, par, desc, 0.03)
    q = t[np.random.default_rng(5).integers(0, 8300, 8200)] ^ np.packbits(np.random.default_rng(6).random((8200, 256)) < 0.02, axis=1, bitorder="little")
    fm = S.FeatureMatching(v, {"Threshold": 50, "NNRatio": 0.6})
    want = O.match_dbow(ov, q, t, 50, 0.6, 1)
    assert len(want) > 1000
    assert np.array_equal(_bytes(fm.MatchByDBoW(q, t, levelsup=1)[1]), _bytes(want))                     # 8300 rows: unsorted path
    assert np.array_equal(_bytes(fm.MatchByDBoW(q[:8192], t[:8192], levelsup=1)[1]), _bytes(O.match_dbow(ov, q[:8192], t[:8192], 50, 0.6, 1)))   # sorted, full LDS
    small = O.match_dbow(ov, q[:900], t[:1000], 50, 0.6, 1)
    assert np.array_equal(_bytes(fm.MatchByDBoW(q[:900], t[:1000], levelsup=1)[1]), _bytes(small))      # sorted path
    monkeypatch.setenv("SSVIO_DBOW_UNSORTED", "1")
    assert np.array_equal(_bytes(fm.MatchByDBoW(q[:900], t[:1000], levelsup=1)[1]), _bytes(small))      # same inputs, unsorted path
    fm.close(); v.close()


def _pose(seed):
    rng = np.random.default_rng(seed)
    a, b, c = rng.uniform(-0.3, 0.3, 3)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1.0, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    T = np.eye(4); T[:3, :3] = Rz @ Ry @ Rx; T[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    return T.astype(np.float32)


@pytest.mark.parametrize("h,w,voxel", [(120, 160, 0.01), (240, 320, 0.01), (240, 320, 0.05), (37, 53, 0.3)])
def test_map_point_cloud_fusion_equals_oracle(mods, h, w, voxel):
    """Map::MapPointCloudFusion (src/Map.cpp:164-197) over four key frames: the map after every iteration, byte for byte.
    Parity unpinned (PCL restated in the oracle).  The zero-depth holes of every frame land on one point: a run of thousands
    of points in one voxel (the wave-wide summation path)."""
    S, synth, O, V, L = mods
    cam = O.camera(160.5912 * w / 320, 120.4792 * h / 240, 253.0589 * w / 320, 254.1649 * h / 240, 10.0)
    fm = S.FeatureMatching(None, {})
    want = np.zeros(0, O.CLOUD_DTYPE)
    for k in range(4):
        depth = synth.depth_image(500 + k, h, w)
        bgr = synth.to_rgb(synth.texture(600 + k, max(h, 16), max(w, 16))[:h, :w], k)
        T = _pose(700 + k)
        cloud = O.rgb_cloud(depth, bgr, float(cam[0]), float(cam[1]), float(cam[2]), float(cam[3]), 10.0)
        want = O.voxel_fuse(want, cloud, T, voxel)
        if k % 2 == 0:
            fm.MapFuseFrame(depth, bgr, cam, 10.0, T, voxel)             # cloud built on the device from depth + BGR
        else:
            fm.MapFuse(cloud, T, voxel)                                  # host cloud
        got = fm.MapFetch()
        assert len(got) == len(want) and np.array_equal(got.view(np.uint8), want.view(np.uint8)), k
    assert 0 < len(want) < 4 * h * w
    # a map loaded from the host continues identically; an empty cloud re-filters the map (idempotent here)
    fm.MapSet(want[: len(want) // 2])
    fm.MapFuse(want[len(want) // 2:], np.eye(4), voxel)
    again = O.voxel_fuse(want[: len(want) // 2], want[len(want) // 2:], np.eye(4), voxel)
    assert np.array_equal(fm.MapFetch().view(np.uint8), again.view(np.uint8))
    fm.MapSet(None)
    assert len(fm.MapFetch()) == 0
    fm.close()


def test_map_fusion_edge_cases(mods):
    import torch
    S, synth, O, V, L = mods
    fm = S.FeatureMatching(None, {})
    I = np.eye(4, dtype=np.float32)
    c = np.zeros(5, O.CLOUD_DTYPE); c["w"] = 1; c["a"] = 255
    c["x"] = [0.0, 100.0, 1e-4, 0.02, 0.0201]; c["y"][1] = 100.0; c["z"][1] = 100.0; c["r"] = [1, 2, 3, 4, 5]
    fm.MapFuse(c, I, 0.01)                                               # 10^4 cells per axis: indices overflow, the input comes back
    assert np.array_equal(fm.MapFetch().view(np.uint8), O.voxel_fuse(c[:0], c, I, 0.01).view(np.uint8)) and len(fm.MapFetch()) == 5
    fm.MapSet(None)
    fm.MapFuse(torch.from_numpy(c.view(np.uint8).reshape(-1, 32)).cuda(), I, 1.0)          # a device cloud
    assert np.array_equal(fm.MapFetch().view(np.uint8), O.voxel_fuse(c[:0], c, I, 1.0).view(np.uint8))
    fm.MapSet(None)
    fm.MapFuse(c[:0], I, 0.01)
    assert len(fm.MapFetch()) == 0
    bad = c.copy(); bad["x"][2] = np.nan
    with pytest.raises(S.SsvioError):
        fm.MapFuse(bad, I, 0.01)
    with pytest.raises(S.SsvioError):
        fm.MapFuse(c, I, 0.0)
    fm.MapSet(None)
    fm.MapFuse(c, I, 0.01)                                               # the handle stays usable
    assert len(fm.MapFetch()) == 5
    fm.close()
