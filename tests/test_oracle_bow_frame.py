"""CPU: known-answer tests that pin the oracle's restatement of MatchByDBoW (src/FeatureMatching.cpp:58-125,
:192-208; DBoW3's transform) and of the steps behind the matcher (src/Frame.cpp:93-108, src/Odometry.cpp:44-64,
:94-108, src/Map.cpp:100-123, :151-162), plus the host halves that live in ssvio_amd/vocabulary.py.

The reference holds no fixtures for these: expectations are derived by hand in the comments, and a
second, literal Python restatement (std::map semantics with dicts) cross-checks the C oracle on random inputs.
"""
import numpy as np
import pytest


def _desc(*first_bytes):
    """32-byte descriptors that differ only in byte 0"""
    d = np.zeros((len(first_bytes), 32), np.uint8)
    d[:, 0] = first_bytes
    return d


@pytest.fixture()
def tiny(oracle):
    # k = 2, L = 2:     0
    #                 /   \
    #               1       2          node descriptors (byte 0): 1: 0x00, 2: 0xFF
    #              / \     / \
    #             3   4   5   6        3: 0x00, 4: 0x0F, 5: 0xF0, 6: 0xFF;  words 0..3;  weights 1, 2, 0 (stopped), 4
    parent = [-1, 0, 0, 1, 1, 2, 2]
    desc = _desc(0, 0x00, 0xFF, 0x00, 0x0F, 0xF0, 0xFF)
    weight = [0, 0, 0, 1.0, 2.0, 0.0, 4.0]
    word = [-1, -1, -1, 0, 1, 2, 3]
    return oracle.Vocabulary(2, 2, parent, desc, weight, word)


def test_transform_descends_to_the_nearest_child_first_child_on_ties(tiny):
    # 0x01: d(1)=1 < d(2)=7 -> node 1; d(3)=1 < d(4)=3 -> leaf 3, word 0
    # 0x0F: d(1)=4 = d(2)=4 tie -> first child 1; d(3)=4 > d(4)=0 -> leaf 4, word 1
    # 0xF8: d(1)=5 > d(2)=3 -> node 2; d(5)=1 < d(6)=3 -> leaf 5, word 2 (weight 0)
    # 0xFF: node 2, leaf 6, word 3;   0x3C: d(1)=4 = d(2)=4 -> node 1; d(3)=4, d(4)=4 -> tie -> leaf 3
    f = _desc(0x01, 0x0F, 0xF8, 0xFF, 0x3C)
    word, node, weight = tiny.transform(f, levelsup=1)             # nid level = L - 1 = 1
    assert word.tolist() == [0, 1, 2, 3, 0]
    assert node.tolist() == [1, 1, 2, 2, 1]
    assert weight.tolist() == [1.0, 2.0, 0.0, 4.0, 1.0]
    _, node0, _ = tiny.transform(f, levelsup=0)                     # nid level = 2: the leaves themselves
    assert node0.tolist() == [3, 4, 5, 6, 3]
    for ls in (2, 4):                                               # nid level <= 0: the root (reference passes 4, :206)
        assert tiny.transform(f, levelsup=ls)[1].tolist() == [0] * 5


def test_bow_vector_tf_idf_l1(tiny):
    # words 0, 1, 2(stopped), 3, 0 with weights 1, 2, 0, 4, 1 -> {0: 2, 1: 2, 3: 4}, L1 norm 8
    w, v = tiny.bow_vector([0, 1, 2, 3, 0], [1.0, 2.0, 0.0, 4.0, 1.0])
    assert w.tolist() == [0, 1, 3] and v.tolist() == [0.25, 0.25, 0.5]


def test_bow_vector_other_weightings_and_scores(oracle, tiny):
    word, weight = [3, 0, 3, 1], [4.0, 1.0, 4.0, 2.0]
    tiny.weighting, tiny.scoring = 3, 5                             # BINARY keeps the first weight, DOT_PRODUCT does not normalise
    w, v = tiny.bow_vector(word, weight)
    assert w.tolist() == [0, 1, 3] and v.tolist() == [1.0, 2.0, 4.0]
    tiny.weighting, tiny.scoring = 1, 5                             # TF without normalisation: divided by the vector's size
    w, v = tiny.bow_vector(word, weight)
    assert v.tolist() == [1.0 / 3, 2.0 / 3, 8.0 / 3]
    tiny.weighting, tiny.scoring = 0, 1                             # L2: sqrt(1 + 4 + 64)
    w, v = tiny.bow_vector(word, weight)
    assert np.array_equal(v, np.array([1.0, 2.0, 8.0]) / np.sqrt(69.0))


def test_match_dbow_order_and_acceptance(oracle, tiny):
    # train features: j0 0x0E (node 1), j1 0xFE (node 2), j2 0x00 (node 1), j3 0xF1 -> leaf 5, stopped (never a candidate)
    t = _desc(0x0E, 0xFE, 0x00, 0xF1)
    # queries: i0 0xFF node 2: candidates {j1}: best 1, second 256 -> kept (1 <= 50, 1 < 0.6 * 256)
    #          i1 0x0F node 1: candidates {j0: 1, j2: 4} -> best 1 (j0), second 4: 1 < 2.4 -> kept
    #          i2 0x01 node 1: {j0: 4, j2: 1} -> best 1 (j2), second 4 -> kept
    #          i3 0xF8 -> leaf 5, stopped -> never matched
    q = _desc(0xFF, 0x0F, 0x01, 0xF8)
    m = oracle.match_dbow(tiny, q, t, 50, 0.6, levelsup=1)
    # std::map order: node 1 first (queries 1, 2), then node 2 (query 0)
    assert m["queryIdx"].tolist() == [1, 2, 0] and m["trainIdx"].tolist() == [0, 2, 1]
    assert m["distance"].tolist() == [1.0, 1.0, 1.0] and (m["imgIdx"] == -1).all()
    # levelsup = 2 -> every feature in node 0: one group, ascending query index; i3 and j3 still excluded (weight 0)
    m = oracle.match_dbow(tiny, q, t, 50, 0.6, levelsup=2)
    assert m["queryIdx"].tolist() == [0, 1, 2] and m["trainIdx"].tolist() == [1, 0, 2]
    # Threshold is inclusive (:104), the ratio test strict (:105): best 1, second 4 fails at ratio 0.25, passes above
    assert oracle.match_dbow(tiny, q, t, 1, 0.6, levelsup=1)["queryIdx"].tolist() == [1, 2, 0]
    assert oracle.match_dbow(tiny, q, t, 0, 0.6, levelsup=1)["queryIdx"].tolist() == []
    assert oracle.match_dbow(tiny, q, t, 50, 0.25, levelsup=1)["queryIdx"].tolist() == [0]
    assert oracle.match_dbow(tiny, q, t, 50, 0.26, levelsup=1)["queryIdx"].tolist() == [1, 2, 0]
    # a node present in one frame only is skipped (:117-122)
    assert oracle.match_dbow(tiny, q, _desc(0xFE), 50, 0.6, levelsup=1)["queryIdx"].tolist() == [0]
    assert len(oracle.match_dbow(tiny, q[:0], t)) == 0 and len(oracle.match_dbow(tiny, q, t[:0])) == 0


def _literal_match_dbow(voc, q, t, th, ratio, levelsup):
    """src/FeatureMatching.cpp:58-125 line by line, FeatureVector = dict iterated in key order"""
    def fv(d):
        _, node, w = voc.transform(d, levelsup)
        out = {}
        for i in range(len(d)):
            if w[i] > 0:
                out.setdefault(int(node[i]), []).append(i)
        return out
    f1, f2 = fv(q), fv(t)
    ham = lambda a, b: int(np.unpackbits(a ^ b).sum())
    res = []
    for nid in sorted(set(f1) & set(f2)):
        for i in f1[nid]:
            b1, b2, bi = 256, 256, -1
            for j in f2[nid]:
                d = ham(q[i], t[j])
                if d < b1: b2, b1, bi = b1, d, j
                elif d < b2: b2 = d
            if b1 <= th and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                res.append((i, bi, b1))
    return res


@pytest.mark.parametrize("seed,k,L,levelsup,early", [(1, 10, 3, 1, 0.0), (2, 4, 4, 2, 0.3), (3, 3, 5, 4, 0.1), (4, 20, 2, 1, 0.0)])
def test_c_oracle_equals_literal_restatement(oracle, seed, k, L, levelsup, early):
    from ssvio_amd import synth
    par, desc, w = synth.vocabulary_tables(seed, k, L, early_leaf_p=early)
    voc = oracle.Vocabulary(k, L, par, desc, w, _words(par))
    t = synth.descriptors_near_leaves(seed + 10, 150, par, desc, 0.03)
    q = t[np.random.default_rng(seed).permutation(150)[:120]].copy()
    q ^= np.packbits(np.random.default_rng(seed + 1).random((120, 256)) < 0.02, axis=1, bitorder="little")
    m = oracle.match_dbow(voc, q, t, 50, 0.6, levelsup)
    want = _literal_match_dbow(voc, q, t, 50, 0.6, levelsup)
    assert len(want) > 10
    assert [(int(a), int(b), int(c)) for a, b, c in zip(m["queryIdx"], m["trainIdx"], m["distance"])] == want


def _words(parent):
    n = len(parent)
    leaf = np.ones(n, bool); leaf[parent[1:]] = False
    word = np.full(n, -1, np.int32); word[leaf] = np.arange(int(leaf.sum()))
    return word


def test_host_vocabulary_maps_and_text_format(oracle, tmp_path):
    """ssvio_amd.vocabulary: BowVector / FeatureVector assembly equals the oracle's, and the text format round-trips"""
    from ssvio_amd import synth
    from ssvio_amd.vocabulary import Vocabulary
    par, desc, w = synth.vocabulary_tables(7, 5, 3, early_leaf_p=0.2)
    v = Vocabulary(5, 3, par, desc, w)
    assert np.array_equal(v.word, _words(par))
    ov = oracle.Vocabulary(5, 3, par, desc, w, v.word)
    d = synth.descriptors_near_leaves(8, 300, par, desc, 0.05)
    word, node, weight = ov.transform(d, 1)
    for scoring, weighting in ((0, 0), (1, 0), (5, 1), (5, 3), (2, 2)):
        v.scoring = ov.scoring = scoring; v.weighting = ov.weighting = weighting
        a, b = v.bow_vector(word, weight), ov.bow_vector(word, weight)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    fv = Vocabulary.feature_vector(node, weight)
    assert list(fv) == sorted(fv) and sum(len(x) for x in fv.values()) == int((weight > 0).sum())
    assert all(x == sorted(x) for x in fv.values())
    v.scoring, v.weighting = 0, 0
    p = tmp_path / "voc.txt"
    v.save_txt(str(p))
    r = Vocabulary.load_txt(str(p))
    assert (r.k, r.L, r.scoring, r.weighting) == (5, 3, 0, 0)
    assert np.array_equal(r.parent[1:], v.parent[1:]) and np.array_equal(r.desc[1:], v.desc[1:])
    assert np.array_equal(r.weight[1:], v.weight[1:]) and np.array_equal(r.word, v.word)


def test_dbow3_binary_stream_round_trip_and_refusals(tmp_path):
    """DBoW3's uncompressed toStream / fromStream layout as ssvio_amd.vocabulary restates it (no .dbow3 file exists here to
    pin it: this checks the reader against the writer, the header-order tolerance and what is refused)"""
    import struct
    from ssvio_amd import synth
    from ssvio_amd.vocabulary import Vocabulary
    par, desc, w = synth.vocabulary_tables(11, 4, 3, early_leaf_p=0.25)
    v = Vocabulary(4, 3, par, desc, w, scoring=1, weighting=2)
    p = str(tmp_path / "v.dbow3")
    v.save_dbow3(p)
    raw = open(p, "rb").read()
    assert struct.unpack_from("<QBI4i", raw, 0) == (88877711233, 0, len(par), 4, 3, 1, 2)
    assert Vocabulary.is_dbow3(p)
    r = Vocabulary.load(p)
    assert (r.k, r.L, r.scoring, r.weighting) == (4, 3, 1, 2)
    assert np.array_equal(r.parent, v.parent) and np.array_equal(r.desc[1:], v.desc[1:]) and np.array_equal(r.weight[1:], v.weight[1:])
    assert np.array_equal(r.word, v.word)
    # records are not in id order in the stream (stack walk): the first record's parent is the root, a later one jumps back
    ids = [struct.unpack_from("<I", raw, 29 + 60 * i)[0] for i in range(len(par) - 1)]
    assert ids != sorted(ids) and sorted(ids) == list(range(1, len(par)))
    # descriptor header written as (type, cols, rows) instead of (cols, rows, type): still understood
    alt = bytearray(raw)
    for i in range(len(par) - 1):
        struct.pack_into("<3i", alt, 29 + 60 * i + 16, 0, 32, 1)
    open(p, "wb").write(bytes(alt))
    assert np.array_equal(Vocabulary.load(p).desc, r.desc)
    # refused: compressed flag, truncated file, a word that names an inner node, a 64-byte descriptor header
    def refused(b):
        open(p, "wb").write(bytes(b))
        with pytest.raises(ValueError):
            Vocabulary.load_dbow3(p)
    b = bytearray(raw); b[8] = 1; refused(b)
    refused(raw[:len(raw) // 2])
    b = bytearray(raw); struct.pack_into("<I", b, len(raw) - 4, 0); refused(b)            # last word -> node 0 (the root)
    b = bytearray(raw); struct.pack_into("<i", b, 29 + 16, 64); refused(b)
    # a text vocabulary goes to the text reader
    t = str(tmp_path / "v.txt"); v.save_txt(t)
    assert not Vocabulary.is_dbow3(t) and np.array_equal(Vocabulary.load(t).parent, v.parent)


# ---------------------------------------------------------------- behind the matcher
def _kps(oracle, pts):
    k = np.zeros(len(pts), oracle.KP_DTYPE)
    k["x"] = [p[0] for p in pts]; k["y"] = [p[1] for p in pts]
    return k


def test_get_3d_point_known_values(oracle):
    cam = oracle.camera(2.0, 1.0, 4.0, 8.0, 65535.0 * 0.5)          # DepthScale/65535 = 0.5 exactly
    assert cam[4] == np.float32(0.5)
    depth = np.arange(12, dtype=np.uint16).reshape(3, 4) * 2        # depth(y, x) = 2 * (4 y + x)
    # keypoint (3.9, 1.5): pixel (int)1, (int)3 -> raw 14 -> d = 7;  x = (3.9 - 2) * 7 * 0.25, y = (1.5 - 1) * 7 * 0.125
    p = oracle.get_3d_points(_kps(oracle, [(3.9, 1.5), (0.0, 0.0)]), depth, cam)
    f = np.float32
    assert p[0].tolist() == [float((f(3.9) - f(2.0)) * f(7.0) * f(0.25)), float(f(0.5) * f(7.0) * f(0.125)), 7.0]
    assert p[1].tolist() == [0.0, 0.0, 0.0]                         # raw 0 -> depth 0 -> the origin
    # the yaml value of the reference: 10.0 / 65535.0 narrowed to float (include/Init.h:18)
    assert oracle.camera(0, 0, 1, 1, 10.0)[4] == np.float32(10.0 / 65535.0)


def test_assemble_correspondences_filters_in_order(oracle):
    cam = oracle.camera(0.0, 0.0, 1.0, 1.0, 65535.0)                # depth value = raw, threshold z >= 1
    last_d = np.array([[5, 0, 7, 9]], np.uint16); cur_d = np.array([[0, 3, 4, 0]], np.uint16)
    last_k = _kps(oracle, [(0, 0), (1, 0), (2, 0), (3, 0)]); cur_k = _kps(oracle, [(0, 0), (1, 0), (2, 0), (3, 0)])
    m = np.zeros(4, oracle.DM_DTYPE)
    m["queryIdx"] = [1, 2, 0, 3]; m["trainIdx"] = [0, 1, 2, 3]; m["distance"] = [10, 11, 12, 13]
    # match 0: last z 5 ok, cur z 3 ok -> both lists;  match 1: last z 0 -> dropped;  match 2: last 7 ok, cur 0 -> first list only
    # match 3: last 9 ok, cur kp 3 z 0 -> first list only
    r = oracle.assemble_correspondences(m, last_k, last_d, cur_k, cur_d, cam)
    assert r["pre_good"]["trainIdx"].tolist() == [0, 2, 3] and r["final_good"]["trainIdx"].tolist() == [0]
    assert r["last3d"][:, 2].tolist() == [5.0, 7.0, 9.0] and r["cur3d"][:, 2].tolist() == [3.0, 0.0, 0.0]
    assert r["cur2d"].tolist() == [[1.0, 0.0], [0.0, 0.0], [3.0, 0.0]]
    assert r["edge_last3d"].tolist() == [[0.0, 0.0, 5.0]] and r["edge_cur3d"].tolist() == [[3.0, 0.0, 3.0]]
    assert r["final_good"]["distance"].tolist() == [10.0]


def test_track_map_points_first_hit_and_defaults(oracle):
    last = np.zeros(4, oracle.DM_DTYPE); cur = np.zeros(5, oracle.DM_DTYPE)
    last["queryIdx"] = [7, 3, 9, 3]; last["trainIdx"] = [70, 30, 90, 31]
    cur["trainIdx"] = [3, 5, 3, 7, 8]; cur["queryIdx"] = [100, 101, 102, 103, 104]
    out = oracle.track_map_points(last, cur)
    # last 0 (q 7) -> cur 3;  last 1 (q 3) -> FIRST cur with train 3 = cur 0;  last 2 (q 9) none;  last 3 (q 3) -> cur 0 again
    assert out["queryIdx"].tolist() == [103, 100, 100] and out["trainIdx"].tolist() == [70, 30, 31]
    assert (out["imgIdx"] == -1).all() and (out["distance"] == np.finfo(np.float32).max).all()
    assert len(oracle.track_map_points(last[:0], cur)) == 0 and len(oracle.track_map_points(last, cur[:0])) == 0


def test_in_match_vec_first_position(oracle):
    v = np.zeros(5, oracle.DM_DTYPE); v["queryIdx"] = [4, 2, 4, 0, 9]
    assert oracle.in_match_vec(v, 6).tolist() == [3, -1, 1, -1, 0, -1]
    assert oracle.in_match_vec(v[:0], 3).tolist() == [-1, -1, -1]


def test_manage_map_points_decisions_and_world_points(oracle):
    # src/Map.cpp:125-149.  5 keypoints; last_match_vec query ids [3, 1, 3]: keypoints 1 and 3 are tracked (3 -> first position 0)
    cam = oracle.camera(0.0, 0.0, 1.0, 1.0, 65535.0)                     # depth value = raw, x = px * d, y = py * d
    depth = np.array([[2, 4, 6, 8, 10]], np.uint16)
    kps = _kps(oracle, [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0)])
    v = np.zeros(3, oracle.DM_DTYPE); v["queryIdx"] = [3, 1, 3]; v["trainIdx"] = [30, 10, 31]
    # previous key frame's observations {map_point_id, point_id}: point 3 seen as map point 77 first, then 78; point 1 never
    obs = [(70, 0), (77, 3), (78, 3), (79, 9)]
    T = np.array([[0, -1, 0, 10], [1, 0, 0, 20], [0, 0, 1, 30], [0, 0, 0, 1]], np.float32)     # rotate 90 deg about z, then translate
    match_num, map_point, world = oracle.manage_map_points(kps, depth, cam, T, v, obs)
    assert match_num.tolist() == [-1, 1, -1, 0, -1]
    assert map_point.tolist() == [-1, -1, -1, 77, -1]                     # keypoint 1: no observation (undefined in the reference) -> -1
    # keypoint 2: p = (2 * 6, 0, 6) -> (-0 + 10, 12 + 20, 6 + 30);  keypoint 4: p = (40, 0, 10) -> (10, 60, 40);  tracked rows stay 0
    assert world.tolist() == [[10.0, 20.0, 32.0], [0.0, 0.0, 0.0], [10.0, 32.0, 36.0], [0.0, 0.0, 0.0], [10.0, 60.0, 40.0]]
    # float order e0 + (e1 + e2), then + t: with e0 = 2^24, e1 = e2 = 1 the pair sum 2 survives, one at a time it would not
    T2 = np.eye(4, dtype=np.float32); T2[0, :3] = [2.0 ** 24, 1.0, 1.0]
    cam2 = oracle.camera(0.0, 0.0, 1.0, 1.0, 65535.0)
    k2 = _kps(oracle, [(1, 1)])                                           # p = (1 * 1, 1 * 1, 1) with depth 1
    _, _, w2 = oracle.manage_map_points(k2, np.array([[0, 0], [0, 1]], np.uint16), cam2, T2, v[:0], [])
    assert w2[0, 0] == np.float32(2.0 ** 24 + 2.0)
    n0 = oracle.manage_map_points(kps[:0], depth, cam, T, v, obs)
    assert all(len(a) == 0 for a in n0)


def test_rgb_cloud_known_values(oracle):
    depth = np.array([[0, 65535], [13107, 100]], np.uint16)             # (y, x)
    bgr = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    c = oracle.rgb_cloud(depth, bgr, 0.5, 0.25, 2.0, 4.0, 10.0)
    # index u*h + v: (u0,v0) (u0,v1) (u1,v0) (u1,v1)
    assert c["z"].tolist() == [0.0, float(np.float32(13107 * (10.0 / 65535.0))), 10.0, float(np.float32(100 * (10.0 / 65535.0)))]
    f = np.float32
    assert c["x"][2] == (f(1.0) - f(0.5)) * f(10.0) * f(0.5) and c["y"][2] == (f(0.0) - f(0.25)) * f(10.0) * f(0.25)
    assert c["b"].tolist() == [0, 6, 3, 9] and c["g"].tolist() == [1, 7, 4, 10] and c["r"].tolist() == [2, 8, 5, 11]
    assert (c["a"] == 255).all() and (c["w"] == 1.0).all() and (c["pad"] == 0).all()
    # the driver multiplies in double and narrows once: differs from the float product for some raw values
    s32 = np.float32(10.0 / 65535.0)
    raw = np.arange(1, 65536, dtype=np.uint16).reshape(1, -1)
    z = oracle.rgb_cloud(raw, np.zeros((1, 65535, 3), np.uint8), 0, 0, 1, 1, 10.0)["z"]
    assert np.array_equal(z, (raw[0].astype(np.float64) * (10.0 / 65535.0)).astype(np.float32))
    assert (z != raw[0].astype(np.float32) * s32).any()


def _cloud(oracle, xyz, rgb=None):
    c = np.zeros(len(xyz), oracle.CLOUD_DTYPE)
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    c["w"] = 1.0; c["a"] = 255
    if rgb is not None:
        rgb = np.asarray(rgb, np.uint8).reshape(-1, 3)
        c["r"], c["g"], c["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    return c


def test_voxel_fuse_known_values(oracle):
    """Map::MapPointCloudFusion (src/Map.cpp:164-197), hand-derived.  Parity unpinned: PCL restated (oracle/frame_oracle.c)."""
    f = np.float32
    I = np.eye(4, dtype=np.float32)
    # two points in one 1-cm voxel -> centroid; colour channels averaged and truncated; data[3] = 1, alpha 255
    c = _cloud(oracle, [[0.001, 0.002, 0.003], [0.004, 0.006, 0.008], [0.5, 0.5, 0.5]], [[10, 20, 30], [11, 21, 33], [1, 2, 3]])
    r = oracle.voxel_fuse(c[:0], c, I, 0.01)
    assert len(r) == 2
    assert r["x"][0] == (f(0.001) + f(0.004)) / f(2) and r["y"][0] == (f(0.002) + f(0.006)) / f(2) and r["z"][0] == (f(0.003) + f(0.008)) / f(2)
    assert (r["r"][0], r["g"][0], r["b"][0], r["a"][0]) == (10, 20, 31, 255) and (r["w"] == 1.0).all() and (r["pad"] == 0).all()
    assert (r["x"][1], r["r"][1]) == (f(0.5), 1)
    # a point exactly on a leaf boundary belongs to the upper voxel (floor): 0.02 * 100 = 2.0 -> cell 2; 0.0199 -> cell 1
    c = _cloud(oracle, [[0.0, 0, 0], [0.0199, 0, 0], [0.02, 0, 0], [0.0201, 0, 0]])
    r = oracle.voxel_fuse(c[:0], c, I, 0.01)
    assert len(r) == 3 and r["x"][2] == (f(0.02) + f(0.0201)) / f(2)
    # output order is ascending voxel index x + y * dx + z * dx * dy, not input order
    c = _cloud(oracle, [[0.0, 0.0, 0.05], [0.0, 0.05, 0.0], [0.05, 0.0, 0.0], [0.0, 0.0, 0.0]])
    r = oracle.voxel_fuse(c[:0], c, I, 0.01)
    assert r["x"].tolist() == [0.0, f(0.05), 0.0, 0.0] and r["y"].tolist() == [0.0, 0.0, f(0.05), 0.0] and r["z"].tolist() == [0.0, 0.0, 0.0, f(0.05)]
    # the transform as written: Affine3f::Identity().rotate(R).translate(t) = R * (p + t), NOT R * p + t (:177-181)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]               # +90 degrees about z
    T[:3, 3] = [1.0, 0.0, 0.0]
    c = _cloud(oracle, [[2.0, 0.0, 0.0]])
    r = oracle.voxel_fuse(c[:0], c, T, 0.01)
    assert (r["x"][0], r["y"][0], r["z"][0]) == (0.0, 3.0, 0.0)      # R * ((2,0,0) + (1,0,0)); R * p + t would be (1, 2, 0)
    # the map's own points are not transformed again, and they come first in a voxel's sum
    m0 = _cloud(oracle, [[0.0, 3.0011, 0.0]])
    r = oracle.voxel_fuse(m0, c, T, 0.01)
    assert len(r) == 1 and r["y"][0] == (f(3.0011) + f(3.0)) / f(2)
    # sums are sequential float additions in cloud order: (2^24 + 1) + 1 = 2^24 in float, (1 + 1) + 2^24 = 2^24 + 2
    big = _cloud(oracle, [[2.0 ** 24, 0, 0], [1.0, 0, 0], [1.0, 0, 0]])
    r = oracle.voxel_fuse(big[:0], big, I, 1e9)
    assert len(r) == 1 and r["x"][0] == f(2.0 ** 24) / f(3)
    r = oracle.voxel_fuse(big[:0], big[::-1], I, 1e9)
    assert len(r) == 1 and r["x"][0] == f(2.0 ** 24 + 2) / f(3)
    # a leaf too small for 32-bit voxel indices: PCL warns and returns the input unchanged
    far = _cloud(oracle, [[0, 0, 0], [100.0, 100.0, 100.0], [0.0001, 0, 0]])
    r = oracle.voxel_fuse(far[:0], far, I, 0.01)
    assert len(r) == 3 and np.array_equal(r["x"], far["x"])
    assert len(oracle.voxel_fuse(c[:0], c[:0], I, 0.01)) == 0
