"""The C++ adapter classes (include/ssvio/Feature.h, FeatureMatching.h) used the way SSVIO's
Odometry/Frame use the reference classes; outputs compared with the oracle bit for bit."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_test")


def test_adapters_compile_and_link_against_the_cabi():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adapters_match_oracle(tmp_path):
    from ssvio_amd import synth
    from oracle import oracle as O
    assert os.path.exists(EXE), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    a, b = synth.frame_pair(4, 240, 320)
    raw = tmp_path / "frames.raw"
    raw.write_bytes(a.tobytes() + b.tobytes())
    out = str(tmp_path / "o")
    from ssvio_amd.vocabulary import Vocabulary
    par, vdesc, vw = synth.vocabulary_tables(9, 10, 5, stop_p=0.02)
    voc = Vocabulary(10, 5, par, vdesc, vw)
    voc.save_txt(str(tmp_path / "voc.txt"))
    depth = np.stack([synth.depth_image(5, 240, 320), synth.depth_image(6, 240, 320)])
    (tmp_path / "depth.raw").write_bytes(depth.tobytes())
    voc.save_dbow3(str(tmp_path / "voc.dbow3"))
    import struct
    good = (tmp_path / "voc.dbow3").read_bytes()
    bad = []
    for i, blob in enumerate([good[:len(good) // 3],                                              # truncated
                           good[:9] + struct.pack("<I", 1 << 30) + good[13:],                  # node count far beyond the file
                           good[:8] + b"\x01" + good[9:],                                      # compressed flag
                           good[:29 + 4] + struct.pack("<I", 7) + good[29 + 8:]]):             # first node's parent is not before it
        (tmp_path / ("bad%d.dbow3" % i)).write_bytes(blob); bad.append(str(tmp_path / ("bad%d.dbow3" % i)))
    r = subprocess.run([EXE, str(raw), "320", "240", "1200", out, str(tmp_path / "voc.txt"), str(tmp_path / "depth.raw"), str(tmp_path / "voc.dbow3")] + bad,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "binary_voc=same" in r.stdout
    assert "pyr0=320x240" in r.stdout
    ex = O.Extractor(1200, 1.2, 8, 20, 7)
    ka, da = ex.extract(a); kb, db = ex.extract(b)
    rd = lambda suf: open(out + suf, "rb").read()
    assert rd(".ka") == ka.tobytes() and rd(".kb") == kb.tobytes()
    assert rd(".da") == da.tobytes() and rd(".db") == db.tobytes()
    assert rd(".m") == O.match_bruteforce(db, da).tobytes()
    assert rd(".m2") == O.match_knn2_ratio(db, da, 50, 0.6).tobytes()
    # MatchByProject with the inputs the adapter test built (reference points dumped by it)
    p3 = np.frombuffer(rd(".p3"), np.float64).reshape(-1, 3)
    K = np.array([[260.0, 0, 160.0], [0, 260.0, 120.0], [0, 0, 1.0]]); R = np.eye(3); t = np.array([0.02, -0.01, 0.0])
    p2 = np.stack([kb["x"].astype(np.float64), kb["y"].astype(np.float64)], 1)
    want = O.match_project(p3, da, p2, db, 12.0, K, R, t, 50, 0.6)
    assert rd(".m4") == want.tobytes() and len(want) > 10
    # the live path: MatchByDBoW on the text vocabulary, correspondence lists, match chaining
    ov = O.Vocabulary(10, 5, par, vdesc, vw, voc.word)
    bow = O.match_dbow(ov, db, da, 50, 0.6, 4)
    assert rd(".bow") == bow.tobytes() and len(bow) > 10
    word, node, weight = ov.transform(db, 4)
    fv = Vocabulary.feature_vector(node, weight)
    flat = []
    for nid, idx in fv.items():
        flat += [nid, len(idx)] + idx
    assert rd(".fv1") == np.array(flat, np.uint32).tobytes()
    bw, bv = ov.bow_vector(word, weight)
    assert rd(".bv1") == np.stack([bw.astype(np.float64), bv], 1).tobytes()
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    c = O.assemble_correspondences(bow, ka, depth[0], kb, depth[1], cam)
    assert rd(".l3") == c["last3d"].tobytes() and rd(".c2") == c["cur2d"].tobytes() and rd(".fin") == c["final_good"].tobytes()
    bf = O.match_bruteforce(db, da)
    assert rd(".trk") == O.track_map_points(bf, c["final_good"]).tobytes()
    assert rd(".inm") == O.in_match_vec(bf, len(kb)).tobytes()
    trk = O.track_map_points(bf, c["final_good"])
    obs = np.stack([1000 + np.arange(0, len(bf), 2), bf["queryIdx"][::2]], axis=1).astype(np.int32)
    T = np.array([[0, -1, 0, 0.25], [1, 0, 0, -0.5], [0, 0, 1, 1.5], [0, 0, 0, 1]], np.float32)
    mnum, mpid, npt = O.manage_map_points(kb, depth[1], cam, T, trk, obs)
    assert rd(".mnum") == mnum.tobytes() and rd(".mpid") == mpid.tobytes() and rd(".npt") == npt.tobytes()
    assert (mnum >= 0).any() and (mpid >= 0).any() and (mnum < 0).any()
    # OpenCV-compat cross-check and two key frames of Map::MapPointCloudFusion through the C++ wrappers
    assert rd(".m5") == O.match_crosscheck_cv(db, da).tobytes()
    gray3 = lambda g: np.repeat(g[:, :, None], 3, axis=2)
    cA = O.rgb_cloud(depth[0], gray3(a), 160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    cB = O.rgb_cloud(depth[1], gray3(b), 160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    m1 = O.voxel_fuse(cA[:0], cA, np.eye(4, dtype=np.float32), 0.05)
    m2 = O.voxel_fuse(m1, cB, T, 0.05)
    assert rd(".map") == m2.tobytes() and 1000 < len(m2) < len(cA) + len(cB)
