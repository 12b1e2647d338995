"""GPU: the HIP path reproduces the committed golden digests (tests/golden/oracle_digests.json, written by the oracle through
tools/gen_oracle_digests.py) on the same seeded inputs -- without calling the oracle.  Together with test_oracle_digests.py
(oracle == fixtures, CPU) this closes the triangle GPU == fixtures == oracle."""
import hashlib
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dg(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    return h.hexdigest()


def test_gpu_outputs_equal_the_committed_digests():
    import ssvio_amd as S
    from ssvio_amd import synth
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")))
    a, b = synth.frame_pair(11, 240, 320)
    vga = synth.frame_stream(1000, 2)
    par, vdesc, vw = synth.vocabulary_tables(5, 10, 4, stop_p=0.02)
    if dg(a, b) != want["input_frame_pair_11_240x320"] or dg(vga) != want["input_stream_1000_vga"] or dg(par, vdesc, vw) != want["input_vocabulary_5_10_4"]:
        pytest.skip("synthetic inputs differ on this machine (numpy / libm), digests not comparable")
    ex = S.ORBextractor(1200, 1.2, 8, 20, 7)
    ka, da = ex.Extract(a); kb, db = ex.Extract(b)
    assert {"n": len(ka), "kps": dg(ka), "desc": dg(da)} == want["extract_1200_a"]
    assert {"n": len(kb), "kps": dg(kb), "desc": dg(db)} == want["extract_1200_b"]
    assert [dg(ex.debug_level_image(l)) for l in range(8)] == want["level_images_b"]
    assert [dg(ex.debug_level_image(l, blurred=True)) for l in range(8)] == want["level_blur_b"]
    rgb = synth.to_rgb(a, 1)
    kr, dr = ex.ExtractRGB(rgb)
    assert {"n": len(kr), "kps": dg(kr), "desc": dg(dr)} == want["extract_rgb_a"]
    ex.close()
    e2 = S.ORBextractor(1000, 1.2, 8, 20, 7)
    k0, d0 = e2.Extract(vga[0]); k1, d1 = e2.Extract(vga[1])
    assert {"n": [len(k0), len(k1)], "kps": dg(k0, k1), "desc": dg(d0, d1)} == want["extract_vga_1000"]
    e2.close()
    voc = S.Vocabulary(10, 4, par, vdesc, vw)
    fm = S.FeatureMatching(voc, {"Threshold": 50, "NNRatio": 0.6})
    m = fm.MatchByBruteForce(db, da)[1]
    assert {"n": len(m), "dg": dg(m)} == want["match_bruteforce"]
    assert dg(fm.MatchByKnnRatio(db, da)[1]) == want["match_knn2_ratio"]
    assert dg(fm.MatchByCrossCheck(db, da)[1]) == want["match_crosscheck"]
    assert dg(*fm.ComputeBoWVector(db, 4)[2]) == want["bow_transform_b"]
    md = fm.MatchByDBoW(db, da)[1]
    assert {"n": len(md), "dg": dg(md)} == want["match_dbow"]
    depth_a, depth_b = synth.depth_image(12, 240, 320), synth.depth_image(13, 240, 320)
    cam = np.array([160.5912, 120.4792, 253.0589, 254.1649, np.float32(np.float64(np.float32(10.0)) / 65535.0)], np.float32)
    c = fm.AssembleCorrespondences(md, ka, depth_a, kb, depth_b, cam)
    assert {k: dg(v) for k, v in sorted(c.items())} == want["assemble"]
    assert dg(fm.TrackMapPoints(m, c["final_good"])) == want["track_map_points"]
    trk = fm.TrackMapPoints(m, c["final_good"])
    obs = np.stack([1000 + np.arange(0, len(m), 2), m["queryIdx"][::2]], axis=1).astype(np.int32)
    T = np.array([[0, -1, 0, 0.25], [1, 0, 0, -0.5], [0, 0, 1, 1.5], [0, 0, 0, 1]], np.float32)
    assert dg(*fm.ManageMapPoints(kb, depth_b, cam, T, trk, obs)) == want["manage_map_points"]
    assert dg(fm.RGBCloud(depth_a, rgb, cam, 10.0)) == want["rgb_cloud"]
    p3 = np.stack([(ka["x"] - 160.0) / 260.0 * 2.0, (ka["y"] - 120.0) / 260.0 * 2.0, np.full(len(ka), 2.0)], 1).astype(np.float64)
    p2 = np.stack([kb["x"], kb["y"]], 1).astype(np.float64)
    K = np.array([[260.0, 0, 160.0], [0, 260.0, 120.0], [0, 0, 1.0]])
    assert dg(fm.MatchByProject(p3, da, p2, db, 12.0, K, np.eye(3), np.array([0.02, -0.01, 0.0]))[1]) == want["match_project"]
    fm.close(); voc.close()
