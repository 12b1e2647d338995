"""Writes tests/golden/oracle_digests.json: SHA-256 digests of the oracle's outputs on seeded synthetic inputs.
The reference ships no fixtures (parity unpinned), so these do not pin the oracle to the reference; they pin it to
ITSELF across rounds and machines: a later edit that changes any output bit of the restatement (or of the synthetic input
generator) shows up as a digest mismatch in tests/test_oracle_digests.py.  Regenerate only for a deliberate change of the
oracle, and say so in the commit.   usage: python tools/gen_oracle_digests.py"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ssvio_amd import synth
from oracle import oracle as O


def dg(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    return h.hexdigest()


def compute():
    out = {}
    a, b = synth.frame_pair(11, 240, 320)
    out["input_frame_pair_11_240x320"] = dg(a, b)
    ex = O.Extractor(1200, 1.2, 8, 20, 7)
    ka, da = ex.extract(a); kb, db = ex.extract(b)
    out["extract_1200_a"] = {"n": len(ka), "kps": dg(ka), "desc": dg(da)}
    out["extract_1200_b"] = {"n": len(kb), "kps": dg(kb), "desc": dg(db)}
    out["level_images_b"] = [dg(ex.level_image(l)) for l in range(8)]
    out["level_blur_b"] = [dg(ex.level_blur(l)) for l in range(8)]
    rgb = synth.to_rgb(a, 1)
    kr, dr = ex.extract_rgb(rgb)
    out["extract_rgb_a"] = {"n": len(kr), "kps": dg(kr), "desc": dg(dr)}
    vga = synth.frame_stream(1000, 2)
    e2 = O.Extractor(1000, 1.2, 8, 20, 7)
    k0, d0 = e2.extract(vga[0]); k1, d1 = e2.extract(vga[1])
    out["input_stream_1000_vga"] = dg(vga)
    out["extract_vga_1000"] = {"n": [len(k0), len(k1)], "kps": dg(k0, k1), "desc": dg(d0, d1)}
    m = O.match_bruteforce(db, da)
    out["match_bruteforce"] = {"n": len(m), "dg": dg(m)}
    out["match_knn2_ratio"] = dg(O.match_knn2_ratio(db, da, 50, 0.6))
    out["match_crosscheck"] = dg(O.match_crosscheck(db, da))
    par, vdesc, vw = synth.vocabulary_tables(5, 10, 4, stop_p=0.02)
    word = np.full(len(par), -1, np.int32); leaf = np.ones(len(par), bool); leaf[par[1:]] = False; word[leaf] = np.arange(int(leaf.sum()))
    voc = O.Vocabulary(10, 4, par, vdesc, vw, word)
    out["input_vocabulary_5_10_4"] = dg(par, vdesc, vw)
    w_, n_, wt_ = voc.transform(db, 4)
    out["bow_transform_b"] = dg(w_, n_, wt_)
    md = O.match_dbow(voc, db, da, 50, 0.6, 4)
    out["match_dbow"] = {"n": len(md), "dg": dg(md)}
    depth_a, depth_b = synth.depth_image(12, 240, 320), synth.depth_image(13, 240, 320)
    cam = O.camera(160.5912, 120.4792, 253.0589, 254.1649, 10.0)
    c = O.assemble_correspondences(md, ka, depth_a, kb, depth_b, cam)
    out["assemble"] = {k: dg(v) for k, v in sorted(c.items())}
    out["track_map_points"] = dg(O.track_map_points(m, c["final_good"]))
    trk = O.track_map_points(m, c["final_good"])
    obs = np.stack([1000 + np.arange(0, len(m), 2), m["queryIdx"][::2]], axis=1).astype(np.int32)
    T = np.array([[0, -1, 0, 0.25], [1, 0, 0, -0.5], [0, 0, 1, 1.5], [0, 0, 0, 1]], np.float32)
    out["manage_map_points"] = dg(*O.manage_map_points(kb, depth_b, cam, T, trk, obs))
    out["rgb_cloud"] = dg(O.rgb_cloud(depth_a, rgb, 160.5912, 120.4792, 253.0589, 254.1649, 10.0))
    p3 = np.stack([(ka["x"] - 160.0) / 260.0 * 2.0, (ka["y"] - 120.0) / 260.0 * 2.0, np.full(len(ka), 2.0)], 1).astype(np.float64)
    p2 = np.stack([kb["x"], kb["y"]], 1).astype(np.float64)
    K = np.array([[260.0, 0, 160.0], [0, 260.0, 120.0], [0, 0, 1.0]])
    out["match_project"] = dg(O.match_project(p3, da, p2, db, 12.0, K, np.eye(3), np.array([0.02, -0.01, 0.0]), 50, 0.6))
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
    json.dump(compute(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
