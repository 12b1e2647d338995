"""GPU-box probe: the floor of k_describe's HBM fetch for the bench's 64-frame batch -- the union of the cache lines its
patches touch, every line counted once per frame (a frame's levels stay in one XCD's L2 during the launch).  Uses the library
for the keypoints and the pyramid layout of DESIGN.md section 2 (row pitch align64(w) + 128, 64 bytes of slack before a row's
first pixel, level offsets multiples of 64 -- only the position of a row inside the line grid matters here).
  python tools/probe/describe_lines.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth

B, W, H = 64, 640, 480
ex = ssvio_amd.ORBextractor(1000, 1.2, 8, 20, 7)
d = torch.from_numpy(synth.frame_stream(1000, B, H, W)).cuda()
ex.extract_batch_dev(d, B, W, H)
kps, _, n = ex.fetch_results(B)
scale = ex.GetScaleFactors()
for line in (128, 64):
    total = 0
    for f in range(B):
        k = kps[f, :n[f]]
        for l in range(8):
            lw = ex.level_geometry(W, H, l)["w"]
            pitch = (lw + 63) // 64 * 64 + 128
            s = k[k["octave"] == l]
            x = np.rint(s["x"] / scale[l]).astype(np.int64); y = np.rint(s["y"] / scale[l]).astype(np.int64)
            for rad_r, c0, c1 in ((18, -18, 18), (14, -16, 19)):        # blurred window 37 x 37 (+ alignment), intensity rows -14..14, 36 bytes
                rows = y[:, None] + np.arange(-rad_r, rad_r + 1)[None, :]
                first = (rows * pitch + 64 + x[:, None] + c0) // line
                last = (rows * pitch + 64 + x[:, None] + c1) // line
                ids = np.concatenate([first.ravel(), last.ravel()])
                total += len(np.unique(ids)) * line
    alg = int(n.sum()) * (600 + 512 + 60)
    print("line %3d B: unique lines of the 64-frame batch %.1f MB; algorithmic %.1f MB (%d keypoints); ratio %.2f" %
          (line, total / 1e6, alg / 1e6, int(n.sum()), total / alg))
