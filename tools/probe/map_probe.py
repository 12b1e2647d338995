"""GPU-box probe: Map::MapPointCloudFusion, 8 key-frame clouds of 640x480 points fused in turn into the resident map, 3 rounds.
Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations:  python tools/probe/map_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth

H, W = 480, 640
fm = ssvio_amd.FeatureMatching(None, {})
cam = np.array([W / 2 + 0.59, H / 2 + 0.48, 0.79 * W, 0.79 * W, np.float32(10.0 / 65535.0)], np.float32)
bgr = synth.to_rgb(synth.texture(5, H, W), 1)
clouds = [torch.from_numpy(fm.RGBCloud(synth.depth_image(300 + k, H, W), bgr, cam, 10.0).view(np.uint8)).cuda() for k in range(8)]
Ts = [np.eye(4, dtype=np.float32) for _ in range(8)]
for k, T in enumerate(Ts):
    T[:3, 3] = (0.05 * k, 0.01 * k, 0.02 * k)
fm.set_profiling(True)
for rep in range(3):
    fm.MapSet(None)
    wall = []
    for c, T in zip(clouds, Ts):
        t0 = time.perf_counter(); fm.MapFuse(c, T, 0.02); wall.append((time.perf_counter() - t0) * 1e3)
    print("round %d: ms per key frame" % rep, " ".join("%.2f" % w for w in wall), "map", len(fm.MapFetch()))
t = fm.kernel_times()
print({k: round(v[0] * 1e3 / max(v[1], 1), 1) for k, v in t.items()})
