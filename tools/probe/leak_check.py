"""GPU-box probe: device memory after repeated create / use / destroy cycles of both handles (a leak shows as a falling free count)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth, _lib as L
frames = torch.from_numpy(synth.frame_stream(5, 8, 480, 640)).cuda()
pairs = np.array([[f, (f - 1) % 8] for f in range(8)], np.int32)
depth = synth.depth_image(1, 480, 640); bgr = synth.to_rgb(synth.texture(5, 480, 640), 1)
cam = np.array([320.5, 240.5, 500, 500, np.float32(10.0 / 65535.0)], np.float32)
def cycle():
    g = ssvio_amd.ORBextractor(1000, 1.2, 8, 20, 7); fm = ssvio_amd.FeatureMatching(None, {})
    fm.set_stream(g.get_stream())
    g.extract_batch_dev(frames, 8, 640, 480, wait=False)
    _, d_desc, d_n, cap = g.device_results()
    for mode in (L.MODE_BRUTEFORCE, L.MODE_KNN2_RATIO, L.MODE_CROSSCHECK, L.MODE_CROSSCHECK_CV):
        fm.match_batch_dev(d_desc, cap, cap, d_n, pairs, mode)
    fm.sync()
    fm.MapSet(None); fm.MapFuseFrame(depth, bgr, cam, 10.0, np.eye(4, dtype=np.float32), 0.05)
    g.close(); fm.close()
for _ in range(3): cycle()
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
for i in range(60): cycle()
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB, after 60 cycles %.1f MB, difference %.2f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
