import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth
for (h, w, nf) in ((240, 320, 500), (480, 640, 1000), (1080, 1920, 4000), (2160, 3840, 8000)):
    img = synth.texture(5, h, w)
    ex = ssvio_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    d = torch.from_numpy(img[None]).cuda()
    for _ in range(5): ex.extract_batch_dev(d, 1, w, h)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): ex.extract_batch_dev(d, 1, w, h)
    dt = (time.perf_counter() - t) / 50
    print("%dx%d device-resident single frame: %.1f us" % (w, h, dt * 1e6))
    ex.close()
