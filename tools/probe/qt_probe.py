import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth, _lib as L
W, H, NF = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (640, 480, 1000)))     # needs a -DSSVIO_QT_TIMING build of kernels_quadtree.hip
frames = synth.frame_stream(1000, 2, H, W)
d = torch.from_numpy(frames).cuda()
ex = ssvio_amd.ORBextractor(NF, 1.2, 8, 20, 7)
for _ in range(5):
    ex.extract_batch_dev(d, 1, W, H)
out = (C.c_ulonglong * 64)()
lib = L.lib()
print("rc", lib.ssvio_debug_qt_times(out))
t = list(out)
t0 = t[0]
print("gather", (t[1]-t0)*10, "ns; roots", (t[2]-t[1])*10)
for r in range(10):
    a, b = t[3+2*r], t[4+2*r]
    if a > t0 and b > a: print("round", r, "node phase", (b-a)*10, "ns", "prev key phase", (a - (t[2] if r == 0 else t[4+2*(r-1)]))*10)
print("final", (t[41]-t[40])*10, "total", (t[41]-t0)*10)
