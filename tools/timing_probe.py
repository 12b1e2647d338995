import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth, _lib as L
B=64
frames = synth.frame_stream(1000, B, 480, 640)
d = torch.from_numpy(frames).cuda()
ex = ssvio_amd.ORBextractor(1000,1.2,8,20,7)
fm = ssvio_amd.FeatureMatching(None, {})
pairs = np.array([[f,(f-1)%B] for f in range(B)], np.int32)
def step(match=True):
    ex.extract_batch_dev(d, B, 640, 480)
    if match:
        kp, de, n, cap = ex.device_results()
        fm.match_batch_dev(de, cap, cap, n, pairs, 0); fm.sync()
for _ in range(3): step()
for match in (True, False):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): step(match)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print("match=%s  ms/step %.3f  fps %.0f"%(match, dt*1e3, B/dt))
for thr in (1,4,8,16):
    os.environ["SSVIO_HOST_THREADS"]=str(thr)
    e2 = ssvio_amd.ORBextractor(1000,1.2,8,20,7)
    for _ in range(3): e2.extract_batch_dev(d, B, 640, 480)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): e2.extract_batch_dev(d, B, 640, 480)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print("threads %d  ms/step %.3f"%(thr, dt*1e3))
    e2.close()
for b in (1,4,16):
    for _ in range(3): ex.extract_batch_dev(d, b, 640, 480)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): ex.extract_batch_dev(d, b, 640, 480)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
    print("batch %d  ms/call %.3f fps %.0f"%(b, dt*1e3, b/dt))
