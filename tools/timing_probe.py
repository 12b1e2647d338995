"""GPU-box probe: wall time per step without per-kernel events, at several batch sizes (run under rocprofv3
--kernel-trace and summarise with tools/trace_summary.py for true per-launch kernel durations)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import synth
BATCHES = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
B = max(BATCHES)
frames = synth.frame_stream(1000, B, 480, 640)
d = torch.from_numpy(frames).cuda()
ex = ssvio_amd.ORBextractor(1000, 1.2, 8, 20, 7)
fm = ssvio_amd.FeatureMatching(None, {})
for b in BATCHES:
    pairs = np.array([[f, (f - 1) % b] for f in range(b)], np.int32)
    def step():
        ex.extract_batch_dev(d, b, 640, 480)
        kp, de, n, cap = ex.device_results()
        fm.match_batch_dev(de, cap, cap, n, pairs, 0); fm.sync()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("batch %d  ms/step %.3f fps %.0f" % (b, dt * 1e3, b / dt))
ex.close(); fm.close()
