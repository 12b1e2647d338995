"""GPU-box probe: per-call cost of the C++ adapter classes (include/ssvio/) on a VGA frame pair, 1000 features, a k = 10, L = 5
vocabulary -- what an unmodified Odometry would pay per call.  usage: python tools/adapter_time.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ssvio_amd import synth
from ssvio_amd.vocabulary import Vocabulary
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_test")
with tempfile.TemporaryDirectory() as d:
    a, b = synth.frame_pair(4, 480, 640)
    open(os.path.join(d, "frames.raw"), "wb").write(a.tobytes() + b.tobytes())
    par, vdesc, vw = synth.vocabulary_tables_full(3, 10, 5)
    Vocabulary(10, 5, par, vdesc, vw).save_dbow3(os.path.join(d, "voc.dbow3"))
    depth = np.stack([synth.depth_image(5, 480, 640), synth.depth_image(6, 480, 640)])
    open(os.path.join(d, "depth.raw"), "wb").write(depth.tobytes())
    r = subprocess.run([EXE, os.path.join(d, "frames.raw"), "640", "480", "1000", os.path.join(d, "o"), os.path.join(d, "voc.dbow3"),
                        os.path.join(d, "depth.raw")], env=dict(os.environ, SSVIO_ADAPTER_TIME="1"), capture_output=True, text=True, timeout=300)
    print(r.stdout, end="")
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(r.returncode)
