"""probe: which library call, made before torch initialises its HIP context, leaves torch without a GPU?"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CASES = {
    "match": "fm = S.FeatureMatching(None, {}); fm.MatchByBruteForce(d, d); fm.close()",
    "vocab_create": "v.handle(0)",
    "vocab_create_destroy": "v.handle(0); v.close()",
    "transform": "fm = S.FeatureMatching(v, {}); fm.ComputeBoWVector(d); fm.close()",
    "nothing": "pass",
}
if len(sys.argv) > 1:
    import numpy as np
    import ssvio_amd as S
    from ssvio_amd import synth
    par, desc, w = synth.vocabulary_tables(1, 4, 3)
    v = S.Vocabulary(4, 3, par, desc, w)
    d = synth.descriptors_near_leaves(1, 64, par, desc)
    exec(CASES[sys.argv[1]])
    import torch
    try:
        torch.zeros(4).cuda(); print(sys.argv[1], "-> torch ok")
    except Exception as e:
        print(sys.argv[1], "-> torch FAILED:", str(e).splitlines()[0])
else:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=120)
        print(r.stdout.strip() or r.stderr.strip()[-300:])
