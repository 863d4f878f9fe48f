"""Where should the pyramid built ahead start?  bench step rate with developer knob 10 = 0 (behind FAST), 1 (behind the quad-tree),
2 (behind the descriptors) and with FrontEnd(prefetch=False).   python tools/gate_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd")
pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf, B = 1241, 376, 1000, 64
pairs = [synth.stereo_pair(w, h, i) for i in range(B)]
L = np.stack([p[0] for p in pairs])
R = np.stack([p[1] for p in pairs])
for name, pf, knob in (("no prefetch", False, 0), ("behind FAST", True, 0), ("behind quad-tree", True, 1), ("behind descriptors", True, 2)):
    pkg.set_default_option(10, knob)
    fe = pl.FrontEnd(w, h, nf, True, B, prefetch=pf).upload(L, R)
    best = 1e9
    for rep in range(3):
        for i in range(10):
            fe.step(i)
        fe.drain()
        t = time.perf_counter()
        for i in range(60):
            fe.step(i)
        fe.drain()
        best = min(best, (time.perf_counter() - t) / 60)
    print("%-20s %.4f ms/step  %.0f frames/s" % (name, best * 1e3, B / best), flush=True)
    del fe
