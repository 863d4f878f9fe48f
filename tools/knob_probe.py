"""Step rate of the pipelined front end under a developer knob:  python tools/knob_probe.py W H NFEAT B stereo(0/1) KNOB V1 V2 ..."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd")
pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf, B, stereo, knob = (int(x) for x in sys.argv[1:7])
vals = [int(x) for x in sys.argv[7:]]
if stereo:
    pairs = [synth.stereo_pair(w, h, i) for i in range(B)]
    L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
else:
    L, R = np.stack([synth.frame(w, h, i) for i in range(B)]), None
for v in vals:
    pkg.set_default_option(knob, v)
    late = os.environ.get("LATE")
    fe = pl.FrontEnd(w, h, nf, bool(stereo), B, stereo_late=None if late is None else bool(int(late))).upload(L, R)
    best = 1e9
    for rep in range(3):
        for i in range(10):
            fe.step(i)
        fe.drain()
        t = time.perf_counter()
        for i in range(40):
            fe.step(i)
        fe.drain()
        best = min(best, (time.perf_counter() - t) / 40)
    print("knob %d = %d: %.4f ms/step  %.0f per s" % (knob, v, best * 1e3, B / best), flush=True)
    del fe
pkg.set_default_option(knob, 0)
