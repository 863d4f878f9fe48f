import sys, os, importlib, time
sys.path.insert(0, "/root/repo")
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import oracle
oracle.build()
for (w, h, nf) in ((3840, 2160, 5000), (4095, 1200, 8000), (2000, 4095, 3000), (4096, 1000, 1000)):
    img = synth.frame(w, h, 7)
    try:
        t0 = time.time(); gk, gd = pkg.ORBextractor(nf, 1.2, 8, 20, 7)(img); t1 = time.time()
    except Exception as e:
        print(w, h, nf, "GPU error:", e); continue
    ok, od = oracle.Extractor(nf, 1.2, 8, 20, 7).extract(img); t2 = time.time()
    same = len(gk) == len(ok) and np.array_equal(gd, od) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8))
    print(w, h, nf, "n", len(gk), len(ok), "identical" if same else "MISMATCH", "gpu %.2fs cpu %.2fs" % (t1 - t0, t2 - t1))
