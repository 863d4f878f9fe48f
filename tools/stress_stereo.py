"""Randomised parity stress of Frame::ComputeStereoMatches: python tools/stress_stereo.py [N] — GPU vs CPU oracle on random
sizes / budgets / pyramid parameters / baselines (not part of the pytest suite)."""
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = 0
t0 = time.time()
for it in range(N):
    w = int(rng.integers(400, 1400)); h = int(rng.integers(300, 700))
    nf = int(rng.choice([300, 1000, 2000, 3000]))
    sf = float(rng.choice([1.2, 1.2, 1.3, 1.5])); nl = int(rng.choice([8, 8, 5, 6]))
    if min(w, h) / (sf ** (nl - 1)) < 64:
        nl = max(1, int(np.log(min(w, h) / 64.0) / np.log(sf)) + 1)
    l, r = synth.stereo_pair_blocky(w, h, 3000 + it)
    fx = float(rng.uniform(300, 900)); mbf = float(rng.uniform(20, 400)); mb = float(np.float32(mbf) / np.float32(fx))
    ol, orr = oracle.Extractor(nf, sf, nl, 20, 7), oracle.Extractor(nf, sf, nl, 20, 7)
    try:
        kl, dl = ol.extract(l); kr, dr = orr.extract(r)
    except RuntimeError:
        continue
    on, our, odp = oracle.stereo_match(kl, dl, kr, dr, [ol.pyramid_level(i) for i in range(nl)], [orr.pyramid_level(i) for i in range(nl)],
                                       ol.scale_factors, ol.inv_scale_factors, mbf, mb)
    exl, exr = pkg.ORBextractor(nf, sf, nl, 20, 7), pkg.ORBextractor(nf, sf, nl, 20, 7)
    gkl, gdl = exl(l); gkr, gdr = exr(r)
    ur, dp, n = pkg.compute_stereo_matches(exl, exr, gkl, gdl, gkr, gdr, mbf, mb)
    same = n == on and ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    if not same:
        bad += 1
        print("MISMATCH", it, w, h, nf, sf, nl, mbf, fx, n, on)
    # round 5: the latency form of the whole stereo frame (orbx_stereo_frame_view: two calls, so that both of the handle's records are used)
    for rep in range(2):
        f = exl.stereo_frame_view(l, r, mbf, mb)
        samev = (f["nmatch"] == on and f["kl"].tobytes() == kl.tobytes() and f["dl"].tobytes() == dl.tobytes() and f["kr"].tobytes() == kr.tobytes() and
                 f["dr"].tobytes() == dr.tobytes() and f["uright"].tobytes() == our.tobytes() and f["depth"].tobytes() == odp.tobytes())
        if not samev:
            bad += 1
            print("MISMATCH (stereo_frame_view, call %d)" % rep, it, w, h, nf, sf, nl, mbf, fx, f["nmatch"], on)
    exl.close(); exr.close()
print("stereo stress: %d configs, %d mismatches, %.1f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
