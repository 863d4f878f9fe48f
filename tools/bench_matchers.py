"""Times the guided-search matchers (host-array C ABI, one frame per call) on the GPU and the
CPU oracle on the same inputs: python tools/bench_matchers.py"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import oracle

def timeit(f, n):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r

w, h = 1241, 376
rng = np.random.default_rng(0)
for nf in (1000, 2000):
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    k1, d1 = orc.extract(synth.frame(w, h, 60))
    img2 = np.roll(synth.frame(w, h, 60), (2, 5), axis=(0, 1))
    k2, d2 = orc.extract(img2)
    sf = orc.scale_factors
    go, gg = oracle.grid_geom(w, h), pkg.grid_geom(w, h)
    prev = np.stack([k1["x"], k1["y"]], 1)
    m = pkg.ORBmatcher(0.9, True)
    tg, rg = timeit(lambda: m.SearchForInitialization(k1, d1, k2, d2, gg, prev, 100), 5)
    tc, rc = timeit(lambda: oracle.search_for_initialization(k1, d1, k2, d2, go, prev, 100, 0.9, True), 5)
    assert rg[0] == rc[0] and (rg[1] == rc[1]).all()
    print("nf=%d SearchForInitialization: GPU %.3f ms  CPU-oracle %.3f ms  (%d matches)" % (nf, tg, tc, rg[0]))
    # map points
    mcount = 2 * len(k1)
    idx = rng.choice(len(k1), mcount, replace=True)
    mps = np.zeros(mcount, oracle.MP_DTYPE)
    mps["in_view"] = 1
    mps["proj_x"] = k1["x"][idx] + rng.normal(0, 1.5, mcount); mps["proj_y"] = k1["y"][idx] + rng.normal(0, 1.5, mcount)
    mps["proj_xr"] = mps["proj_x"] - 5; mps["level"] = k1["octave"][idx]; mps["view_cos"] = 0.999; mps["observations"] = 2
    md = d1[idx] ^ (rng.integers(0, 256, (mcount, 32), dtype=np.uint8) & rng.integers(0, 256, (mcount, 32), dtype=np.uint8) & rng.integers(0, 256, (mcount, 32), dtype=np.uint8))
    ur = np.full(len(k1), -1, np.float32); fm = np.full(len(k1), -1, np.int32)
    m2 = pkg.ORBmatcher(0.8, True)
    tg, rg = timeit(lambda: m2.SearchByProjection(k1, d1, ur, gg, sf, mps, md, fm, None, 3.0), 5)
    tc, rc = timeit(lambda: oracle.search_by_projection_mp(k1, d1, ur, go, sf, mps, md, fm, None, 3.0, 0.8), 5)
    assert rg[0] == rc[0] and (rg[1] == rc[1]).all()
    print("nf=%d SearchByProjection(F,MPs) m=%d: GPU %.3f ms  CPU-oracle %.3f ms  (%d matches)" % (nf, mcount, tg, tc, rg[0]))
