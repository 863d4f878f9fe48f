import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
variant = sys.argv[1]
K = 100
w, h, nf, B = 1241, 376, 1000, 64
pairs = [synth.stereo_pair_blocky(w, h, i % 8) for i in range(B)]
L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
if variant in ("prior_fe", "prior_fe_same"):
    fe0 = pl.FrontEnd(w, h, nf, True, B).upload(L, R)
    for i in range(300): fe0.step(i)
    fe0.drain()
fe = fe0 if variant == "prior_fe_same" else pl.FrontEnd(w, h, nf, True, B).upload(L, R)
if variant == "prof":
    fe.ex.set_profiling(3)
    for i in range(100): fe.step(i)
    fe.drain(); fe.ex.stage_ms(); fe.ex.set_profiling(0)
fe.enable_host_streaming()
pL, pR = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
def run(K):
    fe.submit(0, pL, pR); fe.submit(1, pL, pR)
    for i in range(K):
        fe.step(i)
        if i >= 1: fe.fetch(i - 1)
        if i >= 3: fe.wait(i - 3)
        if i + 2 < K: fe.submit(i + 2, pL, pR)
    fe.fetch(K - 1); fe.wait(K - 1)
run(10); fe.drain()
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); run(K); dt = time.perf_counter() - t
    print("%s rep %d: %.3f ms per step, %.0f frames/s" % (variant, rep, dt / K * 1e3, B * K / dt), flush=True)
    fe.drain()
