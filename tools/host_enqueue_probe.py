"""How long does the HOST need to enqueue one pipelined step (all launches, no synchronisation)?  If it approaches the GPU's
0.67 ms per step the bench becomes host-bound whenever the box's CPUs are busy.   python tools/host_enqueue_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf, B = 1241, 376, 1000, 64
pairs = [synth.stereo_pair(w, h, i) for i in range(B)]
fe = pl.FrontEnd(w, h, nf, True, B).upload(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
for i in range(50):
    fe.step(i)
fe.drain()
for rep in range(3):
    # a long GPU job in front, so that the queue never runs dry while the host enqueues: pure host time
    x = torch.randn((8192, 8192), device="cuda")
    for _ in range(4):
        x = (x @ x) * 1e-4
    t0 = time.perf_counter()
    n = 30
    for i in range(n):
        fe.step(i)
    t1 = time.perf_counter()
    fe.drain()
    t2 = time.perf_counter()
    print("host enqueue %.1f us per step; with the GPU work behind it %.1f us per step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
