"""The host-streaming loop of bench.py's end_to_end block alone (for a rocprofv3 --kernel-trace --memory-copy-trace timeline):
python tools/e2e_probe.py [steps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pl = importlib.import_module("orb_slam2v2-1_amd.pipeline")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w, h, nf, B = 1241, 376, 1000, 64
pairs = [synth.stereo_pair_blocky(w, h, i % 8) for i in range(B)]
L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
fe = pl.FrontEnd(w, h, nf, True, B).upload(L, R)
fe.enable_host_streaming()
pL, pR = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
def run(K):
    fe.submit(0, pL, pR)
    fe.submit(1, pL, pR)
    for i in range(K):
        fe.step(i)
        if i >= 1:
            fe.fetch(i - 1)
        if i >= 3:
            fe.wait(i - 3)
        if i + 2 < K:
            fe.submit(i + 2, pL, pR)
    fe.fetch(K - 1)
    fe.wait(K - 1)
run(5)
fe.drain()
for rep in range(4):
    DEPTH = (0, 2, 3, 5)[rep]
    t = time.perf_counter()
    fe.submit(0, pL, pR)
    fe.submit(1, pL, pR)
    th = [0.0, 0.0, 0.0]
    for i in range(K):
        a = time.perf_counter(); fe.step(i)
        b = time.perf_counter()
        if i >= 1:
            fe.fetch(i - 1)
        if DEPTH and i >= DEPTH:
            fe.wait(i - DEPTH)          # the consumer takes results as they arrive: the host stays DEPTH steps ahead of them
        c = time.perf_counter()
        if i + 2 < K:
            fe.submit(i + 2, pL, pR)
        d = time.perf_counter()
        th[0] += b - a; th[1] += c - b; th[2] += d - c
    tq = time.perf_counter() - t
    fe.fetch(K - 1)
    fe.wait(K - 1)
    dt = time.perf_counter() - t
    print("depth %d: enqueue loop %.3f ms per step (step %.3f fetch %.3f submit %.3f), total %.3f ms per step" % (
        DEPTH, tq / K * 1e3, th[0] / K * 1e3, th[1] / K * 1e3, th[2] / K * 1e3, dt / K * 1e3), flush=True)
    fe.drain()
t = time.perf_counter()
run(K)
dt = time.perf_counter() - t
print("%d steps: %.3f ms per step, %.0f frames/s, H2D %.1f GB/s" % (K, dt / K * 1e3, B * K / dt, 2 * B * w * h * K / dt / 1e9))
fe.drain()
