"""Latency probe of ONE stereo frame (two images, one handle) through the batched kernels, option by option:
    python tools/latency_probe.py              stage times (HIP events) of extract_batch_device at B = 2 under the listed options
With ORBX_LIB=.../liborbx_hip_dev.so it also prints the quad-tree's per-level time stamps at B = 2."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf = 1241, 376, 2000
frames, _ = synth.stereo_sequence(w, h, 4, k=11, step=0.04)
imgs = np.stack([frames[2][0], frames[2][1]])
d = torch.from_numpy(imgs).cuda()
st = torch.cuda.current_stream().cuda_stream


def run(opts, n=200, label=""):
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0])
    cap = ex.max_keypoints()
    k = torch.zeros((2, cap, 7), device="cuda"); de = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    c = torch.zeros(2, dtype=torch.int32, device="cuda")
    for kk, vv in opts:
        ex.set_option(kk, vv)
    call = lambda: ex.extract_batch_device(d.data_ptr(), 2, w, h, w, w * h, k.data_ptr(), de.data_ptr(), c.data_ptr(), cap, st)
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    gpu_us = e0.elapsed_time(e1) * 1e3 / n
    t0 = time.perf_counter()
    for _ in range(n):
        call()
        torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) * 1e6 / n
    ex.set_profiling(1)
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    ms, _ = ex.stage_ms()
    print("%-40s back-to-back %.1f us/call, call+sync %.1f us | stages (events, +~4 us each) pyr %.1f fast %.1f oct %.1f desc %.1f"
          % (label or str(opts), gpu_us, wall_us, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3), flush=True)
    ex.set_profiling(0)
    return ex, (k, de, c, cap)


run([], label="default")
run([(5, 1)], label="pyramid: one fused launch (5=1)")
run([(5, 3)], label="pyramid: hybrid (5=3)")
run([(4, 2)], label="quad-tree: every level multi-workgroup (4=2)")
run([(11, 2)], label="quad-tree: 1024-thread build (11=2)")
run([(6, 3)], label="FAST: strips (6=3)")
run([(5, 1), (4, 2)], label="5=1 + 4=2")
if "dev" in os.environ.get("ORBX_LIB", ""):
    ex, (k, de, c, cap) = run([], label="default (dev build)")
    ex.set_option(7, 8)
    for _ in range(3):
        ex.extract_batch_device(d.data_ptr(), 2, w, h, w, w * h, k.data_ptr(), de.data_ptr(), c.data_ptr(), cap, st)
    torch.cuda.synchronize()
    fb = ex.octree_fallbacks(2).astype(np.int64)
    f = [((fb >> s) & 255) / 4.0 for s in (24, 16, 8, 0)]
    for l in range(8):
        print("  level %d: set-up %.1f us, key sweep %.1f, count pyramid + roots %.1f, passes %.1f (image 0; image 1: %.1f %.1f %.1f %.1f)" % (
            l, f[0][0, l], f[1][0, l], f[2][0, l], f[3][0, l], f[0][1, l], f[1][1, l], f[2][1, l], f[3][1, l]))
