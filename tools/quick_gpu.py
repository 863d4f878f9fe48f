"""Ad-hoc GPU sanity + timing probe (not a test): python tools/quick_gpu.py [B]
The phase-stop options (keys 0, 1, 7) need the developer build: python orb_slam2v2-1_amd/build.py --developer; ORBX_LIB=orb_slam2v2-1_amd/lib/liborbx_hip_dev.so python ..."""
import sys, os, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, nf = 1241, 376, 1000
if "noise" in sys.argv:    # uniform noise: the worst case for any early-rejection scheme, the densest candidate lists
    imgs = np.random.default_rng(8).integers(0, 256, (8, h, w), dtype=np.uint8)
else:
    imgs = synth.batch(w, h, 8, 0)
imgs = np.concatenate([imgs] * ((B + 7) // 8))[:B]
import os as _os
if _os.environ.get("PYR_T"): pkg.set_default_option(3, int(_os.environ["PYR_T"]))
ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
k, d = ex(imgs[0])
print("single:", len(k), k[:3], d[0][:8])
dev = torch.device("cuda:0")
timg = torch.from_numpy(imgs).to(dev)
cap = ex.max_keypoints()
print("cap", cap)
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ex.set_profiling(True)
for it in range(3):
    ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
torch.cuda.synchronize()
t0 = time.time()
K = 10
for it in range(K):
    ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
torch.cuda.synchronize()
dt = (time.time() - t0) / K
print("B=%d  %.3f ms/batch  %.1f us/img  %.0f img/s" % (B, dt * 1e3, dt * 1e6 / B, B / dt))
print("stage ms [pyr, fast, octree, describe, total]:", ex.stage_ms())
print("counts", cnt[:8].tolist())
if len(sys.argv) > 2 and sys.argv[2] == "ablate_oct":
    for ph in (1, 2, 3, 4, 5, 6, 7, 0):
        pkg.set_default_option(1, ph)
        ex.set_profiling(True)
        for it in range(5):
            ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
        torch.cuda.synchronize()
        print("octree dbgStop", ph, "stage ms", ex.stage_ms()[0])
if len(sys.argv) > 2 and sys.argv[2] == "ablate":
    for ph in (1, 2, 3, 0):
        pkg.set_default_option(0, ph)
        ex.set_profiling(True)
        for it in range(5):
            ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
        torch.cuda.synchronize()
        print("fast phaseLimit", ph, "stage ms", ex.stage_ms()[0])

