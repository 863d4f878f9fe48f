"""Where k_octree_pyr spends its time in a BATCH (128 images 1241x376): stage time of the quad-tree with the kernel stopped after
phase n (ORBX_OPT key 7; needs the developer build):
   python orb_slam2v2-1_amd/build.py --developer
   ORBX_LIB=orb_slam2v2-1_amd/lib/liborbx_hip_dev.so python tools/octree_batch_phase_probe.py [natural]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
B, w, h, nf = 128, 1241, 376, 1000
nat = "natural" in sys.argv
imgs = np.stack([(synth.natural if nat else synth.frame)(w, h, i) for i in range(8)])
imgs = np.concatenate([imgs] * 16)
ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
ex(imgs[0])
cap = ex.max_keypoints()
timg = torch.from_numpy(imgs).cuda()
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for ph in (1, 2, 3, 4, 0):
    ex.set_option(7, ph)
    for it in range(3):
        ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
    torch.cuda.synchronize()
    ex.set_profiling(1)
    for it in range(10):
        ex.extract_batch_device(timg.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
    torch.cuda.synchronize()
    ms = ex.stage_ms()[0]
    ex.set_profiling(0)
    print("stop after phase %d: quad-tree stage %.4f ms" % (ph, ms[2]), flush=True)
