import sys, importlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd"); synth = importlib.import_module("orb_slam2v2-1_amd.synth")
for name, gen, w, h, nf in (("dense", synth.frame, 1241, 376, 1000), ("dense2000", synth.frame, 1241, 376, 2000), ("natural", synth.natural, 1241, 376, 1000), ("fullhd", synth.frame, 1920, 1080, 4000), ("euroc", synth.frame, 752, 480, 1000)):
    B = 16
    imgs = np.stack([gen(w, h, 1000 + i) for i in range(B)])
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0]); cap = ex.max_keypoints()
    d = torch.from_numpy(imgs).cuda()
    k = torch.zeros((B, cap, 7), device="cuda"); de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); c = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.set_option(6, 3)
    ex.extract_batch_device(d.data_ptr(), B, w, h, w, w * h, k.data_ptr(), de.data_ptr(), c.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fb = ex.octree_fallbacks(B)
    print(name, "fallback levels per image (of 8):", fb.sum(1).tolist(), "per level:", fb.sum(0).tolist())
