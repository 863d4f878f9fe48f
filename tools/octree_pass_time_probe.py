"""Developer probe (needs lib/liborbx_hip_dev.so: python orb_slam2v2-1_amd/build.py --developer; run with ORBX_LIB pointing at it):
per (image, level) workgroup of k_octree_pyr - number of list passes, time in the pass loop, time since kernel entry (0.1 us units,
from wall_clock64), read through orbx_debug_octree_fallbacks (the developer build stores the stamps in that record)."""
import sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd"); synth = importlib.import_module("orb_slam2v2-1_amd.synth")
SETS = (("dense", synth.frame, 1241, 376, 1000, 128), ("natural", synth.natural, 1241, 376, 1000, 128), ("dense2000", synth.frame, 1241, 376, 2000, 128),
        ("fullhd4000", synth.frame, 1920, 1080, 4000, 32))
if len(sys.argv) > 1:
    SETS = tuple(s for s in SETS if s[0] in sys.argv[1:])
for name, gen, w, h, nf, B in SETS:
    imgs = np.stack([gen(w, h, 1000 + (i % 16)) for i in range(B)])
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0]); cap = ex.max_keypoints()
    ex.set_option(7, 9)     # time stamps in place of the fall-back flags (developer build)
    d = torch.from_numpy(imgs).cuda()
    k = torch.zeros((B, cap, 7), device="cuda"); de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); c = torch.zeros(B, dtype=torch.int32, device="cuda")
    for rep in range(3):
        ex.extract_batch_device(d.data_ptr(), B, w, h, w, w * h, k.data_ptr(), de.data_ptr(), c.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fb = ex.octree_fallbacks(B).astype(np.int64)
    passes, loop, total = fb >> 24, ((fb >> 12) & 4095) / 10.0, (fb & 4095) / 10.0
    print(name)
    for l in range(8):
        print("  level %d: passes %s  pass loop us mean %.1f max %.1f   since entry us mean %.1f max %.1f" % (
            l, sorted(set(passes[:, l].tolist())), loop[:, l].mean(), loop[:, l].max(), total[:, l].mean(), total[:, l].max()))
    ex.set_option(7, 8)     # the finer split: entry -> sweep start -> sweep end -> first pass -> last pass (0.25-us units)
    for rep in range(2):
        ex.extract_batch_device(d.data_ptr(), B, w, h, w, w * h, k.data_ptr(), de.data_ptr(), c.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fb = ex.octree_fallbacks(B).astype(np.int64)
    f = [((fb >> s) & 255) / 4.0 for s in (24, 16, 8, 0)]
    for l in range(8):
        print("  level %d: set-up %.1f us, key sweep %.1f, count pyramid + roots %.1f, passes %.1f (means over %d images)" % (
            l, f[0][:, l].mean(), f[1][:, l].mean(), f[2][:, l].mean(), f[3][:, l].mean(), B))
    ex.set_option(7, 0)
