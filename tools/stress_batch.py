"""Randomised parity stress of the BATCHED device entry point: python tools/stress_batch.py [N] [seed] - random sizes, budgets,
batch sizes 2..12 (both sides of the "fills the GPU" rules), image kinds mixed inside a batch, random alternative paths (split call,
early quad-tree, k_gather instead of the lists read in place, the sparse path forced or by density in its three forms, chunks), every call issued THREE times (the second sees
the first one's per-level verdicts and scratch); every image of every batch against the CPU oracle."""
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = 0
t0 = time.time()
KNOBS = [(), ((15, 2),), ((15, 4),), ((19, 2),), ((19, 3),), ((18, 1),), ((16, 2),), ((8, 2),), ((8, 3),), ((6, 3),), ((6, 3), (19, 2)), ((6, 3), (16, 2), (15, 3)),
         ((6, 3), (20, 1)), ((6, 3), (20, 2)), ((6, 3), (16, 2), (20, 1)), ((6, 3), (16, 2), (20, 2)), ((6, 3), (20, 1), (15, 3)), ((6, 3), (20, 1), (8, 2)),
         ((22, 1),), ((22, 2),), ((6, 3), (22, 2))]
for it in range(N):
    w = int(rng.integers(320, 1300)); h = int(rng.integers(240, 720))
    nf = int(rng.choice([50, 300, 800, 1000, 2000]))
    B = int(rng.integers(2, 13))
    imgs = []
    for b in range(B):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            imgs.append(synth.frame(w, h, 9000 + 13 * it + b))
        elif kind == 1:
            imgs.append(synth.natural(w, h, 9000 + 13 * it + b))
        elif kind == 2:
            imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
        elif kind == 3:
            imgs.append(np.full((h, w), int(rng.integers(0, 256)), np.uint8))
        else:
            im = np.full((h, w), 60, np.uint8)
            x, y = int(rng.integers(40, w - 160)), int(rng.integers(40, h - 120))
            im[y:y + 90, x:x + 130] = synth.frame(130, 90, it + b)
            imgs.append(im)
    imgs = np.stack(imgs)
    try:
        orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
        exp = [orc.extract(imgs[b]) for b in range(B)]
    except RuntimeError:
        continue
    knobs = KNOBS[int(rng.integers(0, len(KNOBS)))]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0])
    cap = ex.max_keypoints()
    d_imgs = torch.from_numpy(imgs).cuda()
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for k, v in knobs:
        pkg.set_default_option(k, v)
    try:
        for rep in range(3):
            ex.extract_batch_device(d_imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
        torch.cuda.synchronize()
    finally:
        for k, v in knobs:
            pkg.set_default_option(k, 0)
    kk = kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    dd = desc.cpu().numpy()
    for b in range(B):
        ok, od = exp[b]
        n = int(cnt[b])
        same = n == len(ok) and kk[b, :n].tobytes() == ok.tobytes() and dd[b, :n].tobytes() == od.tobytes()
        if not same:
            bad += 1
            print("MISMATCH", it, "image", b, "of", B, w, h, nf, knobs, n, len(ok), flush=True)
    ex.close()
    del d_imgs, kps, desc, cnt
    if it % 20 == 19:
        print("  ... %d batches, %d mismatching images, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
print("batch stress: %d batches, %d mismatching images, %.1f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
