"""Probe: do two extractor handles on two streams (64 images each) beat one handle with 128 images?
python tools/overlap_probe.py   (GPU box)"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf = 1241, 376, 1000
imgs = np.stack([synth.frame(w, h, 1000 + i) for i in range(16)])
imgs = np.concatenate([imgs] * 8)
dev = torch.device("cuda:0")
timg = torch.from_numpy(imgs).to(dev)


def run(nh, steps=30):
    B = 128 // nh
    exs = [pkg.ORBextractor(nf, 1.2, 8, 20, 7) for _ in range(nh)]
    for ex in exs:
        ex(imgs[0])
    cap = exs[0].max_keypoints()
    streams = [torch.cuda.Stream() for _ in range(nh)]
    bufs = [(torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
             torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(nh)]
    def once():
        for i, ex in enumerate(exs):
            k, d, c = bufs[i]
            ex.extract_batch_device(timg[i * B:].data_ptr(), B, w, h, w, w * h, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap,
                                    streams[i].cuda_stream)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%d handle(s) x %3d images: %.3f ms per 128 images" % (nh, B, dt * 1e3))


for nh in (1, 2, 4):
    run(nh)
