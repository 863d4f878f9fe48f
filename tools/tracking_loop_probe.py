"""Latency probe of the per-frame front-end of a stereo tracking loop (BASELINE config 5 without the dataset and the
back-end): for every synthetic stereo frame  extract(left) || extract(right)  ->  ComputeStereoMatches  ->
SearchByProjection(current, last) with the 3-D points of the previous frame.  Host arrays in and out, one frame at a time
(replicas only: frame t needs frame t-1).  python tools/tracking_loop_probe.py [frames]"""
import sys, os, time, importlib, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w, h, nf = 1241, 376, 2000            # KITTI: 2000 features per camera
fx, fy, cx, cy, bf = 718.856, 718.856, 607.19, 185.2, 386.1448
mb = float(np.float32(bf) / np.float32(fx))
base_l, base_r = synth.stereo_pair_blocky(w + 80, h, 5)
frames = [(np.ascontiguousarray(base_l[:, 2 * t:2 * t + w]), np.ascontiguousarray(base_r[:, 2 * t:2 * t + w])) for t in range(nframes)]
exl, exr = pkg.ORBextractor(nf, 1.2, 8, 20, 7), pkg.ORBextractor(nf, 1.2, 8, 20, 7)
matcher = pkg.ORBmatcher(0.9, True)
geom = pkg.grid_geom(w, h)
cam = pkg.Camera(fx, fy, cx, cy, bf, mb)
sf = exl.scale_factors if hasattr(exl, "scale_factors") else (np.float32(1.2) ** np.arange(8)).astype(np.float32)
last = None
t_ext, t_st, t_proj, nm_hist = [], [], [], []
for t, (l, r) in enumerate(frames):
    t0 = time.perf_counter()
    out = {}
    th = threading.Thread(target=lambda: out.__setitem__("r", exr(r)))   # the reference runs the two extractors on two threads
    th.start()
    kl, dl = exl(l)
    th.join()
    kr, dr = out["r"]
    t1 = time.perf_counter()
    ur, depth, nst = pkg.compute_stereo_matches(exl, exr, kl, dl, kr, dr, bf, mb)
    t2 = time.perf_counter()
    nmatch = 0
    if last is not None:
        lk, ld, lur, ldepth = last
        pts = np.zeros(len(lk), pkg.LASTPT_DTYPE)
        ok = ldepth > 0
        pts["has_mp"] = ok
        z = np.where(ok, ldepth, 1).astype(np.float32)
        pts["wx"] = (lk["x"] - cx) / fx * z; pts["wy"] = (lk["y"] - cy) / fy * z; pts["wz"] = z
        pts["observations"] = 1; pts["octave"] = lk["octave"]; pts["angle"] = lk["angle"]
        Tl = np.eye(4, dtype=np.float32)
        Tc = np.eye(4, dtype=np.float32)           # constant-velocity guess = identity here; the scene shifts 2 px per frame
        cur = np.full(len(kl), -1, np.int32)
        nmatch, cur = matcher.SearchByProjectionFrame(kl, dl, ur, geom, sf, cam, Tc, Tl, pts, ld, cur, None, 15.0, False)
    t3 = time.perf_counter()
    last = (kl, dl, ur, depth)
    if t >= 5:
        t_ext.append(t1 - t0); t_st.append(t2 - t1); t_proj.append(t3 - t2); nm_hist.append(nmatch)
ms = lambda v: 1e3 * float(np.median(v))
print("stereo tracking front-end, %dx%d, %d features/camera, %d frames (median per frame):" % (w, h, nf, len(t_ext)))
print("  extract L||R %.3f ms + stereo match %.3f ms + SearchByProjection(cur,last) %.3f ms = %.3f ms/frame (%.0f frames/s), %d projection matches"
      % (ms(t_ext), ms(t_st), ms(t_proj), ms(t_ext) + ms(t_st) + ms(t_proj), 1e3 / (ms(t_ext) + ms(t_st) + ms(t_proj)), int(np.median(nm_hist))))
