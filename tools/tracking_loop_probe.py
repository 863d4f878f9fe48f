"""Per-frame latency of the stereo tracking front-end of tests/tracking_chain.py (BASELINE config 5 without its dataset,
optimiser and back-end; replicas only: frame t needs frame t-1) through the host-array entry points and through the
device-resident ones (keypoints / descriptors / mvuRight stay in HBM between extraction and matching).
    python tools/tracking_loop_probe.py [frames]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tracking_chain as tc   # noqa: E402

import time
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# time spent INSIDE the C ABI (ctypes argument conversion included), per entry point: what is left of a call's wall time is the Python harness
_ctime = {}


def _timed(L, name):
    f = getattr(L, name)

    def g(*a):
        t0 = time.perf_counter()
        r = f(*a)
        _ctime.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    setattr(L, name, g)
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""          # "device": the device-resident chain alone (profiler runs)
w, h, nf, step = 1241, 376, 2000, 0.04
synth = importlib.import_module(tc.PKG + ".synth")
frames, _ = synth.stereo_sequence(w, h, T, k=11, step=step)
Ts = tc.poses(T, step)
res = {}
_pkg = importlib.import_module(tc.PKG)
for _n in ("orbx_stereo_frame_view", "orbm_search_by_projection_frame_device", "orbm_search_local_points_device"):
    _timed(_pkg.lib(), _n)
for B in ((tc.GpuDeviceBackend,) if ONLY == "device" else (tc.GpuViewBackend,) if ONLY == "view" else
          (tc.GpuHostBackend, tc.GpuStereoFrameBackend, tc.GpuDeviceBackend, tc.GpuViewBackend)):
    c = tc.Chain(B(w, h, nf), w, h, nf)
    src = frames
    if B in (tc.GpuDeviceBackend, tc.GpuViewBackend):      # the device-resident chain takes its frames from pinned host memory (a capture buffer)
        import torch
        src = [(torch.from_numpy(l).pin_memory(), torch.from_numpy(r).pin_memory()) for l, r in frames]
    for t in range(T):
        c.step(src[t][0], src[t][1], Ts[t])
    log = c.log[5:]
    ms = lambda key: 1e3 * float(np.median([s[key] for s in log]))
    res[c.be.name] = c.log
    if c.be.name == "gpu-view":
        print("           inside the C ABI (median us): " + ", ".join("%s %.1f" % (k, 1e6 * float(np.median(v[5:]))) for k, v in _ctime.items()))
    print("%-10s %dx%d, %d features/camera, %d frames, median per frame: extract L+R + ComputeStereoMatches %.3f ms + "
          "SearchByProjection(cur,last) %.3f ms + SearchLocalPoints %.3f ms = %.3f ms (%.0f frames/s); %d projection / %d local-map matches"
          % (c.be.name, w, h, nf, len(log), ms("t_frame"), ms("t_proj"), ms("t_local"), ms("t_frame") + ms("t_proj") + ms("t_local"),
             1e3 / (ms("t_frame") + ms("t_proj") + ms("t_local")), int(np.median([s["proj_n"] for s in log])),
             int(np.median([s["local_n"] for s in log]))))
if not ONLY:
    # informational: the same latency path when the capture path delivers the images into HBM (device pointers are accepted as they are)
    import torch
    c = tc.Chain(tc.GpuViewBackend(w, h, nf), w, h, nf)
    srcd = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in frames]
    for t in range(T):
        c.step(srcd[t][0], srcd[t][1], Ts[t])
    res["gpu-view-dev"] = c.log
    print("gpu-view, images already in HBM: extract L+R + ComputeStereoMatches %.3f ms per frame (median); chain identical: %s"
          % (1e3 * float(np.median([s["t_frame"] for s in c.log[5:]])), tc.first_difference(res["gpu-host"], c.log) is None))
    print("chains identical:", all(tc.first_difference(res["gpu-host"], res[k]) is None for k in ("gpu-device", "gpu-host-1call", "gpu-view")))
