"""Single-image latency of orbx_extract (host in, host out): wall time per call vs the GPU time of its kernels.
  python tools/single_image_probe.py [n]        (under rocprofv3 --kernel-trace --stats for the kernel side)"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 3:
    pkg.lib().orbx_debug_set(int(sys.argv[2]), int(sys.argv[3]))   # developer knob, e.g. 5 1 = fused pyramid kernel
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
img = synth.frame(1241, 376, 5)
for _ in range(5):
    ex(img)
t0 = time.perf_counter()
for _ in range(n):
    ex(img)
t = (time.perf_counter() - t0) / n
print("orbx_extract 1241x376/1000: %.1f us per call over %d calls" % (t * 1e6, n))
