"""Where k_octree_pyr spends its time on ONE image (its duration is the slowest block's critical path):
  rocprofv3 --kernel-trace --stats -- python3 tools/octree_phase_probe.py W H NFEAT STOP
STOP = developer knob 7 (0 = whole kernel, 1 = after the histogram sweep, 2 = after the count pyramid, 3 = after the
passes, 4 = after the final sweep).
The phase-stop options (keys 0, 1, 7) need the developer build: python orb_slam2v2-1_amd/build.py --developer; ORBX_LIB=orb_slam2v2-1_amd/lib/liborbx_hip_dev.so python ..."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
w, h, nf, stop = (int(x) for x in sys.argv[1:5])
ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
img = synth.frame(w, h, 5)
ex(img)
pkg.set_default_option(7, stop)
for _ in range(20):
    try:
        ex(img)
    except Exception:
        pass
pkg.set_default_option(7, 0)
