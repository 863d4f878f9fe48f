"""Build the HIP shared library (gfx950 only) in-tree: orb_slam2v2-1_amd/lib/liborbx_hip.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off is required for bit-parity of the
float expressions the reference evaluates without FMA.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "liborbx_hip.so")
SOURCES = ["orbx_extract.hip", "orbx_pyramid.hip", "orbx_fast.hip", "orbx_octree.hip", "orbx_octree_wide.hip", "orbx_describe.hip",
           "orbx_match.hip", "orbx_match_fast.hip", "orbx_bow.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-value", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(ROOT, "include", "orbx.h"), os.path.join(ROOT, "include", "orb_pattern_31.inc")]
    return any(os.path.getmtime(d) > t for d in deps)


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "lib", "liborb_host.so")


def build_host(force=False, verbose=False):
    """C++ host classes (ORB_SLAM2::ORBextractor / ORBmatcher mirrors) over the C ABI: g++ only."""
    srcs = [os.path.join(HOST, f) for f in ("ORBextractor.cc", "ORBmatcher.cc", "ORBVocabulary.cc")]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST)] + [LIB]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
           "-o", HOST_LIB] + srcs + ["-L" + os.path.dirname(LIB), "-lorbx_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


def build(force=False, verbose=False):
    if not force and not needs_build():
        build_host(False, verbose)
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_host(True, verbose)
    return LIB


def build_developer(verbose=False):
    """lib/liborbx_hip_dev.so: the same library with -DORBX_DEVELOPER (orbx_set_option accepts the phase-stop keys 0, 1, 7 that the
    ablation probes of tools/ use; outputs are incomplete under them).  Never loaded by the tests or bench.py; ORBX_LIB selects it."""
    out = os.path.join(HERE, "lib", "liborbx_hip_dev.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [hipcc] + FLAGS + ["-DORBX_DEVELOPER", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--developer" in sys.argv:
        print(build_developer(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(LIB)
