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
    return _stale(LIB) or _stale(DEV_LIB)


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


DEV_LIB = os.path.join(HERE, "lib", "liborbx_hip_dev.so")
OBJ = os.path.join(HERE, "lib", "obj")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(ROOT, "include", "orbx.h"), os.path.join(ROOT, "include", "orb_pattern_31.inc")]


def _compile_link(out, defines, verbose):
    """One hipcc -c per source, in parallel (a source takes 5-25 s; nine one after the other took 30 s), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = "dev" if defines else "rel"
    os.makedirs(OBJ, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + defines + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    tdep = max(os.path.getmtime(d) for d in _deps() if not d.endswith(".hip"))      # headers: every object depends on them
    jobs = []
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(OBJ, src.replace(".hip", "." + tag + ".o"))
        if not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), tdep) or \
                (src == "orbx_octree_wide.hip" and os.path.getmtime(op) < os.path.getmtime(os.path.join(CSRC, "orbx_octree.hip"))):
            jobs.append([hipcc] + cflags + ["-c", sp, "-o", op])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for rc, j in zip(ex.map(lambda c: subprocess.call(c), jobs), jobs):
            if rc != 0:
                raise subprocess.CalledProcessError(rc, j)
    objs = [os.path.join(OBJ, s.replace(".hip", "." + tag + ".o")) for s in SOURCES]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def _stale(lib):
    return not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in _deps())


def build(force=False, verbose=False):
    """The product library (no test hooks: nm -D shows no *_debug_* symbol) AND the developer build beside it (the same code with
    -DORBX_DEVELOPER: + the read-only stage hooks orbx_debug_* / orbm_debug_* the staged parity tests look through, + the phase-stop
    and time-stamp option keys of the probes in tools/).  Results are identical; everything timed or end-to-end loads the product
    library."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or _stale(LIB):
        _compile_link(LIB, [], verbose)
        build_host(True, verbose)
    else:
        build_host(False, verbose)
    if force or _stale(DEV_LIB):
        _compile_link(DEV_LIB, ["-DORBX_DEVELOPER"], verbose)
    return LIB


def build_developer(verbose=False):
    if _stale(DEV_LIB):
        _compile_link(DEV_LIB, ["-DORBX_DEVELOPER"], verbose)
    return DEV_LIB


if __name__ == "__main__":
    if "--developer" in sys.argv:
        print(build_developer(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(LIB)
