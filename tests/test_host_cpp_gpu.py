"""GPU: the C++ host classes (orb_slam2v2-1_amd/host: ORB_SLAM2::ORBextractor, ORBmatcher,
ComputeStereoMatchesHIP) called the way Frame.cc / Tracking.cc call the reference classes,
compared with the CPU oracle.  The driver is tests/cpp/host_driver.cc (compiled here with g++
against the cv/Frame shims; in the reference tree the same sources build against OpenCV)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "orb_slam2v2-1_amd", "lib")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import importlib
    importlib.import_module("orb_slam2v2-1_amd.build").build()
    exe = str(tmp_path_factory.mktemp("bin") / "host_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "orb_slam2v2-1_amd", "host"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_driver.cc"), "-L" + LIBDIR, "-lorb_host",
                           "-lorbx_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def _run(exe, *args):
    env = dict(os.environ)
    # one HIP runtime per process: the driver is a plain C++ program, it uses /opt/rocm's
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    return [int(t) for t in out.stdout.split()]


def test_extractor_class(driver, oracle, synth, pkg, tmp_path):
    w, h, nf = 752, 480, 1000
    img = synth.frame(w, h, 21)
    img.tofile(tmp_path / "a.raw")
    n, pw, ph = _run(driver, "extract", tmp_path / "a.raw", w, h, nf, tmp_path / "o")
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    ok, od = orc.extract(img)
    k = np.fromfile(str(tmp_path / "o.kps"), pkg.KP_DTYPE)
    d = np.fromfile(str(tmp_path / "o.desc"), np.uint8).reshape(-1, 32)
    assert n == len(ok) == len(k)
    for f in ok.dtype.names:
        np.testing.assert_array_equal(k[f], ok[f], err_msg=f)
    np.testing.assert_array_equal(d, od)
    p3 = np.fromfile(str(tmp_path / "o.pyr3"), np.uint8).reshape(ph, pw)
    np.testing.assert_array_equal(p3, orc.pyramid_level(3))


def test_stereo_through_frame(driver, oracle, synth, tmp_path):
    w, h, nf = 752, 480, 1000
    l, r = synth.stereo_pair_blocky(w, h, 22)
    l.tofile(tmp_path / "l.raw"); r.tofile(tmp_path / "r.raw")
    mbf, fx = 47.9, 435.2
    N, nm = _run(driver, "stereo", tmp_path / "l.raw", tmp_path / "r.raw", w, h, nf, mbf, fx, tmp_path / "s")
    ol, orr = oracle.Extractor(nf, 1.2, 8, 20, 7), oracle.Extractor(nf, 1.2, 8, 20, 7)
    kl, dl = ol.extract(l); kr, dr = orr.extract(r)
    mb = np.float32(mbf) / np.float32(fx)
    on, our, odp = oracle.stereo_match(kl, dl, kr, dr, [ol.pyramid_level(i) for i in range(8)],
                                       [orr.pyramid_level(i) for i in range(8)], ol.scale_factors,
                                       ol.inv_scale_factors, mbf, mb)
    assert N == len(kl) and nm == on > 20
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "s.uright"), np.float32), our)
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "s.depth"), np.float32), odp)


def test_search_for_initialization_through_frames(driver, oracle, synth, tmp_path):
    w, h, nf = 640, 480, 2000
    a = synth.frame(w, h, 23)
    b = np.roll(a, (2, 4), axis=(0, 1))
    a.tofile(tmp_path / "a.raw"); b.tofile(tmp_path / "b.raw")
    n1, n2, nm = _run(driver, "init", tmp_path / "a.raw", tmp_path / "b.raw", w, h, nf, tmp_path / "i")
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    k1, d1 = orc.extract(a)
    k2, d2 = orc.extract(b)
    prev = np.stack([k1["x"], k1["y"]], 1)
    on, om12, oprev = oracle.search_for_initialization(k1, d1, k2, d2, oracle.grid_geom(w, h), prev, 100, 0.9, True)
    assert (n1, n2) == (len(k1), len(k2)) and nm == on > 30
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "i.m12"), np.int32), om12)
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "i.prev"), np.float32).reshape(-1, 2), oprev)


def test_search_by_projection_through_mappoints(driver, oracle, synth, tmp_path):
    w, h, nf = 1241, 376, 1000
    img = synth.frame(w, h, 24)
    img.tofile(tmp_path / "a.raw")
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    k, d = orc.extract(img)
    rng = np.random.default_rng(3)
    m = 1200
    idx = rng.choice(len(k), m, replace=True)
    mps = np.zeros(m, oracle.MP_DTYPE)
    mps["in_view"] = rng.random(m) > 0.1
    mps["proj_x"] = k["x"][idx] + rng.normal(0, 1.5, m)
    mps["proj_y"] = k["y"][idx] + rng.normal(0, 1.5, m)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = np.clip(k["octave"][idx] + rng.integers(-1, 2, m), 0, 7)
    mps["view_cos"] = rng.uniform(0.99, 1.0, m)
    mps["observations"] = rng.integers(0, 4, m)
    md = d[idx] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                   rng.integers(0, 256, (m, 32), dtype=np.uint8))
    mps.tofile(tmp_path / "mps.bin"); md.tofile(tmp_path / "md.bin")
    N, nm = _run(driver, "projmp", tmp_path / "a.raw", w, h, nf, tmp_path / "mps.bin", tmp_path / "md.bin", tmp_path / "p")
    on, ofm = oracle.search_by_projection_mp(k, d, np.full(len(k), -1, np.float32), oracle.grid_geom(w, h),
                                             orc.scale_factors, mps, md, np.full(len(k), -1, np.int32), None, 3.0, 0.8)
    assert N == len(k) and nm == on > 100
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "p.held"), np.int32), ofm)


def test_search_by_projection_through_keyframe(driver, oracle, synth, tmp_path):
    """SURVEY §8(f) rank 1 through the C++ class: ORBmatcher::SearchByProjection(Frame&, KeyFrame*,
    sAlreadyFound, th, ORBdist) — projection + PredictScale on the host, matching on the GPU."""
    w, h, nf = 1241, 376, 1000
    img = synth.frame(w, h, 25)
    img.tofile(tmp_path / "a.raw")
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    k, d = orc.extract(img)
    sf = orc.scale_factors
    rng = np.random.default_rng(5)
    fx, fy, cx, cy = 718.856, 718.856, 607.19, 185.2
    m = 1300
    idx = rng.choice(len(k), m, replace=True)
    z = rng.uniform(4, 40, m).astype(np.float32)
    kf = np.zeros(m, oracle.KFPOINT_DTYPE)
    kf["valid"] = rng.random(m) > 0.2
    kf["wx"] = ((k["x"][idx] + rng.normal(0, 2, m) - cx) / fx * z).astype(np.float32)
    kf["wy"] = ((k["y"][idx] + rng.normal(0, 2, m) - cy) / fy * z).astype(np.float32)
    kf["wz"] = z
    kf["max_distance"] = z * sf[k["octave"][idx]] * rng.uniform(0.85, 1.15, m)
    kf["min_distance"] = kf["max_distance"] / sf[7] * rng.uniform(0.5, 1.0, m)
    kf["angle"] = (k["angle"][idx] + rng.normal(0, 5, m)) % 360
    kd = d[idx] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                   rng.integers(0, 256, (m, 32), dtype=np.uint8))
    kf.tofile(tmp_path / "kf.bin"); kd.tofile(tmp_path / "kd.bin")
    th, orbdist, tx, tz = 10.0, 100, 0.03, 0.2
    N, nm = _run(driver, "projkf", tmp_path / "a.raw", w, h, nf, tmp_path / "kf.bin", tmp_path / "kd.bin", th, orbdist,
                 "%r,%r,%r,%r" % (fx, fy, cx, cy), "%r,%r" % (tx, tz), tmp_path / "k")
    Tc = np.eye(4, dtype=np.float32)
    Tc[0, 3], Tc[2, 3] = tx, tz
    cam = oracle.Cam(fx, fy, cx, cy, 0.0, 0.0)
    log_sf = np.float32(np.log(np.float32(1.2)))
    on, ocm = oracle.search_by_projection_kf(k, d, oracle.grid_geom(w, h), sf, log_sf, cam, Tc, kf, kd,
                                             np.full(len(k), -1, np.int32), th, orbdist)
    assert N == len(k) and nm == on > 100
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "k.held"), np.int32), ocm)
