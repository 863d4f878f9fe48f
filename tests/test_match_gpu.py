"""GPU parity of the HIP matchers against the CPU oracle through the C ABI (bit-exact indices,
counts and Hamming distances; float outputs compared exactly — same float operations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _features(oracle, img, nf=1000):
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    k, d = orc.extract(img)
    return orc, k, d


def test_hamming_host(pkg, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert pkg.ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert pkg.ORBmatcher.DescriptorDistance(z, ~z) == 256
    assert pkg.ORBmatcher.DescriptorDistance(z, z) == 0


def test_hamming_matrix(pkg):
    import torch
    rng = np.random.default_rng(1)
    for na, nb in [(1, 1), (63, 257), (1000, 1003), (130, 64)]:
        a = rng.integers(0, 256, (na, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        out = torch.zeros((na, nb), dtype=torch.int16, device="cuda")
        rc = pkg.lib().orbm_hamming_matrix_device(ta.data_ptr(), na, tb.data_ptr(), nb, out.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0, pkg.lib().orbx_last_error()
        torch.cuda.synchronize()
        ref = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
        np.testing.assert_array_equal(out.cpu().numpy().astype(np.int32), ref)


@pytest.mark.parametrize("w,h,nf,k", [(1241, 376, 2000, 0), (752, 480, 1000, 1), (640, 480, 1000, 2)])
def test_stereo_parity(pkg, oracle, synth, w, h, nf, k):
    left, right = synth.stereo_pair_blocky(w, h, k)
    exl, exr = pkg.ORBextractor(nf, 1.2, 8, 20, 7), pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = exl(left)
    kr, dr = exr(right)
    orl, okl, odl = _features(oracle, left, nf)
    orr, okr, odr = _features(oracle, right, nf)
    np.testing.assert_array_equal(dl, odl)
    np.testing.assert_array_equal(dr, odr)
    fx = 718.856 if w == 1241 else 435.2
    mbf = 386.1448 if w == 1241 else 47.9
    mb = np.float32(mbf) / np.float32(fx)
    ur, dp, n = pkg.compute_stereo_matches(exl, exr, kl, dl, kr, dr, mbf, mb)
    pl = [orl.pyramid_level(l) for l in range(8)]
    pr = [orr.pyramid_level(l) for l in range(8)]
    on, our, odp = oracle.stereo_match(okl, odl, okr, odr, pl, pr, orl.scale_factors, orl.inv_scale_factors, mbf, mb)
    assert on > 20, "synthetic pair should yield stereo matches (got %d)" % on
    assert n == on
    np.testing.assert_array_equal(ur, our)
    np.testing.assert_array_equal(dp, odp)


def test_stereo_edge_cases(pkg, oracle, synth):
    # no right keypoints / no matches surviving: everything stays -1, count 0 (reference: UB on
    # the empty vector, src/Frame.cc:642 — defined here as "no matches")
    w, h = 640, 480
    left, _ = synth.stereo_pair_blocky(w, h, 5)
    exl, exr = pkg.ORBextractor(500, 1.2, 8, 20, 7), pkg.ORBextractor(500, 1.2, 8, 20, 7)
    kl, dl = exl(left)
    flat = np.full((h, w), 100, np.uint8)
    kr, dr = exr(flat)
    assert len(kr) == 0
    ur, dp, n = pkg.compute_stereo_matches(exl, exr, kl, dl, kr, dr, 40.0, 0.1)
    assert n == 0 and (ur == -1).all() and (dp == -1).all()


@pytest.fixture(params=["fast", "fast_wave", "exact"])
def matcher_path(request, pkg):
    """The implementations of the guided searches: parallel candidates + the whole-workgroup fixed-point resolution (default,
    round 5), parallel candidates + the single-wave speculative resolution (rounds 1-4), and the exact one-workgroup kernels
    both fall back to."""
    pkg.lib().orbm_set_thread_option(2, 1 if request.param == "exact" else 0)
    pkg.lib().orbm_set_thread_option(3, 1 if request.param == "fast_wave" else 0)
    yield request.param
    pkg.lib().orbm_set_thread_option(2, 0)
    pkg.lib().orbm_set_thread_option(3, 0)


def test_search_for_initialization(pkg, oracle, synth, matcher_path):
    w, h = 640, 480
    img1 = synth.frame(w, h, 7)
    img2 = np.roll(img1, (3, 5), axis=(0, 1))  # small camera motion
    rng = np.random.default_rng(5)
    img2 = np.clip(img2.astype(np.int16) + rng.integers(-3, 4, img2.shape), 0, 255).astype(np.uint8)
    _, k1, d1 = _features(oracle, img1, 2000)
    _, k2, d2 = _features(oracle, img2, 2000)
    geom_o, geom_g = oracle.grid_geom(w, h), pkg.grid_geom(w, h)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for ratio, ori, win in [(0.9, True, 100), (0.9, False, 100), (0.6, True, 30)]:
        on, om12, oprev = oracle.search_for_initialization(k1, d1, k2, d2, geom_o, prev, win, ratio, ori)
        m = pkg.ORBmatcher(ratio, ori)
        gn, gm12, gprev = m.SearchForInitialization(k1, d1, k2, d2, geom_g, prev, win)
        assert on > 30
        assert gn == on
        np.testing.assert_array_equal(gm12, om12)
        np.testing.assert_array_equal(gprev, oprev)


def _mappoints_from(oracle, k, d, rng, m):
    idx = rng.choice(len(k), size=m, replace=len(k) < m)
    mps = np.zeros(m, oracle.MP_DTYPE)
    mps["in_view"] = rng.random(m) > 0.1
    mps["proj_x"] = k["x"][idx] + rng.normal(0, 1.5, m)
    mps["proj_y"] = k["y"][idx] + rng.normal(0, 1.5, m)
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(1, 30, m)
    mps["level"] = np.clip(k["octave"][idx] + rng.integers(-1, 2, m), 0, 7)
    mps["view_cos"] = rng.uniform(0.99, 1.0, m)
    mps["observations"] = rng.integers(0, 4, m)
    md = d[idx].copy()
    flip = rng.integers(0, 256, md.shape, dtype=np.uint8) & rng.integers(0, 256, md.shape, dtype=np.uint8) & \
        rng.integers(0, 256, md.shape, dtype=np.uint8) & rng.integers(0, 256, md.shape, dtype=np.uint8)
    return mps, md ^ flip


def test_search_by_projection_mappoints(pkg, oracle, synth, matcher_path):
    w, h = 1241, 376
    _, k, d = _features(oracle, synth.frame(w, h, 11), 1000)
    rng = np.random.default_rng(9)
    sf = oracle.Extractor(1000, 1.2, 8, 20, 7).scale_factors
    for th, ratio in [(1.0, 0.8), (3.0, 0.8), (5.0, 0.6)]:
        mps, md = _mappoints_from(oracle, k, d, rng, 1500)
        uright = np.where(rng.random(len(k)) < 0.5, k["x"] - rng.uniform(1, 30, len(k)), -1).astype(np.float32)
        frame_mp = np.full(len(k), -1, np.int32)
        ext = np.zeros(len(k), np.int32)
        pre = rng.choice(len(k), 50, replace=False)
        frame_mp[pre] = -2
        ext[pre] = rng.integers(0, 2, 50)
        on, ofm = oracle.search_by_projection_mp(k, d, uright, oracle.grid_geom(w, h), sf, mps, md, frame_mp, ext, th, ratio)
        gn, gfm = pkg.ORBmatcher(ratio, True).SearchByProjection(k, d, uright, pkg.grid_geom(w, h), sf, mps, md,
                                                                 frame_mp, ext, th)
        assert on > 100
        assert gn == on
        np.testing.assert_array_equal(gfm, ofm)


def test_search_by_projection_frame(pkg, oracle, synth, matcher_path):
    w, h = 1241, 376
    _, k, d = _features(oracle, synth.frame(w, h, 12), 1000)
    rng = np.random.default_rng(10)
    sf = oracle.Extractor(1000, 1.2, 8, 20, 7).scale_factors
    fx, fy, cx, cy, mbf = 718.856, 718.856, 607.19, 185.2, 386.1448
    cam_o, cam_g = oracle.Cam(fx, fy, cx, cy, mbf, np.float32(mbf) / np.float32(fx)), None
    cam_g = pkg.Camera(fx, fy, cx, cy, mbf, np.float32(mbf) / np.float32(fx))
    n = len(k)
    for mono, dz in [(False, 0.0), (False, 1.5), (False, -1.5), (True, 0.3)]:
        # last frame = same keypoints seen from a camera shifted by (0.05, 0, dz)
        z = rng.uniform(4, 40, n).astype(np.float32)
        last = np.zeros(n, oracle.LASTPT_DTYPE)
        last["has_mp"] = rng.random(n) > 0.2
        last["wx"] = (k["x"] - cx) / fx * z
        last["wy"] = (k["y"] - cy) / fy * z
        last["wz"] = z
        last["observations"] = rng.integers(0, 3, n)
        last["octave"] = k["octave"]
        last["angle"] = (k["angle"] + rng.normal(0, 4, n)) % 360
        ld = d.copy()
        ld ^= (rng.integers(0, 256, ld.shape, dtype=np.uint8) & rng.integers(0, 256, ld.shape, dtype=np.uint8) &
               rng.integers(0, 256, ld.shape, dtype=np.uint8))
        Tc = np.eye(4, dtype=np.float32)
        Tc[0, 3] = 0.02
        Tl = np.eye(4, dtype=np.float32)
        Tl[2, 3] = dz
        uright = np.where(rng.random(n) < 0.5, k["x"] - mbf / z, -1).astype(np.float32)
        cur = np.full(n, -1, np.int32)
        on, ocm = oracle.search_by_projection_frame(k, d, uright, oracle.grid_geom(w, h), sf, cam_o, Tc, Tl, last, ld,
                                                    cur, None, 7.0, mono, True)
        gn, gcm = pkg.ORBmatcher(0.9, True).SearchByProjectionFrame(k, d, uright, pkg.grid_geom(w, h), sf, cam_g, Tc,
                                                                    Tl, last, ld, cur, None, 7.0, mono)
        assert on > 100, on
        assert gn == on
        np.testing.assert_array_equal(gcm, ocm)


def test_guided_search_heavy_contention(pkg, oracle, synth, matcher_path):
    """Many queries competing for few keypoints: long conflict chains in the speculative
    resolver, blocked candidates beyond the kept top-8 (forces the exact fallback)."""
    w, h = 640, 480
    _, k, d = _features(oracle, synth.frame(w, h, 13), 300)
    rng = np.random.default_rng(11)
    sf = oracle.Extractor(300, 1.2, 8, 20, 7).scale_factors
    m = 3000
    idx = rng.choice(min(len(k), 40), m, replace=True)  # everything projects onto the first 40 keypoints
    mps = np.zeros(m, oracle.MP_DTYPE)
    mps["in_view"] = 1
    mps["proj_x"] = k["x"][idx] + rng.normal(0, 6, m)
    mps["proj_y"] = k["y"][idx] + rng.normal(0, 6, m)
    mps["proj_xr"] = mps["proj_x"]
    mps["level"] = k["octave"][idx]
    mps["view_cos"] = 0.9
    mps["observations"] = rng.integers(0, 2, m)
    md = d[idx] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                   rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8))
    ur = np.full(len(k), -1, np.float32)
    fm = np.full(len(k), -1, np.int32)
    on, ofm = oracle.search_by_projection_mp(k, d, ur, oracle.grid_geom(w, h), sf, mps, md, fm, None, 10.0, 0.9)
    gn, gfm = pkg.ORBmatcher(0.9, True).SearchByProjection(k, d, ur, pkg.grid_geom(w, h), sf, mps, md, fm, None, 10.0)
    assert gn == on
    np.testing.assert_array_equal(gfm, ofm)
    # initialization matcher with a tiny second frame: every F1 keypoint fights for the same few F2 keypoints
    k2, d2 = k[:25].copy(), d[:25].copy()
    k1 = np.repeat(k[:25], 40)
    d1 = np.repeat(d[:25], 40, axis=0) ^ (rng.integers(0, 256, (1000, 32), dtype=np.uint8) &
                                           rng.integers(0, 256, (1000, 32), dtype=np.uint8) &
                                           rng.integers(0, 256, (1000, 32), dtype=np.uint8))
    k1["octave"] = 0
    k2["octave"] = 0
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    on, om12, oprev = oracle.search_for_initialization(k1, d1, k2, d2, oracle.grid_geom(w, h), prev, 100, 0.9, True)
    gn, gm12, gprev = pkg.ORBmatcher(0.9, True).SearchForInitialization(k1, d1, k2, d2, pkg.grid_geom(w, h), prev, 100)
    assert gn == on
    np.testing.assert_array_equal(gm12, om12)
    np.testing.assert_array_equal(gprev, oprev)


def _kf_scene(oracle, synth, rng, n_ext=40):
    """A keyframe whose map points project near the current frame's keypoints."""
    w, h = 1241, 376
    _, k, d = _features(oracle, synth.frame(w, h, 14), 1000)
    sf = oracle.Extractor(1000, 1.2, 8, 20, 7).scale_factors
    fx, fy, cx, cy = 718.856, 718.856, 607.19, 185.2
    cam = oracle.Cam(fx, fy, cx, cy, 386.1448, np.float32(386.1448) / np.float32(fx))
    n = len(k)
    m = 1500
    idx = rng.choice(n, m, replace=True)
    z = rng.uniform(4, 40, m).astype(np.float32)
    kf = np.zeros(m, oracle.KFPOINT_DTYPE)
    kf["valid"] = rng.random(m) > 0.15
    kf["wx"] = (k["x"][idx] + rng.normal(0, 2, m) - cx) / fx * z
    kf["wy"] = (k["y"][idx] + rng.normal(0, 2, m) - cy) / fy * z
    kf["wz"] = z
    # scale-invariance range such that the predicted level is around the keypoint's octave
    lvl = k["octave"][idx]
    kf["max_distance"] = z * sf[lvl] * rng.uniform(0.85, 1.15, m)
    kf["min_distance"] = kf["max_distance"] / sf[7] * rng.uniform(0.5, 1.0, m)
    kf["angle"] = (k["angle"][idx] + rng.normal(0, 5, m)) % 360
    kd = d[idx] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                   rng.integers(0, 256, (m, 32), dtype=np.uint8))
    Tc = np.eye(4, dtype=np.float32)
    Tc[0, 3], Tc[2, 3] = 0.03, 0.2
    cur = np.full(n, -1, np.int32)
    cur[rng.choice(n, n_ext, replace=False)] = -2
    return w, h, k, d, sf, cam, kf, kd, Tc, cur


def test_search_by_projection_keyframe(pkg, oracle, synth, matcher_path):
    """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)
    (src/ORBmatcher.cc:1474-1601) = host projection (oracle stage here) + orbm_match_windows."""
    rng = np.random.default_rng(21)
    w, h, k, d, sf, cam, kf, kd, Tc, cur = _kf_scene(oracle, synth, rng)
    log_sf = np.float32(np.log(np.float32(1.2)))
    for th, orbdist in [(10.0, 100), (3.0, 64)]:
        on, ocm = oracle.search_by_projection_kf(k, d, oracle.grid_geom(w, h), sf, log_sf, cam, Tc, kf, kd, cur, th, orbdist)
        q = oracle.kf_window_queries(kf, oracle.grid_geom(w, h), sf, log_sf, cam, Tc, th)
        assert q["valid"].sum() > 500
        gn, gcm = pkg.match_windows(k, d, None, pkg.grid_geom(w, h), q, kd, cur, None, orbdist, True)
        assert on > 100
        assert gn == on
        np.testing.assert_array_equal(gcm, ocm)


def test_match_windows_generic_paths_agree(pkg, oracle, synth, matcher_path):
    """Generic matcher with per-query blocking and the stereo gate: the speculative path
    and the exact one-workgroup path give the same holders."""
    w, h = 1241, 376
    _, k, d = _features(oracle, synth.frame(w, h, 15), 1000)
    rng = np.random.default_rng(22)
    n = len(k)
    q = np.zeros(n, pkg.WINDOW_DTYPE)
    q["valid"] = rng.random(n) > 0.2
    q["u"] = k["x"] + rng.normal(0, 2, n)
    q["v"] = k["y"] + rng.normal(0, 2, n)
    q["radius"] = 7.0 * oracle.Extractor(1000, 1.2, 8, 20, 7).scale_factors[k["octave"]]
    q["min_level"] = k["octave"] - 1
    q["max_level"] = k["octave"] + 1
    q["angle"] = k["angle"]
    q["blocks"] = rng.integers(0, 2, n)
    q["ur_c"] = q["u"] - 10
    q["ur_tol"] = q["radius"]
    uright = np.where(rng.random(n) < 0.5, k["x"] - rng.uniform(5, 15, n), -1).astype(np.float32)
    qd = d ^ (rng.integers(0, 256, d.shape, dtype=np.uint8) & rng.integers(0, 256, d.shape, dtype=np.uint8) &
              rng.integers(0, 256, d.shape, dtype=np.uint8))
    holder = np.full(n, -1, np.int32)
    pkg.lib().orbm_set_thread_option(2, 1)
    en, eh = pkg.match_windows(k, d, uright, pkg.grid_geom(w, h), q, qd, holder, None, 100, True)
    pkg.lib().orbm_set_thread_option(2, 1 if matcher_path == "exact" else 0)
    gn, gh = pkg.match_windows(k, d, uright, pkg.grid_geom(w, h), q, qd, holder, None, 100, True)
    assert gn == en > 100
    np.testing.assert_array_equal(gh, eh)


# ---- SURVEY §8(f) rank 1, KeyFrame side: SearchByProjection(KeyFrame*, Scw), Fuse x2, SearchBySim3
def _kfside(oracle, synth, seed, distorted, m=1800):
    import kf_scene as ks
    w, h = 1241, 376
    rng = np.random.default_rng(seed)
    _, k, d = _features(oracle, synth.frame(w, h, 30 + seed), 1000)
    sf = oracle.Extractor(1000, 1.2, 8, 20, 7).scale_factors
    cam = oracle.Cam(ks.FX, ks.FY, ks.CX, ks.CY, ks.MBF, np.float32(ks.MBF) / np.float32(ks.FX))
    return ks, w, h, rng, k, d, sf, cam, np.float32(np.log(np.float32(1.2))), m


@pytest.mark.parametrize("distorted", [False, True])
@pytest.mark.parametrize("gate", [False, True])
def test_best_in_windows(pkg, oracle, synth, distorted, gate):
    """orbm_best_in_windows (the search of Fuse / SearchBySim3, src/ORBmatcher.cc:905-951,1064-1081,
    1203-1221) against the oracle's grid walk, with and without Fuse's reprojection gate, and with
    a KeyFrame whose int-truncated bounds differ from the bounds its cells were built with."""
    ks, w, h, rng, k, d, sf, cam, log_sf, m = _kfside(oracle, synth, 3, distorted)
    T = ks.pose(rng)
    pts, pd, _ = ks.points_for(oracle, rng, k, d, sf, T, m)
    og, oga, _ = ks.geoms(oracle, w, h, distorted)
    pg, pga, _ = ks.geoms(pkg, w, h, distorted)
    q = oracle.pose_window_queries(pts, og, sf, log_sf, cam, T, 3.0)
    assert 0.5 * m < q["valid"].sum() < m
    uright = np.where(rng.random(len(k)) < 0.5, k["x"] - rng.uniform(1, 40, len(k)), -1).astype(np.float32)
    inv_s2 = (1.0 / (sf * sf)).astype(np.float32) if gate else None
    obi, obd = oracle.best_in_windows(k, d, uright, og, q, pd, inv_s2, oga)
    gbi, gbd = pkg.best_in_windows(k, d, uright, pg, q, pd, inv_s2, 0, pga)
    assert (obi >= 0).sum() > 300
    np.testing.assert_array_equal(gbi, obi)
    np.testing.assert_array_equal(gbd, obd)
    if gate:   # the gate must actually reject something the ungated search accepts
        ubi, _ = oracle.best_in_windows(k, d, uright, og, q, pd, None, oga)
        assert (ubi != obi).sum() > 10


@pytest.mark.parametrize("distorted", [False, True])
def test_search_by_projection_sim3(pkg, oracle, synth, matcher_path, distorted):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:290-403):
    Sim3 projection (oracle stage) + orbm_match_windows with TH_LOW, no orientation check."""
    ks, w, h, rng, k, d, sf, cam, log_sf, m = _kfside(oracle, synth, 4, distorted)
    S = ks.pose(rng, scale=1.07)
    pts, pd, _ = ks.points_for(oracle, rng, k, d, sf, S, m, scale=1.07)
    pts["valid"] = rng.random(m) > 0.1
    og, oga, _ = ks.geoms(oracle, w, h, distorted)
    pg, pga, _ = ks.geoms(pkg, w, h, distorted)
    matched = np.full(len(k), -1, np.int32)
    matched[rng.choice(len(k), 60, replace=False)] = -2
    on, om = oracle.search_by_projection_sim3(k, d, og, sf, log_sf, cam, S, pts, pd, matched, 10, oga)
    q = oracle.sim3_window_queries(pts, og, sf, log_sf, cam, S, 10.0)
    gn, gm = pkg.match_windows(k, d, None, pg, q, pd, matched, None, 50, False, 0, pga)
    assert on > 200
    assert gn == on
    np.testing.assert_array_equal(gm, om)


def test_distinctive_descriptors(pkg, oracle):
    """orbm_distinctive_descriptors (MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:252-317) for a
    batch of map points with 0..300 observations: least-median row, first minimum wins; sizes around the
    64-lane chunks and past the register-cached limit (256) included."""
    rng = np.random.default_rng(31)
    sizes = [0, 1, 2, 3, 4, 5, 17, 63, 64, 65, 128, 129, 255, 256, 257, 300] + list(rng.integers(1, 40, 200))
    blocks, offsets = [], [0]
    for n in sizes:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        # a cluster around one descriptor with a few outliers; coarse noise makes equal medians frequent
        noise = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & \
            rng.integers(0, 256, (n, 32), dtype=np.uint8) & np.uint8(0x0F)
        d = base[None, :] ^ noise
        out = rng.random(n) < 0.15
        d[out] = rng.integers(0, 256, (int(out.sum()), 32), dtype=np.uint8)
        if n > 3:
            d[n // 2] = d[0]            # exact duplicates: equal medians, the first must win
        blocks.append(d)
        offsets.append(offsets[-1] + n)
    desc = np.concatenate(blocks)
    br, bm = pkg.distinctive_descriptors(desc, offsets)
    ties = 0
    for p, d in enumerate(blocks):
        oi, om = oracle.distinctive_descriptor(d)
        assert br[p] == oi, (p, len(d))
        if len(d):
            assert bm[p] == om
            ties += 1 if len(d) > 3 else 0
    assert ties > 100


# ---- Frame::isInFrustum on the device, alone and fused with SearchByProjection(F, MPs)
def _local_map(pkg, oracle, synth, seed, m=3000):
    ks, w, h, rng, k, d, sf, cam, log_sf, _ = _kfside(oracle, synth, seed, False, m)
    T = ks.pose(rng)
    pts3, pd, idx = ks.points_for(oracle, rng, k, d, sf, T, m, bits=3)
    pts3["valid"] = rng.random(m) > 0.1
    # level boundaries: make max_distance/dist land within a few ulps of sf^k for part of the points
    R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    Ow = (-R.T @ t).astype(np.float32)
    dist = np.sqrt(((np.stack([pts3["wx"], pts3["wy"], pts3["wz"]], 1) - Ow).astype(np.float64) ** 2).sum(1)).astype(np.float32)
    edge = rng.random(m) < 0.3
    kk = rng.integers(0, 8, m)
    ulp = rng.integers(-3, 4, m)
    target = (np.float32(1.2) ** kk).astype(np.float32)
    md = (dist * target).astype(np.float32)
    md = (md.view(np.int32) + ulp.astype(np.int32)).view(np.float32)
    pts3["max_distance"] = np.where(edge, md, pts3["max_distance"])
    pts3["min_distance"] = np.where(edge, md * np.float32(0.1), pts3["min_distance"])
    obs = rng.integers(0, 6, m).astype(np.int32)
    wp = np.zeros(m, pkg.WORLDPOINT_DTYPE)
    for f in ("valid", "wx", "wy", "wz", "nx", "ny", "nz", "max_distance", "min_distance"):
        wp[f] = pts3[f]
    wp["observations"] = obs
    return ks, w, h, rng, k, d, sf, cam, log_sf, T, pts3, wp, pd, obs


def test_is_in_frustum(pkg, oracle, synth):
    """orbm_is_in_frustum vs Frame::isInFrustum (src/Frame.cc:284-340) restated with the C library's logf: every
    field identical, including the predicted level of points sitting within 3 ulps of a level boundary."""
    ks, w, h, rng, k, d, sf, cam, log_sf, T, pts3, wp, pd, obs = _local_map(pkg, oracle, synth, 6)
    want = oracle.is_in_frustum(pts3, obs, T, cam, oracle.grid_geom(w, h), 0.5, log_sf, 8)
    thr = pkg.predict_scale_thresholds(log_sf, 8)
    pcam = pkg.Camera(ks.FX, ks.FY, ks.CX, ks.CY, ks.MBF, np.float32(ks.MBF) / np.float32(ks.FX))
    got = pkg.is_in_frustum(wp, T, pcam, pkg.grid_geom(w, h), 0.5, thr, 8)
    assert 0.4 * len(wp) < want["in_view"].sum() < 0.95 * len(wp)
    assert len(set(want["level"][want["in_view"] == 1])) == 8
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)


def test_search_local_points_fused(pkg, oracle, synth, matcher_path):
    """orbm_search_local_points = isInFrustum + SearchByProjection(F, MPs) on the device (Tracking::SearchLocalPoints,
    src/Tracking.cc:1305-1339) against the two oracle stages chained on the host."""
    ks, w, h, rng, k, d, sf, cam, log_sf, T, pts3, wp, pd, obs = _local_map(pkg, oracle, synth, 7, m=2500)
    n = len(k)
    uright = np.where(rng.random(n) < 0.5, k["x"] - rng.uniform(1, 40, n), -1).astype(np.float32)
    frame_mp = np.full(n, -1, np.int32)
    held = rng.choice(n, 150, replace=False)
    frame_mp[held[:75]] = rng.choice(len(wp), 75, replace=False)
    frame_mp[held[75:]] = -2
    ext_obs = rng.integers(0, 3, n).astype(np.int32)
    proj = oracle.is_in_frustum(pts3, obs, T, cam, oracle.grid_geom(w, h), 0.5, log_sf, 8)
    for th in (1.0, 3.0):
        on, ofm = oracle.search_by_projection_mp(k, d, uright, oracle.grid_geom(w, h), sf, proj, pd, frame_mp, ext_obs, th, 0.8)
        thr = pkg.predict_scale_thresholds(log_sf, 8)
        pcam = pkg.Camera(ks.FX, ks.FY, ks.CX, ks.CY, ks.MBF, np.float32(ks.MBF) / np.float32(ks.FX))
        gn, gfm, gproj = pkg.search_local_points(k, d, uright, pkg.grid_geom(w, h), sf, wp, pd, T, pcam, 0.5, thr, frame_mp,
                                                 ext_obs, th, 0.8)
        assert on > 200
        assert gn == on
        np.testing.assert_array_equal(gfm, ofm)
        for f in proj.dtype.names:
            np.testing.assert_array_equal(gproj[f], proj[f], err_msg=f)


# ---- DBoW2 vocabulary descent + ORBmatcher::SearchByBoW
@pytest.fixture(scope="module")
def bow(pkg, oracle):
    import bow_scene as bs
    rng = np.random.default_rng(41)
    voc = bs.make_vocabulary(rng, k=10, L=3)
    ov = oracle.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    gv = pkg.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    return bs, rng, voc, ov, gv


def test_vocabulary_transform(pkg, oracle, bow, tmp_path):
    """orbv_transform vs TemplatedVocabulary::transform(feature, id, weight, nid, levelsup)
    (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1230-1271): word, weight and the node `levelsup` levels above
    the leaf for every feature, with early leaves, stopped words and levelsup beyond the tree depth; the same
    through the ORBvoc.txt text format."""
    bs, rng, voc, ov, gv = bow
    feats = bs.features_near_words(rng, voc, 3000)
    info = gv.info()
    assert info["nnodes"] == len(voc["parent"]) and info["nwords"] == int(voc["is_leaf"].sum())
    bs.write_text(voc, tmp_path / "voc.txt")
    tv = pkg.Vocabulary(path=tmp_path / "voc.txt")
    assert tv.info() == info
    for levelsup in (0, 1, 2, 3, 4):
        w, nid, wt = gv.transform(feats, levelsup)
        w2, nid2, wt2 = tv.transform(feats, levelsup)
        for i in range(0, len(feats), 7):
            ow, owt, onid = ov.transform_one(feats[i], levelsup)
            assert (w[i], nid[i], wt[i]) == (ow, onid, owt), (levelsup, i)
        np.testing.assert_array_equal(w2, w); np.testing.assert_array_equal(nid2, nid); np.testing.assert_array_equal(wt2, wt)
    w, nid, wt = gv.transform(feats, 1)
    assert (wt == 0).sum() > 20 and len(set(nid)) > 50      # stopped words present, many level-2 nodes hit


@pytest.mark.parametrize("variant", ["kf_frame", "kf_kf"])
def test_search_by_bow(pkg, oracle, bow, variant):
    """orbm_search_by_bow vs ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288 KeyFrame-Frame: best <= TH_LOW, all
    candidates; :522-655 KeyFrame-KeyFrame: best < TH_LOW, candidates need a good map point)."""
    bs, rng, voc, ov, gv = bow
    nq, nc = 1000, 1100
    base = bs.features_near_words(rng, voc, 1200, noise_bits=4)
    def frame(n):   # mostly distinct scene points (the ratio test can pass), some seen twice (contention for a candidate)
        nd = n // 7
        src = np.concatenate([rng.permutation(len(base))[:n - nd], rng.integers(0, len(base), nd)])
        rng.shuffle(src)
        d = base[src].copy()
        noise = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & \
            rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & \
            rng.integers(0, 256, (n, 32), dtype=np.uint8)
        ang = (src * 0.3 + rng.normal(0, 4, n)) % 360
        return d ^ noise, ang.astype(np.float32)
    qd, qa = frame(nq)
    cd, ca = frame(nc)
    ca = ((ca + 25) % 360).astype(np.float32)
    _, _, fq = ov.transform(qd, 1)
    _, _, fc = ov.transform(cd, 1)
    nqs, qit, ncs, cit = bs.intersect(fq, fc)
    assert len(nqs) > 50
    qv = (rng.random(nq) > 0.2).astype(np.uint8)
    if variant == "kf_frame":
        cv, strict, maxd = None, 0, 50
    else:
        cv, strict, maxd = (rng.random(nc) > 0.15).astype(np.uint8), 1, 49
    for ratio, ori in ((0.7, True), (0.9, False)):
        on, om = oracle.search_by_bow(qd, qa, qv, cd, ca, cv, nqs, qit, ncs, cit, 50, strict, ratio, ori)
        gn, gm = pkg.search_by_bow(qd, qa, qv, cd, ca, cv, nqs, qit, ncs, cit, maxd, ratio, ori)
        assert on > 150
        assert gn == on
        np.testing.assert_array_equal(gm, om)


def test_search_for_triangulation(pkg, oracle, bow):
    """orbm_search_for_triangulation vs ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-825): node-constrained
    search, epipole and epipolar-line tests, LAST candidate of minimum distance wins, rotation consistency."""
    bs, rng, voc, ov, gv = bow
    n1, n2 = 1000, 1050
    base = bs.features_near_words(rng, voc, 1100, noise_bits=4)
    pos = np.stack([rng.uniform(20, 1220, len(base)), rng.uniform(20, 356, len(base))], 1)

    def frame(n, dx):
        nd = n // 6
        src = np.concatenate([rng.permutation(len(base))[:n - nd], rng.integers(0, len(base), nd)])
        rng.shuffle(src)
        noise = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        for _ in range(4):
            noise &= rng.integers(0, 256, (n, 32), dtype=np.uint8)
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = pos[src, 0] + dx + rng.normal(0, 0.3, n)         # pure horizontal motion: epipolar lines are the rows
        k["y"] = pos[src, 1] + rng.normal(0, 0.8, n)
        k["octave"] = rng.integers(0, 8, n)
        k["angle"] = (src * 0.5 + rng.normal(0, 3, n)) % 360
        flags = (rng.random(n) > 0.2).astype(np.uint8) | ((rng.random(n) < 0.4).astype(np.uint8) << 1)
        return base[src] ^ noise, k, flags
    d1, k1, f1 = frame(n1, 0.0)
    d2, k2, f2 = frame(n2, -12.0)
    # fundamental matrix of a pure x-translation: l = x1'F12 = (0, -1, y1) up to scale
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    ex, ey = 600.0, 180.0                                         # an epipole inside the image: the distance test bites
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sigma2 = (sf * sf).astype(np.float32)
    _, _, fv1 = ov.transform(d1, 1)
    _, _, fv2 = ov.transform(d2, 1)
    nqs, qit, ncs, cit = bs.intersect(fv1, fv2)
    for ori in (True, False):
        on, om = oracle.search_for_triangulation(k1, d1, f1, k2, d2, f2, nqs, qit, ncs, cit, F12, ex, ey, sf, sigma2, 50, ori)
        gn, gm = pkg.search_for_triangulation(k1, d1, f1, k2, d2, f2, nqs, qit, ncs, cit, F12, ex, ey, sf, sigma2, 50, ori)
        assert on > 150
        assert gn == on
        np.testing.assert_array_equal(gm, om)
    # without the epipolar gate (all-zero F12: den == 0) nothing may match
    zn, zm = pkg.search_for_triangulation(k1, d1, f1, k2, d2, f2, nqs, qit, ncs, cit, np.zeros((3, 3), np.float32), ex, ey, sf, sigma2)
    assert zn == 0 and (zm == -1).all()


def test_degenerate_sizes_and_bad_arguments(pkg, oracle):
    """Empty inputs are answers, not errors; malformed inputs are ORBX_ERR_ARG with a message (no crash)."""
    g = pkg.grid_geom(640, 480)
    k0 = np.zeros(0, pkg.KP_DTYPE); d0 = np.zeros((0, 32), np.uint8)
    k3 = np.zeros(3, pkg.KP_DTYPE); k3["x"] = [10, 20, 30]; k3["y"] = [10, 20, 30]
    d3 = np.zeros((3, 32), np.uint8)
    q0 = np.zeros(0, pkg.WINDOW_DTYPE)
    bi, bd = pkg.best_in_windows(k3, d3, None, g, q0, d0)
    assert len(bi) == 0
    q2 = np.zeros(2, pkg.WINDOW_DTYPE)                      # invalid queries only
    bi, bd = pkg.best_in_windows(k3, d3, None, g, q2, np.zeros((2, 32), np.uint8))
    assert list(bi) == [-1, -1] and list(bd) == [256, 256]
    bi, bd = pkg.best_in_windows(k0, d0, None, g, q2, np.zeros((2, 32), np.uint8))
    assert list(bi) == [-1, -1]
    n, h = pkg.match_windows(k3, d3, None, g, q0, d0, np.full(3, -1, np.int32))
    assert n == 0 and list(h) == [-1, -1, -1]
    br, bm = pkg.distinctive_descriptors(d0, [0, 0, 0])
    assert list(br) == [-1, -1]
    n, mq = pkg.search_by_bow(d3, np.zeros(3, np.float32), np.ones(3, np.uint8), d3, np.zeros(3, np.float32), None,
                              [0], np.zeros(0, np.int32), [0], np.zeros(0, np.int32), 50, 0.7)
    assert n == 0 and list(mq) == [-1, -1, -1]
    with pytest.raises(pkg.OrbxError):                      # item index out of range
        pkg.search_by_bow(d3, np.zeros(3, np.float32), np.ones(3, np.uint8), d3, np.zeros(3, np.float32), None,
                          [0, 1], [7], [0, 1], [0], 50, 0.7)
    with pytest.raises(pkg.OrbxError):                      # offsets not monotonic
        pkg.distinctive_descriptors(d3, [0, 2, 1])
    with pytest.raises(pkg.OrbxError):                      # a node whose parent does not precede it
        pkg.Vocabulary(2, 1, 0, 0, [0, 2, 1], [0, 1, 1], np.zeros((3, 32), np.uint8), [0, 1.0, 1.0])
    thr = pkg.predict_scale_thresholds(np.float32(np.log(np.float32(1.2))), 8)
    cam = pkg.Camera(500, 500, 320, 240, 40, 0.08)
    out = pkg.is_in_frustum(np.zeros(0, pkg.WORLDPOINT_DTYPE), np.eye(4, dtype=np.float32), cam, g, 0.5, thr, 8)
    assert len(out) == 0
