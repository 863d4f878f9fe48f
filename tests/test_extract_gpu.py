"""GPU parity of the HIP extractor against the CPU oracle, stage by stage, through the C ABI.

Bar: keypoint sets, order, scores, octaves and descriptors bit-exact; angles within 1e-4
(they are in fact bit-identical: same float operations, no FMA contraction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _staged_tests_use_the_developer_build(hooks):
    """Every test of this module compares stage by stage (orbx_debug_level_points ...): its extractors come from the developer
    build.  test_product_library_end_to_end below runs the same images through the product library."""
    yield


SIZES = [(640, 480, 1000), (1241, 376, 1000), (1241, 376, 2000), (752, 480, 1000), (320, 240, 500)]


def _cands(a):
    return np.stack([a["x"], a["y"], a["score"]], 1).astype(np.int32) if len(a) else np.zeros((0, 3), np.int32)


@pytest.mark.parametrize("w,h,nf", SIZES)
@pytest.mark.parametrize("fast_kernel", ["auto", "strips"])
def test_staged_parity(pkg, oracle, synth, w, h, nf, fast_kernel):
    """fast_kernel: a single image takes k_fast_cells by default (the batch-size rule of orbx_extract.hip); "strips" forces
    k_fast_strips, the kernel a large batch runs on (developer knob 6 = 3)."""
    img = synth.frame(w, h, k=3)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ok, od = None, None
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    ok, od = orc.extract(img)
    pkg.set_default_option(6, 3 if fast_kernel == "strips" else 0)
    try:
        gk, gd = ex(img)
    finally:
        pkg.set_default_option(6, 0)
    for l in range(8):
        np.testing.assert_array_equal(ex.pyramid_level(l, padded=True), orc.pyramid_level(l, padded=True),
                                      err_msg="pyramid level %d" % l)
    for l in range(8):
        np.testing.assert_array_equal(ex.debug_level_points(l, 0), _cands(orc.level_candidates(l)),
                                      err_msg="FAST candidates level %d" % l)
    for l in range(8):
        np.testing.assert_array_equal(ex.debug_level_points(l, 1), _cands(orc.level_keypoints(l)),
                                      err_msg="quad-tree level %d" % l)
    assert len(gk) == len(ok)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(gk[f], ok[f], err_msg=f)
    np.testing.assert_allclose(gk["angle"], ok["angle"], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(gd, od)


def test_batch_matches_single(pkg, oracle, synth):
    w, h = 640, 480
    imgs = synth.batch(w, h, 5, k0=10)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    res = ex.extract_batch(imgs)
    orc = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for b in range(5):
        ok, od = orc.extract(imgs[b])
        gk, gd = res[b]
        assert len(gk) == len(ok)
        np.testing.assert_array_equal(gk[["x", "y", "response", "octave"]], ok[["x", "y", "response", "octave"]])
        np.testing.assert_array_equal(gd, od)


def _compare(pkg, oracle, img, nf, sf=1.2, nl=8, ini=20, mn=7):
    """One image through the HIP path, TWICE - with the FAST kernel a single image takes by default (k_fast_cells, one wave per
    cell) and with the one a large batch takes (k_fast_strips, forced by developer knob 6 = 3) - against the oracle."""
    ex = pkg.ORBextractor(nf, sf, nl, ini, mn)
    orc = oracle.Extractor(nf, sf, nl, ini, mn)
    ok, od = orc.extract(img)
    knob = pkg.set_default_option
    if _KNOB6[0] == 0:
        knob(6, 3)
        try:
            _check_against(ex, orc, ok, od, img, nl)
        finally:
            knob(6, 0)
    return _check_against(ex, orc, ok, od, img, nl)


_KNOB6 = [0]   # tests that set knob 6 themselves (test_fast_cell_kernel_instances) say so here


def _check_against(ex, orc, ok, od, img, nl, staged=True):
    gk, gd = ex(img)
    for l in range(nl if staged else 0):
        np.testing.assert_array_equal(ex.debug_level_points(l, 0), _cands(orc.level_candidates(l)),
                                      err_msg="FAST candidates level %d" % l)
        np.testing.assert_array_equal(ex.debug_level_points(l, 1), _cands(orc.level_keypoints(l)),
                                      err_msg="quad-tree level %d" % l)
    assert len(gk) == len(ok)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(gk[f], ok[f], err_msg=f)
    np.testing.assert_allclose(gk["angle"], ok["angle"], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(gd, od)
    return gk


def test_full_hd_4000_features(pkg, oracle, synth):
    k = _compare(pkg, oracle, synth.frame(1920, 1080, 30), 4000)
    assert len(k) >= 3900


def test_4k_image_and_size_limits(pkg, oracle, synth):
    """3840x2160: level 0 can hold 2.2 M candidates (the best-key election packs response << 24 | ~index);
    sides above 4095 px and portrait images whose level has no quad-tree root are refused with an error."""
    k = _compare(pkg, oracle, synth.frame(3840, 2160, 33), 5000)
    assert len(k) >= 5000
    with pytest.raises(pkg.OrbxError):
        pkg.ORBextractor(500, 1.2, 8, 20, 7)(np.zeros((1000, 4096), np.uint8))
    with pytest.raises(pkg.OrbxError):       # round(w/h) = 0 roots: the reference divides by zero (src/ORBextractor.cc:543-545)
        pkg.ORBextractor(500, 1.2, 4, 20, 7)(np.zeros((2000, 900), np.uint8))


def test_ragged_size_and_small_budget(pkg, oracle, synth):
    _compare(pkg, oracle, synth.frame(641, 479, 31), 300)
    _compare(pkg, oracle, synth.frame(333, 257, 32), 50)


def test_other_pyramid_parameters(pkg, oracle, synth):
    _compare(pkg, oracle, synth.frame(640, 480, 33), 800, sf=1.5, nl=4)
    _compare(pkg, oracle, synth.frame(640, 480, 34), 1000, sf=1.1, nl=12)
    _compare(pkg, oracle, synth.frame(640, 480, 35), 600, ini=5, mn=12)   # iniTh < minTh
    _compare(pkg, oracle, synth.frame(640, 480, 36), 600, ini=40, mn=40)


def test_sparse_image_fewer_candidates_than_budget(pkg, oracle):
    # a flat image with a few isolated squares: most cells fall back to minTh and find nothing,
    # every quad-tree node ends up with one key before N is reached (all bNoMore, :669)
    rng = np.random.default_rng(7)
    img = np.full((480, 640), 90, np.uint8)
    for _ in range(40):
        x, y = rng.integers(30, 600), rng.integers(30, 440)
        img[y:y + 9, x:x + 9] = 90 + rng.integers(10, 120)
    k = _compare(pkg, oracle, img, 1000)
    assert 0 < len(k) < 600


def test_dense_random_texture(pkg, oracle):
    # uniform noise: FAST fires almost everywhere NMS allows -> stresses slot capacity/order
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (376, 620), dtype=np.uint8)
    _compare(pkg, oracle, img, 2000)


def test_extreme_images(pkg, oracle):
    """Saturated 0/255 blocks (the 7-tap blur sums reach their clamp, FAST scores their maximum), a one-pixel
    checkerboard (every pixel a candidate before NMS) and corners hugging the image border."""
    rng = np.random.default_rng(21)
    h, w = 480, 640
    sat = np.zeros((h, w), np.uint8)
    for _ in range(400):
        x, y = rng.integers(0, w - 4), rng.integers(0, h - 4)
        sat[y:y + rng.integers(2, 40), x:x + rng.integers(2, 40)] = 255 if rng.random() < 0.5 else 0
    _compare(pkg, oracle, sat, 1000)
    yy, xx = np.mgrid[0:h, 0:w]
    for pitch in (1, 2, 5):
        chk = (((xx // pitch + yy // pitch) % 2) * 255).astype(np.uint8)
        chk[160:200] = 128
        _compare(pkg, oracle, chk, 1500)
    frame = np.full((h, w), 20, np.uint8)
    frame[:17] = 240; frame[-17:] = 240; frame[:, :17] = 240; frame[:, -17:] = 240
    for _ in range(200):
        frame[rng.integers(0, h), rng.integers(0, w)] = 255
    _compare(pkg, oracle, frame, 500)
    _compare(pkg, oracle, sat, 1)          # a budget of one feature
    _compare(pkg, oracle, sat, 7, nl=1)    # a single level


def test_batch_with_empty_and_flat_frames(pkg, oracle, synth):
    w, h = 640, 480
    imgs = np.stack([synth.frame(w, h, 40), np.full((h, w), 50, np.uint8), synth.frame(w, h, 41)])
    ex = pkg.ORBextractor(700, 1.2, 8, 20, 7)
    res = ex.extract_batch(imgs)
    orc = oracle.Extractor(700, 1.2, 8, 20, 7)
    for b in range(3):
        ok, od = orc.extract(imgs[b])
        assert len(res[b][0]) == len(ok)
        np.testing.assert_array_equal(res[b][1], od)
    assert len(res[1][0]) == 0
    # empty image: silent, outputs empty (src/ORBextractor.cc:1046-1047)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # too small for the reference's cell grid at the last level: argument error, not a crash
    with pytest.raises(pkg.OrbxError) as e:
        ex(np.zeros((100, 100), np.uint8))
    assert e.value.status == pkg.ORBX_ERR_ARG
    # the handle still works afterwards and adapts to a new image size
    k2, _ = ex(synth.frame(752, 480, 42))
    assert len(k2) > 600


def test_determinism_and_batch_size_independence(pkg, synth):
    w, h = 1241, 376
    imgs = synth.batch(w, h, 3, k0=50)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    a = ex.extract_batch(imgs)
    b = ex.extract_batch(np.concatenate([imgs, imgs[::-1]]))
    for i in range(3):
        for j in (i, 5 - i):
            assert a[i][0].tobytes() == b[j][0].tobytes() and a[i][1].tobytes() == b[j][1].tobytes()


def test_two_extractors_on_two_host_threads(pkg, oracle, synth):
    """The reference runs the left and right extractor instances on two std::threads
    (src/Frame.cc:78-81): two handles must be usable concurrently."""
    import threading
    w, h = 752, 480
    left, right = synth.stereo_pair_blocky(w, h, 60)
    exs = [pkg.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(2)]
    out = [None, None]

    def work(i, img):
        for _ in range(5):
            out[i] = exs[i](img)

    ts = [threading.Thread(target=work, args=(i, im)) for i, im in enumerate((left, right))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i, im in enumerate((left, right)):
        ok, od = oracle.Extractor(1000, 1.2, 8, 20, 7).extract(im)
        assert len(out[i][0]) == len(ok)
        np.testing.assert_array_equal(out[i][1], od)
        np.testing.assert_array_equal(out[i][0]["x"], ok["x"])


def test_device_batch_on_a_torch_side_stream(pkg, oracle, synth):
    """orbx_extract_batch_device on a non-default stream with torch-owned buffers."""
    import torch
    w, h, B = 640, 480, 6
    imgs = synth.batch(w, h, B, k0=70)
    ex = pkg.ORBextractor(800, 1.2, 8, 20, 7)
    ex(imgs[0])
    cap = ex.max_keypoints()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = torch.from_numpy(imgs).cuda()
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap,
                                s.cuda_stream)
    s.synchronize()
    orc = oracle.Extractor(800, 1.2, 8, 20, 7)
    for b in range(B):
        ok, od = orc.extract(imgs[b])
        n = int(cnt[b])
        assert n == len(ok)
        np.testing.assert_array_equal(desc[b, :n].cpu().numpy(), od)
        got = kps[b, :n].cpu().numpy().view(np.uint8).reshape(n, 28)
        np.testing.assert_array_equal(got, ok.view(np.uint8).reshape(n, 28))
    # mvImagePyramid of a batch slot
    np.testing.assert_array_equal(ex.pyramid_level(5, b=3), _pyr(oracle, imgs[3], 800, 5))


def test_stage_profiling_modes(pkg, synth):
    """orbx_set_profiling: mode 1 brackets every stage, mode 2 only the FAST stage, mode 3 the FAST stage of every 4th call;
    results do not depend on the mode."""
    imgs = synth.batch(640, 480, 4, k0=90)
    ex = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    base = ex.extract_batch(imgs)
    ex.set_profiling(1)
    for _ in range(3):
        r1 = ex.extract_batch(imgs)
    ms, n = ex.stage_ms()
    assert n == 3 and (ms[:4] > 0).all() and abs(ms[:4].sum() - ms[4]) < 0.2 * ms[4]
    ex.set_profiling(2)
    for _ in range(40):          # more calls than the event ring holds
        r2 = ex.extract_batch(imgs)
    ms2, n2 = ex.stage_ms()
    assert n2 == 40 and ms2[1] > 0 and ms2[0] == 0 and ms2[2] == 0 and ms2[3] == 0 and ms2[4] == 0
    assert 0.3 * ms[1] < ms2[1] < 3 * ms[1]
    ex.set_profiling(3)
    for _ in range(10):          # calls 0, 4, 8 are bracketed
        r3 = ex.extract_batch(imgs)
    ms3, n3 = ex.stage_ms()
    assert n3 == 3 and 0.3 * ms[1] < ms3[1] < 3 * ms[1] and ms3[0] == 0
    ex.set_profiling(0)
    for r in (r1, r2, r3):
        for (k, d), (k0, d0) in zip(r, base):
            np.testing.assert_array_equal(d, d0)
            np.testing.assert_array_equal(k.view(np.uint8), k0.view(np.uint8))
    with pytest.raises(pkg.OrbxError):
        ex.set_profiling(4)


def _pyr(oracle, img, nf, level):
    o = oracle.Extractor(nf, 1.2, 8, 20, 7)
    o.extract(img)
    return o.pyramid_level(level)


def test_fused_pyramid_kernel(pkg, oracle, synth):
    """k_pyramid_fused (all levels in one launch) is kept as the alternative to the level-per-launch
    pyramid (and is used for scale factors > 3): same bytes, frame included."""
    pkg.set_default_option(5, 1)
    try:
        _compare(pkg, oracle, synth.frame(752, 480, 82), 1000)
        img = synth.frame(641, 479, 83)
        ex = pkg.ORBextractor(500, 1.2, 8, 20, 7)
        ex(img)
        o = oracle.Extractor(500, 1.2, 8, 20, 7)
        o.extract(img)
        for lvl in range(8):
            np.testing.assert_array_equal(ex.pyramid_level(lvl, padded=True), o.pyramid_level(lvl, padded=True), err_msg=str(lvl))
    finally:
        pkg.set_default_option(5, 0)


@pytest.mark.parametrize("knob", [2, 3, 4])
def test_pyramid_forms_agree(pkg, oracle, synth, knob):
    """ORBX_OPT_PYRAMID_FORM: 2 = one launch per level (the default of large batches), 3 = the hybrid (levels 1, 2 per launch, levels
    3.. chained through LDS in ONE launch; measured slower, kept as a tested alternative), 4 = level chains (k_pyr_chain, round 5: the
    default of batches of up to four images): every level byte for byte, frame included, for odd sizes, a 3-, 4- and a 10-level
    pyramid, scale factors 1.1 to 1.7, and the key points behind them."""
    pkg.set_default_option(5, knob)
    try:
        for (w, h, nl, sf, seed) in ((641, 479, 8, 1.2, 84), (1241, 376, 8, 1.2, 85), (500, 400, 4, 1.3, 86), (900, 700, 10, 1.2, 87), (320, 240, 3, 1.2, 88),
                                     (1023, 517, 5, 1.5, 89), (770, 431, 9, 1.1, 90), (258, 255, 6, 1.2, 91), (1000, 600, 3, 1.7, 92)):
            img = synth.frame(w, h, seed)
            ex = pkg.ORBextractor(500, sf, nl, 20, 7)
            k, d = ex(img)
            o = oracle.Extractor(500, sf, nl, 20, 7)
            ok, od = o.extract(img)
            assert k.tobytes() == ok.tobytes() and d.tobytes() == od.tobytes(), (w, h, nl)
            for lvl in range(nl):
                np.testing.assert_array_equal(ex.pyramid_level(lvl, padded=True), o.pyramid_level(lvl, padded=True), err_msg=str((w, h, nl, lvl)))
            ex.close()
    finally:
        pkg.set_default_option(5, 0)


@pytest.mark.parametrize("rows", [1, 2])
def test_pyramid_rows_per_wave_agree(pkg, oracle, synth, rows):
    """ORBX_OPT_PYR_ROWS: k_pyr_level with 8 (1) or 16 (2) output rows per wave - by default picked per level from the number of
    waves it leaves the GPU; every level byte for byte, frame included (odd sizes, heights that are no multiple of 16, scale factors
    at which 16 rows need more source rows than the kernel fetches up front and fall to its row-by-row path), single images and a batch."""
    pkg.set_default_option(22, rows)
    try:
        for (w, h, nl, sf, seed) in ((641, 479, 8, 1.2, 84), (1241, 376, 8, 1.2, 85), (500, 400, 4, 1.3, 86), (1023, 517, 5, 1.5, 89), (770, 431, 9, 1.1, 90),
                                     (258, 255, 6, 1.2, 91), (1000, 600, 3, 1.7, 92)):
            img = synth.frame(w, h, seed)
            ex = pkg.ORBextractor(500, sf, nl, 20, 7)
            k, d = ex(img)
            o = oracle.Extractor(500, sf, nl, 20, 7)
            ok, od = o.extract(img)
            assert k.tobytes() == ok.tobytes() and d.tobytes() == od.tobytes(), (w, h, nl)
            for lvl in range(nl):
                np.testing.assert_array_equal(ex.pyramid_level(lvl, padded=True), o.pyramid_level(lvl, padded=True), err_msg=str((w, h, nl, lvl)))
            ex.close()
        imgs = np.stack([synth.frame(752, 480, 300 + i) for i in range(5)])
        ex = pkg.ORBextractor(800, 1.2, 8, 20, 7)
        o = oracle.Extractor(800, 1.2, 8, 20, 7)
        for b, (k, d) in enumerate(ex.extract_batch(imgs)):
            ok, od = o.extract(imgs[b])
            assert k.tobytes() == ok.tobytes() and d.tobytes() == od.tobytes(), b
        ex.close()
    finally:
        pkg.set_default_option(22, 0)


def test_small_budget_many_roots(pkg, oracle, synth):
    """The coarsest level of a 1229x497 image at scale 1.5 has 4 quad-tree roots and a budget of 5: the first pass splits all
    roots (16 nodes) before N is looked at (src/ORBextractor.cc:606-672)."""
    k = _compare(pkg, oracle, synth.frame(1229, 497, 672), 100, sf=1.5, nl=6)
    assert (k["octave"] == 5).sum() == 16


def test_large_scale_factor(pkg, oracle, synth):
    """scaleFactor 2.6 and 3.4: source columns of a lane's pixel pair up to 4 apart (3.4 takes the fused kernel)."""
    _compare(pkg, oracle, synth.frame(1920, 1080, 84), 500, sf=2.6, nl=3)
    _compare(pkg, oracle, synth.frame(1920, 1080, 85), 500, sf=3.4, nl=3)


def test_quadtree_sweep_kernel_alone(pkg, oracle, synth):
    """k_octree (one key sweep per pass) is the exact fallback of k_octree_pyr: run it alone."""
    pkg.set_default_option(4, 1)
    try:
        _compare(pkg, oracle, synth.frame(752, 480, 80), 1000)
        _compare(pkg, oracle, synth.frame(640, 480, 81), 2000)
    finally:
        pkg.set_default_option(4, 0)


def test_quadtree_multi_workgroup_form(pkg, oracle, synth):
    """In a small batch, levels with >= 600 FAST cells (the finest levels of 1920x1080) share their quad-tree between 8 workgroups: partial
    histograms merged by the last workgroup to arrive, best-key election merged by global atomicMax (k_octree_big<1>, <2>).
    Developer knob 4 = 2 forces that form on EVERY level, = 3 forbids it; results must not depend on the choice - including
    levels that outgrow the count pyramid inside the multi-workgroup form (clustered keys), empty levels and a batch."""
    rng = np.random.default_rng(95)
    clustered = np.full((480, 640), 128, np.uint8)
    clustered[200:280, 260:380] = rng.integers(0, 256, (80, 120), dtype=np.uint8)   # all keys in one corner of the tree
    flat = np.full((480, 640), 77, np.uint8)                                          # no keys at all
    for knob in (2, 3):
        pkg.set_default_option(4, knob)
        try:
            _compare(pkg, oracle, synth.frame(1241, 376, 96), 1000)
            _compare(pkg, oracle, synth.frame(640, 480, 97), 2000)
            _compare(pkg, oracle, clustered, 1000)
            _compare(pkg, oracle, flat, 500)
            _compare(pkg, oracle, rng.integers(0, 256, (376, 620), dtype=np.uint8), 2000)
            # a batch: the arrival counters are per (image, level) and must come back to zero for the next call
            imgs = synth.batch(752, 480, 6, k0=98)
            ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
            orc = oracle.Extractor(1000, 1.2, 8, 20, 7)
            for rep in range(2):
                res = ex.extract_batch(imgs)
                for i in range(6):
                    ok, od = orc.extract(imgs[i])
                    assert res[i][0].tobytes() == ok.tobytes() or (len(res[i][0]) == len(ok) and (res[i][1] == od).all()
                                                                     and (res[i][0]["x"] == ok["x"]).all() and (res[i][0]["y"] == ok["y"]).all())
        finally:
            pkg.set_default_option(4, 0)
    _compare(pkg, oracle, synth.frame(1920, 1080, 99), 4000)     # default rule: a single image, the larger levels take the multi-workgroup form


@pytest.mark.parametrize("wide,form", [(2, 0), (2, 1), (2, 2), (1, 0), (1, 2)])
def test_quadtree_workgroup_width(pkg, oracle, synth, wide, form):
    """The quad-tree kernels exist in a 512- and a 1024-thread build (orbx_octree_wide.hip; images with a level of >= 600 FAST cells
    take the wide one).  Developer knob 11 = 2 forces the wide build on small images, = 1 the narrow one on large images; crossed
    with knob 4 (0 default, 1 the exact form alone, 2 every level multi-workgroup).  Same bytes, clustered and empty levels included."""
    rng = np.random.default_rng(195)
    clustered = np.full((480, 640), 128, np.uint8)
    clustered[200:280, 260:380] = rng.integers(0, 256, (80, 120), dtype=np.uint8)
    pkg.set_default_option(11, wide)
    pkg.set_default_option(4, form)
    try:
        _compare(pkg, oracle, synth.frame(1241, 376, 196), 2000)
        _compare(pkg, oracle, clustered, 1000)
        _compare(pkg, oracle, np.full((480, 640), 77, np.uint8), 500)
        _compare(pkg, oracle, synth.frame(1920, 1080, 197), 4000)
        _compare(pkg, oracle, rng.integers(0, 256, (376, 620), dtype=np.uint8), 3000)
    finally:
        pkg.set_default_option(11, 0)
        pkg.set_default_option(4, 0)


def test_fast_cell_kernel_instances(pkg, oracle, synth):
    """FAST runs as k_fast_strips (one wave per strip of four cells) on levels whose cells are at most 32 px wide and as
    k_fast_cells (one wave per cell) on the others.  Developer knob 6 forces k_fast_cells on every level: 1 = its instances
    with compile-time tile strides (44/48/52 dwords), 2 = the run-time-stride instance that serves every other configuration
    (reached naturally by 60-px-tall single-cell levels, stride 60)."""
    for knob in (1, 2):
        pkg.set_default_option(6, knob)
        _KNOB6[0] = knob
        try:
            _compare(pkg, oracle, synth.frame(752, 480, 85), 1000)
            _compare(pkg, oracle, synth.frame(1241, 376, 86), 1000)
        finally:
            _KNOB6[0] = 0
            pkg.set_default_option(6, 0)
    _compare(pkg, oracle, synth.frame(1241, 376, 86), 1000)
    _compare(pkg, oracle, synth.frame(333, 211, 87), 300, sf=1.3, nl=5)


def test_fast_strips_and_cells_mix(pkg, oracle, synth):
    """Sizes whose pyramid mixes strip levels (cells <= 32 px) with wide-cell levels (640x480: levels 5 and 7 have 33- and
    37-px cells), a last strip with one / two / three cells, a last column narrower than its neighbours, and thresholds that
    make the per-cell fallback (src/ORBextractor.cc:809-816) decide differently from cell to cell."""
    _compare(pkg, oracle, synth.frame(640, 480, 90), 1000)
    for w in (406, 437, 468, 499, 531):          # nCols = 12 .. 16 at level 0: every remainder of nCols mod 4
        _compare(pkg, oracle, synth.frame(w, 300, 91 + w), 600)
    rng = np.random.default_rng(93)
    img = np.full((376, 1241), 100, np.uint8)
    for _ in range(300):                                # faint and strong squares: some cells only reach minTh
        x, y, c = rng.integers(20, 1200), rng.integers(20, 340), int(rng.choice([9, 12, 15, 40, 90]))
        img[y:y + 7, x:x + 7] = 100 + c
    _compare(pkg, oracle, img, 1500)
    _compare(pkg, oracle, img, 1500, ini=12, mn=30)     # iniTh < minTh


def test_quadtree_pyramid_overflow_falls_back(pkg, oracle):
    """Clustered candidates: the tree gets deeper than the count pyramid in a few levels, which
    are then redone by k_octree; a large budget on a small textured patch forces it."""
    rng = np.random.default_rng(9)
    img = np.full((480, 640), 100, np.uint8)
    img[200:280, 260:380] = rng.integers(0, 256, (80, 120), dtype=np.uint8)   # all corners in 2 % of the image
    k = _compare(pkg, oracle, img, 3000)
    assert len(k) > 0  # the reference stops early on clustered input (size == prevSize, :669)


def test_host_api_latency_odd_width(pkg, synth):
    """orbx_extract from host memory at 1241x376 (a width that is not a multiple of 4): the image goes up as ONE
    linear copy with the caller's stride kept on the device — a 2-D copy of 1241-byte rows took 3 ms per image.
    Generous bound (measured 0.22 ms)."""
    import time
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img = synth.frame(1241, 376, 90)
    for _ in range(3):
        ex(img)
    t0 = time.perf_counter()
    for _ in range(10):
        k, d = ex(img)
    dt = (time.perf_counter() - t0) / 10
    assert len(k) > 900
    assert dt < 1.5e-3, "orbx_extract took %.3f ms per 1241x376 image" % (dt * 1e3)
    # a strided source (ROI of a wider buffer) gives the same keypoints
    wide = np.zeros((376, 1300), np.uint8)
    wide[:, 13:13 + 1241] = img
    k2, d2 = ex(wide[:, 13:13 + 1241])
    np.testing.assert_array_equal(k2, k)
    np.testing.assert_array_equal(d2, d)


def test_device_sincosf_equals_oracle_and_host_libm(pkg, oracle):
    """src/ORBextractor.cc:113 calls the FLOAT overloads (cosf / sinf).  The descriptor kernel's restatement of glibc's
    algorithm against the oracle's restatement (mode 0) AND this host's libm (mode 1), bit for bit, on 22 M floats of the
    angle domain [0, 2 pi] (every 49th float + dense windows around the algorithm's branch points)."""
    top = int(np.float32(6.2832).view(np.uint32))
    us = [np.arange(0, top, 49, dtype=np.uint32)]
    for centre in (0.0, 2.0 ** -12, 0.78539816, 1.5707964, 2.3561945, 3.1415927, 3.9269908, 4.712389, 5.4977871, 6.2831855):
        u0 = int(np.float32(centre).view(np.uint32))
        us.append(np.arange(max(u0 - 5000, 0), min(u0 + 5000, top), dtype=np.uint32))
    ang = np.concatenate(us).view(np.float32)
    gs, gc = pkg.debug_sincosf(ang)
    for mode in (0, 1):
        oracle.set_sincos_mode(mode)
        os_, oc = oracle.sincosf_array(ang)
        oracle.set_sincos_mode(0)
        np.testing.assert_array_equal(gs.view(np.uint32), os_.view(np.uint32), err_msg="sin, oracle mode %d" % mode)
        np.testing.assert_array_equal(gc.view(np.uint32), oc.view(np.uint32), err_msg="cos, oracle mode %d" % mode)


def test_blurred_patches_direct(pkg, oracle, synth):
    """SURVEY row a8 checked directly: the 7x7 sigma-2 blur is fused into the descriptor kernel and never stored, so a test hook
    dumps the 37x37 blurred block around every keypoint; it must equal the oracle's blurred LEVEL (cv::GaussianBlur of the
    whole level, <= 3.3 fixed point, REFLECT_101) at the same pixels - including keypoints whose block hangs over the level's
    edge (mirrored there) and a saturated image, where the taps' sum of 257 reaches the clamp."""
    rng = np.random.default_rng(31)
    sat = np.zeros((480, 640), np.uint8)
    for _ in range(400):
        x, y = rng.integers(0, 636), rng.integers(0, 476)
        sat[y:y + rng.integers(2, 40), x:x + rng.integers(2, 40)] = 255 if rng.random() < 0.5 else 0
    reach = pkg.blur_reach_mask()
    assert reach.sum() == 1133 and reach[18].all() and reach[:, 18].all() and not reach[0, 0]
    for img, nf in ((synth.frame(752, 480, 70), 1000), (sat, 800)):
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
        orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
        ok, od = orc.extract(img)
        k, d, patches = ex.debug_blur_patches(img)
        np.testing.assert_array_equal(d, od)
        isf = orc.inv_scale_factors
        nedge = 0
        for i in range(len(k)):
            l = int(k["octave"][i])
            bl = orc.blurred_level(l)
            h, w = bl.shape
            cx = int(round(float(k["x"][i]) * float(isf[l]))) if l else int(k["x"][i])
            cy = int(round(float(k["y"][i]) * float(isf[l]))) if l else int(k["y"][i])
            ys, xs = np.arange(cy - 18, cy + 19), np.arange(cx - 18, cx + 19)
            inside = (ys[:, None] >= 0) & (ys[:, None] < h) & (xs[None, :] >= 0) & (xs[None, :] < w)
            nedge += int(not inside.all())
            want = bl[np.clip(ys, 0, h - 1)[:, None], np.clip(xs, 0, w - 1)[None, :]]
            # pixels of the block outside the level are never read by a descriptor tap, nor are those outside the disc the
            # rotated pattern can reach (the kernel does not compute them and the hook reports 0): compare the rest
            inside &= reach
            assert (patches[i][inside] == want[inside]).all(), "keypoint %d level %d" % (i, l)
            assert not patches[i][~reach].any()
        assert len(k) > 500


def test_get_features_in_area_order_direct(pkg, oracle, synth):
    """SURVEY row a12 checked directly: Frame::GetFeaturesInArea (src/Frame.cc:342-395) is never materialised on the device (a
    predicate + a scan-order key inside the matchers); the hook returns what a query would return, in order, against the
    oracle's 64x48 grid lists - level filters (incl. the bCheckLevels quirk minLevel > 0 || maxLevel >= 0), windows hanging over
    the image, empty windows."""
    w, h = 752, 480
    orc = oracle.Extractor(1500, 1.2, 8, 20, 7)
    k, _ = orc.extract(synth.frame(w, h, 71))
    go, gg = oracle.grid_geom(w, h), pkg.grid_geom(w, h)
    rng = np.random.default_rng(32)
    nonempty = 0
    for q in range(300):
        x, y = rng.uniform(-20, w + 20), rng.uniform(-20, h + 20)
        r = float(rng.choice([3.0, 7.5, 15.0, 40.0, 100.0]))
        mn, mx = [(-1, -1), (0, -1), (0, 0), (2, 4), (1, -1), (0, 7), (3, 2)][q % 7]
        want = oracle.grid_query(k, go, x, y, r, mn, mx)
        got = pkg.debug_features_in_area(k, gg, x, y, r, mn, mx)
        np.testing.assert_array_equal(got, want, err_msg="query %d (%.1f, %.1f, r %.1f, levels %d..%d)" % (q, x, y, r, mn, mx))
        nonempty += len(want) > 1
    assert nonempty > 100


def _saturated_image(rng, w, h):
    img = np.zeros((h, w), np.uint8)
    for _ in range(max(50, w * h // 800)):
        x, y = rng.integers(0, w - 4), rng.integers(0, h - 4)
        img[y:y + rng.integers(2, 40), x:x + rng.integers(2, 40)] = 255 if rng.random() < 0.5 else 0
    return img


@pytest.mark.parametrize("form", [1, 2], ids=["per_keypoint", "level_wide"])
@pytest.mark.parametrize("w,h,nf,kind", [(1241, 376, 2000, "synth"), (640, 480, 1000, "sat"), (333, 257, 300, "synth"),
                                         (1920, 1080, 4000, "synth"), (1000, 259, 700, "noise")])
def test_blur_forms_agree_with_oracle(pkg, oracle, synth, form, w, h, nf, kind):
    """Row a8 in both of its forms (developer knob 13): the 7x7 Gaussian per keypoint inside k_describe (1), or whole levels by
    k_blur_levels with k_describe gathering from them (2).  Level-wide: every blurred level equals the oracle's
    cv::GaussianBlur restatement at EVERY pixel - reflected borders (the levels' frames are not written), tiles that end inside a
    dword, saturated neighbourhoods whose taps (sum 257) reach the clamp; final keypoints and descriptors equal the oracle
    in both forms."""
    rng = np.random.default_rng(w + nf)
    img = synth.frame(w, h, k=41) if kind == "synth" else _saturated_image(rng, w, h) if kind == "sat" else \
        rng.integers(0, 256, (h, w), dtype=np.uint8)
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    ok, od = orc.extract(img)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    pkg.set_default_option(13, form)
    try:
        gk, gd = ex(img)
        mask = ex.blurred_mask()
        assert mask == (0xFF if form == 2 else 0)
        if form == 2:
            for l in range(8):
                np.testing.assert_array_equal(ex.blurred_level(l), orc.blurred_level(l), err_msg="blurred level %d" % l)
        k2, d2, patches = ex.debug_blur_patches(img)     # the 37x37 blocks as k_describe saw them, whatever their source
    finally:
        pkg.set_default_option(13, 0)
    assert len(gk) == len(ok) and len(ok) > 100
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(gk[f], ok[f], err_msg=f)
    np.testing.assert_allclose(gk["angle"], ok["angle"], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(gd, od)
    np.testing.assert_array_equal(d2, od)
    isf = orc.inv_scale_factors
    for i in range(0, len(k2), 7):
        l = int(k2["octave"][i])
        bl = orc.blurred_level(l)
        cx = int(round(float(k2["x"][i]) * float(isf[l]))) if l else int(k2["x"][i])
        cy = int(round(float(k2["y"][i]) * float(isf[l]))) if l else int(k2["y"][i])
        np.testing.assert_array_equal(patches[i], bl[cy - 18:cy + 19, cx - 18:cx + 19] * pkg.blur_reach_mask(), err_msg="keypoint %d level %d" % (i, l))


def test_blur_form_rule_and_batch(pkg, oracle, synth):
    """The per-level rule (level-wide iff nfeatures_l * 37^2 * 100 >= thr * w_l * h_l, knob 14 = thr) and a batch in which the
    two forms coexist: same results as the oracle for every image, masks as the rule says."""
    w, h = 752, 480
    imgs = synth.batch(w, h, 5, k0=60)
    for nf, thr in ((1000, 0), (1000, 60), (2500, 120), (400, 250)):
        orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
        pkg.set_default_option(14, thr)
        try:
            res = ex.extract_batch(imgs)
            mask = ex.blurred_mask()
        finally:
            pkg.set_default_option(14, 0)
        nfl = orc.features_per_level
        orc.extract(imgs[0])
        want = 0
        for l in range(8):
            lh, lw = orc.pyramid_level(l).shape
            want |= (1 << l) if thr and int(nfl[l]) * 1369 * 100 >= thr * lw * lh else 0   # default: no level (measured slower)
        assert mask == want, (nf, thr, bin(mask), bin(want))
        for b in range(5):
            ok, od = orc.extract(imgs[b])
            gk, gd = res[b]
            assert len(gk) == len(ok)
            np.testing.assert_array_equal(gk[["x", "y", "response", "octave"]], ok[["x", "y", "response", "octave"]])
            np.testing.assert_array_equal(gd, od)


@pytest.mark.parametrize("split", [0, 2, 3, 5, 7])
def test_split_call_level_groups(pkg, oracle, synth, split):
    """A batch that fills the GPU runs as a split call (developer knob 15: 0 = never (the default: measured slower), a >= 2 = at level a): the
    quad-tree of the levels [0, a) on a second stream beside the quad-tree + descriptors of the levels [a, 8), whose records
    are described into scratch arrays and moved behind those of [0, a) by the second descriptor launch.  Keypoints, their order
    and descriptors must not depend on it - nor on a caller's capacity that cuts the list inside either group."""
    import torch
    w, h, nf, B = 752, 480, 900, 9
    imgs = synth.batch(w, h, B, k0=80)
    imgs[4] = 0                                   # an image without keypoints inside the batch
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    exp = [orc.extract(imgs[b]) for b in range(B)]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0])
    d_imgs = torch.from_numpy(imgs).cuda()
    st = torch.cuda.current_stream().cuda_stream
    pkg.set_default_option(15, split)
    try:
        for cap in (ex.max_keypoints(), 700, 150):
            kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
            desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
            cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
            for rep in range(2):                  # twice: the second call reuses the scratch arrays and the events
                ex.extract_batch_device(d_imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
            torch.cuda.synchronize()
            k = kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
            dd = desc.cpu().numpy()
            for b in range(B):
                ok, od = exp[b]
                n = int(cnt[b])
                assert n == min(len(ok), cap), (split, cap, b, n, len(ok))
                got = np.frombuffer(k[b, :n].tobytes(), pkg.KP_DTYPE)
                for f in ("x", "y", "size", "response", "octave", "class_id"):
                    np.testing.assert_array_equal(got[f], ok[f][:n], err_msg="%s image %d cap %d" % (f, b, cap))
                np.testing.assert_allclose(got["angle"], ok["angle"][:n], atol=1e-4, rtol=0)
                np.testing.assert_array_equal(dd[b, :n], od[:n])
    finally:
        pkg.set_default_option(15, 0)


@pytest.mark.parametrize("kind", ["natural", "dense", "flat_with_one_blob", "noise"])
@pytest.mark.parametrize("form", [0, 1, 2], ids=["row_skip", "strip_compaction", "cell_compaction"])
@pytest.mark.parametrize("mode", [2, 0, 1], ids=["sparse_forced", "sparse_by_density", "sparse_off"])
def test_fast_row_pretest_is_exact(pkg, oracle, synth, kind, mode, form):
    """The FAST stage's paths for corner-sparse levels.  A five-pixel upper bound of the score (every nine-arc of the ring holds
    r[0] or r[8] and r[4] or r[12]) decides where the 76-operation score is computed at all: ORBX_OPT_SPARSE_FORM 0 (default) = rows
    of 128 pixels skipped inside k_fast_strips, 1 = k_fast_strips_sparse (pairs that pass are queued and scored 64 at a time), 2 = the
    same compaction inside the cell kernel; ORBX_OPT_ROW_PRETEST 2 = every level takes the sparse path, 1 = none, 0 = by the candidate
    density the previous call found.  FAST candidates per level (order included), keypoints and descriptors must equal the oracle's
    on corner-sparse scenes, on the dense benchmark frames and on uniform noise (where nearly every pair passes the bound and the
    queue holds the whole cell) and on an image most of whose pairs fail; the second call on a handle is the one that sees the
    first call's verdicts."""
    w, h, nf = 1241, 376, 1000
    if kind == "natural":
        img = synth.natural(w, h, 7)
    elif kind == "dense":
        img = synth.frame(w, h, 7)
    elif kind == "noise":
        img = np.random.default_rng(5).integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = np.full((h, w), 90, np.uint8)
        img[150:230, 500:640] = synth.frame(140, 80, 3)
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    ok, od = orc.extract(img)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_option(6, 3)       # the strip kernel for a single image
    ex.set_option(16, mode)
    ex.set_option(20, form)
    for rep in range(3):
        gk, gd = ex(img)
        for l in range(8):
            np.testing.assert_array_equal(ex.debug_level_points(l, 0), _cands(orc.level_candidates(l)),
                                          err_msg="FAST candidates level %d call %d" % (l, rep))
        assert len(gk) == len(ok)
        np.testing.assert_array_equal(gk[["x", "y", "response", "octave"]], ok[["x", "y", "response", "octave"]])
        np.testing.assert_array_equal(gd, od)


@pytest.mark.parametrize("form", [0, 1, 2], ids=["row_skip", "strip_compaction", "cell_compaction"])
def test_sparse_and_dense_images_in_one_batch(pkg, oracle, synth, form):
    """A batch whose image slots alternate between corner-sparse and dense scenes, three calls on the same slots and then the
    slots swapped: the strip kernel and the compaction kernel share each call's levels by the previous call's verdicts (per image
    slot and level), every image equals the oracle in every call - also right after the swap, when every verdict is wrong."""
    import torch
    w, h, nf, B = 752, 480, 900, 10
    imgs = np.stack([synth.natural(w, h, 40 + b) if b % 2 else synth.frame(w, h, 40 + b) for b in range(B)])
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    exp = [orc.extract(imgs[b]) for b in range(B)]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_option(6, 3)
    ex.set_option(20, form)
    ex(imgs[0])
    cap = ex.max_keypoints()
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    order = list(range(B))
    for call in range(6):
        if call == 3:
            order = order[1:] + order[:1]          # sparse scenes into the slots flagged dense and the other way round
        d_imgs = torch.from_numpy(imgs[order]).cuda()
        ex.extract_batch_device(d_imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
        torch.cuda.synchronize()
        n = cnt.cpu().numpy()
        K = kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
        D = desc.cpu().numpy()
        for s, b in enumerate(order):
            ok, od = exp[b]
            assert n[s] == len(ok), (call, s, b)
            gk = np.frombuffer(K[s, :n[s]].tobytes(), pkg.KP_DTYPE)
            np.testing.assert_array_equal(gk[["x", "y", "response", "octave"]], ok[["x", "y", "response", "octave"]])
            np.testing.assert_array_equal(D[s, :n[s]], od)


@pytest.mark.parametrize("early", [0, 2, 3, 5])
def test_early_quadtree_of_large_levels(pkg, oracle, synth, early):
    """A batch that fills the GPU launches the strips of the levels [0, a) first and starts their quad-tree on a second stream beside
    the FAST of the other levels (developer knob 19: 0 = off - the default, it measured slower - a >= 2 = on).  Same keypoints, order and descriptors - also for
    natural (corner-sparse) frames whose levels fall back to the exact form inside the early launch, and on repeated calls."""
    import torch
    w, h, nf, B = 752, 480, 900, 10
    imgs = np.stack([synth.frame(w, h, 90 + b) if b % 2 == 0 else synth.natural(w, h, 90 + b) for b in range(B)])
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    exp = [orc.extract(imgs[b]) for b in range(B)]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex(imgs[0])
    cap = ex.max_keypoints()
    d_imgs = torch.from_numpy(imgs).cuda()
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pkg.set_default_option(19, early)
    pkg.set_default_option(6, 3)          # strips for this small batch
    try:
        for rep in range(3):
            ex.extract_batch_device(d_imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, st)
        torch.cuda.synchronize()
        k = kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
        dd = desc.cpu().numpy()
        for b in range(B):
            ok, od = exp[b]
            n = int(cnt[b])
            assert n == len(ok), (early, b, n, len(ok))
            got = np.frombuffer(k[b, :n].tobytes(), pkg.KP_DTYPE)
            for f in ("x", "y", "size", "response", "octave", "class_id"):
                np.testing.assert_array_equal(got[f], ok[f], err_msg="%s image %d" % (f, b))
            np.testing.assert_array_equal(dd[b, :n], od)
        for l in (0, 1, 4):                     # the compacted candidates are materialised on demand (k_gather) and still right
            np.testing.assert_array_equal(ex.debug_level_points(l, 0, b=3), _cands_of(oracle, imgs[3], nf, l))
    finally:
        pkg.set_default_option(19, 0)
        pkg.set_default_option(6, 0)


def _cands_of(oracle, img, nf, l):
    o = oracle.Extractor(nf, 1.2, 8, 20, 7)
    o.extract(img)
    return _cands(o.level_candidates(l))


def test_product_library_end_to_end(pkg, oracle, synth):
    """The PRODUCT library (no hooks: what bench.py, the pipeline and the host classes load) on the sizes of test_staged_parity:
    final keypoints and descriptors against the oracle, single images (the small-batch kernels) and one batch of eight (the
    batched ones), and the hooks really are absent from it."""
    assert not any(hasattr(pkg.lib(), n) for n in pkg.DEV_EXPORTS)
    for w, h, nf in SIZES:
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, developer=False)
        with pytest.raises(pkg.OrbxError):
            ex.debug_level_points(0, 0)
        orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
        imgs = [synth.frame(w, h, 900 + i) for i in range(8)]
        exp = [orc.extract(im) for im in imgs]
        _check_against(ex, orc, exp[0][0], exp[0][1], imgs[0], 8, staged=False)
        for (gk, gd), (ok, od) in zip(ex.extract_batch(np.stack(imgs)), exp):
            assert len(gk) == len(ok)
            for f in ("x", "y", "size", "response", "octave"):
                np.testing.assert_array_equal(gk[f], ok[f], err_msg=f)
            np.testing.assert_array_equal(gd, od)
        cand, kept = ex.level_counts(7)
        assert kept.sum() == len(exp[7][0]) and (cand >= kept).all()


def test_small_batch_forms_agree(pkg, oracle, synth):
    """The kernels a batch of up to four images takes by default (round 5: level 0 from source rows staged in LDS, levels in chains,
    the quad-tree fed by the histogram the FAST stage fills at the L2) against the large-batch forms of the same stages switched in
    one by one (ORBX_OPT_PAD_FORM / _PYR_CHAINS / _OCT_HIST = 1) and against the oracle: batches of 1, 2, 3 and 4 images, sizes whose
    byte count is and is not a multiple of the page size, every padded level and every stage list."""
    for (w, h, nf, seed) in ((1241, 376, 2000, 11), (640, 480, 1000, 12), (333, 257, 300, 13), (1920, 1080, 4000, 14)):
        for B in ((1, 2, 3, 4) if w < 1900 else (1, 2)):
            imgs = np.stack([synth.frame(w, h, seed * 10 + i) for i in range(B)])
            orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
            exp = [orc.extract(im) for im in imgs]
            for opts in ((), ((24, 1),), ((25, 1),), ((25, 3),), ((23, 1),), ((24, 1), (25, 1), (23, 1)), ((24, 2), (5, 4)), ((5, 4), (25, 3))):
                ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
                for k_, v_ in opts:
                    ex.set_option(k_, v_)
                got = ex.extract_batch(imgs)
                for b in range(B):
                    assert got[b][0].tobytes() == exp[b][0].tobytes() and got[b][1].tobytes() == exp[b][1].tobytes(), (w, h, B, b, opts)
                orc.extract(imgs[B - 1])
                for l in range(8):
                    np.testing.assert_array_equal(ex.pyramid_level(l, b=B - 1, padded=True), orc.pyramid_level(l, padded=True), err_msg=str((w, h, B, l, opts)))
                    np.testing.assert_array_equal(ex.debug_level_points(l, 1, b=B - 1), _cands(orc.level_keypoints(l)), err_msg=str((w, h, B, l, opts)))
                ex.close()


def test_stereo_frame_view_equals_stereo_frame(pkg, oracle, synth):
    """orbx_stereo_frame_view (the latency path: images read where they lie, record written to pinned host memory by the last kernel,
    two alternating records) == orbx_stereo_frame == the oracle, for pageable, pinned-host and device-resident images; the view of
    the previous call stays intact while the next one is produced."""
    import torch
    w, h, nf = 752, 480, 1000
    mbf, mb = 386.1448, float(np.float32(386.1448) / np.float32(718.856))
    frames, _ = synth.stereo_sequence(w, h, 3, k=21, step=0.04)
    ex, ex2 = pkg.ORBextractor(nf, 1.2, 8, 20, 7, developer=False), pkg.ORBextractor(nf, 1.2, 8, 20, 7, developer=False)
    prev = None
    for t, (l, r) in enumerate(frames):
        ref = ex2.stereo_frame(l, r, mbf, mb)
        if t == 0:
            args = (l, r)                                                                    # pageable numpy arrays
        elif t == 1:
            args = (torch.from_numpy(l).pin_memory(), torch.from_numpy(r).pin_memory())      # pinned host memory
        else:
            args = (torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())                  # device memory
        f = ex.stereo_frame_view(*args, mbf, mb, shape=(h, w))
        for key in ("kl", "dl", "kr", "dr", "uright", "depth"):
            assert f[key].tobytes() == ref[key].tobytes(), (t, key)
        assert f["nmatch"] == ref["nmatch"] > 50
        v = f["view"]
        nl = len(f["kl"])
        dk = torch.zeros(nl * 7, dtype=torch.float32, device="cuda")     # the same record in HBM
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(dk.data_ptr()), C.c_void_p(v.d_kl), C.c_size_t(nl * 28), C.c_int(3)) == 0
        assert dk.cpu().numpy().tobytes() == f["kl"].tobytes()
        if prev is not None:
            for key in ("kl", "dl", "uright", "depth"):
                assert prev[0][key].tobytes() == prev[1][key], "the previous frame's view changed under the next call"
        prev = (f, {key: f[key].tobytes() for key in ("kl", "dl", "uright", "depth")})
    ok, od = oracle.Extractor(nf, 1.2, 8, 20, 7).extract(frames[2][0])
    assert f["kl"].tobytes() == ok.tobytes() and f["dl"].tobytes() == od.tobytes()


def test_quadtree_shared_sweep_in_a_batch(pkg, oracle, synth):
    """ORBX_OPT_OCT_SLICES = 1 (a measured alternative, off by default): in a batch the key sweep of a level with >= 600 FAST cells is shared by two workgroups, >= 1600 by four
    (1920x1080: levels 0-3), each leaving a partial histogram + best keys in global memory for the last one to arrive.  Six images
    of 1920x1080 / 4000 features (more than the four of the small-batch forms) with and without the sharing, 512- and 1024-thread
    builds, three calls each (the arrival counters must be left at zero): keypoints and descriptors of every image == oracle."""
    w, h, nf, B = 1920, 1080, 4000, 6
    imgs = np.stack([synth.frame(w, h, 4100 + i) for i in range(B)])
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    exp = [orc.extract(im) for im in imgs]
    for opts in (((26, 1),), (), ((11, 1), (26, 1)), ((11, 1),)):
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
        for k_, v_ in opts:
            ex.set_option(k_, v_)
        for rep in range(3):
            got = ex.extract_batch(imgs)
            for b in range(B):
                assert got[b][0].tobytes() == exp[b][0].tobytes() and got[b][1].tobytes() == exp[b][1].tobytes(), (opts, rep, b)
        orc.extract(imgs[B - 1])
        for l in range(8):
            np.testing.assert_array_equal(ex.debug_level_points(l, 1, b=B - 1), _cands(orc.level_keypoints(l)), err_msg=str((opts, l)))
        ex.close()


def test_stereo_frame_view_sizes_strides_and_empty(pkg, oracle, synth):
    """orbx_stereo_frame_view over the cases a caller can hand it: image sizes that change from call to call on one handle (the plan,
    the records and the staging buffer are rebuilt), a row stride larger than the width (a cv::Mat ROI; small enough for the LDS-staged
    level 0, and so large that the call falls back to k_pyr_pad), a byte count that is a multiple of the page size, an empty image."""
    import torch
    mbf, mb = 40.0, 0.1
    ex = pkg.ORBextractor(800, 1.2, 8, 20, 7, developer=False)
    ref = pkg.ORBextractor(800, 1.2, 8, 20, 7, developer=False)
    for (w, h, stride, pinned) in ((640, 480, 640, False), (1024, 512, 1024, True), (500, 300, 512, False), (640, 480, 640, True), (333, 257, 20000, False),
                                   (752, 480, 752, True)):
        l, r = synth.stereo_pair_blocky(w, h, 9000 + w + stride)
        want = ref.stereo_frame(l, r, mbf, mb)
        if stride != w:      # rows padded to `stride` bytes, the padding filled with noise the kernels must not read into the image
            rng = np.random.default_rng(stride)
            bl, br = rng.integers(0, 256, (h, stride), dtype=np.uint8), rng.integers(0, 256, (h, stride), dtype=np.uint8)
            bl[:, :w], br[:, :w] = l, r
            l2, r2 = bl, br
        else:
            l2, r2 = l, r
        if pinned:
            tl, tr = torch.from_numpy(np.ascontiguousarray(l2)).pin_memory(), torch.from_numpy(np.ascontiguousarray(r2)).pin_memory()
            f = ex.stereo_frame_view(tl, tr, mbf, mb, shape=(h, w), stride=stride)
        else:
            l2, r2 = np.ascontiguousarray(l2), np.ascontiguousarray(r2)
            f = ex.stereo_frame_view(l2.ctypes.data, r2.ctypes.data, mbf, mb, shape=(h, w), stride=stride)
        for key in ("kl", "dl", "kr", "dr", "uright", "depth"):
            assert f[key].tobytes() == want[key].tobytes(), (w, h, stride, key)
        assert f["nmatch"] == want["nmatch"]
    v = pkg.StereoView()
    assert pkg.lib().orbx_stereo_frame_view(ex._h, None, None, 0, 0, 0, mbf, mb, __import__("ctypes").byref(v)) == 0 and v.nl == 0 and v.nr == 0 and not v.kl
