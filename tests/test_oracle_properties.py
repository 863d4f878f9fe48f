"""CPU: structural properties of the oracle's extraction on synthetic frames."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def extraction(oracle, synth):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    img = synth.frame(640, 480, 4)
    k, d = ex.extract(img)
    return ex, img, k, d


def test_counts_and_order(extraction):
    ex, img, k, d = extraction
    nf = ex.features_per_level
    assert (np.diff(k["octave"]) >= 0).all(), "level-major concatenation (:1076-1104)"
    for l in range(8):
        n = int((k["octave"] == l).sum())
        assert n <= nf[l] + 2, "a level returns at most N+2 nodes"
        assert n == len(ex.level_keypoints(l))
    assert len(k) >= 900 and d.shape == (len(k), 32)
    assert (k["class_id"] == -1).all()


def test_keypoints_inside_and_scaled(extraction):
    ex, img, k, d = extraction
    sf = ex.scale_factors
    for l in range(8):
        lw, lh = ex.pyramid_level(l).shape[::-1]
        kp = ex.level_keypoints(l)
        x, y = kp["x"] + 16, kp["y"] + 16
        assert (x >= 19).all() and (x < lw - 19).all() and (y >= 19).all() and (y < lh - 19).all()
        sel = k[k["octave"] == l]
        np.testing.assert_array_equal(sel["x"], x.astype(np.float32) * (sf[l] if l else np.float32(1)))
        np.testing.assert_array_equal(sel["size"], np.float32(int(np.float32(31) * sf[l])))
        assert ((sel["angle"] >= 0) & (sel["angle"] < 360.0001)).all()


def test_candidates_are_strict_local_maxima(extraction, oracle):
    ex, img, k, d = extraction
    lvl = ex.pyramid_level(2)
    c = ex.level_candidates(2)
    assert len(c) > 100
    # selected keypoints are a subset of the candidates, each candidate location unique
    cs = {(a, b): s for a, b, s in c.tolist()}
    assert len(cs) == len(c)
    for a, b, s in ex.level_keypoints(2).tolist():
        assert cs[(a, b)] == s
    # every candidate is a FAST corner at minTh with exactly that score
    L = oracle.lib()
    lv = np.ascontiguousarray(lvl)
    for a, b, s in c.tolist()[:300]:
        p = lv.ctypes.data + (b + 16) * lv.shape[1] + (a + 16)
        assert L.oracle_fast_is_corner(p, lv.shape[1], 7) == 1
        assert L.oracle_fast_score(p, lv.shape[1], 7) == s


def test_hamming_properties(oracle):
    rng = np.random.default_rng(1)
    a, b, c = (rng.integers(0, 256, 32, dtype=np.uint8) for _ in range(3))
    assert oracle.hamming(a, a) == 0
    assert oracle.hamming(a, b) == oracle.hamming(b, a) <= 256
    assert oracle.hamming(a, c) <= oracle.hamming(a, b) + oracle.hamming(b, c)


def test_three_maxima(oracle):
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    h = [0] * 30
    h[3], h[7], h[9] = 50, 30, 10
    assert oracle.three_maxima(h) == (3, 7, 9)
    h[9] = 4  # < 0.1*50
    assert oracle.three_maxima(h) == (3, 7, -1)
    h[7] = 4
    assert oracle.three_maxima(h) == (3, -1, -1)


def test_grid_query_order_and_filter(oracle, extraction):
    ex, img, k, d = extraction
    g = oracle.grid_geom(640, 480)
    idx = oracle.grid_query(k, g, 320.0, 240.0, 60.0, -1, -1)
    assert len(idx) > 5
    assert (np.abs(k["x"][idx] - 320) < 60).all() and (np.abs(k["y"][idx] - 240) < 60).all()
    # column-major over 64x48 cells, index order inside a cell
    vx = (k["x"][idx] * np.float32(g.inv_w)).astype(np.float32)  # round(): half away from zero (:399-400)
    vy = (k["y"][idx] * np.float32(g.inv_h)).astype(np.float32)
    cx = np.floor(vx.astype(np.float64) + 0.5).astype(int)
    cy = np.floor(vy.astype(np.float64) + 0.5).astype(int)
    key = list(zip(cx.tolist(), cy.tolist(), idx.tolist()))
    assert key == sorted(key)
    lv = oracle.grid_query(k, g, 320.0, 240.0, 60.0, 2, 3)
    assert set(k["octave"][lv].tolist()) <= {2, 3}
    # bCheckLevels quirk (:363): minLevel=0,maxLevel=-1 -> no level filtering at all
    assert len(oracle.grid_query(k, g, 320.0, 240.0, 60.0, 0, -1)) == len(idx)


def test_stereo_oracle_self_consistency(oracle, synth):
    left, right = synth.stereo_pair_blocky(640, 480, 3)
    el, er = oracle.Extractor(800, 1.2, 8, 20, 7), oracle.Extractor(800, 1.2, 8, 20, 7)
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    mbf = 47.9
    mb = np.float32(mbf) / np.float32(435.2)
    n, ur, dp = oracle.stereo_match(kl, dl, kr, dr, [el.pyramid_level(i) for i in range(8)],
                                    [er.pyramid_level(i) for i in range(8)], el.scale_factors, el.inv_scale_factors,
                                    mbf, mb)
    m = ur >= 0
    assert n == m.sum() > 20
    disp = kl["x"][m] - ur[m]
    assert (disp > 0).all() and (disp < 435.2).all()
    np.testing.assert_allclose(dp[m], np.float32(mbf) / disp, rtol=1e-6)
    assert ((dp >= 0) == m).all()


def test_fuse_bookkeeping_hand_built(oracle):
    """Hand-built Fuse (src/ORBmatcher.cc:827-977) on three keypoints: add to an empty slot, lose
    against a better-observed holder, replace a weaker holder, skip a null entry, and replace a
    point that an EARLIER iteration of the same call added (the sequential dependence)."""
    fx = fy = 100.0
    cx, cy, w, h = 320.0, 240.0, 640, 480
    k = np.zeros(3, oracle.KP_DTYPE)
    k["x"], k["y"], k["octave"] = [100, 300, 500], [100, 200, 300], [0, 0, 1]
    d = np.zeros((3, 32), np.uint8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    inv_s2 = (1 / (sf * sf)).astype(np.float32)
    cam = oracle.Cam(fx, fy, cx, cy, 40.0, 0.4)
    g = oracle.grid_geom(w, h)

    def point(kp, z, grow):
        p = np.array([(kp["x"] - cx) / fx * z, (kp["y"] - cy) / fy * z, z], np.float64)
        dist = np.linalg.norm(p)
        return (1, p[0], p[1], p[2], *(p / dist), dist * grow, dist * 0.1)
    pts = np.zeros(5, oracle.MP3D_DTYPE)
    pts[0] = point(k[0], 10.0, 1.0)      # A -> keypoint 0 (empty slot)
    pts[1] = point(k[1], 8.0, 1.0)       # B -> keypoint 1, holder has more observations
    pts[2] = point(k[2], 12.0, 1.15)     # C -> keypoint 2 (octave 1), holder has fewer observations
    pts[3] = point(k[0], 10.0, 1.0); pts[3]["valid"] = 0   # null entry
    pts[4] = point(k[0], 9.0, 1.0)       # E -> keypoint 0 again, now held by A
    pd = np.zeros((5, 32), np.uint8)
    bad = np.zeros(5, np.int32); in_kf = np.zeros(5, np.int32)
    obs = np.array([1, 2, 3, 9, 7], np.int32)
    slot = np.array([-1, -2, -2], np.int32)
    ext_obs = np.array([0, 5, 1], np.int32); ext_bad = np.zeros(3, np.int32)
    uright = np.array([-1, -1, 500 - 40.0 / 12.0 + 0.5], np.float32)   # ur of C is u - bf/z
    T = np.eye(4, dtype=np.float32)
    n, bi, act, st = oracle.fuse(k, d, uright, g, sf, inv_s2, np.float32(np.log(np.float32(1.2))), cam, T, pts, pd, bad,
                                 in_kf, obs, slot, ext_obs, ext_bad, 3.0)
    assert n == 4
    assert list(act) == [1, 2, 3, 0, 3]
    assert list(bi) == [0, 1, 2, -1, 0]
    assert list(st["slot"]) == [4, -2, 2]
    assert list(st["bad"]) == [1, 1, 0, 0, 0]          # A replaced by E, B replaced by its holder
    assert list(st["in_kf"]) == [0, 0, 1, 0, 1]
    assert list(st["obs"]) == [2, 2, 5, 9, 8]          # C gains 2 (stereo keypoint), A and E gain 1
    assert list(st["ext_bad"]) == [0, 0, 1]
    # the stereo gate: move C's right coordinate far from the keypoint's -> candidate rejected
    uright[2] = 300.0
    n2, bi2, act2, _ = oracle.fuse(k, d, uright, g, sf, inv_s2, np.float32(np.log(np.float32(1.2))), cam, T, pts, pd, bad,
                                   in_kf, obs, slot, ext_obs, ext_bad, 3.0)
    assert n2 == 3 and act2[2] == 0 and bi2[2] == -1


def test_search_by_sim3_needs_mutual_agreement(oracle):
    """SearchBySim3 (src/ORBmatcher.cc:1104-1328): identical keyframes, identity Sim3: every good
    slot matches itself; dropping the point of slot j in KF2 removes exactly the pair (j, j)."""
    fx = fy = 100.0
    cx, cy, w, h = 320.0, 240.0, 640, 480
    rng = np.random.default_rng(3)
    n = 40
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"] = rng.uniform(40, 600, n).astype(np.float32); k["y"] = rng.uniform(40, 440, n).astype(np.float32)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    cam = oracle.Cam(fx, fy, cx, cy, 40.0, 0.4)
    g = oracle.grid_geom(w, h)
    z = 10.0
    p = np.zeros(n, oracle.MP3D_DTYPE)
    p["valid"] = 1
    p["wx"], p["wy"], p["wz"] = (k["x"] - cx) / fx * z, (k["y"] - cy) / fy * z, z
    dist = np.sqrt(p["wx"] ** 2 + p["wy"] ** 2 + p["wz"] ** 2)
    p["max_distance"], p["min_distance"] = dist, dist * 0.1
    I4 = np.eye(4, dtype=np.float32); I3 = np.eye(3, dtype=np.float32); z3 = np.zeros(3, np.float32)
    lsf = np.float32(np.log(np.float32(1.2)))
    nf, m12 = oracle.search_by_sim3(k, d, k, d, g, sf, lsf, cam, I4, I4, 1.0, I3, z3, p, d, p, d, 1.0)
    assert nf == n and list(m12) == list(range(n))
    p2 = p.copy(); p2["valid"][7] = 0
    nf, m12 = oracle.search_by_sim3(k, d, k, d, g, sf, lsf, cam, I4, I4, 1.0, I3, z3, p, d, p2, d, 1.0)
    assert nf == n - 1 and m12[7] == -1 and (np.delete(m12, 7) == np.delete(np.arange(n), 7)).all()


def test_distinctive_descriptor_hand_built(oracle):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317): median index (size_t)(0.5*(N-1)),
    self distance included, first minimum wins."""
    def rows(*bits):   # descriptor with the given number of leading one-bits
        out = np.zeros((len(bits), 32), np.uint8)
        for i, b in enumerate(bits):
            out[i] = np.packbits(np.arange(256) < b)
        return out
    # distances |a-b|: rows at 0, 10, 20, 100 -> medians (index 1 of the sorted row): 10, 10, 10, 80
    assert oracle.distinctive_descriptor(rows(0, 10, 20, 100)) == (0, 10)
    # N = 5, index 2: rows 0,4,10,30,100 -> medians 10, 6, 10, 26, 90 -> row 1
    assert oracle.distinctive_descriptor(rows(0, 4, 10, 30, 100)) == (1, 6)
    assert oracle.distinctive_descriptor(rows(7)) == (0, 0)
    assert oracle.distinctive_descriptor(rows(7, 200)) == (0, 0)      # N = 2: index 0 -> the self distance
    assert oracle.distinctive_descriptor(rows()) == (-1, 0)


def test_vocabulary_transform_hand_built(oracle):
    """TemplatedVocabulary::transform on a 2-ary, 2-level tree with known answers: first minimum in child order,
    word ids in node-id order, TF-IDF sums in feature order, L1 normalisation, FeatureVector at level L-levelsup."""
    def bits(n):
        return np.packbits(np.arange(256) < n)
    # nodes: 0 root | 1: 0 bits, 2: 200 bits | children of 1: 3 (0 bits, w 2.0), 4 (20 bits, w 0 = stopped) | of 2: 5 (180, w 1.5), 6 (220, w 4.0)
    parent = [0, 0, 0, 1, 1, 2, 2]
    is_leaf = [0, 0, 0, 1, 1, 1, 1]
    desc = np.stack([bits(0), bits(0), bits(200), bits(0), bits(20), bits(180), bits(220)])
    weight = [0, 0, 0, 2.0, 0.0, 1.5, 4.0]
    v = oracle.Vocabulary(2, 2, 0, 0, parent, is_leaf, desc, weight)
    assert v.transform_one(bits(3), 1) == (0, 2.0, 1)          # word 0 = node 3, node one level up = 1
    assert v.transform_one(bits(10), 1) == (0, 2.0, 1)         # tie 10 vs 10 between nodes 3 and 4: the first child wins
    assert v.transform_one(bits(11), 1) == (1, 0.0, 1)         # node 4: a stopped word
    assert v.transform_one(bits(100), 0) == (1, 0.0, 4)        # tie at the root (100 vs 100): first child; then node 4 (80 < 100); levelsup 0 -> the leaf itself
    assert v.transform_one(bits(201), 2) == (3, 4.0, 0)        # |201-180| = 21 > |201-220| = 19 -> node 6 = word 3; level 0 -> root
    feats = np.stack([bits(3), bits(201), bits(11), bits(1), bits(230)])
    bw, bv, fv = v.transform(feats, 1)
    assert list(bw) == [0, 3]
    np.testing.assert_array_equal(bv, np.array([4.0, 8.0]) / 12.0)   # 2+2 and 4+4, L1-normalised
    assert fv == {1: [0, 3], 2: [1, 4]}                       # the stopped feature 2 is in neither vector


def test_descriptor_taps_stay_inside_the_reach_disc(oracle, pkg):
    """The fused blur of the HIP descriptor kernel computes only the pixels of a keypoint's 37x37 block that a tap can reach
    (pkg.blur_reach_mask: (|row| - 1/2)^2 + (|col| - 1/2)^2 <= 340).  Every tap of every pattern point at 36 000 angles, computed
    the way src/ORBextractor.cc:113-120 does (float products, cvRound), lies inside it - and the disc is tight: the taps reach its
    rim."""
    import re
    txt = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "include", "orb_pattern_31.inc")).read()
    txt = re.sub(r"//.*", "", txt)
    pat = np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.float32).reshape(-1, 2)
    assert pat.shape == (512, 2) and int((pat ** 2).sum(1).max()) == 338
    reach = pkg.blur_reach_mask()
    hit = np.zeros((37, 37), bool)
    ang = (np.arange(36000, dtype=np.float32) * np.float32(0.01)) * np.float32(np.pi / 180.0)
    a, b = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    for x, y in pat:
        row = np.rint(x * b + y * a).astype(int)      # float32 products and sum, round half to even (cvRound)
        col = np.rint(x * a - y * b).astype(int)
        assert np.abs(row).max() <= 18 and np.abs(col).max() <= 18
        hit[row + 18, col + 18] = True
    assert not (hit & ~reach).any()
    assert hit[reach].mean() > 0.97     # nearly every pixel of the disc is read at some angle
