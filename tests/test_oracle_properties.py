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
