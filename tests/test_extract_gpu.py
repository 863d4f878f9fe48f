"""GPU parity of the HIP extractor against the CPU oracle, stage by stage, through the C ABI.

Bar: keypoint sets, order, scores, octaves and descriptors bit-exact; angles within 1e-4
(they are in fact bit-identical: same float operations, no FMA contraction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(640, 480, 1000), (1241, 376, 1000), (1241, 376, 2000), (752, 480, 1000), (320, 240, 500)]


def _cands(a):
    return np.stack([a["x"], a["y"], a["score"]], 1).astype(np.int32) if len(a) else np.zeros((0, 3), np.int32)


@pytest.mark.parametrize("w,h,nf", SIZES)
def test_staged_parity(pkg, oracle, synth, w, h, nf):
    img = synth.frame(w, h, k=3)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ok, od = None, None
    orc = oracle.Extractor(nf, 1.2, 8, 20, 7)
    ok, od = orc.extract(img)
    gk, gd = ex(img)
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
