"""What the HIP path produces for a reference-vector case, under the key set of oracle/refvec.py (test helper; the only place
where product and checker meet)."""
import importlib

import numpy as np

from oracle import refvec

PKG = "orb_slam2v2-1_amd"


def hip_vectors(case, gauss=None):
    """gauss: flavour of the handles ("half_up" / "sse2"; None = the wrapper's default)."""
    pkg = importlib.import_module(PKG)
    name, w, h, nf, stereo, seed, kind, full = case
    left, right = refvec.case_images(case)
    p = name + "/"
    out = {}
    ex = pkg.ORBextractor(nf, refvec.SCALE, refvec.NLEVELS, refvec.INI_TH, refvec.MIN_TH, gauss=gauss, developer=True)   # (stage hooks)
    # the blurred level is never stored by the default path (the Gaussian is fused into the descriptor kernel): one extra extraction
    # with every level blurred as a whole (k_blur_levels, ORBX_OPT_BLUR_FORM = 2) materialises it for the checksum
    ex.set_option(13, 2)
    ex(left)
    blurred = [ex.blurred_level(l) for l in range(refvec.NLEVELS)]
    ex.set_option(13, 0)
    k, d = ex(left)
    out[p + "scale_factors"] = np.asarray(ex.GetScaleFactors(), "<f4")
    out[p + "features_per_level"] = np.asarray(ex.mnFeaturesPerLevel, "<i4")
    out[p + "umax"] = np.asarray(ex.umax, "<i4")
    for l in range(refvec.NLEVELS):
        q = p + "L%d/" % l
        lvl, pad = ex.pyramid_level(l), ex.pyramid_level(l, padded=True)
        if full:
            out[q + "pyramid"], out[q + "padded"], out[q + "blur"] = lvl.copy(), pad.copy(), blurred[l]
        out[q + "crc"] = np.array([refvec.crc(lvl), refvec.crc(pad), refvec.crc(blurred[l])], "<f8")
        pts = np.asarray(ex.debug_level_points(l, 1), "<i4").reshape(-1, 3).copy()
        pts[:, :2] += 16
        out[q + "keypoints"] = pts
        out[q + "angles"] = np.ascontiguousarray(k["angle"][k["octave"] == l], "<f4")
    out[p + "keypoints"] = np.frombuffer(k.tobytes(), np.uint8).reshape(len(k), 28)
    out[p + "descriptors"] = d
    if stereo:
        exr = pkg.ORBextractor(nf, refvec.SCALE, refvec.NLEVELS, refvec.INI_TH, refvec.MIN_TH, gauss=gauss)
        kr, dr = exr(right)
        out[p + "keypoints_right"] = np.frombuffer(kr.tobytes(), np.uint8).reshape(len(kr), 28)
        out[p + "descriptors_right"] = dr
        mb = float(np.float32(refvec.KITTI_BF) / np.float32(refvec.KITTI_FX))
        ur, dp, n = pkg.compute_stereo_matches(ex, exr, k, d, kr, dr, refvec.KITTI_BF, mb)
        out[p + "mvuRight"], out[p + "mvDepth"] = np.asarray(ur, "<f4"), np.asarray(dp, "<f4")
    return out


def compare_hip(ref, got):
    return refvec.compare(ref, got)
