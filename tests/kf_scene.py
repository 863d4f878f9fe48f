"""Synthetic loop-closing / local-mapping scenes for the KeyFrame-side matchers
(SearchByProjection(KeyFrame*, Scw), Fuse x2, SearchBySim3): map points that project near the
keyframe's keypoints through a pose or a Sim3, with the skip conditions of the reference
(behind the camera, outside the image, outside the scale-invariance range, viewing angle > 60 deg)
all represented.  Shared by tests/test_match_gpu.py and tests/test_host_cpp_gpu.py."""
import numpy as np

FX, FY, CX, CY, MBF = 718.856, 718.856, 607.19, 185.2, 386.1448


def geoms(mod, w, h, distorted):
    """(query geometry of the KeyFrame = int-truncated bounds, assignment geometry of its Frame)."""
    if distorted:
        b = np.array([-2.7, -1.4, w + 3.6, h + 2.2], np.float32)
    else:
        b = np.array([0, 0, w, h], np.float32)
    ga = mod.GridGeom()
    ga.min_x, ga.min_y, ga.max_x, ga.max_y = [float(v) for v in b]
    ga.inv_w = np.float32(64) / (b[2] - b[0])
    ga.inv_h = np.float32(48) / (b[3] - b[1])
    g = mod.GridGeom()
    g.min_x, g.min_y, g.max_x, g.max_y = [float(int(v)) for v in b]   # float -> int truncates (src/KeyFrame.cc:41)
    g.inv_w, g.inv_h = ga.inv_w, ga.inv_h
    return g, ga, b


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(rng, scale=1.0):
    """4x4 float32 [s*R | s*t] with a small rotation / translation."""
    R = rot(*rng.normal(0, 0.01, 3))
    t = rng.normal(0, 0.05, 3)
    T = np.eye(4)
    T[:3, :3] = scale * R
    T[:3, 3] = scale * t
    return T.astype(np.float32)


def points_for(oracle, rng, k, d, sf, T, m, scale=1.0, bits=3):
    """m map points (MP3D_DTYPE) + descriptors projecting near random keypoints of k through the
    (Sim3) transform T (camera = T[:3,:3]/scale * p + T[:3,3]/scale)."""
    n = len(k)
    idx = rng.choice(n, m, replace=True)
    z = rng.uniform(4, 40, m)
    pc = np.stack([(k["x"][idx] + rng.normal(0, 2, m) - CX) / FX * z, (k["y"][idx] + rng.normal(0, 2, m) - CY) / FY * z, z], 1)
    R = T[:3, :3].astype(np.float64) / scale
    t = T[:3, 3].astype(np.float64) / scale
    pw = (pc - t) @ R                       # R^T (pc - t)
    Ow = -R.T @ t
    pts = np.zeros(m, oracle.MP3D_DTYPE)
    pts["valid"] = 1
    pts["wx"], pts["wy"], pts["wz"] = pw[:, 0], pw[:, 1], pw[:, 2]
    po = pw - Ow
    dist = np.linalg.norm(po, axis=1)
    nrm = po / dist[:, None] + rng.normal(0, 0.25, (m, 3))
    flip = rng.random(m) < 0.07            # viewing angle test fails
    nrm[flip] *= -1
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    pts["nx"], pts["ny"], pts["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    lvl = k["octave"][idx]
    pts["max_distance"] = dist * sf[lvl] * rng.uniform(0.85, 1.15, m)
    pts["min_distance"] = pts["max_distance"] / sf[7] * rng.uniform(0.5, 1.0, m)
    far = rng.random(m) < 0.05             # outside the scale-invariance range
    pts["max_distance"][far] *= 0.3
    behind = rng.random(m) < 0.03          # behind the camera
    pwb = (pc * np.array([1, 1, -1]) - t) @ R
    for f, c in (("wx", 0), ("wy", 1), ("wz", 2)):
        pts[f][behind] = pwb[behind, c]
    mask = rng.integers(0, 256, (bits, m, 32), dtype=np.uint8)
    flipbits = mask[0]
    for b in range(1, bits):
        flipbits &= mask[b]
    pd = d[idx] ^ flipbits
    return pts, pd, idx


def fuse_state(rng, n, m):
    """bad[m], obs[m], slot[n] (-1 / -2 / list index), ext_obs[n], ext_bad[n]"""
    bad = (rng.random(m) < 0.08).astype(np.int32)
    obs = rng.integers(1, 9, m).astype(np.int32)
    slot = np.full(n, -1, np.int32)
    perm = rng.permutation(n)
    n_list = min(m // 6, n // 4)
    slot[perm[:n_list]] = rng.choice(m, n_list, replace=False)
    n_ext = n // 3
    slot[perm[n_list:n_list + n_ext]] = -2
    ext_obs = rng.integers(1, 9, n).astype(np.int32)
    ext_bad = (rng.random(n) < 0.15).astype(np.int32)
    in_kf = np.zeros(m, np.int32)
    in_kf[slot[slot >= 0]] = 1
    return bad, in_kf, obs, slot, ext_obs, ext_bad
