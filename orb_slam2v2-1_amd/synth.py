"""Seeded synthetic grayscale frames: the inputs of tests and bench.

Low-frequency background (6 random sinusoids, amplitude 40 around 128) + random
axis-aligned rectangles and rotated checker patches with contrast U[25,120] (corners
above iniThFAST=20 and some only above minThFAST=7) + Gaussian noise sigma=2.
Frame k of a sequence uses seed = 1000 + k.  The stereo right image is the left scene
shifted by a per-row-constant disparity U[2,60] px with independent noise.
"""
import numpy as np


def _scene(rng, w, h, pad):
    W = w + pad
    yy, xx = np.mgrid[0:h, 0:W].astype(np.float32)
    img = np.full((h, W), 128.0, np.float32)
    for _ in range(6):
        fx, fy = rng.uniform(-0.02, 0.02, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += (40.0 / 6.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    nshapes = max(8, int(4000 * (w * h) / 466616.0))
    for _ in range(nshapes):
        c = rng.uniform(25, 120) * (1 if rng.random() < 0.5 else -1)
        if rng.random() < 0.15:
            c = rng.uniform(8, 19) * (1 if rng.random() < 0.5 else -1)  # only above minThFAST
        cx, cy = rng.integers(0, W), rng.integers(0, h)
        if rng.random() < 0.6:
            rw, rh = rng.integers(4, 40), rng.integers(4, 40)
            x0, x1 = max(cx - rw // 2, 0), min(cx + rw // 2 + 1, W)
            y0, y1 = max(cy - rh // 2, 0), min(cy + rh // 2 + 1, h)
            img[y0:y1, x0:x1] += c
        else:
            s = int(rng.integers(10, 32))
            th = rng.uniform(0, np.pi)
            cell = rng.uniform(3.0, 8.0)
            x0, x1 = max(cx - s, 0), min(cx + s + 1, W)
            y0, y1 = max(cy - s, 0), min(cy + s + 1, h)
            if x1 <= x0 or y1 <= y0:
                continue
            py, px = np.mgrid[y0:y1, x0:x1].astype(np.float32)
            u = (px - cx) * np.cos(th) + (py - cy) * np.sin(th)
            v = -(px - cx) * np.sin(th) + (py - cy) * np.cos(th)
            inside = (np.abs(u) <= s * 0.7) & (np.abs(v) <= s * 0.7)
            chk = ((np.floor(u / cell) + np.floor(v / cell)) % 2 == 0)
            img[y0:y1, x0:x1] += np.where(inside & chk, c, 0.0).astype(np.float32)
    return img


def _finish(rng, scene):
    out = scene + rng.normal(0.0, 2.0, scene.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def frame(w, h, k=0):
    """Mono frame k (seed 1000+k), uint8 [h, w]."""
    rng = np.random.default_rng(1000 + k)
    return _finish(rng, _scene(rng, w, h, 0))


def stereo_pair(w, h, k=0):
    """(left, right) rectified pair; disparity is constant per row, U[2,60] px."""
    rng = np.random.default_rng(1000 + k)
    pad = 64
    sc = _scene(rng, w, h, pad)
    disp = rng.integers(2, 61, size=h)
    left = sc[:, :w]
    right = np.empty((h, w), np.float32)
    for y in range(h):
        d = int(disp[y])
        # a point at left column u appears at right column u-d  =>  right[x] = scene[x+d]
        right[y] = sc[y, d:d + w]
    # keep vertical structure coherent: smooth the disparity in blocks of 16 rows
    return _finish(rng, left), _finish(rng, right)


def stereo_pair_blocky(w, h, k=0, block=24):
    """Stereo pair whose disparity is constant over blocks of `block` rows, so that 11x11
    SAD windows (src/Frame.cc:577-607) see a coherent shift and matches survive."""
    rng = np.random.default_rng(1000 + k)
    pad = 64
    sc = _scene(rng, w, h, pad)
    nb = (h + block - 1) // block
    dblk = rng.integers(2, 61, size=nb)
    right = np.empty((h, w), np.float32)
    for y in range(h):
        d = int(dblk[y // block])
        right[y] = sc[y, d:d + w]
    return _finish(rng, sc[:, :w]), _finish(rng, right)


def batch(w, h, n, k0=0):
    """n mono frames [n, h, w]."""
    return np.stack([frame(w, h, k0 + i) for i in range(n)])


def stereo_sequence(w, h, nframes, k=0, step=0.04, block=24):
    """Rectified stereo SEQUENCE of a camera that translates along +X through a scene of fronto-parallel bands (depth
    constant over blocks of `block` rows, as in stereo_pair_blocky).  A band of disparity d px moves step*d px per
    frame, so with baseline b the camera advances step*b per frame: frame t shows the band shifted by round(t*step*d).
    Returns (frames, disparity_per_row) with frames = [(left_t, right_t)]; independent sensor noise per image."""
    rng = np.random.default_rng(1000 + k)
    nb = (h + block - 1) // block
    dblk = rng.integers(2, 61, size=nb)
    pad = 64 + int(np.ceil(nframes * step * 60)) + 8
    sc = _scene(rng, w, h, pad)
    drow = np.array([int(dblk[y // block]) for y in range(h)])
    frames = []
    for t in range(nframes):
        left = np.empty((h, w), np.float32)
        right = np.empty((h, w), np.float32)
        for y in range(h):
            o = int(round(t * step * drow[y]))
            left[y] = sc[y, o:o + w]
            right[y] = sc[y, o + drow[y]:o + drow[y] + w]
        frames.append((_finish(rng, left), _finish(rng, right)))
    return frames, drow


def _box3(a):
    """3x3 box filter with edge replication (float32)."""
    p = np.pad(a, 1, mode="edge")
    s = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return s / 9.0


def _natural_scene(rng, w, h, pad):
    """A corner-SPARSE scene, the regime of real driving / indoor footage: large smooth regions (low-frequency shading), a
    few hundred big surfaces with soft edges (every edge is a line of NON-corners for FAST: a corner needs a 9-pixel arc),
    some sharp-edged objects and a little fine texture.  2-5 % of the pixels are FAST corners at t = 7 (the dense `_scene`
    above: 33-53 %)."""
    W = w + pad
    yy, xx = np.mgrid[0:h, 0:W].astype(np.float32)
    img = np.full((h, W), 120.0, np.float32)
    for _ in range(5):
        fx, fy = rng.uniform(-0.004, 0.004, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += 12.0 * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    nsurf = max(6, int(w * h / 6000.0))
    for i in range(nsurf):
        c = rng.uniform(6, 45) * (1 if rng.random() < 0.5 else -1)
        cx, cy = rng.integers(0, W), rng.integers(0, h)
        rw, rh = rng.integers(16, 140), rng.integers(12, 90)
        x0, x1 = max(cx - rw // 2, 0), min(cx + rw // 2 + 1, W)
        y0, y1 = max(cy - rh // 2, 0), min(cy + rh // 2 + 1, h)
        img[y0:y1, x0:x1] += c
    img = _box3(_box3(img))                 # soft edges: most surfaces were out of focus / motion-blurred
    nsharp = max(4, int(w * h / 4000.0))
    for i in range(nsharp):
        c = rng.uniform(12, 90) * (1 if rng.random() < 0.5 else -1)
        cx, cy = rng.integers(0, W), rng.integers(0, h)
        if rng.random() < 0.7:
            rw, rh = rng.integers(3, 30), rng.integers(3, 30)
            x0, x1 = max(cx - rw // 2, 0), min(cx + rw // 2 + 1, W)
            y0, y1 = max(cy - rh // 2, 0), min(cy + rh // 2 + 1, h)
            img[y0:y1, x0:x1] += c
        else:                                # a patch of fine texture (foliage, gravel)
            s = int(rng.integers(6, 20))
            x0, x1 = max(cx - s, 0), min(cx + s + 1, W)
            y0, y1 = max(cy - s, 0), min(cy + s + 1, h)
            if x1 > x0 and y1 > y0:
                img[y0:y1, x0:x1] += rng.normal(0.0, abs(c) / 4.0, (y1 - y0, x1 - x0)).astype(np.float32)
    return img


def _finish_sigma(rng, scene, sigma):
    out = scene + rng.normal(0.0, sigma, scene.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def natural(w, h, k=0):
    """Corner-sparse mono frame k (seed 1000+k), uint8 [h, w]; sensor noise sigma 1."""
    rng = np.random.default_rng(1000 + k)
    return _finish_sigma(rng, _natural_scene(rng, w, h, 0), 1.0)


def natural_pair(w, h, k=0, block=24):
    """Corner-sparse rectified stereo pair, disparity constant over blocks of `block` rows (as stereo_pair_blocky)."""
    rng = np.random.default_rng(1000 + k)
    pad = 64
    sc = _natural_scene(rng, w, h, pad)
    nb = (h + block - 1) // block
    dblk = rng.integers(2, 61, size=nb)
    right = np.empty((h, w), np.float32)
    for y in range(h):
        d = int(dblk[y // block])
        right[y] = sc[y, d:d + w]
    return _finish_sigma(rng, sc[:, :w], 1.0), _finish_sigma(rng, right, 1.0)
