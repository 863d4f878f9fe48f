"""ctypes binding of the CPU ORACLE (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (orb_slam2v2-1_amd) never imports this.
PARITY UNPINNED — see oracle/orb_oracle.h.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
MP_DTYPE = np.dtype([("in_view", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                     ("level", "<i4"), ("view_cos", "<f4"), ("observations", "<i4")])
LASTPT_DTYPE = np.dtype([("has_mp", "<i4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"),
                         ("observations", "<i4"), ("octave", "<i4"), ("angle", "<f4")])
KFPOINT_DTYPE = np.dtype([("valid", "<i4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"), ("max_distance", "<f4"),
                          ("min_distance", "<f4"), ("angle", "<f4")])
MP3D_DTYPE = np.dtype([("valid", "<i4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"), ("nx", "<f4"), ("ny", "<f4"),
                       ("nz", "<f4"), ("max_distance", "<f4"), ("min_distance", "<f4")])
WINDOW_DTYPE = np.dtype([("valid", "<i4"), ("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
                         ("max_level", "<i4"), ("angle", "<f4"), ("blocks", "<i4"), ("ur_c", "<f4"), ("ur_tol", "<f4")])
assert KP_DTYPE.itemsize == 28 and MP_DTYPE.itemsize == 28 and LASTPT_DTYPE.itemsize == 28


class _Img(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("stride", C.c_int32)]


class GridGeom(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float)]


class Cam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mbf", C.c_float), ("mb", C.c_float)]


def grid_geom(w, h):
    """Frame ctor for an undistorted camera (reference: src/Frame.cc:90-105,207-220,469-473)."""
    g = GridGeom()
    g.min_x, g.min_y, g.max_x, g.max_y = 0.0, 0.0, float(w), float(h)
    g.inv_w = np.float32(64) / np.float32(np.float32(w) - np.float32(0))
    g.inv_h = np.float32(48) / np.float32(np.float32(h) - np.float32(0))
    return g


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [i32, f32, i32, i32, i32]
        L.oracle_destroy.argtypes = [vp]
        for name in ("oracle_scale_factors", "oracle_inv_scale_factors", "oracle_level_sigma2",
                     "oracle_inv_level_sigma2", "oracle_features_per_level", "oracle_umax",
                     "oracle_stage_seconds"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        L.oracle_extract.restype = i32
        L.oracle_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32]
        L.oracle_pyramid_level.restype = i32
        L.oracle_pyramid_level.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.oracle_level_candidates.restype = i32
        L.oracle_level_candidates.argtypes = [vp, i32, C.POINTER(vp)]
        L.oracle_level_keypoints.restype = i32
        L.oracle_level_keypoints.argtypes = [vp, i32, C.POINTER(vp)]
        L.oracle_blurred_level.restype = vp
        L.oracle_blurred_level.argtypes = [vp, i32]
        L.oracle_fast_atan2.restype = f32
        L.oracle_fast_atan2.argtypes = [f32, f32]
        L.oracle_set_sincos_mode.argtypes = [i32]
        L.oracle_sincosf.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
        L.oracle_sincosf_array.argtypes = [vp, i32, vp, vp]
        L.oracle_cv_round.restype = i32
        L.oracle_cv_round.argtypes = [C.c_double]
        L.oracle_fast_score.restype = i32
        L.oracle_fast_score.argtypes = [vp, i32, i32]
        L.oracle_fast_is_corner.restype = i32
        L.oracle_fast_is_corner.argtypes = [vp, i32, i32]
        L.oracle_fast_detect.restype = i32
        L.oracle_fast_detect.argtypes = [vp, i32, i32, i32, i32, vp, i32]
        L.oracle_resize_linear.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32]
        L.oracle_gaussian_blur7.argtypes = [vp, i32, i32, i32, vp, i32]
        L.oracle_gaussian_blur7_flavour.argtypes = [vp, i32, i32, i32, vp, i32, i32]
        L.oracle_set_gauss_flavour.restype = i32
        L.oracle_set_gauss_flavour.argtypes = [vp, i32]
        L.oracle_set_gauss_taps.restype = i32
        L.oracle_set_gauss_taps.argtypes = [vp, i32, i32, i32, i32]
        L.oracle_gaussian_blur7_taps.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(C.c_int)]
        L.oracle_gauss_round_half_even.restype = i32
        L.oracle_gauss_round_half_even.argtypes = [i32]
        L.oracle_gauss_round_sse2_literal.restype = i32
        L.oracle_gauss_round_sse2_literal.argtypes = [i32] * 7
        L.oracle_distribute_octtree.restype = i32
        L.oracle_distribute_octtree.argtypes = [vp, i32, i32, i32, i32, vp, i32]
        L.oracle_hamming.restype = i32
        L.oracle_hamming.argtypes = [vp, vp]
        L.oracle_stereo_match.restype = i32
        L.oracle_stereo_match.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, f32, f32, vp, vp]
        L.oracle_grid_build.restype = vp
        L.oracle_grid_build.argtypes = [vp, i32, C.POINTER(GridGeom)]
        L.oracle_grid_free.argtypes = [vp]
        L.oracle_grid_query.restype = i32
        L.oracle_grid_query.argtypes = [vp, f32, f32, f32, i32, i32, vp, i32]
        L.oracle_three_maxima.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.oracle_search_for_initialization.restype = i32
        L.oracle_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, i32, C.POINTER(GridGeom), vp, vp,
                                                       i32, f32, i32]
        L.oracle_search_by_projection_mp.restype = i32
        L.oracle_search_by_projection_mp.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), vp, vp, vp, i32,
                                                     vp, vp, f32, f32]
        L.oracle_search_by_projection_frame.restype = i32
        L.oracle_search_by_projection_frame.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), vp,
                                                        C.POINTER(Cam), vp, vp, vp, vp, i32, vp, vp,
                                                        f32, i32, i32]
        L.oracle_kf_window_queries.argtypes = [vp, i32, C.POINTER(GridGeom), vp, i32, f32, C.POINTER(Cam), vp, f32, vp]
        L.oracle_search_by_projection_kf.restype = i32
        L.oracle_search_by_projection_kf.argtypes = [vp, vp, i32, C.POINTER(GridGeom), vp, i32, f32, C.POINTER(Cam), vp,
                                                     vp, vp, i32, vp, f32, i32, i32]
        G, K = C.POINTER(GridGeom), C.POINTER(Cam)
        L.oracle_sim3_window_queries.argtypes = [vp, i32, G, vp, i32, f32, K, vp, f32, vp]
        L.oracle_pose_window_queries.argtypes = [vp, i32, G, vp, i32, f32, K, vp, f32, vp]
        L.oracle_best_in_windows.argtypes = [vp, vp, vp, i32, G, G, vp, vp, i32, vp, i32, vp, vp]
        L.oracle_search_by_projection_sim3.restype = i32
        L.oracle_search_by_projection_sim3.argtypes = [vp, vp, i32, G, G, vp, i32, f32, K, vp, vp, vp, i32, vp, i32]
        L.oracle_fuse.restype = i32
        L.oracle_fuse.argtypes = [vp, vp, vp, i32, G, G, vp, vp, i32, f32, K, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp]
        L.oracle_fuse_sim3.restype = i32
        L.oracle_fuse_sim3.argtypes = [vp, vp, i32, G, G, vp, i32, f32, K, vp, vp, vp, i32, vp, vp, vp, f32, vp, vp]
        L.oracle_search_by_sim3.restype = i32
        L.oracle_search_by_sim3.argtypes = [vp, vp, i32, vp, vp, i32, G, G, vp, i32, f32, K, vp, vp, f32, vp, vp, vp, vp, vp, vp,
                                            f32, vp]
        L.oracle_predict_scale_ratio.restype = i32
        L.oracle_predict_scale_ratio.argtypes = [f32, f32, i32]
        L.oracle_is_in_frustum.argtypes = [vp, vp, i32, vp, K, G, f32, f32, i32, vp]
        L.oracle_voc_create.restype = vp
        L.oracle_voc_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.oracle_voc_free.argtypes = [vp]
        L.oracle_voc_words.restype = i32
        L.oracle_voc_words.argtypes = [vp]
        L.oracle_voc_transform_one.argtypes = [vp, vp, i32, C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(i32)]
        L.oracle_voc_transform.restype = i32
        L.oracle_voc_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, C.POINTER(i32)]
        L.oracle_search_for_triangulation.restype = i32
        L.oracle_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, f32, f32, vp, vp, i32, i32, vp]
        L.oracle_search_by_bow.restype = i32
        L.oracle_search_by_bow.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
        L.oracle_distinctive_descriptor.restype = i32
        L.oracle_distinctive_descriptor.argtypes = [vp, i32, C.POINTER(i32)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


GAUSS_FLAVOURS = {"half_up": 0, "sse2": 1}


def parse_gauss(g):
    """A Gaussian flavour as the test harness writes it: "half_up" | "sse2" | "taps:k0,k1,k2,k3" (the fixed-point Gaussian of OpenCV >=
    3.4.1 on the build's Q8 taps, centre first).  Returns (code, taps or None)."""
    if isinstance(g, str) and g.startswith("taps:"):
        taps = tuple(int(v) for v in g[5:].split(","))
        if len(taps) != 4:
            raise ValueError("bad gauss flavour %r (taps:k0,k1,k2,k3)" % (g,))
        return 2, taps
    if g not in GAUSS_FLAVOURS:
        raise ValueError("bad gauss flavour %r" % (g,))
    return GAUSS_FLAVOURS[g], None


# flavour an Extractor takes when it is not told (tests that run the whole parity suite under the other flavour set this
# together with the HIP wrapper's default)
# (ORBX_TEST_GAUSS_FLAVOUR in the environment: the same switch as the HIP wrapper's, inherited by spawned workers)
default_gauss_flavour = os.environ.get("ORBX_TEST_GAUSS_FLAVOUR", "half_up")


class Extractor:
    """CPU oracle of ORBextractor (reference: include/ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, gauss=None):
        """gauss: flavour of cv::GaussianBlur on 8U: "half_up" (scalar FixedPtCastEx, the default), "sse2" (SymmColumnVec_32s8u: round
        half to even for the columns x < (w & ~3)) or "taps:k0,k1,k2,k3" (OpenCV >= 3.4.1's fixed-point Gaussian on the build's Q8 taps,
        centre first); None = default_gauss_flavour."""
        self.L = lib()
        self.h = self.L.oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad extractor arguments")
        self.gauss = default_gauss_flavour if gauss is None else gauss
        code, taps = parse_gauss(self.gauss)
        if (self.L.oracle_set_gauss_taps(self.h, *taps) if taps else self.L.oracle_set_gauss_flavour(self.h, code)):
            raise ValueError("bad gauss flavour %r" % (self.gauss,))
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def _tab(self, fn, n, dt):
        return _arr(getattr(self.L, fn)(self.h), n, dt)

    @property
    def scale_factors(self): return self._tab("oracle_scale_factors", self.nlevels, "<f4")
    @property
    def inv_scale_factors(self): return self._tab("oracle_inv_scale_factors", self.nlevels, "<f4")
    @property
    def level_sigma2(self): return self._tab("oracle_level_sigma2", self.nlevels, "<f4")
    @property
    def inv_level_sigma2(self): return self._tab("oracle_inv_level_sigma2", self.nlevels, "<f4")
    @property
    def features_per_level(self): return self._tab("oracle_features_per_level", self.nlevels, "<i4")
    @property
    def umax(self): return self._tab("oracle_umax", 16, "<i4")
    @property
    def stage_seconds(self): return self._tab("oracle_stage_seconds", 6, "<f8")

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 4 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.oracle_extract(self.h, _p(img), w, h, w, _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError("oracle_extract failed: %d" % n)
        return kps[:n].copy(), desc[:n].copy()

    def pyramid_level(self, level, padded=False):
        ptr, w, h, ps = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        if self.L.oracle_pyramid_level(self.h, level, C.byref(ptr), C.byref(w), C.byref(h), C.byref(ps)):
            raise RuntimeError("no pyramid")
        full = _arr(ptr.value, ps.value * (h.value + 38), np.uint8).reshape(h.value + 38, ps.value)
        return full if padded else full[19:19 + h.value, 19:19 + w.value].copy()

    def blurred_level(self, level):
        """The level after GaussianBlur(7x7, sigma 2, REFLECT_101) (src/ORBextractor.cc:1085-1086), [h, w] uint8."""
        ptr, w, h, ps = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        if self.L.oracle_pyramid_level(self.h, level, C.byref(ptr), C.byref(w), C.byref(h), C.byref(ps)):
            raise RuntimeError("no pyramid")
        return _arr(self.L.oracle_blurred_level(self.h, level), w.value * h.value, np.uint8).reshape(h.value, w.value)

    def level_candidates(self, level):
        ptr = C.c_void_p()
        n = self.L.oracle_level_candidates(self.h, level, C.byref(ptr))
        return _arr(ptr.value, n, CAND_DTYPE)

    def level_keypoints(self, level):
        ptr = C.c_void_p()
        n = self.L.oracle_level_keypoints(self.h, level, C.byref(ptr))
        return _arr(ptr.value, n, CAND_DTYPE)

    def pyramid_imgs(self):
        """(array of _Img, keepalive list) for the stereo matcher: inner ROIs."""
        keep, arr = [], (_Img * self.nlevels)()
        for l in range(self.nlevels):
            im = np.ascontiguousarray(self.pyramid_level(l))
            keep.append(im)
            arr[l].ptr, arr[l].w, arr[l].h, arr[l].stride = im.ctypes.data, im.shape[1], im.shape[0], im.shape[1]
        return arr, keep


def imgs_from_levels(levels):
    keep, arr = [], (_Img * len(levels))()
    for l, im in enumerate(levels):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        keep.append(im)
        arr[l].ptr, arr[l].w, arr[l].h, arr[l].stride = im.ctypes.data, im.shape[1], im.shape[0], im.shape[1]
    return arr, keep


def set_sincos_mode(mode):
    """0 restated glibc cosf/sinf (default = the reference's float overloads), 1 host libm cosf/sinf, 2 double rounded."""
    lib().oracle_set_sincos_mode(int(mode))


def sincosf(angle):
    s, c = C.c_float(), C.c_float()
    lib().oracle_sincosf(float(angle), C.byref(s), C.byref(c))
    return s.value, c.value


def sincosf_array(angles):
    a = np.ascontiguousarray(angles, np.float32)
    s, c = np.zeros_like(a), np.zeros_like(a)
    lib().oracle_sincosf_array(_p(a), len(a), _p(s), _p(c))
    return s, c


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_hamming(_p(a), _p(b))


def fast_detect(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(w * h, CAND_DTYPE)
    n = lib().oracle_fast_detect(_p(img), w, w, h, threshold, _p(out), out.size)
    return out[:n].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src, gauss="half_up"):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    code, taps = parse_gauss(gauss)
    if taps:
        lib().oracle_gaussian_blur7_taps(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1], (C.c_int * 4)(*taps))
    else:
        lib().oracle_gaussian_blur7_flavour(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1], code)
    return dst


def distribute_octtree(cands, width, height, N):
    cands = np.ascontiguousarray(cands, CAND_DTYPE)
    out = np.zeros(max(N + 8, 4 * int(round(width / height)) + 8), CAND_DTYPE)   # the first pass splits every root whatever N is
    n = lib().oracle_distribute_octtree(_p(cands), len(cands), width, height, N, _p(out), out.size)
    if n < 0 or n > out.size:
        raise RuntimeError("octtree failed")
    return out[:n].copy()


def stereo_match(kl, dl, kr, dr, pyr_l, pyr_r, sf, isf, mbf, mb):
    """pyr_l / pyr_r: lists of 2-D uint8 inner level images."""
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    sf = np.ascontiguousarray(sf, np.float32); isf = np.ascontiguousarray(isf, np.float32)
    al, keepl = imgs_from_levels(pyr_l)
    ar, keepr = imgs_from_levels(pyr_r)
    ur = np.zeros(len(kl), np.float32); dp = np.zeros(len(kl), np.float32)
    n = lib().oracle_stereo_match(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr),
                                  C.cast(al, C.c_void_p), C.cast(ar, C.c_void_p), len(pyr_l),
                                  _p(sf), _p(isf), mbf, mb, _p(ur), _p(dp))
    return n, ur, dp


def grid_query(kps, geom, x, y, r, min_level, max_level):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    L = lib()
    g = L.oracle_grid_build(_p(kps), len(kps), C.byref(geom))
    out = np.zeros(max(len(kps), 1), np.int32)
    n = L.oracle_grid_query(g, x, y, r, min_level, max_level, _p(out), out.size)
    L.oracle_grid_free(g)
    return out[:n].copy()


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().oracle_three_maxima(_p(sizes), len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def search_for_initialization(k1, d1, k2, d2, geom2, prev_matched, window=100, nnratio=0.9, check_ori=True):
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(len(k1), np.int32)
    n = lib().oracle_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2),
                                               C.byref(geom2), _p(prev), _p(m12), window, nnratio,
                                               int(check_ori))
    return n, m12, prev


def search_by_projection_mp(kun, desc, uright, geom, sf, mps, mp_desc, frame_mp, ext_obs, th, nnratio):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    uright = np.ascontiguousarray(uright, np.float32); sf = np.ascontiguousarray(sf, np.float32)
    mps = np.ascontiguousarray(mps, MP_DTYPE); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    fm = np.ascontiguousarray(frame_mp, np.int32).copy()
    eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
    n = lib().oracle_search_by_projection_mp(_p(kun), _p(desc), _p(uright), len(kun), C.byref(geom), _p(sf),
                                             _p(mps), _p(mp_desc), len(mps), _p(fm), _p(eo), th, nnratio)
    return n, fm


def search_by_projection_frame(kun, desc, uright, geom, sf, cam, Tcw_cur, Tcw_last, last, last_desc,
                               cur_mp, ext_obs, th, mono, check_ori=True):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    uright = np.ascontiguousarray(uright, np.float32); sf = np.ascontiguousarray(sf, np.float32)
    last = np.ascontiguousarray(last, LASTPT_DTYPE); last_desc = np.ascontiguousarray(last_desc, np.uint8)
    Tc = np.ascontiguousarray(Tcw_cur, np.float32); Tl = np.ascontiguousarray(Tcw_last, np.float32)
    cm = np.ascontiguousarray(cur_mp, np.int32).copy()
    eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
    n = lib().oracle_search_by_projection_frame(_p(kun), _p(desc), _p(uright), len(kun), C.byref(geom),
                                                _p(sf), C.byref(cam), _p(Tc), _p(Tl), _p(last),
                                                _p(last_desc), len(last), _p(cm), _p(eo), th, int(mono),
                                                int(check_ori))
    return n, cm


def kf_window_queries(kf, geom, sf, log_sf, cam, Tcw_cur, th):
    kf = np.ascontiguousarray(kf, KFPOINT_DTYPE); sf = np.ascontiguousarray(sf, np.float32)
    Tc = np.ascontiguousarray(Tcw_cur, np.float32)
    q = np.zeros(len(kf), WINDOW_DTYPE)
    lib().oracle_kf_window_queries(_p(kf), len(kf), C.byref(geom), _p(sf), len(sf), float(log_sf), C.byref(cam), _p(Tc),
                                   float(th), _p(q))
    return q


def search_by_projection_kf(kun, desc, geom, sf, log_sf, cam, Tcw_cur, kf, kf_desc, cur_mp, th, orb_dist, check_ori=True):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(sf, np.float32); kf = np.ascontiguousarray(kf, KFPOINT_DTYPE)
    kd = np.ascontiguousarray(kf_desc, np.uint8); Tc = np.ascontiguousarray(Tcw_cur, np.float32)
    cm = np.ascontiguousarray(cur_mp, np.int32).copy()
    n = lib().oracle_search_by_projection_kf(_p(kun), _p(desc), len(kun), C.byref(geom), _p(sf), len(sf), float(log_sf),
                                             C.byref(cam), _p(Tc), _p(kf), _p(kd), len(kf), _p(cm), float(th),
                                             int(orb_dist), int(check_ori))
    return n, cm


# ---- KeyFrame-side matchers (orb_oracle_kfmatch.c)
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _g(geom):
    return None if geom is None else C.byref(geom)


def sim3_window_queries(pts, geom, sf, log_sf, cam, Scw, th):
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); sf = _f32(sf); S = _f32(Scw)
    q = np.zeros(len(pts), WINDOW_DTYPE)
    lib().oracle_sim3_window_queries(_p(pts), len(pts), C.byref(geom), _p(sf), len(sf), float(log_sf), C.byref(cam),
                                     _p(S), float(th), _p(q))
    return q


def pose_window_queries(pts, geom, sf, log_sf, cam, Tcw, th):
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); sf = _f32(sf); T = _f32(Tcw)
    q = np.zeros(len(pts), WINDOW_DTYPE)
    lib().oracle_pose_window_queries(_p(pts), len(pts), C.byref(geom), _p(sf), len(sf), float(log_sf), C.byref(cam),
                                     _p(T), float(th), _p(q))
    return q


def best_in_windows(kun, desc, uright, geom, q, qdesc, inv_sigma2=None, geom_assign=None):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else _f32(uright)
    q = np.ascontiguousarray(q, WINDOW_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
    s2 = None if inv_sigma2 is None else _f32(inv_sigma2)
    bi = np.zeros(len(q), np.int32); bd = np.zeros(len(q), np.int32)
    lib().oracle_best_in_windows(_p(kun), _p(desc), _p(ur), len(kun), C.byref(geom), _g(geom_assign), _p(q), _p(qd), len(q), _p(s2),
                                 0 if s2 is None else len(s2), _p(bi), _p(bd))
    return bi, bd


def search_by_projection_sim3(kun, desc, geom, sf, log_sf, cam, Scw, pts, pdesc, matched, th, geom_assign=None):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); pd = np.ascontiguousarray(pdesc, np.uint8)
    sf = _f32(sf); S = _f32(Scw); mt = np.ascontiguousarray(matched, np.int32).copy()
    n = lib().oracle_search_by_projection_sim3(_p(kun), _p(desc), len(kun), C.byref(geom), _g(geom_assign), _p(sf), len(sf), float(log_sf),
                                               C.byref(cam), _p(S), _p(pts), _p(pd), len(pts), _p(mt), int(th))
    return n, mt


def fuse(kun, desc, uright, geom, sf, inv_sigma2, log_sf, cam, Tcw, pts, pdesc, bad, in_kf, obs, slot, ext_obs, ext_bad, th,
         geom_assign=None):
    """-> (nFused, best_idx, action, state dict after the call)"""
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else _f32(uright)
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); pd = np.ascontiguousarray(pdesc, np.uint8)
    sf = _f32(sf); s2 = _f32(inv_sigma2); T = _f32(Tcw)
    st = {k: np.ascontiguousarray(v, np.int32).copy() for k, v in
          dict(bad=bad, in_kf=in_kf, obs=obs, slot=slot, ext_obs=ext_obs, ext_bad=ext_bad).items()}
    bi = np.zeros(len(pts), np.int32); act = np.zeros(len(pts), np.int32)
    n = lib().oracle_fuse(_p(kun), _p(desc), _p(ur), len(kun), C.byref(geom), _g(geom_assign), _p(sf), _p(s2), len(sf), float(log_sf),
                          C.byref(cam), _p(T), _p(pts), _p(pd), len(pts), _p(st["bad"]), _p(st["in_kf"]), _p(st["obs"]),
                          _p(st["slot"]), _p(st["ext_obs"]), _p(st["ext_bad"]), float(th), _p(bi), _p(act))
    return n, bi, act, st


def fuse_sim3(kun, desc, geom, sf, log_sf, cam, Scw, pts, pdesc, bad, slot, ext_bad, th, geom_assign=None):
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); pd = np.ascontiguousarray(pdesc, np.uint8)
    sf = _f32(sf); S = _f32(Scw)
    bad = np.ascontiguousarray(bad, np.int32); eb = np.ascontiguousarray(ext_bad, np.int32)
    sl = np.ascontiguousarray(slot, np.int32).copy()
    bi = np.zeros(len(pts), np.int32); rep = np.zeros(len(pts), np.int32)
    n = lib().oracle_fuse_sim3(_p(kun), _p(desc), len(kun), C.byref(geom), _g(geom_assign), _p(sf), len(sf), float(log_sf), C.byref(cam),
                               _p(S), _p(pts), _p(pd), len(pts), _p(bad), _p(sl), _p(eb), float(th), _p(bi), _p(rep))
    return n, bi, rep, sl


def search_by_sim3(k1, d1, k2, d2, geom, sf, log_sf, cam, T1w, T2w, s12, R12, t12, pts1, pd1, pts2, pd2, th, geom_assign=None):
    k1 = np.ascontiguousarray(k1, KP_DTYPE); d1 = np.ascontiguousarray(d1, np.uint8)
    k2 = np.ascontiguousarray(k2, KP_DTYPE); d2 = np.ascontiguousarray(d2, np.uint8)
    pts1 = np.ascontiguousarray(pts1, MP3D_DTYPE); pd1 = np.ascontiguousarray(pd1, np.uint8)
    pts2 = np.ascontiguousarray(pts2, MP3D_DTYPE); pd2 = np.ascontiguousarray(pd2, np.uint8)
    sf = _f32(sf); T1 = _f32(T1w); T2 = _f32(T2w); R = _f32(R12); t = _f32(t12)
    m12 = np.zeros(len(k1), np.int32)
    n = lib().oracle_search_by_sim3(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), C.byref(geom), _g(geom_assign), _p(sf), len(sf),
                                    float(log_sf), C.byref(cam), _p(T1), _p(T2), float(s12), _p(R), _p(t), _p(pts1),
                                    _p(pd1), _p(pts2), _p(pd2), float(th), _p(m12))
    return n, m12


def distinctive_descriptor(desc):
    """-> (BestIdx, BestMedian) of MapPoint::ComputeDistinctiveDescriptors for one map point"""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    med = C.c_int(0)
    i = lib().oracle_distinctive_descriptor(_p(desc), len(desc), C.byref(med))
    return i, med.value


def is_in_frustum(pts, obs, Tcw, cam, geom, viewing_cos_limit, log_sf, nlevels):
    """Frame::isInFrustum for a list of map points -> MP_DTYPE records"""
    pts = np.ascontiguousarray(pts, MP3D_DTYPE); T = _f32(Tcw)
    ob = None if obs is None else np.ascontiguousarray(obs, np.int32)
    out = np.zeros(len(pts), MP_DTYPE)
    lib().oracle_is_in_frustum(_p(pts), _p(ob), len(pts), _p(T), C.byref(cam), C.byref(geom), float(viewing_cos_limit),
                               float(log_sf), int(nlevels), _p(out))
    return out


class Vocabulary:
    """DBoW2 vocabulary tree (TemplatedVocabulary) from per-node arrays in file order."""

    def __init__(self, k, L, scoring, weighting, parent, is_leaf, desc, weight):
        self.parent = np.ascontiguousarray(parent, np.int32); self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); self.weight = np.ascontiguousarray(weight, np.float64)
        self.k, self.L = k, L
        self.h = lib().oracle_voc_create(k, L, scoring, weighting, len(self.parent), _p(self.parent), _p(self.is_leaf),
                                         _p(self.desc), _p(self.weight))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_voc_free(self.h)
            self.h = None

    def transform_one(self, feature, levelsup):
        f = np.ascontiguousarray(feature, np.uint8)
        w, nid, wt = C.c_int(0), C.c_int(0), C.c_double(0)
        lib().oracle_voc_transform_one(self.h, _p(f), int(levelsup), C.byref(w), C.byref(wt), C.byref(nid))
        return w.value, wt.value, nid.value

    def transform(self, features, levelsup):
        """-> (bow_word, bow_value, {node: [feature indices]})"""
        f = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(f)
        bw = np.zeros(n + 1, np.int32); bv = np.zeros(n + 1, np.float64)
        fn = np.zeros(n + 1, np.int32); fs = np.zeros(n + 2, np.int32); fi = np.zeros(n + 1, np.int32)
        nfv = C.c_int(0)
        nb = lib().oracle_voc_transform(self.h, _p(f), n, int(levelsup), _p(bw), _p(bv), _p(fn), _p(fs), _p(fi), C.byref(nfv))
        fv = {int(fn[p]): [int(x) for x in fi[fs[p]:fs[p + 1]]] for p in range(nfv.value)}
        return bw[:nb].copy(), bv[:nb].copy(), fv


def search_by_bow(qd, qa, qv, cd, ca, cv, nqs, qit, ncs, cit, th_low, strict_lt, nnratio, check_ori=True):
    qd = np.ascontiguousarray(qd, np.uint8); qa = _f32(qa); qv = np.ascontiguousarray(qv, np.uint8)
    cd = np.ascontiguousarray(cd, np.uint8); ca = _f32(ca)
    cvv = None if cv is None else np.ascontiguousarray(cv, np.uint8)
    nqs = np.ascontiguousarray(nqs, np.int32); qit = np.ascontiguousarray(qit, np.int32)
    ncs = np.ascontiguousarray(ncs, np.int32); cit = np.ascontiguousarray(cit, np.int32)
    mq = np.zeros(len(qa), np.int32)
    n = lib().oracle_search_by_bow(_p(qd), _p(qa), _p(qv), len(qa), _p(cd), _p(ca), _p(cvv), len(ca), _p(nqs), _p(qit), _p(ncs),
                                   _p(cit), len(nqs) - 1, int(th_low), int(strict_lt), float(nnratio), int(check_ori), _p(mq))
    return n, mq


def search_for_triangulation(k1, qd, qf, k2, cd, cf, nqs, qit, ncs, cit, F12, ex, ey, sf, sigma2, th_low=50, check_ori=True):
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    qd = np.ascontiguousarray(qd, np.uint8); cd = np.ascontiguousarray(cd, np.uint8)
    qf = np.ascontiguousarray(qf, np.uint8); cf = np.ascontiguousarray(cf, np.uint8)
    nqs = np.ascontiguousarray(nqs, np.int32); qit = np.ascontiguousarray(qit, np.int32)
    ncs = np.ascontiguousarray(ncs, np.int32); cit = np.ascontiguousarray(cit, np.int32)
    F = _f32(F12); sf = _f32(sf); s2 = _f32(sigma2)
    mq = np.zeros(len(k1), np.int32)
    n = lib().oracle_search_for_triangulation(_p(k1), _p(qd), _p(qf), len(k1), _p(k2), _p(cd), _p(cf), len(k2), _p(nqs), _p(qit),
                                              _p(ncs), _p(cit), len(nqs) - 1, _p(F), float(ex), float(ey), _p(sf), _p(s2),
                                              int(th_low), int(check_ori), _p(mq))
    return n, mq
