"""orb_slam2v2-1_amd — MI355X-native ORB front-end + Hamming matchers (HIP, gfx950).

Python-side binding of the C ABI in include/orbx.h (ctypes; no torch types cross the
boundary).  The classes mirror the reference's operator interface for this path:

  ORBextractor  <- include/ORBextractor.h:45-111   (ctor args, operator(), getters,
                                                     mvImagePyramid)
  ORBmatcher    <- include/ORBmatcher.h:37-102     (TH_LOW/TH_HIGH/HISTO_LENGTH,
                                                     DescriptorDistance, the hot Search*)
  compute_stereo_matches <- Frame::ComputeStereoMatches (src/Frame.cc:481-655)

There is NO CPU fallback: if the HIP library is missing or no GPU is usable the calls
raise OrbxError.  (The CPU oracle lives in /oracle and is test infrastructure only.)
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The PRODUCT library (no test hooks), and the DEVELOPER build of the same sources (-DORBX_DEVELOPER: + the read-only stage hooks of
# include/orbx_dev.h, + the phase-stop / time-stamp option keys of the probes in tools/).  ORBX_LIB names another file for the first.
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "lib", "liborbx_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "lib", "liborbx_hip_dev.so")
# extractors created while this is True come from the developer build (tests that look at intermediate stages: `hooks` fixture)
default_developer = False

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MP_DTYPE = np.dtype([("in_view", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                     ("level", "<i4"), ("view_cos", "<f4"), ("observations", "<i4")])
LASTPT_DTYPE = np.dtype([("has_mp", "<i4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"),
                         ("observations", "<i4"), ("octave", "<i4"), ("angle", "<f4")])
WINDOW_DTYPE = np.dtype([("valid", "<i4"), ("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
                         ("max_level", "<i4"), ("angle", "<f4"), ("blocks", "<i4"), ("ur_c", "<f4"), ("ur_tol", "<f4")])

ORBX_OK, ORBX_ERR_ARG, ORBX_ERR_NO_DEVICE, ORBX_ERR_HIP, ORBX_ERR_CAPACITY, ORBX_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
NUM_STAGES = 5

# every symbol include/orbx.h declares (tests check the library exports all of them)
EXPORTS = [
    "orbx_create", "orbx_destroy", "orbx_get_levels", "orbx_get_scale_factor", "orbx_get_tables",
    "orbx_max_keypoints", "orbx_extract", "orbx_extract_batch", "orbx_extract_batch_device",
    "orbx_pyramid_host", "orbx_pyramid_device", "orbx_level_counts", "orbx_set_profiling",
    "orbx_get_stage_ms", "orbx_create_flavoured", "orbx_get_flavour", "orbx_set_option", "orbx_get_option", "orbm_set_thread_option", "orbm_hamming", "orbm_hamming_matrix_device", "orbm_stereo_batch_device",
    "orbm_stereo", "orbm_search_for_initialization", "orbm_search_by_projection_mp",
    "orbm_search_by_projection_frame", "orbm_match_windows", "orbm_best_in_windows", "orbm_distinctive_descriptors", "orbm_predict_scale_thresholds", "orbm_is_in_frustum",
    "orbm_search_local_points", "orbv_create", "orbv_load_text", "orbv_destroy", "orbv_info", "orbv_transform",
    "orbm_search_by_bow", "orbm_search_for_triangulation", "orbx_last_error", "orbx_version", "orbx_device_count",
    "orbx_record_bytes", "orbx_pack_records_device", "orbx_thread_release_scratch",
    "orbm_search_by_projection_frame_device", "orbm_search_local_points_device", "orbx_fast_kernels", "orbx_extract_batch_device_prefetch", "orbx_stream_wait_fast_stage", "orbx_side_stream", "orbm_stereo_batch_device_prev",
    "orbx_side_stream_for", "orbx_stereo_frame", "orbx_set_pyramid_buffers",
    "orbx_stereo_frame_view", "orbx_host_alloc", "orbx_host_free",
]
# what include/orbx_dev.h declares on top: exported by the developer build only
DEV_EXPORTS = ["orbx_debug_level_points", "orbx_debug_sincosf", "orbx_debug_blur_patches", "orbm_debug_features_in_area",
               "orbx_debug_blurred_level", "orbx_debug_octree_fallbacks"]


class OrbxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("orbx status %d: %s" % (status, msg))
        self.status = status


class StereoView(C.Structure):
    """orbx_stereo_view_t (orbx_stereo_frame_view)."""
    _fields_ = [("nl", C.c_int32), ("nr", C.c_int32), ("nmatch", C.c_int32), ("cap", C.c_int32),
                ("kl", C.c_void_p), ("kr", C.c_void_p), ("dl", C.c_void_p), ("dr", C.c_void_p), ("uright", C.c_void_p), ("depth", C.c_void_p),
                ("d_kl", C.c_void_p), ("d_kr", C.c_void_p), ("d_dl", C.c_void_p), ("d_dr", C.c_void_p), ("d_uright", C.c_void_p),
                ("d_depth", C.c_void_p)]


class GridGeom(C.Structure):
    """orbm_grid_geom_t (src/Frame.cc:90-105)."""
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mbf", C.c_float), ("mb", C.c_float)]


def grid_geom(w, h):
    """Grid of an undistorted camera: mnMinX=0, mnMaxX=cols (src/Frame.cc:469-478)."""
    g = GridGeom()
    g.min_x, g.min_y, g.max_x, g.max_y = 0.0, 0.0, float(w), float(h)
    g.inv_w = np.float32(64) / np.float32(w)
    g.inv_h = np.float32(48) / np.float32(h)
    return g


class Flavour(C.Structure):
    """orbx_flavour_t: which OpenCV build the handle stands in for (include/orbx.h)."""
    _fields_ = [("gauss_rounding", C.c_int32), ("gauss_taps", C.c_int32 * 4), ("reserved", C.c_int32 * 3)]


GAUSS_FLAVOURS = {"half_up": 0, "sse2": 1}
GAUSS_FIXED_TAPS = 2


def parse_gauss(g):
    """"half_up" | "sse2" | "taps:k0,k1,k2,k3" (ORBX_GAUSS_FIXED_TAPS: the build's Q8 taps, centre first) -> Flavour."""
    fl = Flavour()
    if isinstance(g, str) and g.startswith("taps:"):
        taps = [int(v) for v in g[5:].split(",")]
        if len(taps) != 4:
            raise ValueError("bad gauss flavour %r (taps:k0,k1,k2,k3)" % (g,))
        fl.gauss_rounding = GAUSS_FIXED_TAPS
        for i, v in enumerate(taps):
            fl.gauss_taps[i] = v
    elif g in GAUSS_FLAVOURS:
        fl.gauss_rounding = GAUSS_FLAVOURS[g]
    else:
        raise ValueError("bad gauss flavour %r" % (g,))
    return fl
# What an ORBextractor takes when its constructor is not told.  HARNESS state of this Python module (the library itself has no
# process-global switch: flavour and options are per handle): the parity suite is run under both flavours by changing this and the
# oracle's default together, and the tests that cover an alternative kernel set a default option around the code under test.
# ORBX_TEST_GAUSS_FLAVOUR=sse2 in the environment runs a whole test / bench session (spawned oracle workers included) under the
# other flavour.
default_gauss_flavour = os.environ.get("ORBX_TEST_GAUSS_FLAVOUR", "half_up")
_default_options = {}
_live = None


def set_default_option(key, value):
    """Option `key` of every ORBextractor created from now on AND of the live ones (value 0 = the library default)."""
    if value:
        _default_options[int(key)] = int(value)
    else:
        _default_options.pop(int(key), None)
    for e in list(_live or ()):
        if getattr(e, "_h", None):
            e.set_option(key, value)


_lib = None
_dev_lib = None


def build(force=False):
    from . import build as _b
    return _b.build(force=force)


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and
    request it by the un-versioned name, so if our library pulled in /opt/rocm's copy first, a
    later `import torch` would start a SECOND runtime (whose device init then fails and whose
    pointers/streams are foreign to ours).  When torch is installed, load its copy first; our
    library's NEEDED libamdhip64.so.7 then binds to it by soname, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib(developer=False):
    """Load the HIP library (developer = True: the developer build with the stage hooks); fails loudly when it has not been built."""
    global _lib, _dev_lib
    if developer:
        if _dev_lib is None:
            _dev_lib = _load(DEV_LIB_PATH, True)
        return _dev_lib
    if _lib is None:
        _lib = _load(LIB_PATH, os.path.basename(LIB_PATH) == "liborbx_hip_dev.so")
    return _lib


def _load(path, dev):
    if not os.path.exists(path):
        raise OrbxError(ORBX_ERR_NO_DEVICE, "HIP library %s is missing: run `python __graft_entry__.py` "
                                            "(build()) first; there is no CPU fallback" % path)
    _preload_shared_hip_runtime()
    L = C.CDLL(path)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbx_create.restype = i32
    L.orbx_create.argtypes = [i32, f32, i32, i32, i32, i32, C.POINTER(vp)]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_create_flavoured.argtypes = [i32, f32, i32, i32, i32, i32, C.POINTER(Flavour), C.POINTER(vp)]
    L.orbx_get_flavour.argtypes = [vp, C.POINTER(Flavour)]
    L.orbx_set_option.argtypes = [vp, i32, i32]
    L.orbx_get_option.argtypes = [vp, i32, C.POINTER(i32)]
    L.orbm_set_thread_option.argtypes = [i32, i32]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f32
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.orbx_max_keypoints.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    L.orbx_extract_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, sz, vp, vp, vp, i32, vp]
    L.orbx_extract_batch_device_prefetch.argtypes = [vp, vp, i32, i32, i32, i32, sz, vp]
    L.orbx_stream_wait_fast_stage.argtypes = [vp, vp]
    L.orbx_side_stream.argtypes = [vp]
    L.orbx_side_stream.restype = vp
    L.orbx_stereo_frame.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, C.POINTER(i32), vp, vp, C.POINTER(i32), vp, vp,
                                    C.POINTER(i32)]
    L.orbx_stereo_frame_view.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32, vp]
    L.orbx_host_alloc.argtypes = [sz]
    L.orbx_host_alloc.restype = vp
    L.orbx_host_free.argtypes = [vp]
    L.orbx_host_free.restype = None
    L.orbx_set_pyramid_buffers.argtypes = [vp, i32]
    L.orbx_side_stream_for.argtypes = [vp, vp]
    L.orbx_side_stream_for.restype = vp
    L.orbx_pyramid_host.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.orbx_pyramid_device.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_level_counts.argtypes = [vp, i32, vp, vp]
    L.orbx_set_profiling.argtypes = [vp, i32]
    L.orbx_fast_kernels.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_record_bytes.argtypes = [i32]
    L.orbx_pack_records_device.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp]
    L.orbx_get_stage_ms.argtypes = [vp, vp, C.POINTER(i32)]
    L.orbm_hamming.argtypes = [vp, vp]
    L.orbm_hamming_matrix_device.argtypes = [vp, i32, vp, i32, vp, vp]
    L.orbm_stereo_batch_device.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp]
    L.orbm_stereo_batch_device_prev.argtypes = L.orbm_stereo_batch_device.argtypes
    L.orbm_stereo.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, C.POINTER(i32)]
    L.orbm_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, i32, C.POINTER(GridGeom), vp, vp, i32, f32,
                                                 i32, i32, C.POINTER(i32)]
    L.orbm_search_by_projection_mp.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), vp, i32, vp, vp, i32, vp, vp,
                                               f32, f32, i32, C.POINTER(i32)]
    L.orbm_search_by_projection_frame.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), vp, i32, C.POINTER(Camera),
                                                  vp, vp, vp, vp, i32, vp, vp, f32, i32, i32, i32, C.POINTER(i32)]
    L.orbm_search_by_projection_frame_device.argtypes = L.orbm_search_by_projection_frame.argtypes + [vp]
    L.orbm_match_windows.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), C.POINTER(GridGeom), vp, vp, i32, vp, vp, i32, i32, i32, C.POINTER(i32)]
    L.orbm_distinctive_descriptors.argtypes = [vp, vp, i32, vp, vp, i32]
    L.orbm_predict_scale_thresholds.argtypes = [f32, i32, vp]
    L.orbv_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, C.POINTER(vp)]
    L.orbv_load_text.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.orbv_destroy.argtypes = [vp]
    L.orbv_destroy.restype = None
    L.orbv_info.argtypes = [vp] + [C.POINTER(i32)] * 6
    L.orbv_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbm_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, f32, f32, vp, vp, i32, i32, i32, vp,
                                                C.POINTER(i32), i32]
    L.orbm_search_by_bow.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, i32, vp, C.POINTER(i32), i32]
    L.orbm_is_in_frustum.argtypes = [vp, i32, vp, C.POINTER(Camera), C.POINTER(GridGeom), f32, vp, i32, vp, i32]
    L.orbm_search_local_points.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), vp, i32, vp, vp, i32, vp, C.POINTER(Camera), f32,
                                           vp, vp, vp, f32, f32, i32, C.POINTER(i32), vp]
    L.orbm_search_local_points_device.argtypes = L.orbm_search_local_points.argtypes + [vp]
    L.orbm_best_in_windows.argtypes = [vp, vp, vp, i32, C.POINTER(GridGeom), C.POINTER(GridGeom), vp, vp, i32, vp, i32, vp, vp, i32]
    if dev:
        L.orbx_debug_level_points.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
        L.orbx_debug_sincosf.argtypes = [vp, i32, vp, vp, i32]
        L.orbx_debug_blur_patches.argtypes = [vp, i32, vp, i32]
        L.orbx_debug_octree_fallbacks.argtypes = [vp, vp, i32]
        L.orbx_debug_blurred_level.argtypes = [vp, i32, i32, vp, i32, vp]
        L.orbm_debug_features_in_area.argtypes = [vp, i32, C.POINTER(GridGeom), f32, f32, f32, i32, i32, vp, C.POINTER(i32), i32]
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_version.restype = C.c_char_p
    L._orbx_developer = bool(dev)
    return L


def _check(rc, L=None):
    if rc != 0:
        raise OrbxError(rc, (L or lib()).orbx_last_error().decode())


def _p(a):
    return a.ctypes.data if a is not None else None      # (an int: every pointer parameter is declared c_void_p; data_as() costs a microsecond more)


def blur_reach_mask():
    """[37, 37] bool: the pixels of a keypoint's blurred block a descriptor tap can reach - the rounded rotation of a pattern point
    with x^2 + y^2 <= 338 satisfies (|row| - 1/2)^2 + (|col| - 1/2)^2 <= 338 (+ 2 of slack), 1133 of 1369 pixels.  The fused blur of
    the descriptor kernel computes nothing else, and ORBextractor.debug_blur_patches reports the rest as 0."""
    d = np.abs(np.arange(-18, 19))
    t = np.where(d > 0, 2 * d - 1, 0)
    return (t[:, None] ** 2 + t[None, :] ** 2) <= 4 * 340


def device_count():
    return lib().orbx_device_count()


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference: include/ORBextractor.h:45-111).

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); calling the object
    on an 8-bit gray image returns (keypoints[KP_DTYPE], descriptors[N,32] uint8); the mask
    argument of the reference is ignored there (src/ORBextractor.cc:1043) and absent here.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0, gauss=None, developer=None):
        """gauss: "half_up" / "sse2" / "taps:k0,k1,k2,k3" = orbx_flavour_t (include/orbx.h); None = default_gauss_flavour.
        developer: True = this handle lives in the developer build of the library (stage hooks debug_*); None = default_developer."""
        global _live
        self._L = lib(default_developer if developer is None else developer)
        h = C.c_void_p()
        self.gauss = default_gauss_flavour if gauss is None else gauss
        fl = parse_gauss(self.gauss)
        self._ck(self._L.orbx_create_flavoured(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST),
                                             int(minThFAST), int(device), C.byref(fl), C.byref(h)))
        self._h = h
        self.nfeatures, self.nlevels, self.device = int(nfeatures), int(nlevels), int(device)
        self._shape = None
        if _live is None:
            import weakref
            _live = weakref.WeakSet()
        _live.add(self)
        for k, v in _default_options.items():
            self.set_option(k, v)

    def _ck(self, rc):
        _check(rc, self._L)

    def _hooks(self):
        if not self._L._orbx_developer:
            raise OrbxError(ORBX_ERR_ARG, "stage hooks (debug_*) exist in the developer build only: ORBextractor(..., developer=True)")

    def level_counts(self, b=0):
        """(FAST candidates per level, keypoints kept per level) of image slot b of the last call (orbx_level_counts)."""
        c, k = np.zeros(self.nlevels, np.int32), np.zeros(self.nlevels, np.int32)
        self._ck(self._L.orbx_level_counts(self._h, int(b), _p(c), _p(k)))
        return c, k

    def set_option(self, key, value):
        """orbx_set_option: per-handle choice among kernels / arrangements with identical results (ORBX_OPT_* of include/orbx.h)."""
        self._ck(self._L.orbx_set_option(self._h, int(key), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        self._ck(self._L.orbx_get_option(self._h, int(key), C.byref(v)))
        return v.value

    def flavour(self):
        fl = Flavour()
        self._ck(self._L.orbx_get_flavour(self._h, C.byref(fl)))
        if fl.gauss_rounding == GAUSS_FIXED_TAPS:
            return "taps:" + ",".join(str(int(v)) for v in fl.gauss_taps)
        return {v: k for k, v in GAUSS_FLAVOURS.items()}[fl.gauss_rounding]

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters (include/ORBextractor.h:62-83)
    def GetLevels(self):
        return self._L.orbx_get_levels(self._h)

    def GetScaleFactor(self):
        return self._L.orbx_get_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32), np.zeros(16, np.int32)]
        self._ck(self._L.orbx_get_tables(self._h, *[_p(a) for a in t]))
        return t

    def GetScaleFactors(self): return self._tables()[0]
    def GetInverseScaleFactors(self): return self._tables()[1]
    def GetScaleSigmaSquares(self): return self._tables()[2]
    def GetInverseScaleSigmaSquares(self): return self._tables()[3]
    @property
    def mnFeaturesPerLevel(self): return self._tables()[4]
    @property
    def umax(self): return self._tables()[5]

    def max_keypoints(self):
        return self._L.orbx_max_keypoints(self._h)

    # -- operator()
    def __call__(self, image):
        image = np.asarray(image)
        if image.size == 0:  # empty image: silent return (src/ORBextractor.cc:1046-1047)
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (:1050)"
        if image.strides[1] != 1 or image.strides[0] < image.shape[1]:
            image = np.ascontiguousarray(image)   # a row-strided view (ROI of a wider image) is passed as it is
        h, w = image.shape
        cap = max(self.nfeatures + 8 * self.nlevels, 64) + 260
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._ck(self._L.orbx_extract(self._h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        self._shape = (h, w)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """Batched many-frame mode on host arrays [B,h,w]: list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, h, w = images.shape
        cap = max(self.nfeatures + 8 * self.nlevels, 64) + 260
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[images[b].ctypes.data for b in range(B)])
        self._ck(self._L.orbx_extract_batch(self._h, C.cast(ptrs, C.c_void_p), B, w, h, w, _p(kps), _p(desc), cap, _p(n)))
        self._shape = (h, w)
        return [(kps[b, :n[b]].copy(), desc[b, :n[b]].copy()) for b in range(B)]

    def extract_batch_device(self, d_imgs, B, w, h, stride, image_stride, d_kps, d_desc, d_counts, cap, stream=0):
        """Device-resident batched mode; all d_* are raw device pointers (ints)."""
        self._ck(self._L.orbx_extract_batch_device(self._h, d_imgs, B, w, h, stride, image_stride, d_kps, d_desc,
                                                 d_counts, cap, stream))
        self._shape = (h, w)

    def prefetch_batch_device(self, d_imgs, B, w, h, stride, image_stride, side_stream=0):
        """Start the pyramid of the NEXT batch now (orbx_extract_batch_device_prefetch): the images must be complete in HBM."""
        self._ck(self._L.orbx_extract_batch_device_prefetch(self._h, d_imgs, B, w, h, stride, image_stride, side_stream))

    def side_stream(self):
        """hipStream_t of the handle's own side stream (orbx_side_stream)."""
        return self._L.orbx_side_stream(self._h)

    def stereo_frame(self, left, right, mbf, mb):
        """One stereo frame host to host in one call (orbx_stereo_frame): -> dict(kl, dl, kr, dr, uright, depth, nmatch)."""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == right.shape and left.ndim == 2
        hgt, w = left.shape
        cap = (self.max_keypoints() if self._shape == (hgt, w) else self.nfeatures + 3 * self.nlevels + 8 * 64) + 8
        kl, kr = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
        ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        nl, nr, nm = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.orbx_stereo_frame(self._h, _p(left), _p(right), w, hgt, w, float(mbf), float(mb), cap, _p(kl), _p(dl), C.byref(nl),
                                         _p(kr), _p(dr), C.byref(nr), _p(ur), _p(dp), C.byref(nm)))
        self._shape = (hgt, w)
        a, b = nl.value, nr.value
        return {"kl": kl[:a].copy(), "dl": dl[:a].copy(), "kr": kr[:b].copy(), "dr": dr[:b].copy(), "uright": ur[:a].copy(),
                "depth": dp[:a].copy(), "nmatch": nm.value}

    def stereo_frame_view(self, left, right, mbf, mb, shape=None, stride=None):
        """The latency form of stereo_frame (orbx_stereo_frame_view): no copy commands, results in the handle's pinned record.
        left / right: uint8 numpy arrays [h, w] (pageable: staged by the call), or objects with data_ptr() (torch tensors - pinned
        host or device memory), or raw addresses (then shape = (h, w)).  Returns dict(kl, dl, kr, dr, uright, depth, nmatch, view):
        numpy VIEWS of the pinned record (valid until the call after the next one on this handle) and the StereoView with the
        device pointers (d_kl, d_dl, d_uright, ...)."""
        def addr(a):
            if isinstance(a, int):
                return a, None
            if hasattr(a, "data_ptr"):
                return a.data_ptr(), tuple(a.shape)
            a = np.ascontiguousarray(a, np.uint8)
            return a.ctypes.data, a.shape, a
        la, ra = addr(left), addr(right)
        hgt, w = shape if shape is not None else la[1]
        v = StereoView()
        self._ck(self._L.orbx_stereo_frame_view(self._h, la[0], ra[0], w, hgt, w if stride is None else int(stride), float(mbf), float(mb), C.byref(v)))
        self._shape = (hgt, w)
        a, b = v.nl, v.nr
        view = lambda p, n, dt: np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(p), dt) if n > 0 else np.zeros(0, dt)
        return {"kl": view(v.kl, a, KP_DTYPE), "dl": view(v.dl, a * 32, np.uint8).reshape(a, 32), "kr": view(v.kr, b, KP_DTYPE),
                "dr": view(v.dr, b * 32, np.uint8).reshape(b, 32), "uright": view(v.uright, a, np.float32),
                "depth": view(v.depth, a, np.float32), "nmatch": v.nmatch, "view": v}

    def set_pyramid_buffers(self, n):
        """2 (default) or 3 pyramid buffers (orbx_set_pyramid_buffers)."""
        self._ck(self._L.orbx_set_pyramid_buffers(self._h, int(n)))

    def side_stream_for(self, main_stream):
        """The side stream, probed (and replaced if need be) so that it does not share a hardware queue with main_stream."""
        return self._L.orbx_side_stream_for(self._h, main_stream)

    def stream_wait_fast_stage(self, stream):
        """Order `stream` behind the FAST stage of the last extraction call (orbx_stream_wait_fast_stage)."""
        self._ck(self._L.orbx_stream_wait_fast_stage(self._h, stream))

    # -- mvImagePyramid (include/ORBextractor.h:85)
    def pyramid_level(self, level, b=0, padded=False):
        w, h = C.c_int(), C.c_int()
        self._ck(self._L.orbx_pyramid_host(self._h, b, level, int(padded), None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._ck(self._L.orbx_pyramid_host(self._h, b, level, int(padded), _p(out), w.value, C.byref(w), C.byref(h)))
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def pyramid_device(self, level, b=0):
        ptr, w, h, s = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.orbx_pyramid_device(self._h, b, level, C.byref(ptr), C.byref(w), C.byref(h), C.byref(s)))
        return ptr.value, w.value, h.value, s.value

    # -- test / profiling hooks
    def debug_level_points(self, level, stage, b=0):
        self._hooks()
        n = C.c_int()
        self._ck(self._L.orbx_debug_level_points(self._h, b, level, stage, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        if n.value:
            self._ck(self._L.orbx_debug_level_points(self._h, b, level, stage, _p(out), n.value, C.byref(n)))
        return out[:n.value]

    def fast_kernels(self, B):
        """names of the FAST kernel(s) a batch of B images of the planned size runs"""
        st, ce, ipl = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self._L.orbx_fast_kernels(self._h, int(B), C.byref(st), C.byref(ce), C.byref(ipl)))
        self.fast_images_per_launch = ipl.value
        return [n for n, f in (("k_fast_strips", st.value), ("k_fast_cells", ce.value)) if f]

    def octree_fallbacks(self, B=1):
        """[B, nlevels] int32: 1 where the last call's quad-tree of that (image, level) was redone by the exact form."""
        self._hooks()
        out = np.zeros((B, self.nlevels), np.int32)
        self._ck(self._L.orbx_debug_octree_fallbacks(self._h, _p(out), B * self.nlevels))
        return out

    def blurred_mask(self):
        """Levels of the last call that were blurred as a whole by k_blur_levels (bit l)."""
        m = C.c_uint()
        self._ck(self._L.orbx_debug_blurred_level(self._h, 0, 0, None, 0, C.byref(m)))
        return m.value

    def blurred_level(self, level, b=0):
        """Level `level` of image b after GaussianBlur(7x7, sigma 2) as k_blur_levels wrote it (levels in blurred_mask())."""
        self._hooks()
        w, h = C.c_int(), C.c_int()
        self._ck(self._L.orbx_pyramid_host(self._h, b, level, 0, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._ck(self._L.orbx_debug_blurred_level(self._h, b, level, _p(out), w.value, None))
        return out

    def debug_blur_patches(self, image):
        """Test hook: extract one image and also return the 37x37 blurred block around every keypoint [N,37,37]."""
        self._hooks()
        self._ck(self._L.orbx_debug_blur_patches(self._h, 1, None, 0))
        try:
            k, d = self(image)
            out = np.zeros((max(len(k), 1), 37, 37), np.uint8)
            if len(k):
                self._ck(self._L.orbx_debug_blur_patches(self._h, 1, _p(out), len(k)))
        finally:
            self._L.orbx_debug_blur_patches(self._h, 0, None, 0)
        return k, d, out[:len(k)]

    def set_profiling(self, mode=1):
        """0/False off, 1/True events at every stage boundary, 2 only around k_fast_cells (see orbx.h)."""
        self._ck(self._L.orbx_set_profiling(self._h, int(mode)))

    def stage_ms(self):
        """(average ms per call [pyramid, FAST, quad-tree, describe, total], calls averaged)"""
        ms = np.zeros(NUM_STAGES, np.float32)
        n = C.c_int(0)
        self._ck(self._L.orbx_get_stage_ms(self._h, _p(ms), C.byref(n)))
        return ms, n.value


def debug_features_in_area(kun, geom, x, y, r, min_level=-1, max_level=-1, device=0):
    """Test hook: Frame::GetFeaturesInArea as the matchers see it -> indices in the reference's order."""
    kun = np.ascontiguousarray(kun, KP_DTYPE)
    out = np.zeros(max(len(kun), 1), np.int32)
    n = C.c_int(0)
    _check(lib(True).orbm_debug_features_in_area(_p(kun), len(kun), C.byref(geom), float(x), float(y), float(r), int(min_level),
                                             int(max_level), _p(out), C.byref(n), int(device)), lib(True))
    return out[:n.value].copy()


def debug_sincosf(angles, device=0):
    """Test hook: the device's cosf / sinf restatement -> (sin, cos) float32 arrays."""
    a = np.ascontiguousarray(angles, np.float32)
    s, c = np.zeros_like(a), np.zeros_like(a)
    _check(lib(True).orbx_debug_sincosf(_p(a), len(a), _p(s), _p(c), int(device)), lib(True))
    return s, c


def pack_records_device(d_kps, d_desc, d_uright, d_depth, d_counts, B, cap, d_records, stream=0):
    """One record per frame for the result all-gather (raw device pointers; layout in orbx.h / batching.py)."""
    _check(lib().orbx_pack_records_device(d_kps, d_desc, d_uright, d_depth, d_counts, B, cap, d_records, stream))


def stereo_batch_device(ex_left, ex_right, B, left_slot0, right_slot0, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap,
                        mbf, mb, d_uright, d_depth, d_nmatch, stream=0, prev=False):
    """Device-resident Frame::ComputeStereoMatches for B frames (raw device pointers).  prev: on the pyramids of the
    extraction call before the last one (orbm_stereo_batch_device_prev)."""
    fn = lib().orbm_stereo_batch_device_prev if prev else lib().orbm_stereo_batch_device
    _check(fn(ex_left._h, ex_right._h, B, left_slot0, right_slot0, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, float(mbf), float(mb),
              d_uright, d_depth, d_nmatch, stream))


def compute_stereo_matches(ex_left, ex_right, kl, dl, kr, dr, mbf, mb):
    """Frame::ComputeStereoMatches (src/Frame.cc:481-655) -> (mvuRight, mvDepth, nmatches).

    ex_left / ex_right must have just extracted the left / right image (their pyramids are
    read on the device)."""
    L = lib()
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    ur = np.full(len(kl), -1, np.float32); dp = np.full(len(kl), -1, np.float32)
    n = C.c_int(0)
    _check(L.orbm_stereo(ex_left._h, ex_right._h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr),
                         float(mbf), float(mb), _p(ur), _p(dp), C.byref(n)))
    return ur, dp, n.value


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher for the hot routines (include/ORBmatcher.h:37-102)."""
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30  # src/ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.device = int(device)

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return lib().orbm_hamming(_p(a), _p(b))

    def SearchForInitialization(self, k1, d1, k2, d2, geom2, vbPrevMatched, windowSize=10):
        """(src/ORBmatcher.cc:405-520) -> (nmatches, vnMatches12, vbPrevMatched')"""
        k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.full(len(k1), -1, np.int32)
        n = C.c_int(0)
        _check(lib().orbm_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), C.byref(geom2),
                                                    _p(prev), _p(m12), int(windowSize), self.mfNNratio,
                                                    int(self.mbCheckOrientation), self.device, C.byref(n)))
        return n.value, m12, prev

    def SearchByProjection(self, kun, desc, uright, geom, scale_factors, mps, mp_desc, frame_mp, ext_obs=None, th=1.0):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:45-129)
        -> (nmatches, frame_mp')"""
        kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        uright = np.ascontiguousarray(uright, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
        mps = np.ascontiguousarray(mps, MP_DTYPE); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        fm = np.ascontiguousarray(frame_mp, np.int32).copy()
        eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
        n = C.c_int(0)
        _check(lib().orbm_search_by_projection_mp(_p(kun), _p(desc), _p(uright), len(kun), C.byref(geom), _p(sf),
                                                  len(sf), _p(mps), _p(mp_desc), len(mps), _p(fm), _p(eo), float(th),
                                                  self.mfNNratio, self.device, C.byref(n)))
        return n.value, fm

    def SearchByProjectionFrame(self, kun, desc, uright, geom, scale_factors, cam, Tcw_cur, Tcw_last, last, last_desc,
                                cur_mp, ext_obs=None, th=7.0, bMono=False):
        """SearchByProjection(Frame &cur, const Frame &last, th, bMono) (src/ORBmatcher.cc:1330-1472)
        -> (nmatches, cur_mp')"""
        kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        uright = np.ascontiguousarray(uright, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
        last = np.ascontiguousarray(last, LASTPT_DTYPE); last_desc = np.ascontiguousarray(last_desc, np.uint8)
        Tc = np.ascontiguousarray(Tcw_cur, np.float32); Tl = np.ascontiguousarray(Tcw_last, np.float32)
        cm = np.ascontiguousarray(cur_mp, np.int32).copy()
        eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
        n = C.c_int(0)
        _check(lib().orbm_search_by_projection_frame(_p(kun), _p(desc), _p(uright), len(kun), C.byref(geom), _p(sf),
                                                     len(sf), C.byref(cam), _p(Tc), _p(Tl), _p(last), _p(last_desc),
                                                     len(last), _p(cm), _p(eo), float(th), int(bMono),
                                                     int(self.mbCheckOrientation), self.device, C.byref(n)))
        return n.value, cm


def search_by_projection_frame_device(d_kun, d_desc, d_uright, n, geom, scale_factors, cam, Tcw_cur, Tcw_last, last, d_last_desc,
                                      cur_mp, ext_obs=None, th=7.0, bMono=False, check_orientation=True, device=0, stream=0):
    """orbm_search_by_projection_frame_device: d_* are raw device pointers (the extractor's / stereo matcher's outputs in HBM)
    -> (nmatches, cur_mp')"""
    sf = np.ascontiguousarray(scale_factors, np.float32)
    last = np.ascontiguousarray(last, LASTPT_DTYPE)
    Tc = np.ascontiguousarray(Tcw_cur, np.float32); Tl = np.ascontiguousarray(Tcw_last, np.float32)
    cm = np.ascontiguousarray(cur_mp, np.int32).copy()
    eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
    nm = C.c_int(0)
    _check(lib().orbm_search_by_projection_frame_device(d_kun, d_desc, d_uright, int(n), C.byref(geom), _p(sf), len(sf), C.byref(cam),
                                                        _p(Tc), _p(Tl), _p(last), d_last_desc, len(last), _p(cm), _p(eo), float(th),
                                                        int(bMono), int(check_orientation), int(device), C.byref(nm), stream))
    return nm.value, cm


def search_local_points_device(d_kun, d_desc, d_uright, n, geom, sf, pts, mp_desc, Tcw, cam, viewing_cos_limit, thresholds, frame_mp,
                               ext_obs, th, nnratio, device=0, stream=0):
    """orbm_search_local_points_device -> (nmatches, frame_mp', projections)"""
    sf = np.ascontiguousarray(sf, np.float32)
    pts = np.ascontiguousarray(pts, WORLDPOINT_DTYPE); md = np.ascontiguousarray(mp_desc, np.uint8)
    T = np.ascontiguousarray(Tcw, np.float32); thr = np.ascontiguousarray(thresholds, np.float32)
    fm = np.ascontiguousarray(frame_mp, np.int32).copy()
    eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
    proj = np.zeros(len(pts), MP_DTYPE)
    nm = C.c_int(0)
    _check(lib().orbm_search_local_points_device(d_kun, d_desc, d_uright, int(n), C.byref(geom), _p(sf), len(sf), _p(pts), _p(md),
                                                 len(pts), _p(T), C.byref(cam), float(viewing_cos_limit), _p(thr), _p(fm), _p(eo),
                                                 float(th), float(nnratio), int(device), C.byref(nm), _p(proj), stream))
    return nm.value, fm, proj


def match_windows(kun, desc, uright, geom, queries, query_desc, holder, ext_blocks=None, max_dist=100,
                  check_orientation=True, device=0, geom_assign=None):
    """orbm_match_windows: the projected-window matcher behind the SearchByProjection family
    -> (nmatches, holder')"""
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    q = np.ascontiguousarray(queries, WINDOW_DTYPE); qd = np.ascontiguousarray(query_desc, np.uint8)
    h = np.ascontiguousarray(holder, np.int32).copy()
    eb = None if ext_blocks is None else np.ascontiguousarray(ext_blocks, np.int32)
    n = C.c_int(0)
    _check(lib().orbm_match_windows(_p(kun), _p(desc), _p(ur), len(kun), C.byref(geom),
                                    None if geom_assign is None else C.byref(geom_assign), _p(q), _p(qd), len(q), _p(h),
                                    _p(eb), int(max_dist), int(check_orientation), int(device), C.byref(n)))
    return n.value, h


def best_in_windows(kun, desc, uright, geom, queries, query_desc, inv_level_sigma2=None, device=0, geom_assign=None):
    """orbm_best_in_windows: stateless window search behind Fuse / SearchBySim3 -> (best_idx, best_dist)"""
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    q = np.ascontiguousarray(queries, WINDOW_DTYPE); qd = np.ascontiguousarray(query_desc, np.uint8)
    s2 = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    bi = np.full(len(q), -1, np.int32); bd = np.full(len(q), 256, np.int32)
    _check(lib().orbm_best_in_windows(_p(kun), _p(desc), _p(ur), len(kun), C.byref(geom),
                                      None if geom_assign is None else C.byref(geom_assign), _p(q), _p(qd), len(q), _p(s2),
                                      0 if s2 is None else len(s2), _p(bi), _p(bd), int(device)))
    return bi, bd


def distinctive_descriptors(desc, offsets, device=0):
    """orbm_distinctive_descriptors: MapPoint::ComputeDistinctiveDescriptors batched -> (best_row, best_median)"""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(offsets, np.int32)
    m = len(off) - 1
    br = np.zeros(m, np.int32); bm = np.zeros(m, np.int32)
    _check(lib().orbm_distinctive_descriptors(_p(desc), _p(off), m, _p(br), _p(bm), int(device)))
    return br, bm


WORLDPOINT_DTYPE = np.dtype([("valid", "<i4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"), ("nx", "<f4"), ("ny", "<f4"),
                             ("nz", "<f4"), ("max_distance", "<f4"), ("min_distance", "<f4"), ("observations", "<i4")])


def predict_scale_thresholds(log_scale_factor, nlevels):
    """orbm_predict_scale_thresholds (host code, no GPU): MapPoint::PredictScale as nlevels-1 float thresholds"""
    t = np.zeros(max(nlevels - 1, 1), np.float32)
    _check(lib().orbm_predict_scale_thresholds(float(log_scale_factor), int(nlevels), _p(t)))
    return t[:nlevels - 1]


def is_in_frustum(pts, Tcw, cam, geom, viewing_cos_limit, thresholds, nlevels, device=0):
    """orbm_is_in_frustum: Frame::isInFrustum for a list of map points -> MP_DTYPE records"""
    pts = np.ascontiguousarray(pts, WORLDPOINT_DTYPE); T = np.ascontiguousarray(Tcw, np.float32)
    thr = np.ascontiguousarray(thresholds, np.float32)
    out = np.zeros(len(pts), MP_DTYPE)
    _check(lib().orbm_is_in_frustum(_p(pts), len(pts), _p(T), C.byref(cam), C.byref(geom), float(viewing_cos_limit), _p(thr),
                                    int(nlevels), _p(out), int(device)))
    return out


def search_local_points(kun, desc, uright, geom, sf, pts, mp_desc, Tcw, cam, viewing_cos_limit, thresholds, frame_mp,
                        ext_obs, th, nnratio, device=0):
    """orbm_search_local_points: isInFrustum + SearchByProjection(F, MPs) -> (nmatches, frame_mp', projections)"""
    kun = np.ascontiguousarray(kun, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    ur = np.ascontiguousarray(uright, np.float32); sf = np.ascontiguousarray(sf, np.float32)
    pts = np.ascontiguousarray(pts, WORLDPOINT_DTYPE); md = np.ascontiguousarray(mp_desc, np.uint8)
    T = np.ascontiguousarray(Tcw, np.float32); thr = np.ascontiguousarray(thresholds, np.float32)
    fm = np.ascontiguousarray(frame_mp, np.int32).copy()
    eo = None if ext_obs is None else np.ascontiguousarray(ext_obs, np.int32)
    proj = np.zeros(len(pts), MP_DTYPE)
    n = C.c_int(0)
    _check(lib().orbm_search_local_points(_p(kun), _p(desc), _p(ur), len(kun), C.byref(geom), _p(sf), len(sf), _p(pts), _p(md),
                                          len(pts), _p(T), C.byref(cam), float(viewing_cos_limit), _p(thr), _p(fm), _p(eo),
                                          float(th), float(nnratio), int(device), C.byref(n), _p(proj)))
    return n.value, fm, proj


class Vocabulary:
    """DBoW2 vocabulary on the GPU (orbv_*): from per-node arrays in file order, or from an ORBvoc.txt file."""

    def _ck(self, rc):
        _check(rc, self._L)

    def __init__(self, k=None, L=None, scoring=0, weighting=0, parent=None, is_leaf=None, desc=None, weight=None, path=None,
                 device=0):
        self._L = lib()
        h = C.c_void_p()
        if path is not None:
            self._ck(self._L.orbv_load_text(str(path).encode(), int(device), C.byref(h)))
        else:
            parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); weight = np.ascontiguousarray(weight, np.float64)
            self._ck(self._L.orbv_create(int(k), int(L), int(scoring), int(weighting), len(parent), _p(parent), _p(is_leaf), _p(desc),
                                       _p(weight), int(device), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orbv_destroy(self._h)
            self._h = None

    def info(self):
        v = [C.c_int(0) for _ in range(6)]
        self._ck(self._L.orbv_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("k", "L", "scoring", "weighting", "nnodes", "nwords"), [x.value for x in v]))

    def transform(self, desc, levelsup=4):
        """orbv_transform -> (word_id, node_id, weight) per descriptor"""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        w = np.zeros(n, np.int32); nid = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
        self._ck(self._L.orbv_transform(self._h, _p(d), n, int(levelsup), _p(w), _p(nid), _p(wt)))
        return w, nid, wt


def search_by_bow(q_desc, q_angle, q_valid, c_desc, c_angle, c_valid, node_qstart, q_items, node_cstart, c_items, max_dist,
                  nnratio, check_orientation=True, device=0):
    """orbm_search_by_bow -> (nmatches, match_q)"""
    qd = np.ascontiguousarray(q_desc, np.uint8); qa = np.ascontiguousarray(q_angle, np.float32)
    qv = np.ascontiguousarray(q_valid, np.uint8)
    cd = np.ascontiguousarray(c_desc, np.uint8); ca = np.ascontiguousarray(c_angle, np.float32)
    cv = None if c_valid is None else np.ascontiguousarray(c_valid, np.uint8)
    nqs = np.ascontiguousarray(node_qstart, np.int32); qi = np.ascontiguousarray(q_items, np.int32)
    ncs = np.ascontiguousarray(node_cstart, np.int32); ci = np.ascontiguousarray(c_items, np.int32)
    mq = np.zeros(len(qa), np.int32)
    n = C.c_int(0)
    _check(lib().orbm_search_by_bow(_p(qd), _p(qa), _p(qv), len(qa), _p(cd), _p(ca), _p(cv), len(ca), _p(nqs), _p(qi), _p(ncs),
                                    _p(ci), len(nqs) - 1, int(max_dist), float(nnratio), int(check_orientation), _p(mq),
                                    C.byref(n), int(device)))
    return n.value, mq


def search_for_triangulation(kp1, q_desc, q_flags, kp2, c_desc, c_flags, node_qstart, q_items, node_cstart, c_items, F12, ex, ey,
                             scale_factors, level_sigma2, max_dist=50, check_orientation=True, device=0):
    """orbm_search_for_triangulation -> (nmatches, match_q)"""
    k1 = np.ascontiguousarray(kp1, KP_DTYPE); k2 = np.ascontiguousarray(kp2, KP_DTYPE)
    qd = np.ascontiguousarray(q_desc, np.uint8); cd = np.ascontiguousarray(c_desc, np.uint8)
    qf = np.ascontiguousarray(q_flags, np.uint8); cf = np.ascontiguousarray(c_flags, np.uint8)
    nqs = np.ascontiguousarray(node_qstart, np.int32); qi = np.ascontiguousarray(q_items, np.int32)
    ncs = np.ascontiguousarray(node_cstart, np.int32); ci = np.ascontiguousarray(c_items, np.int32)
    F = np.ascontiguousarray(F12, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    s2 = np.ascontiguousarray(level_sigma2, np.float32)
    mq = np.zeros(len(k1), np.int32)
    n = C.c_int(0)
    _check(lib().orbm_search_for_triangulation(_p(k1), _p(qd), _p(qf), len(k1), _p(k2), _p(cd), _p(cf), len(k2), _p(nqs), _p(qi),
                                               _p(ncs), _p(ci), len(nqs) - 1, _p(F), float(ex), float(ey), _p(sf), _p(s2), len(sf),
                                               int(max_dist), int(check_orientation), _p(mq), C.byref(n), int(device)))
    return n.value, mq
