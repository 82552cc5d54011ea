"""ctypes front door to the CPU oracle (oracle/orb_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and the bench's
cpu_baseline leg; never by the product package.  Parity unpinned (see the C++ header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "camera_oracle.cpp", "camera_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off -fno-fast-math"
PORTABLE_FLAGS = "-O2 -ffp-contract=off -fno-fast-math"
build_flags = PORTABLE_FLAGS


def use_build(kind: str) -> str:
    """Switch this module to another build of the same sources: "native" (-O3 -march=native, built on this machine:
    what the bench's CPU baseline times), "asan" (AddressSanitizer + UBSan; the process must have the sanitizer
    runtime preloaded) or "portable" (the default).  Returns the flags of the build now in use."""
    global _SO, _lib, build_flags
    if kind == "portable":
        so, flags = os.path.join(_HERE, "_build", "liborb_oracle.so"), PORTABLE_FLAGS
        build()
    else:
        so = os.path.join(_HERE, "_build", f"liborb_oracle_{kind}.so")
        flags = NATIVE_FLAGS if kind == "native" else "-O1 -fsanitize=address,undefined"
        subprocess.check_call(["make", "-C", _HERE, "-s", kind])
    _SO, _lib, build_flags = so, None, flags
    return flags


_lib = None


def lib():
    global _lib
    if _lib is None:
        kind = os.environ.get("ORB_ORACLE_BUILD", "")          # tests/test_sanitizers.py: run the tests on the asan build
        if kind in ("native", "asan") and not _SO.endswith(f"_{kind}.so"):
            use_build(kind)
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_pyramid_level.restype = C.c_int
        L.orc_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_level_candidates.restype = C.c_int
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_level_keypoints.restype = C.c_int
        L.orc_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_gaussian7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.orc_resize_linear_cn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_remap_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p]
        L.orc_fast.restype = C.c_int
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_strength.restype = C.c_int
        L.orc_fast_strength.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_sincos_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sincos_vs_libm.restype = C.c_long
        L.orc_sincos_vs_libm.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.orc_distribute.restype = C.c_int
        L.orc_distribute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int]
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int]
        L.orc_stereo_match.restype = C.c_int
        L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.orc_grid_build.restype = C.c_void_p
        L.orc_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_grid_destroy.argtypes = [C.c_void_p]
        L.orc_grid_csr.restype = C.c_int
        L.orc_grid_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_features_in_area.restype = C.c_int
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                           C.c_int, C.c_void_p, C.c_int]
        L.orc_search_window_best.restype = C.c_int
        L.orc_search_window_best.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float] + [C.c_void_p] * 3 + [C.c_float, C.c_void_p]
        L.orc_fuse_search.restype = C.c_int
        L.orc_fuse_search.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_search_by_sim3.restype = C.c_int
        L.orc_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + ([C.c_void_p] * 6 + [C.c_int]) * 2 + [C.c_void_p]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_bow.argtypes = [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_bow_keyframes.restype = C.c_int
        L.orc_search_by_bow_keyframes.argtypes = ([C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]) * 2 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_for_triangulation.restype = C.c_int
        L.orc_search_for_triangulation.argtypes = ([C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p]) * 2 + [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_bow_transform.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3
        L.orc_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection.restype = C.c_int
        L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_bucket_best.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_stereo.restype = C.c_int
        L.orc_search_by_projection_stereo.argtypes = ([C.c_void_p] * 3 + [C.c_int]) * 2 + [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p]
        L.orc_search_window_best_stereo.restype = C.c_int
        L.orc_search_window_best_stereo.argtypes = ([C.c_void_p] * 3 + [C.c_int]) * 2 + [C.c_void_p] * 11 + [C.c_int, C.c_float] + [C.c_void_p] * 3 + [C.c_float, C.c_void_p]
        L.orc_search_for_triangulation_cameras.restype = C.c_int
        L.orc_search_for_triangulation_cameras.argtypes = ([C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]) * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_float] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
        L.orc_kb8_triangulate_matches.restype = C.c_float
        L.orc_kb8_triangulate_matches.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.orc_stereo_fisheye_matches.restype = C.c_int
        L.orc_stereo_fisheye_matches.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_int]) * 2 + [C.c_void_p] * 10
        L.orc_triangulate_matches.restype = C.c_double
        L.orc_triangulate_matches.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_epipolar_constrain_margin.restype = C.c_int
        L.orc_epipolar_constrain_margin.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_camera_project.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_camera_unproject.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_epipolar_constrain.restype = C.c_int
        L.orc_epipolar_constrain.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.orc_camera_gates_many.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Extractor:
    """Mirror of ORB_SLAM3::ORBextractor (ORBextractor.h:42-121) on the CPU oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        n = nlevels
        self.scale_factors = np.zeros(n, np.float32)
        self.inv_scale_factors = np.zeros(n, np.float32)
        self.level_sigma2 = np.zeros(n, np.float32)
        self.inv_level_sigma2 = np.zeros(n, np.float32)
        self.features_per_level = np.zeros(n, np.int32)
        self.umax = np.zeros(16, np.int32)
        self.L.orc_tables(self.h, _p(self.scale_factors), _p(self.inv_scale_factors), _p(self.level_sigma2),
                          _p(self.inv_level_sigma2), _p(self.features_per_level), _p(self.umax))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self.L.orc_level_size(self.h, w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def __call__(self, image: np.ndarray, lapping=(0, 0), dynamic_rects=None):
        """Returns (keypoints[KP_DTYPE], descriptors[n,32] u8, monoIndex)."""
        assert image.dtype == np.uint8 and image.ndim == 2
        h, w = image.shape
        cap = self.nfeatures + 16 * self.nlevels + 64
        n = C.c_int()
        rects = None
        nr = 0
        if dynamic_rects is not None and len(dynamic_rects):
            rects = np.ascontiguousarray(dynamic_rects, np.int32).reshape(-1, 4)
            nr = len(rects)
        while True:
            kps = np.zeros(cap, KP_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            mono = self.L.orc_extract(self.h, image.ctypes.data, w, h, image.strides[0], int(lapping[0]),
                                      int(lapping[1]), _p(rects), nr, _p(kps), _p(desc), cap, C.byref(n))
            if n.value <= cap:
                break
            # a level whose quota is below its number of quadtree roots keeps one keypoint per non-empty root
            # (ORBextractor.cc:665-700): far more than nfeatures for a tiny quota on a wide image
            cap = n.value
        return kps[:n.value].copy(), desc[:n.value].copy(), mono

    def pyramid_level(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.orc_pyramid_level(self.h, level, None, 1 << 30, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.orc_pyramid_level(self.h, level, _p(out), out.size, C.byref(w), C.byref(h))
        return out

    def level_candidates(self, level):
        n = self.L.orc_level_candidates(self.h, level, None, 0)
        out = np.zeros((n, 3), np.float32)
        self.L.orc_level_candidates(self.h, level, _p(out), n)
        return out

    def level_keypoints(self, level):
        n = self.L.orc_level_keypoints(self.h, level, None, 0)
        out = np.zeros(n, KP_DTYPE)
        self.L.orc_level_keypoints(self.h, level, _p(out), n)
        return out


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def _hwc(img):
    img = np.ascontiguousarray(img, np.uint8)
    return img.reshape(img.shape[0], img.shape[1], -1)


def cvt_gray(img: np.ndarray, rgb: bool) -> np.ndarray:
    """cv::cvtColor(img, COLOR_{RGB|BGR}[A]2GRAY) for an H x W x {3,4} uint8 image."""
    img = _hwc(img)
    h, w, cn = img.shape
    dst = np.zeros((h, w), np.uint8)
    lib().orc_cvt_gray(_p(img), w, h, w * cn, cn, int(rgb), _p(dst))
    return dst


def resize_linear_cn(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize(img, Size(dw, dh)) for H x W [x C] uint8."""
    src = _hwc(img)
    h, w, cn = src.shape
    dst = np.zeros((dh, dw, cn), np.uint8)
    lib().orc_resize_linear_cn(_p(src), w, h, w * cn, cn, _p(dst), dw, dh)
    return dst if np.ndim(img) == 3 else dst[:, :, 0]


def remap_linear(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv::remap(img, map_x, map_y, INTER_LINEAR) for H x W [x C] uint8 and float32 maps."""
    src = _hwc(img)
    h, w, cn = src.shape
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    dh, dw = mx.shape
    dst = np.zeros((dh, dw, cn), np.uint8)
    lib().orc_remap_linear(_p(src), w, h, w * cn, cn, _p(mx), _p(my), dw, dh, _p(dst))
    return dst if np.ndim(img) == 3 else dst[:, :, 0]


def stage(img: np.ndarray, rgb=False, maps=None, size=None) -> np.ndarray:
    """The image ORBextractor receives: remap OR resize of the delivered image, then grey conversion."""
    out = img
    if maps is not None:
        out = remap_linear(out, maps[0], maps[1])
    elif size is not None:
        out = resize_linear_cn(out, size[0], size[1])
    if np.ndim(out) == 3 and out.shape[2] >= 3:
        out = cvt_gray(out, rgb)
    elif np.ndim(out) == 3:
        out = out[:, :, 0]
    return np.ascontiguousarray(out)


def undistort_points(xy, K, D):
    """cv::undistortPoints(xy, xy, K, D, Mat(), K) for float32 [n, 2] points; K = (fx, fy, cx, cy)."""
    xy = np.ascontiguousarray(xy, np.float32)
    K = np.ascontiguousarray(K, np.float32)
    D = np.ascontiguousarray(D, np.float32)
    out = np.zeros_like(xy)
    lib().orc_undistort_points(_p(xy), len(xy), _p(K), _p(D), len(D), _p(out))
    return out


def image_bounds(w, h, K, D):
    """Frame::ComputeImageBounds: (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    K = np.ascontiguousarray(K, np.float32)
    D = np.ascontiguousarray(D, np.float32)
    b = np.zeros(4, np.float32)
    lib().orc_image_bounds(w, h, _p(K), _p(D), len(D), _p(b))
    return b


def gaussian7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    lib().orc_gaussian7(_p(src), src.shape[1], src.shape[0], _p(dst))
    return dst


def fast(img: np.ndarray, threshold: int) -> np.ndarray:
    """cv::FAST(img, ..., threshold, true): rows of (x, y, score) in emission order."""
    cap = img.size // 4 + 16
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], threshold, _p(out), cap)
    return out[:n].copy()


def fast_strength(img: np.ndarray, x: int, y: int) -> int:
    return lib().orc_fast_strength(img.ctypes.data, img.strides[0], x, y)


def fast_atan2(y: float, x: float) -> float:
    return float(lib().orc_fast_atan2(y, x))


def sincos(x: np.ndarray):
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    lib().orc_sincos_many(_p(x), x.size, _p(s), _p(c))
    return s, c


def sincos_vs_libm(lo: float, hi: float, stride: int):
    n = C.c_long()
    bad = lib().orc_sincos_vs_libm(lo, hi, stride, C.byref(n))
    return bad, n.value


def distribute(xyr: np.ndarray, minX, maxX, minY, maxY, N) -> np.ndarray:
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    cap = max(4 * N + 64, 256)
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_distribute(_p(xyr), len(xyr), minX, maxX, minY, maxY, N, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def knn2(q: np.ndarray, t: np.ndarray, fast_popcount: bool = True) -> np.ndarray:
    """Rows of (idx0, dist0, idx1, dist1); missing = (-1, INT_MAX)."""
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros((len(q), 4), np.int32)
    lib().orc_knn2(_p(q), len(q), _p(t), len(t), _p(out), 1 if fast_popcount else 0)
    return out


def stereo_match(kpsL, descL, kpsR, descR, pyrL, pyrR, sf, inv_sf, mbf, mb):
    """Frame::ComputeStereoMatches.  pyrL/pyrR: lists of contiguous level images.
    Returns (uRight, depth, bestDist, bestIdxR)."""
    nl = len(pyrL)
    PL = (C.c_void_p * nl)(*[p.ctypes.data for p in pyrL])
    PR = (C.c_void_p * nl)(*[p.ctypes.data for p in pyrR])
    lw = np.array([p.shape[1] for p in pyrL], np.int32)
    lh = np.array([p.shape[0] for p in pyrL], np.int32)
    for p in list(pyrL) + list(pyrR):
        assert p.flags.c_contiguous
    n = len(kpsL)
    uR = np.zeros(n, np.float32)
    depth = np.zeros(n, np.float32)
    bd = np.zeros(n, np.int32)
    bi = np.zeros(n, np.int32)
    kpsL = np.ascontiguousarray(kpsL)
    kpsR = np.ascontiguousarray(kpsR)
    descL = np.ascontiguousarray(descL)
    descR = np.ascontiguousarray(descR)
    sf = np.ascontiguousarray(sf, np.float32)
    inv_sf = np.ascontiguousarray(inv_sf, np.float32)
    lib().orc_stereo_match(_p(kpsL), _p(descL), n, _p(kpsR), _p(descR), len(kpsR), PL, PR, _p(lw), _p(lh),
                           _p(sf), _p(inv_sf), nl, mbf, mb, _p(uR), _p(depth), _p(bd), _p(bi))
    return uR, depth, bd, bi


class Grid:
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea (64 x 48 cells)."""

    def __init__(self, kps, minX, maxX, minY, maxY):
        self.kps = np.ascontiguousarray(kps)
        self.g = lib().orc_grid_build(_p(self.kps), len(self.kps), minX, maxX, minY, maxY)

    def __del__(self):
        try:
            lib().orc_grid_destroy(self.g)
        except Exception:
            pass

    def csr(self):
        """(cell_start[3073], cell_idx[]) with cell = column * 48 + row."""
        start = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(max(len(self.kps), 1), np.int32)
        n = lib().orc_grid_csr(self.g, _p(start), _p(idx), len(idx))
        return start, idx[:n].copy()

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(len(self.kps), np.int32)
        n = lib().orc_features_in_area(self.g, _p(self.kps), x, y, r, min_level, max_level, _p(out), len(out))
        return out[:n].copy()


def search_by_projection(grid: Grid, fdesc, uRight, qx, qy, qxr, qlevel, qradius, qdesc, taken, nnratio):
    fdesc = np.ascontiguousarray(fdesc, np.uint8)
    qx = np.ascontiguousarray(qx, np.float32)
    qy = np.ascontiguousarray(qy, np.float32)
    qxr = np.ascontiguousarray(qxr, np.float32)
    qlevel = np.ascontiguousarray(qlevel, np.int32)
    qradius = np.ascontiguousarray(qradius, np.float32)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
    taken = np.ascontiguousarray(taken, np.uint8).copy()
    out = np.zeros(len(qx), np.int32)
    n = lib().orc_search_by_projection(grid.g, _p(grid.kps), _p(fdesc), _p(uR), len(grid.kps), _p(qx), _p(qy),
                                       _p(qxr), _p(qlevel), _p(qradius), _p(qdesc), len(qx), _p(taken),
                                       nnratio, _p(out))
    return out, taken, n


def bucket_best(d1, idx1, off1, d2, idx2, off2):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    off1 = np.ascontiguousarray(off1, np.int32)
    off2 = np.ascontiguousarray(off2, np.int32)
    bi = np.zeros(len(idx1), np.int32)
    bd = np.zeros(len(idx1), np.int32)
    lib().orc_bucket_best(_p(d1), _p(idx1), _p(off1), _p(d2), _p(idx2), _p(off2), len(off1) - 1, _p(bi), _p(bd))
    return bi, bd


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, np.uint8)


def search_window_best(grid: Grid, fdesc, uRight, qx, qy, qxr, qradius, qmin, qmax, qdesc, max_dist,
                       taken=None, marks=None, qangle=None, factor=1.0 / 30):
    """Best-only projection searches; returns (kp_match, taken_after, nmatches)."""
    fdesc, qdesc = _u8(fdesc), _u8(qdesc)
    qx, qy, qxr, qradius = _f32(qx), _f32(qy), _f32(qxr), _f32(qradius)
    qmin, qmax = _i32(qmin), _i32(qmax)
    uR = None if uRight is None else _f32(uRight)
    tk = None if taken is None else _u8(taken).copy()
    mk = _u8(marks)
    qa = None if qangle is None else _f32(qangle)
    kpm = np.zeros(len(grid.kps), np.int32)
    n = lib().orc_search_window_best(grid.g, _p(grid.kps), _p(fdesc), _p(uR), len(grid.kps), _p(qx), _p(qy), _p(qxr),
                                     _p(qradius), _p(qmin), _p(qmax), _p(qdesc), len(qx), max_dist, _p(tk), _p(mk),
                                     _p(qa), factor, _p(kpm))
    return kpm, tk, n


def fuse_search(grid: Grid, fdesc, uRight, qu, qv, qur, qradius, qlevel, qdesc, inv_level_sigma2):
    fdesc, qdesc = _u8(fdesc), _u8(qdesc)
    qu, qv, qur, qradius = _f32(qu), _f32(qv), _f32(qur), _f32(qradius)
    qlevel = _i32(qlevel)
    uR = None if uRight is None else _f32(uRight)
    sig = _f32(inv_level_sigma2)
    out = np.zeros(len(qu), np.int32)
    n = lib().orc_fuse_search(grid.g, _p(grid.kps), _p(fdesc), _p(uR), len(grid.kps), _p(qu), _p(qv), _p(qur),
                              _p(qradius), _p(qlevel), _p(qdesc), len(qu), _p(sig), _p(out))
    return out, n


def search_by_sim3(grid1: Grid, desc1, grid2: Grid, desc2, into2, into1):
    """into2 / into1 = (u, v, radius, level, desc, source) of the points of KF1 searched in KF2 and vice versa."""
    desc1, desc2 = _u8(desc1), _u8(desc2)

    def pack(t):
        u, v, r, l, d, s = t
        return [_f32(u), _f32(v), _f32(r), _i32(l), _u8(d), _i32(s)]
    a, b = pack(into2), pack(into1)
    out = np.zeros(len(grid1.kps), np.int32)
    n = lib().orc_search_by_sim3(grid1.g, _p(grid1.kps), _p(desc1), len(grid1.kps), grid2.g, _p(grid2.kps), _p(desc2),
                                 len(grid2.kps), *[_p(x) for x in a], len(a[0]), *[_p(x) for x in b], len(b[0]), _p(out))
    return out, n


def search_for_initialization(kps1, desc1, grid2: Grid, desc2, prev, window, nnratio, check_ori):
    kps1 = np.ascontiguousarray(kps1)
    desc1, desc2 = _u8(desc1), _u8(desc2)
    prev = _f32(prev).copy()
    out = np.zeros(len(kps1), np.int32)
    n = lib().orc_search_for_initialization(_p(kps1), _p(desc1), len(kps1), grid2.g, _p(grid2.kps), _p(desc2),
                                            len(grid2.kps), _p(prev), window, nnratio, int(check_ori), _p(out))
    return out, prev, n


def search_by_bow(descKF, angleKF, validKF, idxKF, offKF, descF, angleF, n_left, idxF, offF, nnratio, check_ori):
    descKF, descF = _u8(descKF), _u8(descF)
    out = np.zeros(len(descF), np.int32)
    a = [_f32(angleKF), _u8(validKF), _i32(idxKF), _i32(offKF)]
    b = [_f32(angleF), _i32(idxF), _i32(offF)]
    n = lib().orc_search_by_bow(_p(descKF), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(descF), _p(b[0]), len(descF),
                                n_left, _p(b[1]), _p(b[2]), len(a[3]) - 1, nnratio, int(check_ori), _p(out))
    return out, n


def search_by_bow_keyframes(desc1, angle1, valid1, idx1, off1, desc2, angle2, valid2, idx2, off2, nnratio, check_ori):
    desc1, desc2 = _u8(desc1), _u8(desc2)
    a = [_f32(angle1), _u8(valid1), _i32(idx1), _i32(off1)]
    b = [_f32(angle2), _u8(valid2), _i32(idx2), _i32(off2)]
    out = np.zeros(len(desc1), np.int32)
    n = lib().orc_search_by_bow_keyframes(_p(desc1), _p(a[0]), _p(a[1]), len(desc1), _p(a[2]), _p(a[3]),
                                          _p(desc2), _p(b[0]), _p(b[1]), len(desc2), _p(b[2]), _p(b[3]),
                                          len(a[3]) - 1, nnratio, int(check_ori), _p(out))
    return out, n


def search_for_triangulation(kps1, desc1, free1, ur1, idx1, off1, kps2, desc2, free2, ur2, idx2, off2,
                             F12, ep, scale_factors2, level_sigma2_2, only_stereo, coarse, check_ori):
    kps1, kps2 = np.ascontiguousarray(kps1), np.ascontiguousarray(kps2)
    desc1, desc2 = _u8(desc1), _u8(desc2)
    a = [_u8(free1), None if ur1 is None else _f32(ur1), _i32(idx1), _i32(off1)]
    b = [_u8(free2), None if ur2 is None else _f32(ur2), _i32(idx2), _i32(off2)]
    F = _f32(F12).reshape(9)
    sf, sg = _f32(scale_factors2), _f32(level_sigma2_2)
    out = np.zeros(len(kps1), np.int32)
    n = lib().orc_search_for_triangulation(_p(kps1), _p(desc1), _p(a[0]), _p(a[1]), len(kps1), _p(a[2]), _p(a[3]),
                                           _p(kps2), _p(desc2), _p(b[0]), _p(b[1]), len(kps2), _p(b[2]), _p(b[3]),
                                           len(a[3]) - 1, _p(F), float(ep[0]), float(ep[1]), _p(sf), _p(sg),
                                           int(only_stereo), int(coarse), int(check_ori), _p(out))
    return out, n


def bow_transform(child_off, child_idx, node_desc, word_id, weight, depth, desc, levelsup):
    """Per descriptor (word id, weight, feature-vector node) of DBoW2's tree descent."""
    co, ci = _i32(child_off), _i32(child_idx)
    nd, wi = _u8(node_desc), _i32(word_id)
    w = np.ascontiguousarray(weight, np.float64)
    desc = _u8(desc)
    n = len(desc)
    wo = np.zeros(n, np.int32)
    we = np.zeros(n, np.float64)
    ni = np.zeros(n, np.int32)
    lib().orc_bow_transform(_p(co), _p(ci), _p(nd), _p(wi), _p(w), depth, _p(desc), n, levelsup, _p(wo), _p(we), _p(ni))
    return wo, we, ni


def distinctive_descriptors(desc, off):
    desc, off = _u8(desc), _i32(off)
    n = len(off) - 1
    bi = np.zeros(n, np.int32)
    bm = np.zeros(n, np.int32)
    lib().orc_distinctive_descriptors(_p(desc), _p(off), n, _p(bi), _p(bm))
    return bi, bm


# ---- second-camera (Nleft != -1) matcher paths ------------------------------------------------------------------

def search_by_projection_stereo(gridL: Grid, descL, gridR, descR, left, right, qdesc, l2r, r2l, taken, marks, nnratio):
    """SearchByProjection(Frame&, vector<MapPoint*>&, ...) with both camera passes.  left / right =
    (x, y, radius, level) arrays (radius <= 0: not in that camera's view); gridR may be None (one camera).
    Returns (kp_query over the combined index, taken_after, nmatches)."""
    nL = len(gridL.kps)
    nR = 0 if gridR is None else len(gridR.kps)
    descL, qdesc = _u8(descL), _u8(qdesc)
    descR = None if gridR is None else _u8(descR)
    L = [_f32(left[0]), _f32(left[1]), _f32(left[2]), _i32(left[3])]
    R = [_f32(right[0]), _f32(right[1]), _f32(right[2]), _i32(right[3])]
    a = None if l2r is None else _i32(l2r)
    b = None if r2l is None else _i32(r2l)
    tk = _u8(taken).copy()
    mk = _u8(marks)
    out = np.zeros(nL + nR, np.int32)
    n = lib().orc_search_by_projection_stereo(gridL.g, _p(gridL.kps), _p(descL), nL,
                                              None if gridR is None else gridR.g, None if gridR is None else _p(gridR.kps),
                                              _p(descR), nR, *[_p(x) for x in L], *[_p(x) for x in R], _p(qdesc), len(L[0]),
                                              _p(a), _p(b), _p(tk), _p(mk), nnratio, _p(out))
    return out, tk, n


def search_window_best_stereo(gridL: Grid, descL, gridR, descR, left, right, qdesc, max_dist, taken, marks, qangle,
                              factor=1.0 / 30):
    """SearchByProjection(Frame& Cur, const Frame& Last, ...) with the right-image pass.  left / right =
    (x, y, radius, min_level, max_level)."""
    nL = len(gridL.kps)
    nR = 0 if gridR is None else len(gridR.kps)
    descL, qdesc = _u8(descL), _u8(qdesc)
    descR = None if gridR is None else _u8(descR)
    L = [_f32(left[0]), _f32(left[1]), _f32(left[2]), _i32(left[3]), _i32(left[4])]
    R = [_f32(right[0]), _f32(right[1]), _f32(right[2]), _i32(right[3]), _i32(right[4])]
    tk = _u8(taken).copy()
    mk = _u8(marks)
    qa = None if qangle is None else _f32(qangle)
    out = np.zeros(nL + nR, np.int32)
    n = lib().orc_search_window_best_stereo(gridL.g, _p(gridL.kps), _p(descL), nL,
                                            None if gridR is None else gridR.g, None if gridR is None else _p(gridR.kps),
                                            _p(descR), nR, *[_p(x) for x in L], *[_p(x) for x in R], _p(qdesc), len(L[0]),
                                            max_dist, _p(tk), _p(mk), _p(qa), factor, _p(out))
    return out, tk, n


def search_for_triangulation_cameras(kps1, desc1, free1, ur1, n_left1, idx1, off1, kps2, desc2, free2, ur2, n_left2,
                                     idx2, off2, cam_types, cam_params, poses, ep, scale_factors2, level_sigma2_1,
                                     level_sigma2_2, only_stereo, coarse, check_ori, with_margin=False):
    """cam_types[4], cam_params[4][8] = {cam1 left, cam1 right, cam2 left, cam2 right}; poses[4][12] = ll, lr, rl, rr as
    (R12 row-major, t12).  with_margin: also the per-keypoint rounding margin of the gates evaluated for it (< 1: the
    keypoint's outcome may differ under binary32 rounding, see camera_oracle.cpp)."""
    kps1, kps2 = np.ascontiguousarray(kps1), np.ascontiguousarray(kps2)
    desc1, desc2 = _u8(desc1), _u8(desc2)
    a = [_u8(free1), None if ur1 is None else _f32(ur1), _i32(idx1), _i32(off1)]
    b = [_u8(free2), None if ur2 is None else _f32(ur2), _i32(idx2), _i32(off2)]
    ct, cp, ps = _i32(cam_types), _f32(cam_params).reshape(32), _f32(poses).reshape(48)
    sf, s1, s2 = _f32(scale_factors2), _f32(level_sigma2_1), _f32(level_sigma2_2)
    out = np.zeros(len(kps1), np.int32)
    margin = np.zeros(len(kps1), np.float64)
    n = lib().orc_search_for_triangulation_cameras(
        _p(kps1), _p(desc1), _p(a[0]), _p(a[1]), len(kps1), n_left1, _p(a[2]), _p(a[3]),
        _p(kps2), _p(desc2), _p(b[0]), _p(b[1]), len(kps2), n_left2, _p(b[2]), _p(b[3]), len(a[3]) - 1,
        _p(ct), _p(cp), _p(ps), float(ep[0]), float(ep[1]), _p(sf), _p(s1), _p(s2),
        int(only_stereo), int(coarse), int(check_ori), _p(out), _p(margin))
    return (out, n, margin) if with_margin else (out, n)


def camera_project(cam_type, params, xyz):
    uv = np.zeros(2, np.float32)
    lib().orc_camera_project(cam_type, _p(_f32(params)), _p(_f32(xyz)), _p(uv))
    return uv


def camera_unproject(cam_type, params, uv):
    xyz = np.zeros(3, np.float32)
    lib().orc_camera_unproject(cam_type, _p(_f32(params)), _p(_f32(uv)), _p(xyz))
    return xyz


def kb8_triangulate_matches(cam1, cam2, p1, p2, R12, t12, sigma1, sigma2):
    return float(lib().orc_kb8_triangulate_matches(_p(_f32(cam1)), _p(_f32(cam2)), float(p1[0]), float(p1[1]), float(p2[0]),
                                                   float(p2[1]), _p(_f32(R12).reshape(9)), _p(_f32(t12)), sigma1, sigma2))


def stereo_fisheye_matches(kpsL, descL, mono_left, kpsR, descR, mono_right, cam1, cam2, Rlr, tlr, level_sigma2,
                           with_margin=False):
    """Frame::ComputeStereoFishEyeMatches: (left_to_right, right_to_left, depth, points3d, nMatches[, margin])."""
    kpsL, kpsR = np.ascontiguousarray(kpsL), np.ascontiguousarray(kpsR)
    descL, descR = _u8(descL), _u8(descR)
    nL, nR = len(kpsL), len(kpsR)
    l2r, r2l = np.zeros(nL, np.int32), np.zeros(nR, np.int32)
    depth, p3d = np.zeros(nL, np.float32), np.zeros((nL, 3), np.float32)
    margin = np.zeros(nL, np.float64)
    n = lib().orc_stereo_fisheye_matches(_p(kpsL), _p(descL), nL, mono_left, _p(kpsR), _p(descR), nR, mono_right,
                                         _p(_f32(cam1)), _p(_f32(cam2)), _p(_f32(Rlr).reshape(9)), _p(_f32(tlr)),
                                         _p(_f32(level_sigma2)), _p(l2r), _p(r2l), _p(depth), _p(p3d), _p(margin))
    return (l2r, r2l, depth, p3d, n, margin) if with_margin else (l2r, r2l, depth, p3d, n)


def triangulate_matches(type1, cam1, type2, cam2, p1, p2, R12, t12, sigma1, sigma2):
    """TriangulateMatches in binary64: (depth or rejection code, point[3], rounding margin)."""
    p3d = np.zeros(3, np.float64)
    m = np.zeros(1, np.float64)
    z = lib().orc_triangulate_matches(type1, _p(_f32(cam1)), type2, _p(_f32(cam2)), float(p1[0]), float(p1[1]), float(p2[0]),
                                      float(p2[1]), _p(_f32(R12).reshape(9)), _p(_f32(t12)), sigma1, sigma2, _p(p3d), _p(m))
    return float(z), p3d, float(m[0])


def epipolar_constrain_margin(type1, cam1, type2, cam2, p1, p2, R12, t12, sigma1, sigma2):
    m = np.zeros(1, np.float64)
    ok = lib().orc_epipolar_constrain_margin(type1, _p(_f32(cam1)), type2, _p(_f32(cam2)), float(p1[0]), float(p1[1]),
                                             float(p2[0]), float(p2[1]), _p(_f32(R12).reshape(9)), _p(_f32(t12)), sigma1,
                                             sigma2, _p(m))
    return bool(ok), float(m[0])


def epipolar_constrain(type1, cam1, type2, cam2, p1, p2, R12, t12, sigma1, sigma2):
    return bool(lib().orc_epipolar_constrain(type1, _p(_f32(cam1)), type2, _p(_f32(cam2)), float(p1[0]), float(p1[1]),
                                             float(p2[0]), float(p2[1]), _p(_f32(R12).reshape(9)), _p(_f32(t12)), sigma1, sigma2))


def camera_gates_many(type1, cam1, type2, cam2, uv1, uv2, R12, t12, sigma1, sigma2):
    """The binary64 gates for n pairs: dict(depth, tri_margin, p3d, accepted, epi_margin)."""
    uv1, uv2 = _f32(uv1).reshape(-1, 2), _f32(uv2).reshape(-1, 2)
    n = len(uv1)
    s1 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma1, np.float32), (n,)))
    s2 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma2, np.float32), (n,)))
    depth, tm, em = np.zeros(n), np.full(n, 1e300), np.full(n, 1e300)
    p3d, acc = np.zeros((n, 3)), np.zeros(n, np.uint8)
    lib().orc_camera_gates_many(type1, _p(_f32(cam1)), type2, _p(_f32(cam2)), _p(uv1), _p(uv2), n, _p(_f32(R12).reshape(9)),
                                _p(_f32(t12)), _p(s1), _p(s2), _p(depth), _p(tm), _p(p3d), _p(acc), _p(em))
    return dict(depth=depth, tri_margin=tm, p3d=p3d, accepted=acc.astype(bool), epi_margin=em)
