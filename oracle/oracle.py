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
    src = os.path.join(_HERE, "orb_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
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
        L.orc_features_in_area.restype = C.c_int
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                           C.c_int, C.c_void_p, C.c_int]
        L.orc_search_by_projection.restype = C.c_int
        L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_bucket_best.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]
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
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        rects = None
        nr = 0
        if dynamic_rects is not None and len(dynamic_rects):
            rects = np.ascontiguousarray(dynamic_rects, np.int32).reshape(-1, 4)
            nr = len(rects)
        mono = self.L.orc_extract(self.h, image.ctypes.data, w, h, image.strides[0], int(lapping[0]),
                                  int(lapping[1]), _p(rects), nr, _p(kps), _p(desc), cap, C.byref(n))
        assert n.value <= cap
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
