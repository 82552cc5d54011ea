"""Loader of the HIP library (liborbx.so) and ctypes prototypes of its C ABI.

There is no CPU fallback: if the library is missing or cannot be loaded this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORBX_LIB points the loader at another build of the same library (the diagnostic build with phase stamps)
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "lib", "liborbx.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MATCH2_DTYPE = np.dtype([("idx0", "<i4"), ("dist0", "<i4"), ("idx1", "<i4"), ("dist1", "<i4")])
QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("xr", "<f4"), ("radius", "<f4"),
                        ("min_level", "<i4"), ("max_level", "<i4")])
RECT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])

ORBX_OK, ORBX_ERR_INVALID, ORBX_ERR_CAPACITY, ORBX_ERR_HIP, ORBX_ERR_EMPTY = 0, -1, -2, -3, -4


class OrbxParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("max_width", C.c_int),
                ("max_height", C.c_int), ("max_batch", C.c_int), ("device", C.c_int)]


class OrbxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"orbx status {status}: {msg}")
        self.status = status


# Every symbol include/orbx.h declares, with its prototype.
PROTOTYPES = {
    "orbx_last_error": (C.c_char_p, []),
    "orbx_device_count": (C.c_int, []),
    "orbx_extractor_create": (C.c_int, [C.POINTER(OrbxParams), C.POINTER(C.c_void_p)]),
    "orbx_extractor_destroy": (None, [C.c_void_p]),
    "orbx_extractor_tables": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "orbx_extractor_max_keypoints": (C.c_int, [C.c_void_p]),
    "orbx_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "orbx_extract_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_extract_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t,
                                            C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    "orbx_extract_status": (C.c_int, [C.c_void_p]),
    "orbx_extractor_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "orbx_extractor_get_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_long)]),
    "orbx_pyramid_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int)]),
    "orbx_pyramid_level_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_ssize_t)]),
    "orbx_debug_blurred_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int)]),
    "orbx_debug_level_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_int)]),
    "orbx_search_window_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_fuse_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_search_by_sim3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_search_for_initialization": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_bucket_allpairs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "orbx_search_by_bow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_search_by_bow_keyframes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_search_for_triangulation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_release_cached_memory": (None, []),
    "orbx_stager_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "orbx_stager_destroy": (None, [C.c_void_p]),
    "orbx_stager_output_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "orbx_stage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t]),
    "orbx_stage_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_void_p]),
    "orbx_extract_batch_staged_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_frame_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "orbx_frame_destroy": (None, [C.c_void_p]),
    "orbx_frame_assign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "orbx_frame_assign_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "orbx_frame_size": (C.c_int, [C.c_void_p]),
    "orbx_image_bounds": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "orbx_frame_assign_distorted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "orbx_frame_assign_distorted_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "orbx_frame_keypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orbx_frame_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "orbx_frame_window_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "orbx_frame_search_by_projection": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_search_window_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_search_by_projection_stereo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_search_window_best_stereo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_search_for_triangulation_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_frame_search_by_bow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_search_by_bow_keyframes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_search_for_triangulation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_frame_fuse_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_distinctive_descriptors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_vocabulary_create": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "orbx_vocabulary_destroy": (None, [C.c_void_p]),
    "orbx_bow_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_bow_vectors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_scale_tables": (C.c_int, [C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_stereo_fisheye_matches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_camera_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orbx_camera_unproject": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orbx_camera_triangulate_matches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_camera_epipolar_constrain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_archive_read_matrix": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    "orbx_archive_write_matrix": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "orbx_archive_read_keypoints": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "orbx_archive_write_keypoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "orbx_archive_scan_descriptors": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "orbx_archive_copy_descriptors": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_void_p]),
    "orbx_debug_wave_sort": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_debug_level_keypoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "orbx_descriptor_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orbx_hamming_knn2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "orbx_knn2_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "orbx_hamming_knn2_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "orbx_knn2_static_max_queries": (C.c_size_t, []),
    "orbx_hamming_knn2_device_static": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                  C.c_void_p]),
    "orbx_knn2_merge_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orbx_stereo_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "orbx_stereo_match_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbx_window_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.POINTER(C.c_int)]),
    "orbx_search_by_projection": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                            C.c_void_p, C.POINTER(C.c_int)]),
    "orbx_bucket_best": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C orb_slam3_amd/csrc` "
                              "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch's ROCm wheel bundles its own libamdhip64 (same
        # SONAME as the system one).  Importing torch first makes the dynamic loader resolve
        # liborbx.so's dependency to that already-loaded runtime, so device pointers and
        # streams can be shared with torch tensors.  Without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)   # raises AttributeError on a missing export
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != ORBX_OK:
        raise OrbxError(status, lib().orbx_last_error().decode("utf-8", "replace"))


def ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
