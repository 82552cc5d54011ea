"""The C ABI library loads without a GPU and exports every symbol include/orbx.h declares;
entry points that need a device fail with a status, never crash."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "orbx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_header_and_prototypes_agree():
    from orb_slam3_amd import _lib
    assert set(_declared()) == set(_lib.PROTOTYPES.keys())


def test_library_exports_every_declared_symbol():
    from orb_slam3_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "liborbx.so not built (make -C orb_slam3_amd/csrc)"
    L = C.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), f"missing export {name}"
    _lib.lib()   # prototypes attach


def test_struct_layouts():
    from orb_slam3_amd import _lib
    assert _lib.KP_DTYPE.itemsize == 28          # cv::KeyPoint
    assert _lib.MATCH2_DTYPE.itemsize == 16
    assert _lib.QUERY_DTYPE.itemsize == 24
    assert _lib.RECT_DTYPE.itemsize == 16
    assert C.sizeof(_lib.OrbxParams) == 36


def test_no_cpu_fallback_in_product_package():
    # the product package must not import or call the oracle
    pkg = os.path.join(ROOT, "orb_slam3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from orb_slam3_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_argument_validation_without_device():
    # invalid arguments are rejected before any device work
    from orb_slam3_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    p = _lib.OrbxParams(1000, 1.2, 99, 20, 7, 752, 480, 1, -1)
    assert L.orbx_extractor_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID
    assert b"nlevels" in L.orbx_last_error()
    p = _lib.OrbxParams(0, 1.2, 8, 20, 7, 752, 480, 1, -1)
    assert L.orbx_extractor_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID
    out = np.zeros(4, np.int32)
    assert L.orbx_hamming_knn2(None, -1, None, 0, None) == _lib.ORBX_ERR_INVALID
    assert L.orbx_descriptor_distance(None, None, 0, None) == _lib.ORBX_OK


def test_new_handles_validate_without_device(oracle):
    """Stager / frame / camera arguments are checked before any device work; the image bounds are host arithmetic
    and equal the oracle's ComputeImageBounds."""
    import orb_slam3_amd as A
    from orb_slam3_amd import _lib
    from orb_slam3_amd.staging import StageParams
    L = _lib.lib()
    h = C.c_void_p()
    p = StageParams(752, 480, 2, 0, 0, 0, 0, None, None, 0)
    assert L.orbx_stager_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID and b"channels" in L.orbx_last_error()
    p = StageParams(752, 480, 1, 0, 1, 752, 480, None, None, 0)          # remap without maps
    assert L.orbx_stager_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID and b"maps" in L.orbx_last_error()
    p = StageParams(752, 480, 1, 0, 2, 0, 480, None, None, 0)            # resize to nothing
    assert L.orbx_stager_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID
    p = StageParams(0, 480, 1, 0, 0, 0, 0, None, None, 0)
    assert L.orbx_stager_create(C.byref(p), C.byref(h)) == _lib.ORBX_ERR_INVALID
    K = (458.654, 457.296, 367.215, 248.375)
    for D in [(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), (0.0, 0.1, 0.0, 0.0), (),
              (0.12, -0.2, 1e-3, -5e-4, 0.05)]:
        assert np.array_equal(A.Camera(*K, D).image_bounds(752, 480), oracle.image_bounds(752, 480, K, D))
    cam = A.Camera(0.0, 457.0, 367.0, 248.0, (0.1,))
    b = np.zeros(4, np.float32)
    assert L.orbx_image_bounds(C.byref(cam), 752, 480, b.ctypes.data_as(C.c_void_p)) == _lib.ORBX_ERR_INVALID
    assert L.orbx_frame_assign(None, None, None, None, 0, 0.0, 1.0, 0.0, 1.0) == _lib.ORBX_ERR_INVALID
    assert L.orbx_frame_size(None) == 0


def _declared_dist():
    txt = open(os.path.join(ROOT, "include", "orbx_dist.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", txt)))


def test_dist_library_exports_and_validates_without_device():
    """liborbx_dist.so (the multi-GPU entry points over RCCL) loads without a GPU, exports what include/orbx_dist.h
    declares and rejects bad arguments before touching a device or a communicator."""
    from orb_slam3_amd import _lib
    _lib.lib()
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "liborbx_dist.so")
    assert os.path.exists(path), "liborbx_dist.so not built (make -C orb_slam3_amd/csrc)"
    D = C.CDLL(path)
    names = _declared_dist()
    assert len(names) == 9
    for name in names:
        assert hasattr(D, name), f"missing export {name}"
    D.orbx_knn2_sharded_workspace_bytes.restype = C.c_size_t
    D.orbx_knn2_sharded_workspace_bytes.argtypes = [C.c_int, C.c_int64, C.c_int]
    L = _lib.lib()
    one = D.orbx_knn2_sharded_workspace_bytes(5000, 1_250_000, 1)
    eight = D.orbx_knn2_sharded_workspace_bytes(5000, 1_250_000, 8)
    assert eight - one == 7 * 80128 and one > L.orbx_knn2_workspace_bytes(5000, 1_250_000)
    assert D.orbx_knn2_sharded_workspace_bytes(-1, 10, 1) == 0 and D.orbx_knn2_sharded_workspace_bytes(10, 10, 0) == 0
    D.orbx_hamming_knn2_sharded_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    assert D.orbx_hamming_knn2_sharded_device(None, 10, None, 10, 0, None, None, None, 0, None) == _lib.ORBX_ERR_INVALID
    assert b"communicator" in L.orbx_last_error()
    assert D.orbx_extract_pool_devices(None) == 0
    assert D.orbx_rccl_unique_id(None) == _lib.ORBX_ERR_INVALID


def test_scale_tables_need_no_device(oracle):
    """orbx_scale_tables is host arithmetic (ORBextractor.cc:481-512): equal to the oracle's tables, no handle, no GPU."""
    from orb_slam3_amd import _lib
    L = _lib.lib()
    for nfeat, sf, nl in [(1000, 1.2, 8), (1200, 1.2, 8), (5000, 1.2, 8), (2000, 1.5, 5), (500, 1.1, 12)]:
        a = [np.zeros(nl, np.float32) for _ in range(4)]
        q = np.zeros(nl, np.int32)
        assert L.orbx_scale_tables(nfeat, sf, nl, *[x.ctypes.data_as(C.c_void_p) for x in a], q.ctypes.data_as(C.c_void_p)) == 0
        e = oracle.Extractor(nfeat, sf, nl, 20, 7)
        assert np.array_equal(a[0], e.scale_factors) and np.array_equal(a[1], e.inv_scale_factors)
        assert q.sum() == nfeat and (q >= 0).all()
    assert L.orbx_scale_tables(1000, 1.0, 8, None, None, None, None, None) == _lib.ORBX_ERR_INVALID
