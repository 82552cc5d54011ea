"""Saved-map records: the layouts through which the reference stores keypoints and descriptors in its Atlas archives
(include/SerializationUtils.h:76-169, include/KeyFrame.h:118-126, System.cc:1568-1672), read and written through the
C ABI (orbx_archive_*), and a loader that pulls every descriptor matrix out of an archive image for the large-map
brute-force matcher.  Parity unpinned: the reference tree holds neither a saved map nor boost."""
from __future__ import annotations

import ctypes as C
import mmap
import os

import numpy as np

from ._lib import KP_DTYPE, check, lib, ptr


def _buf(a):
    a = np.frombuffer(a, np.uint8) if not isinstance(a, np.ndarray) else a
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a


def write_matrix(mat: np.ndarray, text: bool, cv_type: int = 0) -> bytes:
    """serializeMatrix of a continuous cv::Mat (cv_type 0 = CV_8UC1)."""
    mat = np.ascontiguousarray(mat)
    rows, cols = mat.shape[:2]
    raw = mat.view(np.uint8).reshape(-1)
    n = C.c_size_t()
    check(lib().orbx_archive_write_matrix(ptr(raw), rows, cols, cv_type, int(text), None, 0, C.byref(n)))
    out = np.zeros(n.value, np.uint8)
    check(lib().orbx_archive_write_matrix(ptr(raw), rows, cols, cv_type, int(text), ptr(out), len(out), C.byref(n)))
    return out.tobytes()


def read_matrix(buf, text: bool, offset: int = 0):
    """(raw bytes as uint8 [rows, cols * elemSize], cv_type, bytes consumed) of the serializeMatrix record at offset."""
    b = _buf(buf)[offset:]
    used, rows, cols, typ = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
    check(lib().orbx_archive_read_matrix(ptr(b), len(b), int(text), C.byref(used), C.byref(rows), C.byref(cols),
                                         C.byref(typ), None, 0))
    depth_bytes = [1, 1, 2, 2, 4, 4, 8, 2][typ.value & 7] * ((typ.value >> 3) + 1)
    out = np.zeros(max(rows.value * cols.value * depth_bytes, 1), np.uint8)
    check(lib().orbx_archive_read_matrix(ptr(b), len(b), int(text), C.byref(used), C.byref(rows), C.byref(cols),
                                         C.byref(typ), ptr(out), len(out)))
    return out[:rows.value * cols.value * depth_bytes].reshape(rows.value, cols.value * depth_bytes), typ.value, used.value


def write_keypoints(kps: np.ndarray, text: bool) -> bytes:
    """serializeVectorKeyPoints."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    n = C.c_size_t()
    check(lib().orbx_archive_write_keypoints(ptr(kps), len(kps), int(text), None, 0, C.byref(n)))
    out = np.zeros(n.value, np.uint8)
    check(lib().orbx_archive_write_keypoints(ptr(kps), len(kps), int(text), ptr(out), len(out), C.byref(n)))
    return out.tobytes()


def read_keypoints(buf, text: bool, offset: int = 0):
    b = _buf(buf)[offset:]
    used, n = C.c_size_t(), C.c_int()
    check(lib().orbx_archive_read_keypoints(ptr(b), len(b), int(text), C.byref(used), None, 0, C.byref(n)))
    kps = np.zeros(max(n.value, 1), KP_DTYPE)
    check(lib().orbx_archive_read_keypoints(ptr(b), len(b), int(text), C.byref(used), ptr(kps), len(kps), C.byref(n)))
    return kps[:n.value].copy(), used.value


def scan_descriptors(buf, text: bool, min_rows: int = 2):
    """(payload offsets, rows) of every descriptor matrix (32 columns, CV_8UC1, continuous) in an archive image."""
    b = _buf(buf)
    n, total = C.c_int(), C.c_longlong()
    check(lib().orbx_archive_scan_descriptors(ptr(b), len(b), int(text), min_rows, None, None, 0, C.byref(n), C.byref(total)))
    off = np.zeros(max(n.value, 1), np.uint64)
    rows = np.zeros(max(n.value, 1), np.int32)
    check(lib().orbx_archive_scan_descriptors(ptr(b), len(b), int(text), min_rows, ptr(off), ptr(rows), n.value,
                                              C.byref(n), C.byref(total)))
    return off[:n.value].astype(np.int64), rows[:n.value].copy()


class MapDescriptors:
    """The descriptor matrices of a saved map (an Atlas archive file), memory-mapped.  Iterating `chunks(max_rows)`
    yields (global first row, uint8 [rows, 32]) blocks in file order: the contiguous index ranges a sharded
    brute-force search hands to its ranks (the file is never read whole)."""

    def __init__(self, path: str, text: bool = False, min_rows: int = 2):
        self.path, self.text = path, text
        self._f = open(path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
        self._buf = np.frombuffer(self._mm, np.uint8) if size else np.zeros(0, np.uint8)
        self.offsets, self.rows = scan_descriptors(self._buf, text, min_rows) if size else (np.zeros(0, np.int64), np.zeros(0, np.int32))
        self.first_row = np.concatenate([[0], np.cumsum(self.rows)]).astype(np.int64)

    def __len__(self):
        return int(self.first_row[-1])

    def matrix(self, i: int) -> np.ndarray:
        rows = int(self.rows[i])
        if not self.text:
            o = int(self.offsets[i])
            return self._buf[o:o + rows * 32].reshape(rows, 32)
        out = np.zeros((max(rows, 1), 32), np.uint8)
        check(lib().orbx_archive_copy_descriptors(ptr(self._buf), len(self._buf), 1, int(self.offsets[i]), rows, ptr(out)))
        return out[:rows]

    def chunks(self, max_rows: int):
        pend, n, first = [], 0, 0
        for i in range(len(self.rows)):
            m = self.matrix(i)
            s = 0
            while s < len(m):
                take = min(len(m) - s, max_rows - n)
                pend.append(m[s:s + take]); n += take; s += take
                if n == max_rows:
                    yield first, np.concatenate(pend)
                    first += n; pend, n = [], 0
        if n:
            yield first, np.concatenate(pend)

    def shard(self, rank: int, world: int) -> tuple:
        """(global first row, rows) of the contiguous range rank `rank` of `world` searches."""
        total = len(self)
        per = total // world
        b = rank * per
        e = total if rank == world - 1 else b + per
        parts = []
        for i in range(len(self.rows)):
            lo, hi = max(b, int(self.first_row[i])), min(e, int(self.first_row[i + 1]))
            if lo < hi:
                parts.append(self.matrix(i)[lo - int(self.first_row[i]):hi - int(self.first_row[i])])
        return b, (np.concatenate(parts) if parts else np.zeros((0, 32), np.uint8))

    def close(self):
        self._buf = None
        if self._mm is not None:
            self._mm.close()
        self._f.close()
