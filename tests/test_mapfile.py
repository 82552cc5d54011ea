"""Saved-map records (serializeMatrix / serializeVectorKeyPoints, include/SerializationUtils.h:76-169): writer, reader and
the descriptor scan over an archive image, in both archive encodings.  Host code only (no GPU needed).  The layouts
follow the two templates and boost's primitive encoding; the reference tree holds no saved map, so the byte strings
below are derived by hand from that description: parity unpinned."""
import struct

import numpy as np
import pytest

from orb_slam3_amd import mapfile, synth
from orb_slam3_amd._lib import OrbxError


def _kps(rng, n):
    k = np.zeros(n, synth.KP_DTYPE)
    k["x"] = rng.uniform(0, 752, n); k["y"] = rng.uniform(0, 480, n)
    k["size"] = 31 * 1.2 ** rng.integers(0, 8, n)
    k["angle"] = rng.uniform(0, 360, n); k["response"] = rng.integers(7, 255, n)
    k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    return k


def test_binary_layout_by_hand():
    d = np.arange(64, dtype=np.uint8).reshape(2, 32)
    rec = mapfile.write_matrix(d, text=False)
    # cols, rows, type as native ints, `continuous` as one byte, then the bytes of the matrix
    assert rec == struct.pack("<iii", 32, 2, 0) + b"\x01" + d.tobytes()
    k = np.zeros(1, synth.KP_DTYPE)
    k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"], k["class_id"] = 10.5, 20.25, 31, 90, 50, 3, -1
    rec = mapfile.write_keypoints(k, text=False)
    # NumEl, then angle, response, size, pt.x, pt.y, class_id, octave
    assert rec == struct.pack("<i", 1) + struct.pack("<fffffii", 90, 50, 31, 10.5, 20.25, -1, 3)


def test_text_layout_by_hand():
    d = np.array([[7, 255] + [0] * 30], np.uint8)
    rec = mapfile.write_matrix(d, text=True).decode()
    assert rec.split() == ["32", "1", "0", "1", "7", "255"] + ["0"] * 30
    k = np.zeros(1, synth.KP_DTYPE)
    k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"], k["class_id"] = 10.5, 20.25, 31, 90, 50, 3, -1
    tok = mapfile.write_keypoints(k, text=True).decode().split()
    assert tok[0] == "1" and [float(t) for t in tok[1:6]] == [90, 50, 31, 10.5, 20.25] and tok[6:] == ["-1", "3"]


@pytest.mark.parametrize("text", [False, True])
def test_round_trips(text):
    rng = np.random.default_rng(1)
    d = synth.random_descriptors(137, 3)
    rec = mapfile.write_matrix(d, text)
    m, typ, used = mapfile.read_matrix(rec + b" 99 tail", text)
    assert typ == 0 and used == len(rec) and np.array_equal(m, d)
    dist = rng.normal(0, 0.1, (4, 1)).astype(np.float32)                     # mDistCoef: CV_32F (type 5)
    rec = mapfile.write_matrix(dist, text, cv_type=5)
    m, typ, used = mapfile.read_matrix(rec, text)
    assert typ == 5 and np.array_equal(m.view(np.float32).reshape(4, 1), dist)
    k = _kps(rng, 211)
    k["x"][0] = np.float32(1) / np.float32(3)                                # needs all 9 digits in a text archive
    rec = mapfile.write_keypoints(k, text)
    k2, used = mapfile.read_keypoints(rec, text)
    assert used == len(rec) and np.array_equal(k2, k)
    e, used = mapfile.read_keypoints(mapfile.write_keypoints(k[:0], text), text)
    assert len(e) == 0
    m, _, _ = mapfile.read_matrix(mapfile.write_matrix(d[:0], text), text)
    assert m.shape[0] == 0


@pytest.mark.parametrize("text", [False, True])
def test_truncated_records_are_refused(text):
    d = synth.random_descriptors(10, 3)
    rec = mapfile.write_matrix(d, text)
    with pytest.raises(OrbxError):
        mapfile.read_matrix(rec[:len(rec) // 2], text)
    rec = mapfile.write_keypoints(_kps(np.random.default_rng(2), 5), text)
    with pytest.raises(OrbxError):
        mapfile.read_keypoints(rec[:len(rec) - 6], text)


def _fake_archive(rng, text, n_kf, with_map_points):
    """An archive image shaped like what KeyFrame::serialize emits around the records of interest (KeyFrame.h:107-126):
    unrelated primitives, mDistCoef, N, mvKeys, mvKeysUn, mvuRight, mvDepth, mDescriptors, more primitives."""
    def prim_i(v):
        return (" %d" % v).encode() if text else struct.pack("<i", v)

    def prim_f(v):
        return (" %.9e" % v).encode() if text else struct.pack("<f", v)
    out = [b"22 serialization::archive 17 0 0" if text else b"\x16\0\0\0\0\0\0\0serialization::archive\x11\0\x04\x08\x04\x08\x01\0\0\0"]
    descs = []
    for f in range(n_kf):
        n = int(rng.integers(40, 300))
        for v in rng.uniform(0, 500, 9):
            out.append(prim_f(float(v)))
        out.append(mapfile.write_matrix(rng.normal(0, 0.1, (4, 1)).astype(np.float32), text, cv_type=5))
        out.append(prim_i(n))
        k = _kps(rng, n)
        out.append(mapfile.write_keypoints(k, text))
        out.append(mapfile.write_keypoints(k, text))
        for _ in range(2):                                        # mvuRight, mvDepth: count + floats
            out.append(prim_i(n))
            out += [prim_f(-1.0)] * n
        d = synth.random_descriptors(n, 100 + f)
        d[0, :4] = [32, 0, 0, 0]                                  # payload bytes that look like the start of a header
        descs.append(d)
        out.append(mapfile.write_matrix(d, text))
        for v in rng.integers(0, 1000, 20):
            out.append(prim_i(int(v)))
    mp = []
    if with_map_points:
        for p in range(25):
            d = synth.random_descriptors(1, 900 + p)
            mp.append(d)
            out.append(prim_i(p))
            out.append(mapfile.write_matrix(d, text))
    return b"".join(out), descs, mp


@pytest.mark.parametrize("text", [False, True])
def test_scan_finds_every_descriptor_matrix(text, tmp_path):
    rng = np.random.default_rng(7)
    img, descs, mp = _fake_archive(rng, text, 5, True)
    off, rows = mapfile.scan_descriptors(img, text, min_rows=2)
    assert list(rows) == [len(d) for d in descs]
    off1, rows1 = mapfile.scan_descriptors(img, text, min_rows=1)
    assert list(rows1) == [len(d) for d in descs] + [1] * len(mp)
    path = tmp_path / ("map.osa.txt" if text else "map.osa")
    path.write_bytes(img)
    md = mapfile.MapDescriptors(str(path), text)
    whole = np.concatenate(descs)
    assert len(md) == len(whole)
    for i, d in enumerate(descs):
        assert np.array_equal(md.matrix(i), d)
    got = np.concatenate([c for _, c in md.chunks(97)])
    firsts = [f for f, _ in md.chunks(97)]
    assert np.array_equal(got, whole) and firsts == list(range(0, len(whole), 97))
    parts = [md.shard(r, 3) for r in range(3)]
    assert [p[0] for p in parts] == [0, len(whole) // 3, 2 * (len(whole) // 3)]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), whole)
    md.close()


def test_scan_of_an_archive_without_descriptors():
    off, rows = mapfile.scan_descriptors(b"no matrices in here at all, just text 32 5 0 0 1 2 3", True)
    assert len(rows) == 0
    off, rows = mapfile.scan_descriptors(bytes(range(256)) * 4, False)
    assert len(rows) == 0
