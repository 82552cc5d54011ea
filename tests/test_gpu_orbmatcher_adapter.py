"""The flattening / write-back templates of include/orbx/orbmatcher_adapter.hpp, run as a native program on stand-in
Frame / KeyFrame / MapPoint types (tests/cpp/orbmatcher_adapter_demo.cpp) and compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "orb_slam3_amd", "lib", "orbmatcher_adapter_demo")
W, H = 752, 480
f32 = np.float32


def _residents(n):
    """(slot tag before the search, taken flag): every 17th slot holds a map point with observations, the next one a
    map point without."""
    tag = np.full(n, -1, np.int32)
    j = np.arange(n)
    tag[j % 17 <= 1] = -2
    return tag, (j % 17 == 0).astype(np.uint8)


def _replay(tag, kp_query, src):
    out = tag.copy()
    hit = kp_query >= 0
    out[hit] = src[kp_query[hit]]
    return out


def _feature_vector(desc, drop):
    node = desc[:, 0] & 31
    return {int(k): np.nonzero(node == k)[0] for k in np.unique(node) if k % drop != 0}


def _shared(fv1, fv2):
    idx1, off1, idx2, off2 = [], [0], [], [0]
    for k in sorted(set(fv1) & set(fv2)):
        idx1 += list(fv1[k]); idx2 += list(fv2[k])
        off1.append(len(idx1)); off2.append(len(idx2))
    return np.array(idx1, np.int32), np.array(off1, np.int32), np.array(idx2, np.int32), np.array(off2, np.int32)


def test_adapter_templates_match_oracle(oracle, tmp_path):
    assert os.path.exists(DEMO), "orbmatcher_adapter_demo not built (make -C orb_slam3_amd/csrc)"
    from orb_slam3_amd import KP_DTYPE
    L, R = synth.stereo_pair(W, H, 11)
    (tmp_path / "l.raw").write_bytes(L.tobytes())
    (tmp_path / "r.raw").write_bytes(R.tobytes())
    pre = str(tmp_path / "o")
    run = subprocess.run([DEMO, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(W), str(H), pre],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    rd = lambda name, dt: np.fromfile(pre + name, dtype=dt)
    eL, eR = oracle.Extractor(1200), oracle.Extractor(1200)
    kL, dL, _ = eL(L, (0, 0))
    kR, dR, _ = eR(R, (0, 0))
    assert np.array_equal(rd("_kL.bin", KP_DTYPE), kL) and np.array_equal(rd("_kR.bin", KP_DTYPE), kR)
    nL, nR = len(kL), len(kR)
    sf = eL.scale_factors.astype(f32)
    i = np.arange(nL)
    lvl = kL["octave"].astype(np.int32)
    px = (kL["x"] + f32(0.25) * (i % 5).astype(f32)).astype(f32)
    py = (kL["y"] - f32(0.5) * (i % 3).astype(f32)).astype(f32)
    pxr = (px - f32(5.0)).astype(f32)
    view = np.where(i % 2 == 1, f32(2.5), f32(4.0)).astype(f32)          # RadiusByViewingCos of 0.999 / 0.9
    in_view = (i % 7 != 0) & (i % 11 != 0)                               # mbTrackInView and not bad
    depth = 10.0 + 5.0 * (i % 13)
    gR = oracle.Grid(kR, 0.0, float(W), 0.0, float(H))
    gL = oracle.Grid(kL, 0.0, float(W), 0.0, float(H))

    # A: mvuRight gate, far points dropped, th = 3, every candidate observed
    keep = np.nonzero(in_view & ~(depth > 50.0))[0]
    uR = np.full(nR, -1, f32)
    j = np.arange(0, nR, 3)
    uR[j] = (kR["x"][j] - f32(5.0) + f32(3.0) * ((j % 4).astype(f32) - f32(1.5))).astype(f32)
    tag, taken = _residents(nR)
    radius = ((view * f32(3.0)).astype(f32) * sf[lvl]).astype(f32)
    out, _, n = oracle.search_by_projection(gR, dR, uR, px[keep], py[keep], pxr[keep], lvl[keep], radius[keep], dL[keep], taken, 0.8)
    want = tag.copy()
    for q, kp in enumerate(out):
        if kp >= 0:
            want[kp] = keep[q]
    assert f"A={n} cacheHit=1" in run.stdout and n > 50
    assert np.array_equal(rd("_A.bin", np.int32), want)

    # B: no mvuRight, th = 1, candidates without observations do not block their keypoint
    keep = np.nonzero(in_view)[0]
    marks = (i % 3 != 0).astype(np.uint8)
    radius1 = (view * sf[lvl]).astype(f32)
    zeros = np.zeros(len(keep), f32)
    left = (px[keep], py[keep], radius1[keep], lvl[keep])
    right = (zeros, zeros, zeros, np.zeros(len(keep), np.int32))
    kpq, _, n = oracle.search_by_projection_stereo(gR, dR, None, None, left, right, dL[keep], None, None, taken, marks[keep], 0.9)
    assert f"B={n}" in run.stdout and n > 10
    assert np.array_equal(rd("_B.bin", np.int32), _replay(tag, kpq, keep))

    # C: two cameras
    knn = oracle.knn2(dL, dR)
    t = knn[:, 0]
    has_r = (i % 2 == 0) & (i % 10 != 0) & (i % 11 != 0)                 # mbTrackInViewR, level != -1, not bad
    lvl_r = kR["octave"][t].astype(np.int32)
    view_r = np.where(i % 4 != 0, f32(4.0), f32(2.5)).astype(f32)
    rad_l = np.where(in_view, radius, f32(0)).astype(f32)
    rad_r = np.where(has_r, (view_r * sf[lvl_r]).astype(f32), f32(0)).astype(f32)
    keep = np.nonzero((rad_l > 0) | (rad_r > 0))[0]
    left = (px[keep], py[keep], rad_l[keep], lvl[keep])
    right = ((kR["x"][t] + f32(0.5)).astype(f32)[keep], kR["y"][t][keep], rad_r[keep], lvl_r[keep])
    l2r = np.full(nL, -1, np.int32)
    r2l = np.full(nR, -1, np.int32)
    m = np.arange(0, min(nL, nR), 5)
    l2r[m] = m
    r2l[m] = m
    tag2, taken2 = _residents(nL + nR)
    kpq, _, n = oracle.search_by_projection_stereo(gL, dL, gR, dR, left, right, dL[keep], l2r, r2l, taken2, marks[keep], 0.9)
    assert f"C={n} sizes={nL},{nR}" in run.stdout and n > 100
    assert np.array_equal(rd("_C.bin", np.int32), _replay(tag2, kpq, keep))
    assert (kpq[nL:] >= 0).sum() > 20, "the right-camera pass found nothing"

    # D / E: bag-of-words searches over the nodes the two feature vectors share
    fv1, fv2 = _feature_vector(dL, 5), _feature_vector(dR, 7)
    idx1, off1, idx2, off2 = _shared(fv1, fv2)
    v1 = ((i % 4 != 0) & (i % 9 != 0)).astype(np.uint8)
    jR = np.arange(nR)
    v2 = ((jR % 6 != 0) & (jR % 5 != 0)).astype(np.uint8)
    mF, nD = oracle.search_by_bow(dL, kL["angle"], v1, idx1, off1, dR, kR["angle"], -1, idx2, off2, 0.7, True)
    m12, nE = oracle.search_by_bow_keyframes(dL, kL["angle"], v1, idx1, off1, dR, kR["angle"], v2, idx2, off2, 0.8, True)
    assert f"D={nD} E={nE} nodes={len(off1) - 1}" in run.stdout and nD > 10 and nE > 10
    assert np.array_equal(rd("_D.bin", np.int32), mF)                    # tags of keyframe 1 are its keypoint indices
    assert np.array_equal(rd("_E.bin", np.int32), np.where(m12 >= 0, m12 + 100000, -1))
