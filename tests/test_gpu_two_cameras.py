"""GPU parity of the matcher paths of frames that carry a second camera (Frame::Nleft != -1, the TUM-VI fisheye
stereo configuration) against the CPU oracle's restatements: SearchByProjection(Frame&, MapPoints) with the
right-grid pass (ORBmatcher.cc:146-213), SearchByProjection(Cur, Last) with the right-image pass and the shared
rotation histogram (:1879-1969), Fuse(..., bRight) (:1214-1407), SearchForTriangulation with the per-candidate
camera pair and KannalaBrandt8::epipolarConstrain (:1099-1139).  Fixture: a 512 x 512 synthetic stereo pair run
through the extractor with the TUM-VI lapping area.  Bar: identical assignment arrays and return values."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

W = H = 512
BOUNDS = (0.0, float(W), 0.0, float(H))
SF = np.array([1.2 ** i for i in range(8)], np.float32)
# TUM-VI-like KannalaBrandt8 cameras (Examples/Stereo-Inertial/TUM-VI.yaml)
KB_L = np.array([190.978, 190.973, 254.93, 256.90, 0.0034, 0.00076, -0.0021, 0.00024], np.float32)
KB_R = np.array([190.442, 190.434, 252.60, 254.92, 0.0034, 0.00071, -0.0020, 0.00020], np.float32)


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


@pytest.fixture(scope="module")
def fisheye_frame():
    """Left / right keypoints and descriptors of one 512 x 512 stereo pair + mvLeftToRightMatch / mvRightToLeftMatch
    from the brute-force ratio matches (Frame::ComputeStereoFishEyeMatches, Frame.cc:1286-1338, without the triangulation
    gates, which do not matter for the index bookkeeping under test)."""
    A = _gpu()
    L, R = synth.stereo_pair(W, H, 77)
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
    kL, dL, _ = ex(L, (0, 511))
    kR, dR, _ = ex(R, (0, 511))
    m2 = A.knn_match2(dL, dR)
    ok = A.fisheye_ratio_matches(m2, 0.95)    # looser than the reference's 0.7: more stereo partners to exercise
    l2r = np.full(len(kL), -1, np.int32)
    r2l = np.full(len(kR), -1, np.int32)
    for i in ok:
        j = int(m2["idx0"][i])
        if r2l[j] == -1:
            l2r[i] = j
            r2l[j] = i
    assert (l2r >= 0).sum() > 30
    return kL, dL, kR, dR, l2r, r2l


def _queries_two_cameras(rng, kL, dL, kR, dR, l2r, n, th):
    """Map points seen near left keypoints; their right projections sit near the stereo partner when there is one,
    near a random right keypoint otherwise.  Some are visible in one camera only."""
    src = rng.integers(0, len(kL), n)
    ql = np.zeros(n, synth.QUERY_DTYPE)
    qr = np.zeros(n, synth.QUERY_DTYPE)
    ql["x"] = kL["x"][src] + rng.normal(0, 3, n).astype(np.float32)
    ql["y"] = kL["y"][src] + rng.normal(0, 3, n).astype(np.float32)
    lvl = np.clip(kL["octave"][src] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    view_cos = rng.random(n) < 0.5
    r = np.where(view_cos, 2.5, 4.0).astype(np.float32)
    ql["radius"] = (r * np.float32(th) * SF[lvl]).astype(np.float32)
    ql["min_level"] = lvl - 1
    ql["max_level"] = lvl
    partner = np.where(l2r[src] >= 0, l2r[src], rng.integers(0, len(kR), n))
    qr["x"] = kR["x"][partner] + rng.normal(0, 3, n).astype(np.float32)
    qr["y"] = kR["y"][partner] + rng.normal(0, 3, n).astype(np.float32)
    lvr = np.clip(kR["octave"][partner] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    qr["radius"] = (r * SF[lvr]).astype(np.float32)            # no th in the right pass (ORBmatcher.cc:151-154)
    qr["min_level"] = lvr - 1
    qr["max_level"] = lvr
    vis = rng.random(n)
    ql["radius"][vis < 0.15] = 0            # mbTrackInView false
    qr["radius"][vis > 0.85] = 0            # mbTrackInViewR false
    qdesc = synth.noisy_copy(rng, dL[src], 50)
    return ql, lvl, qr, lvr, qdesc, src


@pytest.mark.parametrize("seed,th,two", [(1, 1.0, True), (2, 3.0, True), (3, 1.0, False), (4, 2.0, True), (5, 0.5, True),
                                         (6, 5.0, True), (7, 3.0, False), (8, 1.5, True)])
def test_search_by_projection_both_cameras(oracle, fisheye_frame, seed, th, two):
    A = _gpu()
    kL, dL, kR, dR, l2r, r2l = fisheye_frame
    rng = np.random.default_rng(seed)
    ql, lvl, qr, lvr, qdesc, _ = _queries_two_cameras(rng, kL, dL, kR, dR, l2r, 1200, th)
    nL, nR = len(kL), (len(kR) if two else 0)
    taken0 = (rng.random(nL + nR) < 0.15).astype(np.uint8)
    marks = (rng.random(len(ql)) < 0.85).astype(np.uint8)         # map points without observations do not occupy
    gL = oracle.Grid(kL, *BOUNDS)
    gR = oracle.Grid(kR, *BOUNDS) if two else None
    kpq_o, taken_o, n_o = oracle.search_by_projection_stereo(
        gL, dL, gR, dR if two else None, (ql["x"], ql["y"], ql["radius"], lvl), (qr["x"], qr["y"], qr["radius"], lvr),
        qdesc, l2r if two else None, r2l if two else None, taken0, marks, 0.8)
    fL = A.FrameFeatures(kL, dL, None, BOUNDS)
    fR = A.FrameFeatures(kR, dR, None, BOUNDS) if two else None
    taken = taken0.copy()
    kpq, n = A.ORBmatcher(0.8).SearchByProjectionStereoIn(fL, fR, ql, qr, qdesc, taken, l2r if two else None,
                                                         r2l if two else None, marks)
    assert n == n_o
    assert np.array_equal(kpq, kpq_o)
    assert np.array_equal(taken, taken_o)
    assert (kpq[:nL] >= 0).sum() > 100
    if two:
        assert (kpq[nL:] >= 0).sum() > 100
        # a left keypoint with a stereo partner hands its map point to the partner unless a later query took it
        both = np.nonzero((l2r >= 0) & (kpq[:nL] >= 0))[0]
        assert len(both) > 5


def test_left_ratio_failure_skips_the_right_camera_pass(oracle):
    """ORBmatcher.cc:125-126 leaves the map-point loop: a left best within TH_HIGH that fails the ratio test gets no
    right-camera pass.  Fixture: 200 map points with two equally good left candidates on one level and one clear
    right match each (tests/replay_two_cameras.py).  The library must equal the oracle and an independent Python
    transcription of :42-216; the transcription with the exit misread must differ."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import replay_two_cameras as R
    A = _gpu()
    kL, dL, kR, dR, ql, qr, qdesc, l2r, r2l, half = R.ambiguous_left_fixture(3, 400, synth.KP_DTYPE, synth.QUERY_DTYPE)
    gL, gR = oracle.Grid(kL, *BOUNDS), oracle.Grid(kR, *BOUNDS)
    rng = np.random.default_rng(9)
    taken0 = (rng.random(len(kL) + len(kR)) < 0.05).astype(np.uint8)
    marks = (rng.random(len(ql)) < 0.9).astype(np.uint8)
    lvl = ql["max_level"].astype(np.int32)
    kpq_o, taken_o, n_o = oracle.search_by_projection_stereo(
        gL, dL, gR, dR, (ql["x"], ql["y"], ql["radius"], lvl), (qr["x"], qr["y"], qr["radius"], lvl), qdesc, l2r, r2l,
        taken0, marks, 0.8)
    kpq_p, taken_p, n_p, n_exit = R.search_by_projection_two_cameras(gL, kL, dL, gR, kR, dR, ql, qr, qdesc, l2r, r2l,
                                                                     taken0, marks, 0.8)
    kpq_w, _, n_w, _ = R.search_by_projection_two_cameras(gL, kL, dL, gR, kR, dR, ql, qr, qdesc, l2r, r2l, taken0, marks,
                                                          0.8, skip_right_after_left_ratio_failure=False)
    fL, fR = A.FrameFeatures(kL, dL, None, BOUNDS), A.FrameFeatures(kR, dR, None, BOUNDS)
    taken = taken0.copy()
    kpq, n = A.ORBmatcher(0.8).SearchByProjectionStereoIn(fL, fR, ql, qr, qdesc, taken, l2r, r2l, marks)
    assert n_exit >= 150
    assert n == n_o == n_p
    assert np.array_equal(kpq, kpq_o) and np.array_equal(kpq, kpq_p)
    assert np.array_equal(taken, taken_o) and np.array_equal(taken, taken_p)
    assert n_w >= n + 100 and not np.array_equal(kpq_w, kpq)


def test_search_by_projection_one_camera_equals_existing_entry_point(oracle, fisheye_frame):
    """right = NULL and all marks 1 is the mono loop of orbx_frame_search_by_projection."""
    A = _gpu()
    kL, dL, kR, dR, l2r, r2l = fisheye_frame
    rng = np.random.default_rng(5)
    ql, lvl, qr, lvr, qdesc, _ = _queries_two_cameras(rng, kL, dL, kR, dR, l2r, 800, 2.0)
    ql["radius"] = np.maximum(ql["radius"], 1.0)
    fL = A.FrameFeatures(kL, dL, None, BOUNDS)
    t1 = np.zeros(len(kL), np.uint8)
    t2 = np.zeros(len(kL), np.uint8)
    m = A.ORBmatcher(0.8)
    kpq, n = m.SearchByProjectionStereoIn(fL, None, ql, None, qdesc, t1)
    out, n2 = m.SearchByProjectionIn(fL, ql, qdesc, t2)
    assert n == n2 and np.array_equal(t1, t2)
    own = np.nonzero(out >= 0)[0]
    assert np.array_equal(kpq[out[own]], own)


@pytest.mark.parametrize("seed,use_rot", [(1, True), (2, False), (3, True), (4, False), (5, True), (6, True)])
def test_search_window_best_both_cameras(oracle, fisheye_frame, seed, use_rot):
    A = _gpu()
    kL, dL, kR, dR, l2r, r2l = fisheye_frame
    rng = np.random.default_rng(10 + seed)
    n = 1100
    ql, lvl, qr, lvr, qdesc, src = _queries_two_cameras(rng, kL, dL, kR, dR, l2r, n, 7.0)
    ql["radius"] = np.where(ql["radius"] > 0, (7.0 * SF[lvl]), 0).astype(np.float32)
    qr["radius"] = (7.0 * SF[lvl]).astype(np.float32)                 # same radius, last octave (ORBmatcher.cc:1888)
    mode = rng.integers(0, 3, n)                                      # forward / backward / default level windows
    for q_, l_ in ((ql, lvl), (qr, lvl)):
        q_["min_level"] = np.where(mode == 0, l_, np.where(mode == 1, 0, l_ - 1))
        q_["max_level"] = np.where(mode == 0, -1, np.where(mode == 1, l_, l_ + 1))
    # a few left projections far outside the image: their windows are empty, which skips the right pass as well
    ql["x"][:40] = 5000.0
    qangle = ((kL["angle"][src] + rng.normal(30, 20, n)) % 360).astype(np.float32) if use_rot else None
    nL, nR = len(kL), len(kR)
    taken0 = (rng.random(nL + nR) < 0.1).astype(np.uint8)
    marks = (rng.random(n) < 0.9).astype(np.uint8)
    gL, gR = oracle.Grid(kL, *BOUNDS), oracle.Grid(kR, *BOUNDS)
    kpq_o, taken_o, n_o = oracle.search_window_best_stereo(
        gL, dL, gR, dR, (ql["x"], ql["y"], ql["radius"], ql["min_level"], ql["max_level"]),
        (qr["x"], qr["y"], qr["radius"], qr["min_level"], qr["max_level"]), qdesc, 100.0, taken0, marks, qangle, 1.0 / 30)
    fL, fR = A.FrameFeatures(kL, dL, None, BOUNDS), A.FrameFeatures(kR, dR, None, BOUNDS)
    taken = taken0.copy()
    kpq, nm = A.ORBmatcher(0.9, use_rot).SearchWindowBestStereoIn(fL, fR, ql, qr, qdesc, 100.0, taken, marks, qangle)
    assert nm == n_o
    assert np.array_equal(kpq, kpq_o)
    assert np.array_equal(taken, taken_o)
    assert (kpq[:nL] >= 0).sum() > 100 and (kpq[nL:] >= 0).sum() > 50
    # the queries whose left window was empty own nothing, in either camera
    assert not np.isin(kpq, np.arange(40)).any()


def test_fuse_in_the_right_camera(oracle, fisheye_frame):
    """Fuse(pKF, vpMapPoints, th, bRight = true): the right camera's grid and keypoints, mono reprojection gate
    (mvuRight is -1 throughout a two-camera keyframe), result index + NLeft (ORBmatcher.cc:1315, :1332-1368)."""
    A = _gpu()
    kL, dL, kR, dR, l2r, r2l = fisheye_frame
    rng = np.random.default_rng(21)
    n = 700
    src = rng.integers(0, len(kR), n)
    q = np.zeros(n, synth.QUERY_DTYPE)
    q["x"] = kR["x"][src] + rng.normal(0, 1.5, n).astype(np.float32)
    q["y"] = kR["y"][src] + rng.normal(0, 1.5, n).astype(np.float32)
    lvl = np.clip(kR["octave"][src] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
    q["radius"] = (3.0 * SF[lvl]).astype(np.float32)
    q["min_level"] = lvl - 1
    q["max_level"] = lvl
    qdesc = synth.noisy_copy(rng, dR[src], 45)
    inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
    gR = oracle.Grid(kR, *BOUNDS)
    out_o, n_o = oracle.fuse_search(gR, dR, None, q["x"], q["y"], q["xr"], q["radius"], lvl, qdesc, inv_sigma2)
    fR = A.FrameFeatures(kR, dR, None, BOUNDS)
    out, nf = A.ORBmatcher().FuseIn(fR, q, qdesc, inv_sigma2)
    assert nf == n_o and np.array_equal(out, out_o)
    assert nf > 100
    best_idx = np.where(out >= 0, out + len(kL), -1)       # bestIdx as the reference reports it
    assert best_idx.max() >= len(kL)


def _rot(rng, scale):
    w = rng.normal(0, scale, 3)
    th = np.linalg.norm(w)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _two_camera_keyframes(rng, oracle, n_pts=900):
    """Two keyframes with two KannalaBrandt8 cameras each looking at one cloud of points; keypoints are the
    projections (+ noise), descriptors noisy copies of one descriptor per point, feature vectors put a point's
    observations into the same vocabulary node.  Returns what the search takes plus the four relative poses."""
    X = np.column_stack([rng.uniform(-3, 3, n_pts), rng.uniform(-2, 2, n_pts), rng.uniform(2, 9, n_pts)])
    # camera poses Tcw (R, t): x_c = R x_w + t
    Tlr = (_rot(rng, 0.01), np.array([-0.101, 0.002, 0.001]))                 # right camera from left camera
    kf = []
    for k in range(2):
        Rl, tl = (np.eye(3), np.zeros(3)) if k == 0 else (_rot(rng, 0.05), np.array([0.35, -0.05, 0.12]))
        Rr, tr = Tlr[0] @ Rl, Tlr[0] @ tl + Tlr[1]
        kf.append(((Rl, tl), (Rr, tr)))
    base = synth.random_descriptors(n_pts, 4242)
    node = rng.integers(0, 60, n_pts)
    out = []
    for k in range(2):
        rows = []
        for cam, (R, t), prm in ((0, kf[k][0], KB_L), (1, kf[k][1], KB_R)):
            seen = np.nonzero(rng.random(n_pts) < 0.55)[0]
            for p in seen:
                xc = (R @ X[p] + t).astype(np.float32)
                uv = oracle.camera_project(1, prm, xc)
                if not (20 < uv[0] < W - 20 and 20 < uv[1] < H - 20):
                    continue
                noise = rng.normal(0, 0.4, 2) if rng.random() < 0.8 else rng.normal(0, 6.0, 2)   # some bad observations
                rows.append((cam, p, uv[0] + noise[0], uv[1] + noise[1]))
        rows.sort(key=lambda r: r[0])                      # left keypoints first
        n_left = sum(1 for r in rows if r[0] == 0)
        kps = np.zeros(len(rows), synth.KP_DTYPE)
        kps["x"] = [r[2] for r in rows]
        kps["y"] = [r[3] for r in rows]
        kps["octave"] = rng.integers(0, 4, len(rows))
        kps["angle"] = rng.uniform(0, 360, len(rows)).astype(np.float32)
        pid = np.array([r[1] for r in rows])
        desc = synth.noisy_copy(rng, base[pid], 25)
        nd = node[pid]
        order = np.argsort(nd, kind="stable").astype(np.int32)
        off = np.zeros(61, np.int32)
        np.add.at(off, nd + 1, 1)
        free = (rng.random(len(rows)) < 0.9).astype(np.uint8)
        out.append((kps, desc, free, n_left, (order, np.cumsum(off).astype(np.int32)), pid))
    # relative poses T12 = T1w * Tw2 for the four camera pairs: x_1 = R12 x_2 + t12
    poses = np.zeros((4, 12), np.float32)
    for a in range(2):
        for b in range(2):
            R1, t1 = kf[0][a]
            R2, t2 = kf[1][b]
            R12 = R1 @ R2.T
            t12 = t1 - R12 @ t2
            poses[2 * a + b, :9] = R12.reshape(9)
            poses[2 * a + b, 9:] = t12
    return out[0], out[1], poses


@pytest.mark.parametrize("coarse,check_ori", [(False, False), (True, False), (False, True)])
def test_search_for_triangulation_camera_pairs(oracle, coarse, check_ori):
    A = _gpu()
    rng = np.random.default_rng(31)
    (k1, d1, f1, nl1, fv1, pid1), (k2, d2, f2, nl2, fv2, pid2), poses = _two_camera_keyframes(rng, oracle)
    assert 0 < nl1 < len(k1) and 0 < nl2 < len(k2)
    cam_types = [1, 1, 1, 1]
    cam_params = np.stack([KB_L, KB_R, KB_L, KB_R])
    sig2 = (SF * SF).astype(np.float32)
    args = (k1, d1, f1, None, nl1, fv1[0], fv1[1], k2, d2, f2, None, nl2, fv2[0], fv2[1])
    out_o, n_o, margin = oracle.search_for_triangulation_cameras(*args, cam_types, cam_params, poses, (0.0, 0.0), SF, sig2,
                                                                 sig2, False, coarse, check_ori, with_margin=True)
    out, n = A.ORBmatcher(0.6, check_ori).SearchForTriangulationCameras(
        k1, d1, f1, None, nl1, fv1, k2, d2, f2, None, nl2, fv2, cam_types, cam_params, poses, (0.0, 0.0), SF, sig2, sig2,
        False, coarse)
    # The oracle's gates are an independent binary64 restatement (oracle/camera_oracle.cpp); a keypoint's outcome may
    # differ from it only if one of the gates evaluated for it lay inside the rounding band of its threshold
    # (margin < 1; one keypoint of this fixture, at 0.94 of the band).
    in_band = margin < 1
    print(f"triangulation gates: {int(in_band.sum())} of {len(margin)} keypoints had a gate inside its rounding band")
    differ = out != out_o
    assert not (differ & ~in_band).any(), "a match differs although no gate of that keypoint was within rounding of its threshold"
    if not check_ori:
        assert n - int(differ.sum()) <= n_o <= n + int(differ.sum())
    if not differ.any():
        assert n == n_o
    # the same search on keyframes whose (combined) descriptor arrays are resident on the device
    from orb_slam3_amd.matcher import triangulation_geometry
    kf1, kf2 = A.FrameFeatures(k1, d1, None, BOUNDS), A.FrameFeatures(k2, d2, None, BOUNDS)
    geom = triangulation_geometry(nl1, nl2, cam_types, cam_params, poses, (0.0, 0.0), SF, sig2, sig2)
    out_f, n_f = A.ORBmatcher(0.6, check_ori).SearchForTriangulationIn(kf1, f1, fv1, kf2, f2, fv2, geom, False, coarse)
    assert n_f == n and np.array_equal(out_f, out)
    m = np.nonzero(out >= 0)[0]
    assert len(m) > 60
    # all four camera pairs occur among the matches
    pairs = {(int(i >= nl1), int(out[i] >= nl2)) for i in m}
    assert pairs == {(0, 0), (0, 1), (1, 0), (1, 1)}
    if not coarse and not check_ori:
        # the gate does work: it removes matches that the descriptor test alone accepts, and what it keeps is
        # overwhelmingly the same 3-D point seen twice
        out_c, n_c = A.ORBmatcher(0.6, False).SearchForTriangulationCameras(
            k1, d1, f1, None, nl1, fv1, k2, d2, f2, None, nl2, fv2, cam_types, cam_params, poses, (0.0, 0.0), SF, sig2,
            sig2, False, True)
        assert n_c > n
        assert (pid1[m] == pid2[out[m]]).mean() > 0.9


def test_search_for_triangulation_cameras_single_camera_equals_existing_entry_point(oracle):
    """n_left = -1 on both sides with pinhole models reproduces orbx_search_for_triangulation given the same F12."""
    A = _gpu()
    rng = np.random.default_rng(41)
    n = 600
    pin = np.array([435.2, 435.2, 367.2, 252.2, 0, 0, 0, 0], np.float32)
    R12 = _rot(rng, 0.04).astype(np.float32)
    t12 = np.array([0.3, 0.02, -0.05], np.float32)
    X2 = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 8, n)]).astype(np.float32)
    X1 = (X2 @ R12.T + t12).astype(np.float32)
    k1 = np.zeros(n, synth.KP_DTYPE); k2 = np.zeros(n, synth.KP_DTYPE)
    for i in range(n):
        u1 = oracle.camera_project(0, pin, X1[i]); u2 = oracle.camera_project(0, pin, X2[i])
        k1["x"][i], k1["y"][i] = u1 + rng.normal(0, 0.5, 2)
        k2["x"][i], k2["y"][i] = u2 + rng.normal(0, 2.5, 2)
    k1["octave"] = rng.integers(0, 8, n); k2["octave"] = rng.integers(0, 8, n)
    k1["angle"] = rng.uniform(0, 360, n); k2["angle"] = rng.uniform(0, 360, n)
    base = synth.random_descriptors(n, 99)
    d1, d2 = synth.noisy_copy(rng, base, 20), synth.noisy_copy(rng, base, 20)
    node = rng.integers(0, 40, n)
    order = np.argsort(node, kind="stable").astype(np.int32)
    off = np.zeros(41, np.int32); np.add.at(off, node + 1, 1); off = np.cumsum(off).astype(np.int32)
    fv = (order, off)
    free = np.ones(n, np.uint8)
    sig2 = (SF * SF).astype(np.float32)
    ep = oracle.camera_project(0, pin, (R12.T @ (-t12)).astype(np.float32))      # camera centre 1 in image 2
    poses = np.zeros((4, 12), np.float32); poses[0, :9] = R12.reshape(9); poses[0, 9:] = t12
    m = A.ORBmatcher(0.6, False)
    out, nm = m.SearchForTriangulationCameras(k1, d1, free, None, -1, fv, k2, d2, free, None, -1, fv, [0, 0, 0, 0],
                                              np.stack([pin] * 4), poses, ep, SF, sig2, sig2)
    out_o, n_o, margin = oracle.search_for_triangulation_cameras(k1, d1, free, None, -1, fv[0], fv[1], k2, d2, free, None, -1,
                                                                 fv[0], fv[1], [0, 0, 0, 0], np.stack([pin] * 4), poses, ep,
                                                                 SF, sig2, sig2, False, False, False, with_margin=True)
    differ = out != out_o
    assert not (differ & ~(margin < 1)).any()           # Pinhole::epipolarConstrain against the binary64 checker
    assert differ.sum() <= 3 and abs(nm - n_o) <= differ.sum()
    # F12 = K1^-T [t12]x R12 K2^-1 in double, rounded: the existing entry point agrees wherever the line distance
    # is not within rounding of its threshold
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]], np.float64)
    K = np.array([[pin[0], 0, pin[2]], [0, pin[1], pin[3]], [0, 0, 1]], np.float64)
    F12 = (np.linalg.inv(K.T) @ tx @ R12.astype(np.float64) @ np.linalg.inv(K)).astype(np.float32)
    out2, nm2 = m.SearchForTriangulation(k1, d1, free, None, fv, k2, d2, free, None, fv, F12, ep, SF, sig2)
    assert (out == out2).mean() > 0.98
    assert nm > 100


def test_compute_stereo_fisheye_matches(oracle):
    """Frame::ComputeStereoFishEyeMatches as one call (Frame.cc:1286-1338): brute-force search of the lapping-area
    descriptors, ratio test, KannalaBrandt8::TriangulateMatches, on a synthetic two-camera scene (3-D points seen by
    both cameras, a fifth of them with a wrong right position, distractor descriptors, mono keypoints in front)."""
    A = _gpu()
    from orb_slam3_amd.matcher import CameraModel, compute_stereo_fisheye_matches
    rng = np.random.default_rng(21)
    n, mono_l, mono_r, extra = 600, 50, 30, 200
    cam = np.array([190.98, 190.97, 254.93, 256.9, 0.0034, 0.0007, -0.0021, 0.0002], np.float32)
    cam2 = np.array([190.44, 190.43, 252.6, 254.9, 0.0034, 0.0018, -0.0027, 0.0004], np.float32)
    ang = 0.02
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    t = np.array([0.101, 0.0002, -0.001], np.float32)
    P = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(1, 6, n)], 1).astype(np.float32)
    Pr = (P - t) @ R                                            # x_r = R^T (x_l - t)
    uvL = np.array([oracle.camera_project(1, cam, p) for p in P], np.float32)
    uvR = np.array([oracle.camera_project(1, cam2, p) for p in Pr], np.float32)
    uvR += rng.normal(0, 0.3, uvR.shape).astype(np.float32)
    bad = rng.random(n) < 0.2
    uvR[bad] = rng.uniform(20, 490, (int(bad.sum()), 2)).astype(np.float32)

    def kps(uv, count):
        k = np.zeros(count, A.KP_DTYPE)
        k["x"][:len(uv)] = uv[:, 0]; k["y"][:len(uv)] = uv[:, 1]
        k["octave"] = rng.integers(0, 8, count)
        k["class_id"] = -1
        return k
    dL = synth.random_descriptors(n, 31)
    dR = dL.copy()
    flips = rng.integers(0, 256, (n, 12))
    for i in range(n):
        for b in flips[i, :rng.integers(0, 12)]:
            dR[i, b >> 3] ^= np.uint8(1 << (b & 7))
    perm = rng.permutation(n + extra)                           # right keypoints in another order, with distractors
    kR_all = kps(np.concatenate([uvR, rng.uniform(20, 490, (extra, 2)).astype(np.float32)]), n + extra)[perm]
    dR_all = np.concatenate([dR, synth.random_descriptors(extra, 32)])[perm]
    kL_full = np.concatenate([kps(rng.uniform(20, 490, (mono_l, 2)).astype(np.float32), mono_l), kps(uvL, n)])
    dL_full = np.concatenate([synth.random_descriptors(mono_l, 33), dL])
    kR_full = np.concatenate([kps(rng.uniform(20, 490, (mono_r, 2)).astype(np.float32), mono_r), kR_all])
    dR_full = np.concatenate([synth.random_descriptors(mono_r, 34), dR_all])
    dR_full[mono_r + 5] = dR_full[mono_r + 9]                   # two equal right descriptors: the lower index wins or the ratio fails
    sigma2 = (np.float32(1.2) ** (2 * np.arange(8))).astype(np.float32)
    want = oracle.stereo_fisheye_matches(kL_full, dL_full, mono_l, kR_full, dR_full, mono_r, cam, cam2, R, t, sigma2,
                                         with_margin=True)
    c1, c2 = CameraModel(1, (C.c_float * 8)(*cam)), CameraModel(1, (C.c_float * 8)(*cam2))
    got = compute_stereo_fisheye_matches(kL_full, dL_full, mono_l, kR_full, dR_full, mono_r, c1, c2, R, t, sigma2)
    # The oracle triangulates in binary64 with its own null-vector routine (oracle/camera_oracle.cpp): the match
    # tables are identical wherever no gate lay within rounding of its threshold (margin >= 1), depths and points
    # agree to 5e-3 relative (binary32 SVD in the library, as in the reference).
    in_band = want[5] < 1
    print(f"fisheye stereo: {int(in_band.sum())} of {int((want[5] < 1e299).sum())} triangulated pairs inside a rounding band")
    differ = got[0] != want[0]
    assert not (differ & ~in_band).any()
    assert abs(got[4] - want[4]) <= differ.sum() and 200 < got[4] < n
    if not differ.any():
        assert np.array_equal(got[1], want[1])
    ok = (got[0] >= 0) & ~differ
    assert np.allclose(got[2][ok], want[2][ok], rtol=5e-3)
    assert np.allclose(got[3][ok], want[3][ok], rtol=5e-3, atol=5e-3)
    assert (got[2][~(got[0] >= 0)] == -1).all()
    assert (got[0][:mono_l] == -1).all() and (got[1][:mono_r] == -1).all()
    # the triangulated points are the scene's points (camera-1 coordinates) up to the pixel noise
    src = np.nonzero(ok)[0] - mono_l
    assert np.median(np.linalg.norm(got[3][ok] - P[src], axis=1) / P[src][:, 2]) < 0.05
    # nothing to match: every output is "none"
    e = compute_stereo_fisheye_matches(kL_full, dL_full, len(kL_full), kR_full, dR_full, mono_r, c1, c2, R, t, sigma2)
    assert e[4] == 0 and (e[0] == -1).all() and (e[1] == -1).all() and (e[2] == -1).all()
