"""Known answers for the camera-model gates of SearchForTriangulation (Pinhole / KannalaBrandt8 epipolarConstrain,
Pinhole.cpp:118-141, KannalaBrandt8.cpp:117-146, :227-232, :323-397) as restated for the oracle and the library
(one shared header: parity with the reference's Eigen / libm arithmetic cannot be pinned to the bit, so the gates are
pinned by their geometry instead)."""
import numpy as np

KB = np.array([190.978, 190.973, 254.93, 256.90, 0.0034, 0.00076, -0.0021, 0.00024], np.float32)
PIN = np.array([435.2, 435.2, 367.2, 252.2, 0, 0, 0, 0], np.float32)


def test_project_unproject_round_trip(oracle):
    rng = np.random.default_rng(0)
    for typ, prm in ((0, PIN), (1, KB)):
        for _ in range(200):
            X = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1, 6)], np.float32)
            uv = oracle.camera_project(typ, prm, X)
            ray = oracle.camera_unproject(typ, prm, uv)
            assert ray[2] == 1.0
            assert np.allclose(ray[:2], X[:2] / X[2], atol=2e-4)


def test_kb8_projection_of_the_optical_axis_and_a_known_angle(oracle):
    uv = oracle.camera_project(1, KB, np.array([0, 0, 1], np.float32))
    assert np.allclose(uv, KB[2:4])
    # a ray at 45 degrees in the x-z plane: r = theta + k0 theta^3 + ..., u = fx r + cx
    th = np.float64(np.pi / 4)
    r = th + KB[4] * th ** 3 + KB[5] * th ** 5 + KB[6] * th ** 7 + KB[7] * th ** 9
    uv = oracle.camera_project(1, KB, np.array([1, 0, 1], np.float32))
    assert abs(uv[0] - (KB[0] * r + KB[2])) < 1e-3 and abs(uv[1] - KB[3]) < 1e-4


def _pair(rng, typ, prm, X1, R12, t12):
    X2 = (R12.T @ (X1 - t12)).astype(np.float32)
    return X2


def test_triangulate_matches_codes(oracle):
    R = np.eye(3, dtype=np.float32)
    t = np.array([0.1, 0, 0], np.float32)              # x1 = R12 x2 + t12
    X1 = np.array([0.3, -0.2, 2.0], np.float32)
    X2 = _pair(None, 1, KB, X1, R, t)
    p1, p2 = oracle.camera_project(1, KB, X1), oracle.camera_project(1, KB, X2)
    z = oracle.kb8_triangulate_matches(KB, KB, p1, p2, R, t, 1.0, 1.0)
    assert abs(z - 2.0) < 1e-3                                          # depth in camera 1
    assert oracle.epipolar_constrain(1, KB, 1, KB, p1, p2, R, t, 1.0, 1.0)
    # same ray in both cameras (a point at infinity): not enough parallax
    assert oracle.kb8_triangulate_matches(KB, KB, p1, p1, R, t, 1.0, 1.0) == -1
    # the observation in image 2 moved along the baseline direction past the epipole: negative depth
    far = oracle.camera_project(1, KB, np.array([0.6, -0.2, 2.0], np.float32))
    assert oracle.kb8_triangulate_matches(KB, KB, p1, far, R, t, 1.0, 1.0) in (-2.0, -3.0)
    # moved across the epipolar line by 6 px: reprojection error in image 1 or 2
    off = p2 + np.float32([0, 6])
    assert oracle.kb8_triangulate_matches(KB, KB, p1, off, R, t, 1.0, 1.0) in (-4.0, -5.0)
    assert not oracle.epipolar_constrain(1, KB, 1, KB, p1, off, R, t, 1.0, 1.0)
    # ... which a coarse pyramid level tolerates (sigma^2 of level 7 = 12.8)
    assert oracle.kb8_triangulate_matches(KB, KB, p1, off, R, t, 12.8, 12.8) > 0


def test_pinhole_epipolar_distance_threshold(oracle):
    rng = np.random.default_rng(3)
    R = np.eye(3, dtype=np.float32)
    t = np.array([0.2, 0, 0], np.float32)              # pure x translation: epipolar lines are image rows
    X1 = np.array([0.4, 0.1, 3.0], np.float32)
    X2 = _pair(rng, 0, PIN, X1, R, t)
    p1, p2 = oracle.camera_project(0, PIN, X1), oracle.camera_project(0, PIN, X2)
    assert abs(p1[1] - p2[1]) < 1e-3
    # dsqr = dy^2 < 3.84 sigma^2
    for dy, sig2, expect in ((1.9, 1.0, True), (2.0, 1.0, False), (3.0, 2.9859846, True), (3.5, 2.9859846, False)):
        q = p2 + np.float32([rng.uniform(-20, 20), dy])
        assert oracle.epipolar_constrain(0, PIN, 0, PIN, p1, q, R, t, 1.0, sig2) == expect


def test_stereo_fisheye_matches_recovers_the_scene(oracle):
    """The oracle's Frame::ComputeStereoFishEyeMatches on exact projections of known 3-D points: every pair is matched
    and triangulated back to its point; a right keypoint moved off its ray is rejected by the reprojection gate."""
    rng = np.random.default_rng(5)
    n = 60
    cam = np.array([190.98, 190.97, 254.93, 256.9, 0.0034, 0.0007, -0.0021, 0.0002], np.float32)
    R = np.eye(3, dtype=np.float32)
    t = np.array([0.1, 0.0, 0.0], np.float32)
    P = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1, 4, n)], 1).astype(np.float32)
    uvL = np.array([oracle.camera_project(1, cam, p) for p in P], np.float32)
    uvR = np.array([oracle.camera_project(1, cam, p - t) for p in P], np.float32)
    uvR[7] += np.float32(25.0)                                  # off the epipolar geometry
    KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    kL, kR = np.zeros(n, KP), np.zeros(n, KP)
    kL["x"], kL["y"], kR["x"], kR["y"] = uvL[:, 0], uvL[:, 1], uvR[:, 0], uvR[:, 1]
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sigma2 = np.ones(8, np.float32)
    l2r, r2l, depth, p3d, nm = oracle.stereo_fisheye_matches(kL, d, 0, kR, d, 0, cam, cam, R, t, sigma2)
    assert nm == n - 1 and l2r[7] == -1 and depth[7] == -1
    ok = l2r >= 0
    assert np.array_equal(l2r[ok], np.nonzero(ok)[0]) and np.array_equal(r2l[ok], np.nonzero(ok)[0])
    assert np.allclose(p3d[ok], P[ok], rtol=2e-3, atol=2e-3) and np.allclose(depth[ok], P[ok][:, 2], rtol=2e-3)
