"""The library's camera-model gates (host arithmetic in binary32, orb_slam3_amd/csrc/orbx_camera_gates.h, reached through
the C ABI: orbx_camera_project / _unproject / _triangulate_matches / _epipolar_constrain) against the oracle's independent
binary64 restatement (oracle/camera_oracle.cpp) of KannalaBrandt8::TriangulateMatches / epipolarConstrain
(KannalaBrandt8.cpp:227-232, :323-430) and Pinhole::epipolarConstrain (Pinhole.cpp:118-141).

Bar: the accept / reject decision and the rejection code are IDENTICAL for every pair whose tested quantities all lie
outside the rounding band of their threshold (oracle margin >= 1; bands: |cos - 0.9998| < 2e-6, |z| / |X| < 1e-4,
|e^2 - 5.991 s| / (5.991 s) < 2e-3, |dsqr - 3.84 s| / (3.84 s) < 5e-3); in-band pairs are counted and must stay below
1 % of the sweep; accepted depths agree to 5e-3 relative (binary32 SVD at parallaxes down to 1.15 degrees) and
projections to 2e-3 px.  No GPU involved: these entry points run on the calling thread."""
import numpy as np
import pytest

from orb_slam3_amd import matcher as M

KB_L = np.array([190.978, 190.973, 254.93, 256.90, 0.0034, 0.00076, -0.0021, 0.00024], np.float32)
KB_R = np.array([190.442, 190.434, 252.60, 254.92, 0.0034, 0.00071, -0.0020, 0.00020], np.float32)
PIN = np.array([435.2, 435.2, 367.2, 252.2, 0, 0, 0, 0], np.float32)
SIGMA2 = (np.float32(1.2) ** (2 * np.arange(8))).astype(np.float32)


def _rot(rng, s):
    w = rng.normal(0, s, 3)
    th = np.linalg.norm(w)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _scene(seed, typ, n):
    """n pairs: projections of one point cloud into two cameras, the second observation displaced by a mixture of
    0.2 .. 6 px noise so that every gate (parallax, depth signs, both reprojection tests) fires often."""
    rng = np.random.default_rng(seed)
    p1, p2 = (KB_L, KB_R) if typ else (PIN, PIN)
    R12 = _rot(rng, 0.05).astype(np.float32)
    t12 = rng.normal(0, 0.2, 3).astype(np.float32)
    X2 = np.column_stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(0.5, 30, n)]).astype(np.float32)
    X1 = (X2 @ R12.T.astype(np.float64) + t12).astype(np.float32)
    c1, c2 = M.camera_model(typ, p1), M.camera_model(typ, p2)
    uv1, uv2 = M.camera_project(c1, X1), M.camera_project(c2, X2)
    uv2 = (uv2 + rng.normal(0, 1, (n, 2)) * rng.choice([0.2, 1.0, 2.5, 6.0], n)[:, None]).astype(np.float32)
    s1, s2 = SIGMA2[rng.integers(0, 8, n)], SIGMA2[rng.integers(0, 8, n)]
    return p1, p2, c1, c2, R12, t12, uv1, uv2, s1, s2, X1


@pytest.mark.parametrize("seed", range(6))
def test_kannala_brandt_gates_equal_the_binary64_checker_outside_the_band(oracle, seed):
    n = 20000
    p1, p2, c1, c2, R12, t12, uv1, uv2, s1, s2, _ = _scene(seed, 1, n)
    z, P = M.camera_triangulate_matches(c1, c2, uv1, uv2, R12, t12, s1, s2)
    ok = M.camera_epipolar_constrain(c1, c2, uv1, uv2, R12, t12, s1, s2)
    o = oracle.camera_gates_many(1, p1, 1, p2, uv1, uv2, R12, t12, s1, s2)
    code = np.where(z > 0, 1.0, z)
    code_o = np.where(o["depth"] > 0, 1.0, o["depth"])
    differ = code != code_o
    in_band = o["tri_margin"] < 1
    assert not (differ & ~in_band).any(), "decision differs outside the rounding band"
    assert not ((ok != o["accepted"]) & ~(o["epi_margin"] < 1)).any()
    assert np.array_equal(ok, z > 0.0001)                          # epipolarConstrain is TriangulateMatches > 0.0001
    assert in_band.mean() < 0.01
    acc = (z > 0) & (o["depth"] > 0)
    assert acc.sum() > 2000 and (code_o == -1).sum() > 500 and (code_o <= -4).sum() > 500    # every kind of outcome occurs
    assert (np.abs(z[acc] - o["depth"][acc]) / o["depth"][acc]).max() < 5e-3
    assert (np.linalg.norm(P[acc] - o["p3d"][acc], axis=1) / np.linalg.norm(o["p3d"][acc], axis=1)).max() < 5e-3
    print(f"seed {seed}: {n} pairs, {int(acc.sum())} accepted, {int(in_band.sum())} in band, {int(differ.sum())} of them decided differently")


@pytest.mark.parametrize("seed", range(6))
def test_pinhole_epipolar_gate_equals_the_binary64_checker_outside_the_band(oracle, seed):
    n = 20000
    p1, p2, c1, c2, R12, t12, uv1, uv2, s1, s2, _ = _scene(100 + seed, 0, n)
    ok = M.camera_epipolar_constrain(c1, c2, uv1, uv2, R12, t12, s1, s2)
    o = oracle.camera_gates_many(0, p1, 0, p2, uv1, uv2, R12, t12, s1, s2)
    in_band = o["epi_margin"] < 1
    assert not ((ok != o["accepted"]) & ~in_band).any()
    assert in_band.mean() < 0.01 and 0.2 < ok.mean() < 0.95
    print(f"seed {seed}: {n} pairs, {int(ok.sum())} accepted, {int(in_band.sum())} in band, {int((ok != o['accepted']).sum())} of them decided differently")


def test_project_and_unproject_agree_with_the_binary64_checker(oracle):
    rng = np.random.default_rng(7)
    n = 5000
    X = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), rng.uniform(0.3, 10, n)]).astype(np.float32)
    for typ, prm in ((0, PIN), (1, KB_L)):
        c = M.camera_model(typ, prm)
        uv = M.camera_project(c, X)
        uv_o = np.array([oracle.camera_project(typ, prm, x) for x in X])
        assert np.abs(uv - uv_o).max() < 2e-3
        rays = M.camera_unproject(c, uv)
        rays_o = np.array([oracle.camera_unproject(typ, prm, p) for p in uv])
        assert np.abs(rays - rays_o).max() < 5e-5 * np.abs(rays_o).max()
        assert np.abs(rays[:, :2] - X[:, :2] / X[:, 2:3]).max() < 2e-4 * max(1.0, np.abs(rays).max())


def test_bad_camera_model_is_refused():
    c = M.camera_model(7, PIN)
    with pytest.raises(Exception):
        M.camera_project(c, [[0, 0, 1]])
