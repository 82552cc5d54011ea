"""Device-resident frame features: the grid of Frame::AssignFeaturesToGrid built on the GPU and the
windowed searches run on it, against the CPU oracle.  Bar: identical grids, candidate lists (order
included) and match arrays."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

BOUNDS = (0.0, 752.0, 0.0, 480.0)


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _clustered(rng, n):
    """Keypoints with heavy clusters (long cells), points on cell borders and points outside the bounds."""
    kps, desc = synth.matcher_frame(rng, n, synth.random_descriptors(200, 9), 30)
    k = n // 4
    kps["x"][:k] = rng.normal(300, 6, k).astype(np.float32)          # one dense blob: cells with dozens of entries
    kps["y"][:k] = rng.normal(200, 6, k).astype(np.float32)
    kps["x"][k:k + 50] = (np.arange(50) * 11.75 + 5.875).astype(np.float32)   # x on cell boundaries (.5 cells)
    kps["x"][k + 50:k + 60] = np.float32(-7.0)                       # left of the grid: round(-0.6) = -1
    kps["x"][k + 60:k + 70] = np.float32(751.9)                      # rounds to column 64: outside
    kps["y"][k + 70:k + 80] = np.float32(479.9)                      # rounds to row 48: outside
    kps["x"][k + 80:k + 90] = np.float32(-5.0)                       # round(-0.43) = 0: inside
    return kps, desc


@pytest.mark.parametrize("seed,n,bounds", [(1, 1500, BOUNDS), (2, 5000, BOUNDS), (3, 37, BOUNDS), (4, 0, BOUNDS),
                                           (5, 3000, (-12.5, 770.25, -9.75, 492.0))])
def test_grid_built_on_device_equals_assign_features_to_grid(oracle, seed, n, bounds):
    A = _gpu()
    rng = np.random.default_rng(seed)
    if n:
        kps, desc = _clustered(rng, n) if n >= 200 else synth.matcher_frame(rng, n, None, 30)
    else:
        kps, desc = np.zeros(0, A.KP_DTYPE), np.zeros((0, 32), np.uint8)
    f = A.FrameFeatures(kps, desc, None, bounds)
    assert len(f) == n
    start, idx = f.grid()
    start_o, idx_o = oracle.Grid(kps, *bounds).csr()
    assert np.array_equal(start, start_o)
    assert np.array_equal(idx, idx_o)
    if n >= 200:
        assert np.diff(start).max() > 16                          # long cells
        assert (len(idx) < n) == (bounds == BOUNDS)               # some keypoints fall outside the image-sized grid


def test_frame_searches_equal_host_array_searches_and_oracle(oracle):
    A = _gpu()
    rng = np.random.default_rng(11)
    kps, desc = _clustered(rng, 2000)
    ur = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(1, 40, len(kps)), -1).astype(np.float32)
    f = A.FrameFeatures(kps, desc, ur, BOUNDS)
    g = oracle.Grid(kps, *BOUNDS)
    m = A.ORBmatcher(0.8, True)
    # several searches of the same frame, as Tracking does (TrackWithMotionModel, then SearchLocalPoints)
    for rep in range(3):
        q, lvl, src = synth.projected_queries(rng, kps, 700 + 100 * rep, 7.0, 6.0)
        q["xr"] = q["x"] - rng.uniform(1, 40, len(q)).astype(np.float32)
        qdesc = synth.noisy_copy(rng, desc[src], 60)
        # candidate lists: order and distances
        off, ci, cd = f.window_candidates(q, qdesc)
        off2, ci2, cd2 = A.window_candidates(kps, desc, ur, BOUNDS, q, qdesc)
        assert np.array_equal(off, off2) and np.array_equal(ci, ci2) and np.array_equal(cd, cd2)
        for k in range(0, len(q), 97):
            ref = g.features_in_area(float(q["x"][k]), float(q["y"][k]), float(q["radius"][k]), int(q["min_level"][k]),
                                     int(q["max_level"][k]))
            mine = ci[off[k]:off[k + 1]]
            # the stereo gate only removes entries; the order of the rest is the grid order
            assert np.array_equal(mine, ref[np.isin(ref, mine)])
        # SearchByProjection(Frame, MapPoints)
        taken0 = (rng.random(len(kps)) < 0.2).astype(np.uint8)
        t1, t2 = taken0.copy(), taken0.copy()
        out1, n1 = m.SearchByProjectionIn(f, q, qdesc, t1)
        out2, n2 = m.SearchByProjection(kps, desc, ur, BOUNDS, q, qdesc, t2)
        out_o, taken_o, n_o = oracle.search_by_projection(g, desc, ur, q["x"], q["y"], q["xr"], q["max_level"],
                                                          q["radius"], qdesc, taken0, m.mfNNratio)
        assert n1 == n2 == n_o and n1 > 30
        assert np.array_equal(out1, out2) and np.array_equal(out1, out_o)
        assert np.array_equal(t1, t2) and np.array_equal(t1, taken_o)
        # SearchByProjection(Cur, Last) loop with rotation histogram
        qangle = ((kps["angle"][src] + rng.normal(40, 25, len(q))) % 360).astype(np.float32)
        t1, t2 = taken0.copy(), taken0.copy()
        r1 = m.SearchWindowBestIn(f, q, qdesc, 100.0, t1, None, qangle)
        r2 = m.SearchWindowBest(kps, desc, ur, BOUNDS, q, qdesc, 100.0, t2, None, qangle)
        kpm_o, tk_o, nb_o = oracle.search_window_best(g, desc, ur, q["x"], q["y"], q["xr"], q["radius"], q["min_level"],
                                                      q["max_level"], qdesc, 100.0, taken0, None, qangle, 1.0 / 30)
        assert r1[2] == r2[2] == nb_o and np.array_equal(r1[1], r2[1]) and np.array_equal(r1[1], kpm_o)
        assert np.array_equal(r1[0], r2[0]) and np.array_equal(t1, t2) and np.array_equal(t1, tk_o)
    # Fuse
    sig = (1.0 / (synth.SCALE_FACTORS ** 2)).astype(np.float32)
    q, lvl, src = synth.projected_queries(rng, kps, 900, 3.0, 1.5)
    q["xr"] = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.5, len(q)), q["x"] - 10).astype(np.float32)
    qdesc = synth.noisy_copy(rng, desc[src], 40)
    o1, n1 = m.FuseIn(f, q, qdesc, sig)
    o2, n2 = m.Fuse(kps, desc, ur, BOUNDS, q, qdesc, sig)
    assert n1 == n2 and np.array_equal(o1, o2) and n1 > 30


def test_frame_reassign_and_growing_lists(oracle):
    """One handle reused for frames of different sizes; a search whose lists outgrow the staging arena."""
    A = _gpu()
    rng = np.random.default_rng(21)
    f = A.FrameFeatures()
    for n in (3000, 200, 4500):
        kps, desc = _clustered(rng, n)
        f.assign(kps, desc, None, BOUNDS)
        start, idx = f.grid()
        start_o, idx_o = oracle.Grid(kps, *BOUNDS).csr()
        assert np.array_equal(start, start_o) and np.array_equal(idx, idx_o)
    # every query covers the whole image: n_query * n candidates (>> the first-guess capacity)
    q = np.zeros(400, A.QUERY_DTYPE)
    q["x"], q["y"], q["radius"], q["min_level"], q["max_level"] = 376.0, 240.0, 2000.0, -1, -1
    qdesc = synth.random_descriptors(len(q), 5)
    off, ci, cd = f.window_candidates(q, qdesc)
    inside = len(idx)
    assert np.array_equal(np.diff(off), np.full(len(q), inside))
    assert np.array_equal(ci[:inside], idx)            # whole-grid order = the CSR order
    k = 123
    ref = np.unpackbits(desc[ci[off[k]:off[k + 1]]] ^ qdesc[k], axis=1).sum(1)
    assert np.array_equal(cd[off[k]:off[k + 1]], ref)


def test_frame_searched_from_two_threads(oracle):
    A = _gpu()
    rng = np.random.default_rng(31)
    kps, desc = _clustered(rng, 2500)
    f = A.FrameFeatures(kps, desc, None, BOUNDS)
    jobs = []
    for t in range(2):
        q, lvl, src = synth.projected_queries(np.random.default_rng(40 + t), kps, 800, 7.0, 6.0)
        jobs.append((q, synth.noisy_copy(np.random.default_rng(50 + t), desc[src], 50)))
    want = [A.window_candidates(kps, desc, None, BOUNDS, q, qd) for q, qd in jobs]
    got = [[None] * 20 for _ in jobs]

    def work(t):
        for r in range(20):
            got[t][r] = f.window_candidates(*jobs[t])

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    for t in range(2):
        for r in range(20):
            assert all(np.array_equal(a, b) for a, b in zip(got[t][r], want[t]))


def test_frame_from_device_arrays(oracle):
    A = _gpu()
    import torch
    rng = np.random.default_rng(41)
    kps, desc = _clustered(rng, 1800)
    ur = np.where(rng.random(len(kps)) < 0.5, kps["x"] - 10, -1).astype(np.float32)
    dk = torch.from_numpy(kps.view(np.uint8).copy()).cuda()
    dd = torch.from_numpy(desc.copy()).cuda()
    du = torch.from_numpy(ur.copy()).cuda()
    torch.cuda.synchronize()
    f = A.FrameFeatures().assign_device(dk, dd, du, len(kps), BOUNDS)
    start, idx = f.grid()
    start_o, idx_o = oracle.Grid(kps, *BOUNDS).csr()
    assert np.array_equal(start, start_o) and np.array_equal(idx, idx_o)
    q, lvl, src = synth.projected_queries(rng, kps, 600, 7.0, 6.0)
    q["xr"] = q["x"] - 10
    qdesc = synth.noisy_copy(rng, desc[src], 50)
    a = f.window_candidates(q, qdesc)
    b = A.window_candidates(kps, desc, ur, BOUNDS, q, qdesc)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    m = A.ORBmatcher(0.8, True)
    taken = np.zeros(len(kps), np.uint8)
    out, n = m.SearchByProjectionIn(f, q, qdesc, taken)      # the level of the best / second best comes from the host mirror
    out2, n2 = m.SearchByProjection(kps, desc, ur, BOUNDS, q, qdesc, np.zeros(len(kps), np.uint8))
    assert n == n2 and np.array_equal(out, out2)


EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


@pytest.mark.parametrize("dist", [EUROC_D, EUROC_D + (0.011,), (0.0, 0.0, 0.0, 0.0), (0.12, -0.2, 1e-3, -5e-4, 0.05, 0.01, -0.02, 0.003)])
def test_frame_from_distorted_keypoints(oracle, dist):
    """UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid on the device: bit-identical coordinates
    (binary64 arithmetic in the same order on both sides), bounds and grid."""
    A = _gpu()
    rng = np.random.default_rng(51)
    kps, desc = _clustered(rng, 3000)
    kps["x"] = np.clip(kps["x"], 0, 751)
    kps["y"] = np.clip(kps["y"], 0, 479)
    cam = A.Camera(*EUROC_K, dist)
    want_xy = oracle.undistort_points(np.stack([kps["x"], kps["y"]], 1), EUROC_K, dist) if dist[0] != 0 else \
        np.stack([kps["x"], kps["y"]], 1)
    want_b = oracle.image_bounds(752, 480, EUROC_K, dist)
    assert np.array_equal(cam.image_bounds(752, 480), want_b)
    f = A.FrameFeatures().assign_distorted(cam, 752, 480, kps, desc)
    un, b = f.keypoints()
    assert np.array_equal(b, want_b)
    assert np.array_equal(np.stack([un["x"], un["y"]], 1), want_xy)
    for name in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(un[name], kps[name])
    if dist[0] != 0:
        assert np.abs(want_xy - np.stack([kps["x"], kps["y"]], 1)).max() > 1.0       # the model does move points
    want_un = kps.copy()
    want_un["x"], want_un["y"] = want_xy[:, 0], want_xy[:, 1]
    start_o, idx_o = oracle.Grid(want_un, *map(float, want_b)).csr()
    start, idx = f.grid()
    assert np.array_equal(start, start_o) and np.array_equal(idx, idx_o)
    # searches on the frame use the undistorted keypoints
    q, lvl, src = synth.projected_queries(rng, want_un, 500, 7.0, 6.0)
    qdesc = synth.noisy_copy(rng, desc[src], 50)
    a = f.window_candidates(q, qdesc)
    ref = A.window_candidates(want_un, desc, None, tuple(map(float, want_b)), q, qdesc)
    assert all(np.array_equal(x, y) for x, y in zip(a, ref))
    # device-resident source
    import torch
    dk = torch.from_numpy(kps.view(np.uint8).copy()).cuda()
    dd = torch.from_numpy(desc.copy()).cuda()
    torch.cuda.synchronize()
    f2 = A.FrameFeatures().assign_distorted_device(cam, 752, 480, dk, dd, None, len(kps))
    un2, b2 = f2.keypoints()
    assert np.array_equal(un2, un) and np.array_equal(b2, b)
    assert all(np.array_equal(x, y) for x, y in zip(f2.grid(), (start, idx)))


def test_random_cameras_and_keypoint_sets(oracle):
    """Calibrations (focal lengths, principal points, 4 / 5 / 8 distortion coefficients of random size), image
    sizes and keypoint counts drawn at random: undistorted coordinates, image bounds and grid against the oracle."""
    A = _gpu()
    rng = np.random.default_rng(717)
    for case in range(16):
        w, h = int(rng.integers(160, 2000)), int(rng.integers(120, 1300))
        f = float(rng.uniform(0.4, 1.2)) * w
        K = (f, f * float(rng.uniform(0.97, 1.03)), w * float(rng.uniform(0.45, 0.55)), h * float(rng.uniform(0.45, 0.55)))
        nd = int(rng.choice([4, 5, 8]))
        scale = np.array([0.3, 0.1, 2e-3, 2e-3, 0.05, 0.02, 0.02, 0.01])[:nd]
        dist = tuple(float(v) for v in rng.normal(0, 1, nd) * scale * float(rng.uniform(0.1, 1.0)))
        n = int(rng.integers(0, 6000)) if case % 4 else int(rng.integers(0, 20))
        kps = np.zeros(n, A.KP_DTYPE)
        kps["x"] = rng.uniform(0, w - 1, n).astype(np.float32)
        kps["y"] = rng.uniform(0, h - 1, n).astype(np.float32)
        kps["octave"] = rng.integers(0, 8, n)
        kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        desc = synth.random_descriptors(n, 800 + case)
        cam = A.Camera(*K, dist)
        what = f"case {case}: {w}x{h}, K={K}, dist={dist}, {n} keypoints"
        want_b = oracle.image_bounds(w, h, K, dist)
        assert np.array_equal(cam.image_bounds(w, h), want_b), what
        xy = np.stack([kps["x"], kps["y"]], 1)
        want_xy = oracle.undistort_points(xy, K, dist) if dist[0] != 0 and n else xy
        fr = A.FrameFeatures().assign_distorted(cam, w, h, kps, desc)
        un, b = fr.keypoints()
        assert np.array_equal(b, want_b), what
        assert np.array_equal(np.stack([un["x"], un["y"]], 1), want_xy), what
        want_un = kps.copy()
        if n:
            want_un["x"], want_un["y"] = want_xy[:, 0], want_xy[:, 1]
        start_o, idx_o = oracle.Grid(want_un, *map(float, want_b)).csr()
        start, idx = fr.grid()
        assert np.array_equal(start, start_o) and np.array_equal(idx, idx_o), what
