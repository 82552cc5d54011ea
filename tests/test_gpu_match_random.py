"""Seed sweeps of the order-dependent ORBmatcher loops: list sizes, query counts, vocabulary node counts, noise
levels and gates drawn at random, every case against the CPU oracle's restatement of the reference loop.
Bar: identical match arrays and identical return values."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

BOUNDS = (0.0, 752.0, 0.0, 480.0)
SF = np.array([1.2 ** i for i in range(8)], np.float32)


# ORBX_SWEEP=n adds n more seeds to every sweep (a hunt run by hand; the default set runs with the suite)
_EXTRA = int(os.environ.get("ORBX_SWEEP", "0"))


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _fail(name, seed, detail, e):
    raise AssertionError(f"{name} seed {seed} ({detail}): {e}")


@pytest.mark.parametrize("seed", range(201, 209 + _EXTRA))
def test_window_searches(oracle, seed):
    """SearchByProjection-style best(/second) searches with the "taken" rule (ORBmatcher.cc:42, :476, :582, :1761,
    :1974) and Fuse (:1214)."""
    A = _gpu()
    rng = np.random.default_rng(seed)
    hi = 40 if seed % 4 == 0 else 4000                  # every fourth case: lists of a few entries
    n_kp, n_q = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    base = synth.random_descriptors(int(rng.integers(20, 600)), seed)
    kps, desc = synth.matcher_frame(rng, n_kp, base, int(rng.integers(5, 60)))
    use_ur, use_rot = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    use_marks = [None, False, True][int(rng.integers(0, 3))]
    max_dist = float(rng.choice([50.0, 64.0, 75.0, 100.0]))
    ur = None
    if use_ur:
        ur = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(1, 40, len(kps)), -1).astype(np.float32)
    q, lvl, src = synth.projected_queries(rng, kps, n_q, float(rng.uniform(2, 15)), float(rng.uniform(0.5, 10)))
    mode = rng.integers(0, 3, len(q))
    q["min_level"] = np.where(mode == 0, lvl, np.where(mode == 1, 0, lvl - 1))
    q["max_level"] = np.where(mode == 0, -1, np.where(mode == 1, lvl, lvl + 1))
    q["xr"] = q["x"] - rng.uniform(1, 40, len(q)).astype(np.float32)
    qdesc = synth.noisy_copy(rng, desc[src], int(rng.integers(1, 90)))
    qangle = ((kps["angle"][src] + rng.normal(40, 25, len(q))) % 360).astype(np.float32) if use_rot else None
    taken0 = None if use_marks is None else (rng.random(len(kps)) < 0.2).astype(np.uint8)
    marks = (rng.random(len(q)) < 0.8).astype(np.uint8) if use_marks else None
    detail = f"{len(kps)} keypoints, {len(q)} queries, ur={use_ur} rot={use_rot} marks={use_marks} max_dist={max_dist}"
    try:
        g = oracle.Grid(kps, *BOUNDS)
        kpm_o, taken_o, n_o = oracle.search_window_best(g, desc, ur, q["x"], q["y"], q["xr"], q["radius"], q["min_level"],
                                                        q["max_level"], qdesc, max_dist, taken0, marks, qangle, 1.0 / 30)
        taken = None if taken0 is None else taken0.copy()
        out, kpm, n = A.ORBmatcher(0.9, True).SearchWindowBest(kps, desc, ur, BOUNDS, q, qdesc, max_dist, taken, marks, qangle)
        assert n == n_o, "count"
        assert np.array_equal(kpm, kpm_o), "matches"
        if taken0 is not None:
            assert np.array_equal(taken, taken_o), "taken"
        # Fuse on the same lists: levels nPredictedLevel - 1 .. nPredictedLevel (ORBmatcher.cc:1338)
        q["min_level"], q["max_level"] = lvl - 1, lvl
        inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
        urf = ur if ur is not None else np.full(len(kps), -1, np.float32)
        out_o, nf_o = oracle.fuse_search(g, desc, urf, q["x"], q["y"], q["xr"], q["radius"], lvl, qdesc, inv_sigma2)
        out_f, nf = A.ORBmatcher().Fuse(kps, desc, urf, BOUNDS, q, qdesc, inv_sigma2)
        assert nf == nf_o and np.array_equal(out_f, out_o), "fuse"
    except AssertionError as e:
        _fail("window", seed, detail, e)


@pytest.mark.parametrize("seed", range(301, 307 + _EXTRA))
def test_initialization_and_bow(oracle, seed):
    """SearchForInitialization (ORBmatcher.cc:695), SearchByBoW for a frame (:235) and for two keyframes (:824)."""
    A = _gpu()
    rng = np.random.default_rng(seed)
    check_ori = bool(rng.integers(0, 2))
    base = synth.random_descriptors(int(rng.integers(30, 600)), seed)
    hi = 40 if seed % 3 == 0 else 3000                  # every third case: lists of a few entries
    n1, n2 = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    flips = int(rng.integers(4, 40))
    kps1, desc1 = synth.matcher_frame(rng, n1, base, flips)
    kps2, desc2 = synth.matcher_frame(rng, n2, base, flips)
    n1, n2 = len(kps1), len(kps2)
    detail = f"{n1} and {n2} keypoints, check_ori={check_ori}"
    try:
        k1, k2 = kps1.copy(), kps2.copy()
        k1["octave"] = rng.integers(0, 3, n1)
        k2["octave"] = rng.integers(0, 3, n2)
        m = min(n1, n2)
        k2["x"][:m] = np.clip(k1["x"][:m] + rng.normal(4, 3, m), 1, 750)
        k2["y"][:m] = np.clip(k1["y"][:m] + rng.normal(-3, 3, m), 1, 478)
        k2["angle"][:m] = (k1["angle"][:m] + rng.normal(5, 8, m)) % 360
        k2["octave"][:m] = k1["octave"][:m]
        d2 = desc2.copy()
        d2[:m] = synth.noisy_copy(rng, desc1[:m], 30)
        prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
        win = int(rng.choice([30, 100, 200]))
        out_o, prev_o, n_o = oracle.search_for_initialization(k1, desc1, oracle.Grid(k2, *BOUNDS), d2, prev, win, 0.9, check_ori)
        prev_g = prev.copy()
        out, n = A.ORBmatcher(0.9, check_ori).SearchForInitialization(k1, desc1, k2, d2, BOUNDS, prev_g, win)
        assert n == n_o and np.array_equal(out, out_o) and np.array_equal(prev_g, prev_o), "initialization"

        nodes = int(rng.integers(1, 80))
        v1 = (rng.random(n1) < 0.7).astype(np.uint8)
        v2 = (rng.random(n2) < 0.7).astype(np.uint8)
        fv1, fv2 = synth.feature_vector(rng, n1, nodes), synth.feature_vector(rng, n2, nodes)
        n_left = -1 if rng.integers(0, 2) else int(rng.integers(0, n2 + 1))
        ratio = float(rng.choice([0.6, 0.7, 0.9]))
        out_o, n_o = oracle.search_by_bow(desc1, kps1["angle"], v1, fv1[0], fv1[1], desc2, kps2["angle"], n_left,
                                          fv2[0], fv2[1], ratio, check_ori)
        out, n = A.ORBmatcher(ratio, check_ori).SearchByBoW(desc1, kps1["angle"], v1, fv1, desc2, kps2["angle"], fv2, n_left)
        assert n == n_o and np.array_equal(out, out_o), f"bow frame (nodes={nodes}, n_left={n_left}, ratio={ratio})"
        out_o, n_o = oracle.search_by_bow_keyframes(desc1, kps1["angle"], v1, fv1[0], fv1[1], desc2, kps2["angle"], v2,
                                                    fv2[0], fv2[1], ratio, check_ori)
        out, n = A.ORBmatcher(ratio, check_ori).SearchByBoWKeyFrames(desc1, kps1["angle"], v1, fv1, desc2, kps2["angle"], v2, fv2)
        assert n == n_o and np.array_equal(out, out_o), f"bow keyframes (nodes={nodes}, ratio={ratio})"
    except AssertionError as e:
        _fail("initialization / bow", seed, detail, e)


@pytest.mark.parametrize("seed", range(401, 407 + _EXTRA))
def test_triangulation(oracle, seed):
    """SearchForTriangulation (ORBmatcher.cc:968): per shared node all pairs, epipole and epipolar gates."""
    A = _gpu()
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 40 if seed % 3 == 0 else 3000))     # every third case: lists of a few entries
    nodes = int(rng.integers(1, 60))
    base = synth.random_descriptors(int(rng.integers(20, 400)), seed)
    kps1, desc1 = synth.matcher_frame(rng, n, base, int(rng.integers(4, 30)))
    n = len(kps1)
    only_stereo, coarse, check_ori = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    detail = f"{n} keypoints, {nodes} nodes, only_stereo={only_stereo} coarse={coarse} check_ori={check_ori}"
    try:
        kps2 = kps1.copy()
        kps2["x"] = np.clip(kps1["x"] - rng.uniform(2, 60, n), 1, 750).astype(np.float32)
        kps2["y"] = (kps1["y"] + rng.normal(0, 1.5, n)).astype(np.float32)
        kps2["angle"] = ((kps1["angle"] + rng.normal(3, 6, n)) % 360).astype(np.float32)
        desc2 = synth.noisy_copy(rng, desc1, int(rng.integers(1, 70)))
        f1 = (rng.random(n) < 0.7).astype(np.uint8)
        f2 = (rng.random(n) < 0.7).astype(np.uint8)
        ur1 = np.where(rng.random(n) < 0.5, kps1["x"] - 5, -1).astype(np.float32)
        ur2 = np.where(rng.random(n) < 0.5, kps2["x"] - 5, -1).astype(np.float32)
        node = rng.integers(0, nodes, n)
        node2 = np.where(rng.random(n) < 0.9, node, rng.integers(0, nodes, n))

        def fv(nd):
            order = np.argsort(nd, kind="stable")
            off = np.zeros(nodes + 1, np.int32)
            np.add.at(off, nd + 1, 1)
            return order.astype(np.int32), np.cumsum(off).astype(np.int32)
        fv1, fv2 = fv(node), fv(node2)
        F12 = np.array([0, 0, 0, 0, 0, -1, float(rng.uniform(-0.001, 0.001)), 1, float(rng.uniform(-0.5, 0.5))], np.float32)
        ep = (float(rng.choice(kps2["x"])) + 3.0, float(rng.choice(kps2["y"])) + 2.0)
        sig2 = (SF * SF).astype(np.float32)
        out_o, n_o = oracle.search_for_triangulation(kps1, desc1, f1, ur1, fv1[0], fv1[1], kps2, desc2, f2, ur2, fv2[0],
                                                     fv2[1], F12, ep, SF, sig2, only_stereo, coarse, check_ori)
        out, n_g = A.ORBmatcher(0.6, check_ori).SearchForTriangulation(kps1, desc1, f1, ur1, fv1, kps2, desc2, f2, ur2, fv2,
                                                                       F12, ep, SF, sig2, only_stereo, coarse)
        assert n_g == n_o and np.array_equal(out, out_o)
    except AssertionError as e:
        _fail("triangulation", seed, detail, e)
