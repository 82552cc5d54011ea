"""GPU parity of the order-dependent ORBmatcher loops (device candidate lists + in-order replay)
against the CPU oracle's literal restatements of the reference loops.  Bar: identical match
arrays and identical return values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

BOUNDS = (0.0, 752.0, 0.0, 480.0)
SF = np.array([1.2 ** i for i in range(8)], np.float32)


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _frame(rng, n, A, base_desc=None, flips=30):
    return synth.matcher_frame(rng, n, base_desc, flips)


def _noisy(rng, desc, max_flips):
    return synth.noisy_copy(rng, desc, max_flips)


def _queries(rng, A, kps, n, th=7.0, jitter=6.0):
    return synth.projected_queries(rng, kps, n, th, jitter)


@pytest.mark.parametrize("seed,use_ur,use_rot,use_marks,max_dist", [
    (1, True, True, True, 100.0),     # SearchByProjection(Cur, Last): stereo gate, rotation, obs marks
    (2, False, True, False, 64.0),    # SearchByProjection(Cur, KF, sFound, th, ORBdist = 64)
    (3, False, False, False, 75.0),   # SearchByProjection(KF, Sim3, points, matched, th, 1.5): 50 * 1.5
    (4, False, False, None, 50.0),    # Fuse(KF, Sim3, ...): stateless
])
def test_search_window_best(oracle, seed, use_ur, use_rot, use_marks, max_dist):
    A = _gpu()
    rng = np.random.default_rng(seed)
    base = synth.random_descriptors(300, 50 + seed)
    kps, desc = _frame(rng, 1500, A, base)
    ur = None
    if use_ur:
        ur = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(1, 40, len(kps)), -1).astype(np.float32)
    q, lvl, src = _queries(rng, A, kps, 900)
    if seed == 1:   # level windows of the forward / backward / default cases (ORBmatcher.cc:1813-1818)
        mode = rng.integers(0, 3, len(q))
        q["min_level"] = np.where(mode == 0, lvl, np.where(mode == 1, 0, lvl - 1))
        q["max_level"] = np.where(mode == 0, -1, np.where(mode == 1, lvl, lvl + 1))
    q["xr"] = q["x"] - rng.uniform(1, 40, len(q)).astype(np.float32)
    qdesc = _noisy(rng, desc[src], 60)
    qangle = ((kps["angle"][src] + rng.normal(40, 25, len(q))) % 360).astype(np.float32) if use_rot else None
    taken0 = None if use_marks is None else (rng.random(len(kps)) < 0.2).astype(np.uint8)
    marks = (rng.random(len(q)) < 0.8).astype(np.uint8) if use_marks else None

    g = oracle.Grid(kps, *BOUNDS)
    kpm_o, taken_o, n_o = oracle.search_window_best(g, desc, ur, q["x"], q["y"], q["xr"], q["radius"], q["min_level"],
                                                    q["max_level"], qdesc, max_dist, taken0, marks, qangle, 1.0 / 30)
    m = A.ORBmatcher(0.9, True)
    taken = None if taken0 is None else taken0.copy()
    out, kpm, n = m.SearchWindowBest(kps, desc, ur, BOUNDS, q, qdesc, max_dist, taken, marks, qangle)
    assert n == n_o
    assert np.array_equal(kpm, kpm_o)
    if taken0 is not None:
        assert np.array_equal(taken, taken_o)
    assert (kpm >= 0).sum() > 50
    own = np.nonzero(out >= 0)[0]
    assert np.array_equal(kpm[out[own]], own)


def test_fuse_search(oracle):
    A = _gpu()
    rng = np.random.default_rng(5)
    base = synth.random_descriptors(300, 77)
    kps, desc = _frame(rng, 1500, A, base)
    ur = np.where(rng.random(len(kps)) < 0.5, kps["x"] - rng.uniform(1, 40, len(kps)), -1).astype(np.float32)
    q, lvl, src = _queries(rng, A, kps, 1200, th=3.0, jitter=1.5)
    q["xr"] = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.5, len(q)), q["x"] - 10).astype(np.float32)
    qdesc = _noisy(rng, desc[src], 50)
    inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
    g = oracle.Grid(kps, *BOUNDS)
    out_o, n_o = oracle.fuse_search(g, desc, ur, q["x"], q["y"], q["xr"], q["radius"], lvl, qdesc, inv_sigma2)
    out, n = A.ORBmatcher().Fuse(kps, desc, ur, BOUNDS, q, qdesc, inv_sigma2)
    assert n == n_o and np.array_equal(out, out_o)
    assert n > 30


def test_search_by_sim3(oracle):
    A = _gpu()
    rng = np.random.default_rng(6)
    base = synth.random_descriptors(400, 78)
    kps1, desc1 = _frame(rng, 1200, A, base)
    kps2 = kps1.copy()
    kps2["x"] = np.clip(kps1["x"] + rng.normal(0, 2, len(kps1)), 1, 750).astype(np.float32)
    kps2["y"] = np.clip(kps1["y"] + rng.normal(0, 2, len(kps1)), 1, 478).astype(np.float32)
    perm = rng.permutation(len(kps1))
    kps2, desc2 = kps2[perm], desc1[perm].copy()
    for i in range(len(desc2)):
        for b in rng.integers(0, 256, int(rng.integers(0, 25))):
            desc2[i, b >> 3] ^= np.uint8(1 << (b & 7))
    inv = np.argsort(perm)
    s1 = np.nonzero(rng.random(len(kps1)) < 0.7)[0].astype(np.int32)      # KF1 keypoints with a map point
    s2 = np.nonzero(rng.random(len(kps2)) < 0.7)[0].astype(np.int32)

    def mk(src_kps, src_desc, sel, tgt_kps, tgt_of):
        q = np.zeros(len(sel), A.QUERY_DTYPE)
        t = tgt_kps[tgt_of[sel]]
        q["x"] = t["x"] + rng.normal(0, 3, len(sel)).astype(np.float32)
        q["y"] = t["y"] + rng.normal(0, 3, len(sel)).astype(np.float32)
        lvl = np.clip(t["octave"] + rng.integers(0, 2, len(sel)), 0, 7).astype(np.int32)
        q["radius"] = (7.5 * SF[lvl]).astype(np.float32)
        q["min_level"], q["max_level"] = lvl - 1, lvl
        return q, lvl, src_desc[sel].copy()
    q2, l2, d2 = mk(kps1, desc1, s1, kps2, inv)
    q1, l1, d1 = mk(kps2, desc2, s2, kps1, perm)
    g1, g2 = oracle.Grid(kps1, *BOUNDS), oracle.Grid(kps2, *BOUNDS)
    out_o, n_o = oracle.search_by_sim3(g1, desc1, g2, desc2, (q2["x"], q2["y"], q2["radius"], l2, d2, s1),
                                       (q1["x"], q1["y"], q1["radius"], l1, d1, s2))
    out, n = A.ORBmatcher().SearchBySim3(kps1, desc1, BOUNDS, kps2, desc2, BOUNDS, (q2, d2, s1), (q1, d1, s2))
    assert n == n_o and np.array_equal(out, out_o)
    assert n > 30


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_for_initialization(oracle, check_ori):
    A = _gpu()
    rng = np.random.default_rng(7)
    base = synth.random_descriptors(500, 79)
    kps1, desc1 = _frame(rng, 1500, A, base, flips=12)
    kps1["octave"] = rng.integers(0, 3, len(kps1))
    kps2, desc2 = _frame(rng, 1600, A, base, flips=12)
    kps2["octave"] = rng.integers(0, 3, len(kps2))
    # second frame = first frame moved by a few pixels for most points (consistent rotation)
    m = min(len(kps1), len(kps2))
    kps2["x"][:m] = np.clip(kps1["x"][:m] + rng.normal(4, 3, m), 1, 750)
    kps2["y"][:m] = np.clip(kps1["y"][:m] + rng.normal(-3, 3, m), 1, 478)
    kps2["angle"][:m] = (kps1["angle"][:m] + rng.normal(5, 8, m)) % 360
    kps2["octave"][:m] = kps1["octave"][:m]
    desc2[:m] = _noisy(rng, desc1[:m], 30)
    prev = np.stack([kps1["x"], kps1["y"]], 1).astype(np.float32)
    g2 = oracle.Grid(kps2, *BOUNDS)
    out_o, prev_o, n_o = oracle.search_for_initialization(kps1, desc1, g2, desc2, prev, 100, 0.9, check_ori)
    prev_g = prev.copy()
    out, n = A.ORBmatcher(0.9, check_ori).SearchForInitialization(kps1, desc1, kps2, desc2, BOUNDS, prev_g, 100)
    assert n == n_o and np.array_equal(out, out_o) and np.array_equal(prev_g, prev_o)
    assert n > 20


def _featvec(rng, n, n_nodes):
    return synth.feature_vector(rng, n, n_nodes)


@pytest.mark.parametrize("n_left,check_ori", [(-1, True), (-1, False), (700, True)])
def test_search_by_bow(oracle, n_left, check_ori):
    A = _gpu()
    rng = np.random.default_rng(8)
    base = synth.random_descriptors(120, 80)
    kpsK, descK = _frame(rng, 1000, A, base, flips=16)
    kpsF, descF = _frame(rng, 1100, A, base, flips=16)
    validK = (rng.random(len(kpsK)) < 0.7).astype(np.uint8)
    fvK, fvF = _featvec(rng, len(kpsK), 40), _featvec(rng, len(kpsF), 40)
    out_o, n_o = oracle.search_by_bow(descK, kpsK["angle"], validK, fvK[0], fvK[1], descF, kpsF["angle"], n_left,
                                      fvF[0], fvF[1], 0.7, check_ori)
    out, n = A.ORBmatcher(0.7, check_ori).SearchByBoW(descK, kpsK["angle"], validK, fvK, descF, kpsF["angle"], fvF, n_left)
    assert n == n_o and np.array_equal(out, out_o)
    assert (out >= 0).sum() > 10
    # the same search on frames whose descriptors are resident on the device
    fK, fF = A.FrameFeatures(kpsK, descK, None, BOUNDS), A.FrameFeatures(kpsF, descF, None, BOUNDS)
    out2, n2 = A.ORBmatcher(0.7, check_ori).SearchByBoWIn(fK, validK, fvK, fF, fvF, n_left)
    assert n2 == n_o and np.array_equal(out2, out_o)


def test_search_by_bow_keyframes(oracle):
    A = _gpu()
    rng = np.random.default_rng(9)
    base = synth.random_descriptors(120, 81)
    kps1, desc1 = _frame(rng, 900, A, base, flips=16)
    kps2, desc2 = _frame(rng, 950, A, base, flips=16)
    v1 = (rng.random(len(kps1)) < 0.7).astype(np.uint8)
    v2 = (rng.random(len(kps2)) < 0.7).astype(np.uint8)
    fv1, fv2 = _featvec(rng, len(kps1), 35), _featvec(rng, len(kps2), 35)
    out_o, n_o = oracle.search_by_bow_keyframes(desc1, kps1["angle"], v1, fv1[0], fv1[1], desc2, kps2["angle"], v2,
                                                fv2[0], fv2[1], 0.9, True)
    out, n = A.ORBmatcher(0.9, True).SearchByBoWKeyFrames(desc1, kps1["angle"], v1, fv1, desc2, kps2["angle"], v2, fv2)
    assert n == n_o and np.array_equal(out, out_o)
    assert (out >= 0).sum() > 10
    f1, f2 = A.FrameFeatures(kps1, desc1, None, BOUNDS), A.FrameFeatures(kps2, desc2, None, BOUNDS)
    out2, n2 = A.ORBmatcher(0.9, True).SearchByBoWKeyFramesIn(f1, v1, fv1, f2, v2, fv2)
    assert n2 == n_o and np.array_equal(out2, out_o)


def _triangulation_case(rng, A):
    base = synth.random_descriptors(150, 82)
    n = 1000
    kps1, desc1 = _frame(rng, n, A, base, flips=14)
    # image 2 = image 1 moved sideways (same row up to noise), most descriptors noisy copies
    kps2 = kps1.copy()
    kps2["x"] = np.clip(kps1["x"] - rng.uniform(2, 60, n), 1, 750).astype(np.float32)
    kps2["y"] = (kps1["y"] + rng.normal(0, 1.5, n)).astype(np.float32)
    kps2["angle"] = ((kps1["angle"] + rng.normal(3, 6, n)) % 360).astype(np.float32)
    desc2 = _noisy(rng, desc1, 40)
    f1 = (rng.random(n) < 0.7).astype(np.uint8)
    f2 = (rng.random(n) < 0.7).astype(np.uint8)
    ur1 = np.where(rng.random(n) < 0.5, kps1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n) < 0.5, kps2["x"] - 5, -1).astype(np.float32)
    node = rng.integers(0, 30, n)                      # the same feature falls into the same node in both
    node2 = np.where(rng.random(n) < 0.9, node, rng.integers(0, 30, n))

    def fv(nd):
        order = np.argsort(nd, kind="stable")
        off = np.zeros(31, np.int32)
        np.add.at(off, nd + 1, 1)
        return order.astype(np.int32), np.cumsum(off).astype(np.int32)
    # F12 of a pure sideways translation: the epipolar line of (x, y) is the row y (plus a slight tilt)
    F12 = np.array([0, 0, 0, 0, 0, -1, 0.0005, 1, -0.1], np.float32)
    ep = (rng.choice(kps2["x"]) + 3.0, rng.choice(kps2["y"]) + 2.0)     # an epipole inside the image, near keypoints
    return kps1, desc1, f1, ur1, fv(node), kps2, desc2, f2, ur2, fv(node2), F12, ep


@pytest.mark.parametrize("only_stereo,coarse,check_ori", [(False, False, False), (True, False, True), (False, True, True)])
def test_search_for_triangulation(oracle, only_stereo, coarse, check_ori):
    A = _gpu()
    rng = np.random.default_rng(10)
    kps1, desc1, f1, ur1, fv1, kps2, desc2, f2, ur2, fv2, F12, ep = _triangulation_case(rng, A)
    sig2 = (SF * SF).astype(np.float32)
    out_o, n_o = oracle.search_for_triangulation(kps1, desc1, f1, ur1, fv1[0], fv1[1], kps2, desc2, f2, ur2, fv2[0],
                                                 fv2[1], F12, ep, SF, sig2, only_stereo, coarse, check_ori)
    out, n = A.ORBmatcher(0.6, check_ori).SearchForTriangulation(kps1, desc1, f1, ur1, fv1, kps2, desc2, f2, ur2, fv2,
                                                                 F12, ep, SF, sig2, only_stereo, coarse)
    assert n == n_o and np.array_equal(out, out_o)
    assert (out >= 0).sum() > 40


def test_bucket_allpairs(oracle):
    A = _gpu()
    rng = np.random.default_rng(11)
    d1 = synth.random_descriptors(300, 1)
    d2 = synth.random_descriptors(280, 2)
    d2[7] = ~d1[3]                                   # a pair at distance 256
    fv1, fv2 = _featvec(rng, len(d1), 9), _featvec(rng, len(d2), 9)
    fv1[0][0], fv2[0][0] = 3, 7
    fv1 = (fv1[0], fv1[1]); fv2 = (fv2[0], fv2[1])
    rs, dist = A.bucket_allpairs(d1, fv1[0], fv1[1], d2, fv2[0], fv2[1])
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for b in range(9):
        for e in range(fv1[1][b], fv1[1][b + 1]):
            js = fv2[0][fv2[1][b]:fv2[1][b + 1]]
            exp = pop[d1[fv1[0][e]][None, :] ^ d2[js]].sum(1)
            assert np.array_equal(dist[rs[e]:rs[e] + len(js)], exp.astype(np.uint16))
    assert rs[-1] == len(dist)


def test_empty_inputs():
    """No queries, no keypoints, no shared vocabulary nodes: every entry point returns cleanly."""
    A = _gpu()
    rng = np.random.default_rng(12)
    kps, desc = _frame(rng, 50, A)
    m = A.ORBmatcher(0.9, True)
    q0 = np.zeros(0, A.QUERY_DTYPE)
    d0 = np.zeros((0, 32), np.uint8)
    out, kpm, n = m.SearchWindowBest(kps, desc, None, BOUNDS, q0, d0, 100.0, np.zeros(50, np.uint8))
    assert n == 0 and len(out) == 0 and np.all(kpm == -1)
    q, _, _ = _queries(rng, A, kps, 10)
    out, kpm, n = m.SearchWindowBest(kps[:0], d0, None, BOUNDS, q, desc[:10], 100.0, None)
    assert n == 0 and np.all(out == -1) and len(kpm) == 0
    out, n = m.Fuse(kps, desc, None, BOUNDS, q0, d0, np.ones(8, np.float32))
    assert n == 0 and len(out) == 0
    z = np.zeros(0, np.int32)
    off0 = np.zeros(1, np.int32)
    out, n = m.SearchByBoW(desc, kps["angle"], np.ones(50, np.uint8), (z, off0), desc, kps["angle"], (z, off0))
    assert n == 0 and np.all(out == -1)
    out, n = m.SearchByBoWKeyFrames(desc, kps["angle"], np.ones(50, np.uint8), (z, off0), desc, kps["angle"],
                                    np.ones(50, np.uint8), (z, off0))
    assert n == 0 and np.all(out == -1)
    # nodes present but empty on one side
    off = np.array([0, 3, 3], np.int32)
    offe = np.array([0, 0, 2], np.int32)
    out, n = m.SearchByBoWKeyFrames(desc, kps["angle"], np.ones(50, np.uint8), (np.array([0, 1, 2], np.int32), off), desc,
                                    kps["angle"], np.ones(50, np.uint8), (np.array([4, 5], np.int32), offe))
    assert n == 0 and np.all(out == -1)
    prev = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    out, n = m.SearchForInitialization(kps, desc, kps[:0], d0, BOUNDS, prev, 100)
    assert n == 0 and np.all(out == -1)
    A.lib().orbx_release_cached_memory()


def test_matchers_against_committed_golden(golden_dir):
    """The HIP path on the seeded cases of synth.matcher_golden_cases() against tests/golden/matchers.npz."""
    import os
    A = _gpu()
    g = np.load(os.path.join(golden_dir, "matchers.npz"))
    c = synth.matcher_golden_cases()
    w = c["window"]
    bounds = tuple(float(v) for v in w["bounds"])
    taken = w["taken"].copy()
    _, kpm, n = A.ORBmatcher(0.9, True).SearchWindowBest(w["kps"], w["desc"], w["ur"], bounds, w["q"], w["qdesc"], 100.0,
                                                         taken, w["marks"], w["qangle"])
    assert n == int(g["window_n"]) and np.array_equal(kpm, g["window_kp_match"]) and np.array_equal(taken, g["window_taken"])
    q = w["q"].copy()
    q["min_level"], q["max_level"] = w["lvl"] - 1, w["lvl"]
    m, n = A.ORBmatcher(0.8).SearchByProjection(w["kps"], w["desc"], w["ur"], bounds, q, w["qdesc"], w["taken"].copy())
    assert n == int(g["projection_n"]) and np.array_equal(m, g["projection_match"])
    b = c["bow"]
    m12, n = A.ORBmatcher(0.9, True).SearchByBoWKeyFrames(b["d1"], b["a1"], b["v1"], (b["i1"], b["o1"]), b["d2"], b["a2"],
                                                          b["v2"], (b["i2"], b["o2"]))
    assert n == int(g["bow_kf_n"]) and np.array_equal(m12, g["bow_kf_matches12"])
    mf, n = A.ORBmatcher(0.7, True).SearchByBoW(b["d1"], b["a1"], b["v1"], (b["i1"], b["o1"]), b["d2"], b["a2"],
                                                (b["i2"], b["o2"]))
    assert n == int(g["bow_frame_n"]) and np.array_equal(mf, g["bow_frame_match"])
    v = c["vocabulary"]
    V = A.Vocabulary(v["child_off"], v["child_idx"], v["node_desc"], v["word_id"], v["weight"], int(v["depth"]))
    wo, we, ni = V.transform_features(v["features"], 1)
    assert np.array_equal(wo, g["voc_word"]) and np.array_equal(we, g["voc_weight"]) and np.array_equal(ni, g["voc_node"])
    d = c["distinctive"]
    bi, bm = A.distinctive_descriptors(d["desc"], d["off"])
    assert np.array_equal(bi, g["distinct_idx"]) and np.array_equal(bm, g["distinct_median"])


def test_matcher_entry_points_from_concurrent_threads(golden_dir):
    """Tracking, LocalMapping and LoopClosing call the matchers at the same time: four threads run the golden
    cases side by side (each entry point works on the calling thread's own stream and staging); every result must
    equal the committed vector."""
    import threading
    errors = []

    def work():
        try:
            for _ in range(5):
                test_matchers_against_committed_golden(golden_dir)
        except BaseException as e:      # noqa: BLE001 - collected and re-raised in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    if errors:
        raise errors[0]
