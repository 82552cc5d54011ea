"""GPU parity of the matching kernels against the CPU oracle, through the C ABI (bit-exact:
indices and integer distances; stereo uRight / depth are float32 results of the same
individually rounded operations and are compared exactly)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

MBF, FX = 47.90639384423901, 435.2046959714599


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _m2(a):
    return np.stack([a["idx0"], a["dist0"], a["idx1"], a["dist1"]], 1)


def test_descriptor_distance(oracle):
    A = _gpu()
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    b[:10] = a[:10]
    b[10:20] = ~a[10:20]
    d = A.ORBmatcher.DescriptorDistance(a, b)
    ref = np.unpackbits(a ^ b, axis=1).sum(1)
    assert np.array_equal(d, ref)
    assert d[:10].max() == 0 and d[10:20].min() == 256
    assert A.ORBmatcher.DescriptorDistance(a[0], b[0]) == oracle.descriptor_distance(a[0], b[0])


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (3, 0), (64, 64), (130, 1000), (1000, 1003), (257, 20000), (5, 70001)])
def test_knn2_matches_oracle(oracle, nq, nt):
    A = _gpu()
    q = synth.random_descriptors(nq, 5)
    t, _ = synth.planted_map(q, nt, 6) if nt else (np.zeros((0, 32), np.uint8), None)
    if nt > 300:
        t[100] = t[200]
        t[7] = q[0]; t[9] = q[0]           # two exact copies: tie at distance 0 -> lower index first
    assert np.array_equal(_m2(A.knn_match2(q, t)), oracle.knn2(q, t))


def test_knn2_random_sizes(oracle):
    # Query and map counts around the tile edges of the search kernel, plain and low-entropy (many exact ties),
    # with a shard offset: best / second best (distance, then lower index) against the oracle.
    A = _gpu()
    import torch
    rng = np.random.default_rng(909)
    spans_q = [(1, 70), (120, 140), (250, 260), (1000, 1100)]
    spans_t = [(1, 130), (250, 260), (1000, 1100), (16380, 16390), (30000, 70000)]
    for case in range(16):
        a, b = spans_q[int(rng.integers(0, len(spans_q)))]
        nq = int(rng.integers(a, b + 1))
        a, b = spans_t[int(rng.integers(0, len(spans_t)))]
        nt = int(rng.integers(a, b + 1))
        if case % 4 == 3:      # few distinct values: ties everywhere
            q = np.zeros((nq, 32), np.uint8); q[:, :2] = rng.integers(0, 4, (nq, 2))
            t = np.zeros((nt, 32), np.uint8); t[:, :2] = rng.integers(0, 4, (nt, 2))
        else:
            q = synth.random_descriptors(nq, 1000 + case)
            t, _ = synth.planted_map(q, nt, 2000 + case)
        ref = oracle.knn2(q, t)
        assert np.array_equal(_m2(A.knn_match2(q, t)), ref), f"case {case}: {nq} x {nt}"
        base = int(rng.integers(0, 1 << 20))
        r, _ = A.knn_match2_device(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), base_index=base)
        torch.cuda.synchronize()
        r = r.cpu().numpy()
        exp = ref.copy()
        exp[:, 0] = np.where(ref[:, 0] >= 0, ref[:, 0] + base, ref[:, 0])
        exp[:, 2] = np.where(ref[:, 2] >= 0, ref[:, 2] + base, ref[:, 2])
        assert np.array_equal(r, exp), f"case {case}: {nq} x {nt} at base {base}"


def test_knn2_golden(golden_dir):
    A = _gpu()
    g = np.load(os.path.join(golden_dir, "knn2_planted.npz"))
    q = synth.random_descriptors(int(g["nq"]), int(g["q_seed"]))
    t, _ = synth.planted_map(q, int(g["nt"]), int(g["t_seed"]))
    t[100] = t[200]
    assert np.array_equal(_m2(A.knn_match2(q, t)), g["result"])


def test_knn2_low_entropy_many_ties(oracle):
    A = _gpu()
    rng = np.random.default_rng(3)
    q = np.zeros((200, 32), np.uint8); q[:, 0] = rng.integers(0, 4, 200)
    t = np.zeros((5000, 32), np.uint8); t[:, 0] = rng.integers(0, 4, 5000)
    assert np.array_equal(_m2(A.knn_match2(q, t)), oracle.knn2(q, t))


def test_knn2_sharded_merge_equals_single(oracle):
    # contiguous shards + lexicographic merge reproduce the single-device result (ties included)
    A = _gpu()
    import torch
    q = synth.random_descriptors(500, 7)
    t, _ = synth.planted_map(q, 40000, 8)
    t[1000:1100] = t[30000:30100]          # duplicates across shards
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    full, _ = A.knn_match2_device(dq, dt)
    parts = []
    bounds = [0, 9000, 9001, 25000, 40000]
    for a, b in zip(bounds[:-1], bounds[1:]):
        p, _ = A.knn_match2_device(dq, dt[a:b], base_index=a)
        parts.append(p)
    merged = A.knn2_merge_device(torch.stack(parts))
    torch.cuda.synchronize()
    assert torch.equal(full, merged)
    assert np.array_equal(full.cpu().numpy(), oracle.knn2(q, t))


@pytest.mark.parametrize("text", [False, True])
def test_knn2_over_a_saved_map(oracle, tmp_path, text):
    # descriptors streamed out of a saved-map image (serializeMatrix records, SerializationUtils.h:76-108) into the
    # sharded search: contiguous shards of the file's descriptor rows, local top-2, merge = the search over all rows
    A = _gpu()
    import torch
    from orb_slam3_amd import mapfile
    q = synth.random_descriptors(300, 17)
    t, _ = synth.planted_map(q, 6000, 18)
    cuts = [0, 700, 701, 2600, 6000]                       # four keyframes' descriptor matrices
    other = b" 5 0 other members" if text else b"\x05\0\0\0other members"      # what surrounds the records
    img = b"".join(other * 3 + mapfile.write_matrix(t[a:b], text) for a, b in zip(cuts[:-1], cuts[1:]))
    path = tmp_path / "map.osa"
    path.write_bytes(img)
    md = mapfile.MapDescriptors(str(path), text, min_rows=1)
    assert len(md) == len(t)
    dq = torch.from_numpy(q).cuda()
    parts = []
    for r in range(3):
        base, rows = md.shard(r, 3)
        p, _ = A.knn_match2_device(dq, torch.from_numpy(np.ascontiguousarray(rows)).cuda(), base_index=base)
        parts.append(p)
    merged = A.knn2_merge_device(torch.stack(parts))
    torch.cuda.synchronize()
    assert np.array_equal(merged.cpu().numpy(), oracle.knn2(q, t))
    md.close()


def test_knn2_full_size_properties():
    # the large-map search shape on one GPU shard of eight: 5000 queries against 1.25 M map descriptors
    # (one eighth of 10 M).  Too large for the oracle: checked through properties.
    A = _gpu()
    import torch
    nq, nt = 5000, 1_250_000
    q = synth.random_descriptors(nq, 5)
    t, rows = synth.planted_map(q, nt, 6)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    r, _ = A.knn_match2_device(dq, dt)
    torch.cuda.synchronize()
    r = r.cpu().numpy()
    # reported distances are the true distances of the reported rows
    for col in (0, 2):
        idx = r[:, col]
        d = np.unpackbits(q ^ t[idx], axis=1).sum(1)
        assert np.array_equal(d, r[:, col + 1])
    assert np.all((r[:, 1] < r[:, 3]) | ((r[:, 1] == r[:, 3]) & (r[:, 0] < r[:, 2])))
    # the planted copy is at least as far as the best
    planted = np.unpackbits(q ^ t[rows], axis=1).sum(1)
    assert np.all(r[:, 1] <= planted)
    # uniformly random 256-bit rows are ~128 apart: whenever the planted copy has fewer than 60
    # flipped bits it must be the winner (or tie with a lower index)
    strong = planted < 60
    assert np.all((r[strong, 0] == rows[strong]) | (r[strong, 1] < planted[strong]) |
                  ((r[strong, 1] == planted[strong]) & (r[strong, 0] < rows[strong])))
    # splitting the map and merging changes nothing
    half = nt // 2
    a, _ = A.knn_match2_device(dq, dt[:half])
    b, _ = A.knn_match2_device(dq, dt[half:], base_index=half)
    m = A.knn2_merge_device(torch.stack([a, b]))
    torch.cuda.synchronize()
    assert np.array_equal(m.cpu().numpy(), r)


def _stereo_inputs(oracle, seed, nf=1200):
    L, R = synth.stereo_pair(752, 480, seed)
    eL, eR = oracle.Extractor(nf), oracle.Extractor(nf)
    kL, dL, _ = eL(L, (0, 0))
    kR, dR, _ = eR(R, (0, 0))
    pl = [eL.pyramid_level(l) for l in range(8)]
    pr = [eR.pyramid_level(l) for l in range(8)]
    ref = oracle.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, MBF, MBF / FX)
    return L, R, kL, dL, kR, dR, ref


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_stereo_match_matches_oracle(oracle, seed):
    A = _gpu()
    L, R, kL, dL, kR, dR, (uR, depth, _, _) = _stereo_inputs(oracle, seed)
    gl = A.ORBextractor(1200, 1.2, 8, 20, 7)
    gr = A.ORBextractor(1200, 1.2, 8, 20, 7)
    kgl, dgl, _ = gl(L, (0, 0))
    kgr, dgr, _ = gr(R, (0, 0))
    assert np.array_equal(kgl, kL) and np.array_equal(kgr, kR)
    u, d = A.compute_stereo_matches(gl, gr, kgl, dgl, kgr, dgr, MBF, MBF / FX)
    assert (uR >= 0).sum() > 50
    assert np.array_equal(u, uR)
    assert np.array_equal(d, depth)


def _random_stereo_cases():
    rng = np.random.default_rng(424242)
    for case in range(10):
        w = int(rng.integers(320, 1300))
        h = int(rng.integers(200, min(800, w * 6 // 5) + 1))
        sf = float(rng.choice([1.1, 1.2, 1.5]))
        nl = int(rng.integers(2, 9))
        while nl > 1 and min(w, h) / sf ** (nl - 1) < 12:
            nl -= 1
        nf = int(rng.integers(300, 2500))
        mbf, fx = float(rng.uniform(20, 500)), float(rng.uniform(300, 800))
        yield case, w, h, sf, nl, nf, mbf, fx


def test_stereo_random_geometries(oracle):
    # Frame sizes, pyramid shapes, quotas and calibrations drawn at random: row bands, the disparity range
    # (minD = 0, maxD = mbf / b with b = mbf / fx) and the 11 x 11 SAD windows near the level borders all depend on them.
    A = _gpu()
    matched = 0
    for case, w, h, sf, nl, nf, mbf, fx in _random_stereo_cases():
        L, R = synth.stereo_pair(w, h, 300 + case)
        eL, eR = oracle.Extractor(nf, sf, nl, 20, 7), oracle.Extractor(nf, sf, nl, 20, 7)
        kL, dL, _ = eL(L, (0, 0))
        kR, dR, _ = eR(R, (0, 0))
        pl = [eL.pyramid_level(l) for l in range(nl)]
        pr = [eR.pyramid_level(l) for l in range(nl)]
        uR, depth, _, _ = oracle.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, mbf, mbf / fx)
        gl = A.ORBextractor(nf, sf, nl, 20, 7, w, h)
        gr = A.ORBextractor(nf, sf, nl, 20, 7, w, h)
        try:
            kgl, dgl, _ = gl(L, (0, 0))
            kgr, dgr, _ = gr(R, (0, 0))
            assert np.array_equal(kgl, kL) and np.array_equal(kgr, kR), "keypoints"
            u, d = A.compute_stereo_matches(gl, gr, kgl, dgl, kgr, dgr, mbf, mbf / fx)
            assert np.array_equal(u, uR), "uRight"
            assert np.array_equal(d, depth), "depth"
        except AssertionError as e:
            raise AssertionError(f"case {case}: {w}x{h} sf={sf} nl={nl} nf={nf} mbf={mbf} fx={fx}: {e}")
        finally:
            gl.close()
            gr.close()
        matched += int((uR >= 0).sum())
    assert matched > 500


def test_kitti_style_stereo_and_tumvi_fisheye_knn(oracle):
    """The other two stereo configurations of the metric: 1241 x 376 with 2000 features through the row-band
    search, and 512 x 512 fisheye
    frames matched by brute force with the 0.7 ratio test (Frame.cc:1309-1320)."""
    A = _gpu()
    L, R = synth.stereo_pair(1241, 376, 17)
    eL, eR = oracle.Extractor(2000), oracle.Extractor(2000)
    kL, dL, _ = eL(L, (0, 0))
    kR, dR, _ = eR(R, (0, 0))
    pl = [eL.pyramid_level(l) for l in range(8)]
    pr = [eR.pyramid_level(l) for l in range(8)]
    mbf, fx = 386.1448, 718.856                      # KITTI 00-02 calibration
    uR, depth, _, _ = oracle.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, mbf, mbf / fx)
    gl = A.ORBextractor(2000, 1.2, 8, 20, 7, 1241, 376)
    gr = A.ORBextractor(2000, 1.2, 8, 20, 7, 1241, 376)
    kgl, dgl, _ = gl(L, (0, 0))
    kgr, dgr, _ = gr(R, (0, 0))
    assert np.array_equal(kgl, kL) and np.array_equal(kgr, kR) and len(kR) > 1500
    u, d = A.compute_stereo_matches(gl, gr, kgl, dgl, kgr, dgr, mbf, mbf / fx)
    assert (uR >= 0).sum() > 50 and np.array_equal(u, uR) and np.array_equal(d, depth)
    # TUM-VI style: two 512 x 512 frames, knnMatch(k = 2) + ratio 0.7
    F1, F2 = synth.stereo_pair(512, 512, 18)
    o1, o2 = oracle.Extractor(1000), oracle.Extractor(1000)
    k1, d1, _ = o1(F1, (0, 511))
    k2, d2, _ = o2(F2, (0, 511))
    g1 = A.ORBextractor(1000, 1.2, 8, 20, 7, 512, 512)
    a1, b1, m1 = g1(F1, (0, 511))
    a2, b2, m2 = g1(F2, (0, 511))
    assert np.array_equal(b1, d1) and np.array_equal(b2, d2)
    ref = oracle.knn2(d1, d2)
    got = A.knn_match2(b1, b2)
    assert np.array_equal(np.stack([got["idx0"], got["dist0"], got["idx1"], got["dist1"]], 1), ref)
    good = A.fisheye_ratio_matches(got, 0.7)
    want = np.nonzero((ref[:, 2] >= 0) & (ref[:, 1].astype(np.float64) < ref[:, 3].astype(np.float64) * 0.7))[0]
    assert np.array_equal(good, want) and len(want) > 20


def test_stereo_golden(golden_dir):
    A = _gpu()
    g = np.load(os.path.join(golden_dir, "stereo_euroc.npz"))
    L, R = synth.stereo_pair(752, 480, int(g["seed"]))
    gl = A.ORBextractor(1200, 1.2, 8, 20, 7)
    gr = A.ORBextractor(1200, 1.2, 8, 20, 7)
    kL, dL, _ = gl(L, (0, 0))
    kR, dR, _ = gr(R, (0, 0))
    mbf, fx = float(g["mbf"]), float(g["fx"])
    u, d = A.compute_stereo_matches(gl, gr, kL, dL, kR, dR, mbf, mbf / fx)
    assert np.array_equal(u, g["uRight"]) and np.array_equal(d, g["depth"])


def test_stereo_batch_device(oracle):
    A = _gpu()
    import torch
    seeds = [7, 8, 9, 10]
    pairs = [synth.stereo_pair(752, 480, s) for s in seeds]
    gl = A.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=4)
    gr = A.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=4)
    dL = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dR = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    oL = gl.extract_batch_device(dL, (0, 0))
    oR = gr.extract_batch_device(dR, (0, 0))
    u, d = A.compute_stereo_matches_device(gl, gr, oL, oR, MBF, MBF / FX)
    torch.cuda.synchronize()
    gl.status(); gr.status()
    u, d = u.cpu().numpy(), d.cpu().numpy()
    for f, s in enumerate(seeds[:2]):
        _, _, kL, _, _, _, (uR, depth, _, _) = _stereo_inputs(oracle, s)
        n = len(kL)
        assert int(oL[2][f, 0]) == n
        assert np.array_equal(u[f, :n], uR) and np.array_equal(d[f, :n], depth)


def test_pipelined_handle_sets_equal_sequential_results(oracle):
    # The throughput pattern of the integration notes: consecutive batches alternate between two sets of (left handle,
    # right handle, result arrays, stream pair) with no synchronisation in between; a set's next extraction only follows
    # that set's previous stereo match in stream order.  Every batch must come out as if it had been run alone.
    A = _gpu()
    import torch
    dev = torch.device("cuda", 0)
    nb, nsteps = 3, 6
    batches = []
    for st in range(nsteps):
        pairs = [synth.stereo_pair(752, 480, 300 + 10 * st + i) for i in range(nb)]
        batches.append((torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev),
                        torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)))
    sets = []
    for k in range(2):
        eL = A.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
        eR = A.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
        sets.append(dict(eL=eL, eR=eR, a=torch.cuda.Stream(dev), b=torch.cuda.Stream(dev)))
    cap = sets[0]["eL"].max_keypoints
    mk = lambda: (torch.empty((nb, cap, 28), dtype=torch.uint8, device=dev),
                  torch.empty((nb, cap, 32), dtype=torch.uint8, device=dev),
                  torch.empty((nb, 2), dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    results = []
    for st in range(nsteps):
        S = sets[st % 2]
        oL, oR = mk(), mk()          # fresh result arrays per step so that every step can be checked afterwards
        S["b"].wait_stream(S["a"])
        with torch.cuda.stream(S["a"]):
            S["eL"].extract_batch_device(batches[st][0], (0, 0), out=oL)
        with torch.cuda.stream(S["b"]):
            S["eR"].extract_batch_device(batches[st][1], (0, 0), out=oR)
        S["a"].wait_stream(S["b"])
        with torch.cuda.stream(S["a"]):
            u, d = A.compute_stereo_matches_device(S["eL"], S["eR"], oL, oR, MBF, MBF / FX)
            # the stereo matcher's outputs belong to the handles: copy them out in stream order
            results.append((oL, oR, u.clone(), d.clone()))
    torch.cuda.synchronize()
    for S in sets:
        S["eL"].status(); S["eR"].status()
    # the same batches one at a time on one more pair of handles
    rL = A.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
    rR = A.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
    for st in range(nsteps):
        qL = rL.extract_batch_device(batches[st][0], (0, 0))
        qR = rR.extract_batch_device(batches[st][1], (0, 0))
        u, d = A.compute_stereo_matches_device(rL, rR, qL, qR, MBF, MBF / FX)
        torch.cuda.synchronize()
        oL, oR, pu, pd = results[st]
        assert torch.equal(oL[2], qL[2]) and torch.equal(oR[2], qR[2]), f"counts of step {st}"
        for f in range(nb):
            n = int(qL[2][f, 0]); m = int(qR[2][f, 0])
            assert n > 500 and m > 500
            assert torch.equal(oL[0][f, :n], qL[0][f, :n]) and torch.equal(oL[1][f, :n], qL[1][f, :n]), f"left {st}/{f}"
            assert torch.equal(oR[0][f, :m], qR[0][f, :m]) and torch.equal(oR[1][f, :m], qR[1][f, :m]), f"right {st}/{f}"
            assert torch.equal(pu[f, :n], u[f, :n]) and torch.equal(pd[f, :n], d[f, :n]), f"stereo {st}/{f}"


def test_stereo_no_matches_is_safe():
    # unrelated left / right images: the reference would read an empty vector; we return none
    A = _gpu()
    gl = A.ORBextractor(500, 1.2, 8, 20, 7)
    gr = A.ORBextractor(500, 1.2, 8, 20, 7)
    kL, dL, _ = gl(synth.scene(752, 480, 50), (0, 0))
    flat = np.full((480, 752), 90, np.uint8)
    kR, dR, _ = gr(flat, (0, 0))
    u, d = A.compute_stereo_matches(gl, gr, kL, dL, kR, dR, MBF, MBF / FX)
    assert np.all(u == -1) and np.all(d == -1)


def _frame(oracle, seed=61, nf=1500):
    img = synth.scene(752, 480, seed)
    e = oracle.Extractor(nf)
    k, d, _ = e(img, (0, 0))
    return e, k, d


def test_window_candidates_match_get_features_in_area(oracle):
    A = _gpu()
    e, k, d = _frame(oracle)
    bounds = (0.0, 752.0, 0.0, 480.0)
    grid = oracle.Grid(k, *bounds)
    rng = np.random.default_rng(1)
    nq = 400
    q = np.zeros(nq, A.QUERY_DTYPE)
    q["x"] = rng.uniform(-30, 780, nq); q["y"] = rng.uniform(-30, 510, nq)
    lvl = rng.integers(0, 8, nq)
    q["radius"] = rng.choice([2.5, 4.0, 10.0, 40.0], nq) * e.scale_factors[lvl]
    q["min_level"] = lvl - 1; q["max_level"] = lvl
    q["min_level"][:50] = -1; q["max_level"][:50] = -1        # no level check
    q["min_level"][50:80] = 0                                   # checkLevels with min 0
    qd = synth.random_descriptors(nq, 2)
    off, idx, dist = A.window_candidates(k, d, None, bounds, q, qd)
    for i in range(nq):
        ref = grid.features_in_area(float(q["x"][i]), float(q["y"][i]), float(q["radius"][i]),
                                    int(q["min_level"][i]), int(q["max_level"][i]))
        got = idx[off[i]:off[i + 1]]
        assert np.array_equal(ref, got), i
        dd = np.unpackbits(d[got] ^ qd[i], axis=1).sum(1) if len(got) else np.zeros(0, int)
        assert np.array_equal(dd, dist[off[i]:off[i + 1]])
    assert off[-1] > 1000


def test_short_lived_threads_and_explicit_release(oracle):
    """The per-thread staging state (cached blocks, pinned arena, scratch frames, streams) is given back when a thread
    ends and by orbx_release_cached_memory(); searches before and after either must agree."""
    import threading
    A = _gpu()
    from orb_slam3_amd._lib import lib as load
    e, k, d = _frame(oracle)
    bounds = (0.0, 752.0, 0.0, 480.0)
    rng = np.random.default_rng(3)
    q = np.zeros(300, A.QUERY_DTYPE)
    q["x"] = rng.uniform(0, 752, 300); q["y"] = rng.uniform(0, 480, 300)
    q["radius"] = 25.0; q["min_level"] = -1; q["max_level"] = -1
    qd = synth.random_descriptors(300, 5)
    want = A.window_candidates(k, d, None, bounds, q, qd)
    results = []

    def work():
        results.append(A.window_candidates(k, d, None, bounds, q, qd))
        results.append(A.knn_match2(qd, d))
    for _ in range(3):
        t = threading.Thread(target=work)
        t.start(); t.join()
    load().orbx_release_cached_memory()
    load().orbx_release_cached_memory()
    again = A.window_candidates(k, d, None, bounds, q, qd)
    for got in results[0::2] + [again]:
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    for got in results[1::2]:
        assert np.array_equal(got, A.knn_match2(qd, d))


@pytest.mark.parametrize("stereo", [False, True])
def test_search_by_projection_matches_oracle(oracle, stereo):
    A = _gpu()
    e, k, d = _frame(oracle, 62)
    bounds = (0.0, 752.0, 0.0, 480.0)
    rng = np.random.default_rng(4)
    # queries = noisy copies of existing keypoints ("projected map points"), in random order,
    # several per keypoint so that the greedy 'taken' rule matters
    sel = rng.integers(0, len(k), 1200)
    nq = len(sel)
    q = np.zeros(nq, A.QUERY_DTYPE)
    q["x"] = k["x"][sel] + rng.normal(0, 1.5, nq)
    q["y"] = k["y"][sel] + rng.normal(0, 1.5, nq)
    lvl = np.clip(k["octave"][sel] + rng.integers(0, 2, nq), 0, 7)
    th = 3.0
    q["radius"] = (rng.choice([2.5, 4.0], nq) * th).astype(np.float32) * e.scale_factors[lvl]
    q["min_level"] = lvl - 1; q["max_level"] = lvl
    qd = d[sel].copy()
    flips = rng.integers(0, 256, (nq, 25))
    for i in range(nq):
        for b in flips[i, :rng.integers(0, 25)]:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    uright = None
    if stereo:
        uright = np.where(rng.random(len(k)) < 0.6, k["x"] - rng.uniform(1, 40, len(k)), -1).astype(np.float32)
        q["xr"] = q["x"] - rng.uniform(1, 40, nq)
    taken0 = (rng.random(len(k)) < 0.1).astype(np.uint8)
    grid = oracle.Grid(k, *bounds)
    ref, ref_taken, ref_n = oracle.search_by_projection(grid, d, uright, q["x"], q["y"], q["xr"], lvl, q["radius"], qd,
                                                        taken0, 0.8)
    taken = taken0.copy()
    got, n = A.ORBmatcher(0.8).SearchByProjection(k, d, uright, bounds, q, qd, taken)
    assert n == ref_n and n > 200
    assert np.array_equal(got, ref)
    assert np.array_equal(taken, ref_taken)


def test_bucket_best_matches_oracle(oracle):
    A = _gpu()
    rng = np.random.default_rng(6)
    n1, n2, nb = 900, 1100, 120
    d1 = synth.random_descriptors(n1, 10)
    d2 = synth.random_descriptors(n2, 11)
    # plant near-duplicates (and exact duplicates: the later one must win)
    b1 = rng.integers(0, nb, n1); b2 = rng.integers(0, nb, n2)
    for i in range(0, 600):
        j = int(rng.integers(0, n2))
        d2[j] = d1[i]
        b2[j] = b1[i]
        for b in rng.integers(0, 256, int(rng.integers(0, 45))):
            d2[j, b >> 3] ^= np.uint8(1 << (b & 7))
    idx1 = np.argsort(b1, kind="stable").astype(np.int32)
    idx2 = np.argsort(b2, kind="stable").astype(np.int32)
    off1 = np.concatenate([[0], np.cumsum(np.bincount(b1, minlength=nb))]).astype(np.int32)
    off2 = np.concatenate([[0], np.cumsum(np.bincount(b2, minlength=nb))]).astype(np.int32)
    d2[idx2[off2[3]:off2[3 + 1]]] = d2[idx2[off2[3]]]          # a bucket full of identical descriptors
    ri, rd = oracle.bucket_best(d1, idx1, off1, d2, idx2, off2)
    gi, gd = A.bucket_best(d1, idx1, off1, d2, idx2, off2)
    assert np.array_equal(ri, gi) and np.array_equal(rd, gd)
    assert (ri >= 0).sum() > 200
