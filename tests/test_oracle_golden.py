"""The oracle against the committed golden vectors (tests/golden, made by tools/make_golden.py
from the oracle itself: the reference ships none -- parity unpinned).  Detects drift of the
oracle or of the seeded image generator."""
import glob
import os

import numpy as np
import pytest

from orb_slam3_amd import synth


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "extract_*.npz")))


def test_golden_files_present(golden_dir):
    assert len(_cases(golden_dir)) >= 5
    for f in ("knn2_planted.npz", "stereo_euroc.npz"):
        assert os.path.exists(os.path.join(golden_dir, f))


@pytest.mark.parametrize("name", ["euroc_mono", "euroc_stereo_left", "tumvi_fisheye", "small_4lvl", "rgbd_dynamic"])
def test_extract_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"extract_{name}.npz"))
    w, h, seed = int(g["width"]), int(g["height"]), int(g["seed"])
    img = synth.scene(w, h, seed)
    assert synth.checksum(img) == int(g["image_crc"]), "synthetic image generator drifted"
    p = g["params"]
    e = oracle.Extractor(int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]))
    rects = g["rects"] if len(g["rects"]) else None
    kps, desc, mono = e(img, tuple(int(v) for v in g["lapping"]), rects)
    assert mono == int(g["mono_index"])
    assert np.array_equal(kps, g["kps"])
    assert np.array_equal(desc, g["desc"])
    assert [len(e.level_keypoints(l)) for l in range(int(p[2]))] == g["level_counts"].tolist()
    assert [len(e.level_candidates(l)) for l in range(int(p[2]))] == g["cand_counts"].tolist()


def test_dynamic_rects_remove_keypoints(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "extract_rgbd_dynamic.npz"))
    k = g["kps"]
    for (x, y, w, h) in g["rects"]:
        px, py = np.rint(k["x"]), np.rint(k["y"])
        inside = (px >= x) & (px < x + w) & (py >= y) & (py < y + h)
        # level-0 keypoints can never lie inside an erase rectangle; coarser levels are tested on
        # their scaled coordinates at extraction time, which are exactly the output coordinates
        assert not inside.any()


def test_knn2_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "knn2_planted.npz"))
    q = synth.random_descriptors(int(g["nq"]), int(g["q_seed"]))
    t, rows = synth.planted_map(q, int(g["nt"]), int(g["t_seed"]))
    t[100] = t[200]
    assert synth.checksum(q) == int(g["q_crc"]) and synth.checksum(t) == int(g["t_crc"])
    r = oracle.knn2(q, t)
    assert np.array_equal(r, g["result"])
    # the planted copy is found unless something even closer exists
    for qi, row in enumerate(rows):
        assert r[qi, 1] <= int(np.unpackbits(q[qi] ^ t[row]).sum())


def test_stereo_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "stereo_euroc.npz"))
    L, R = synth.stereo_pair(752, 480, int(g["seed"]))
    assert synth.checksum(L) == int(g["left_crc"]) and synth.checksum(R) == int(g["right_crc"])
    eL, eR = oracle.Extractor(1200), oracle.Extractor(1200)
    kL, dL, _ = eL(L, (0, 0))
    kR, dR, _ = eR(R, (0, 0))
    assert len(kL) == int(g["n_left"]) and len(kR) == int(g["n_right"])
    pl = [eL.pyramid_level(l) for l in range(8)]
    pr = [eR.pyramid_level(l) for l in range(8)]
    mbf, fx = float(g["mbf"]), float(g["fx"])
    uR, depth, bd, bi = oracle.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, mbf, mbf / fx)
    assert np.array_equal(uR, g["uRight"]) and np.array_equal(depth, g["depth"])
    m = uR >= 0
    assert m.sum() > 100
    # matched points: positive disparity below maxD = fx, depth = mbf / disparity
    disp = kL["x"][m] - uR[m]
    assert np.all(disp > 0) and np.all(disp < fx)
    assert np.allclose(depth[m], mbf / disp, rtol=1e-5)


def test_matchers_golden(oracle, golden_dir):
    """The oracle's matcher loops, vocabulary descent and distinctive-descriptor choice still give the
    committed results on the seeded cases of synth.matcher_golden_cases()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(golden_dir), "..", "tools",
                                                                              "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(golden_dir, "matchers.npz"))
    now = mg.matcher_outputs(oracle)
    assert sorted(now) == sorted(g.files)
    for k in g.files:
        assert np.array_equal(np.asarray(now[k]), g[k]), k
    assert int(g["window_n"]) > 100 and int(g["bow_kf_n"]) > 10 and len(np.unique(g["voc_word"])) > 50


def _make_golden(golden_dir):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(golden_dir), "..", "tools",
                                                                              "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_staging_golden(oracle, golden_dir):
    """Input staging, undistortion, bounds and grid of the oracle against tests/golden/staging.npz."""
    g = np.load(os.path.join(golden_dir, "staging.npz"))
    out = _make_golden(golden_dir).staging_outputs(oracle)
    assert sorted(out) == sorted(g.files)
    for k in out:
        assert np.array_equal(out[k], g[k]), k
    assert (g["remap_bgr"] == 0).mean() > 0.02 and g["bounds"][0] < 0 and g["bounds"][1] > 752   # border pixels; barrel distortion
