#!/usr/bin/env python3
"""Writes the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures for this path and cannot be built here (OpenCV absent),
so these vectors pin the ORACLE's behaviour ("parity unpinned" with respect to the
reference itself): they detect drift of the oracle, of the image generator and of the
HIP path against each other.  Run from the repo root:  python tools/make_golden.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from orb_slam3_amd import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name, width, height, seed, (nfeatures, scale, nlevels, ini, min), lapping, dynamic rects
EXTRACT_CASES = [
    ("euroc_mono", 752, 480, 1, (1000, 1.2, 8, 20, 7), (0, 1000), None),
    ("euroc_stereo_left", 752, 480, 2, (1200, 1.2, 8, 20, 7), (0, 0), None),
    ("tumvi_fisheye", 512, 512, 3, (1000, 1.2, 8, 20, 7), (0, 511), None),
    ("small_4lvl", 200, 160, 4, (300, 1.2, 4, 20, 7), (50, 120), None),
    ("rgbd_dynamic", 640, 480, 5, (1000, 1.2, 8, 20, 7), (0, 0), [(100, 80, 200, 160), (400, 300, 120, 100)]),
]

def main():
    os.makedirs(OUT, exist_ok=True)
    for name, w, h, seed, prm, lap, rects in EXTRACT_CASES:
        img = synth.scene(w, h, seed)
        ex = O.Extractor(*prm)
        kps, desc, mono = ex(img, lap, rects)
        np.savez_compressed(os.path.join(OUT, f"extract_{name}.npz"), width=w, height=h, seed=seed,
                            params=np.array(prm, np.float64), lapping=np.array(lap), mono_index=mono,
                            rects=np.array(rects if rects else np.zeros((0, 4)), np.int32),
                            image_crc=np.uint32(synth.checksum(img)), kps=kps, desc=desc,
                            level_counts=np.array([len(ex.level_keypoints(l)) for l in range(prm[2])]),
                            cand_counts=np.array([len(ex.level_candidates(l)) for l in range(prm[2])]))
        print(name, len(kps), mono)
    # brute-force matching with planted neighbours and ties
    q = synth.random_descriptors(200, 5)
    t, rows = synth.planted_map(q, 5000, 6)
    t[100] = t[200]            # exact duplicates: tie on every query
    np.savez_compressed(os.path.join(OUT, "knn2_planted.npz"), q_seed=5, t_seed=6, nq=200, nt=5000,
                        result=O.knn2(q, t), q_crc=np.uint32(synth.checksum(q)), t_crc=np.uint32(synth.checksum(t)))
    # stereo matching on a synthetic rectified pair
    L, R = synth.stereo_pair(752, 480, 7)
    eL, eR = O.Extractor(1200, 1.2, 8, 20, 7), O.Extractor(1200, 1.2, 8, 20, 7)
    kL, dL, _ = eL(L, (0, 0)); kR, dR, _ = eR(R, (0, 0))
    pl = [eL.pyramid_level(l) for l in range(8)]; pr = [eR.pyramid_level(l) for l in range(8)]
    mbf, fx = 47.90639384423901, 435.2046959714599
    uR, depth, bd, bi = O.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, mbf, mbf / fx)
    np.savez_compressed(os.path.join(OUT, "stereo_euroc.npz"), seed=7, mbf=mbf, fx=fx, uRight=uR, depth=depth,
                        n_left=len(kL), n_right=len(kR), left_crc=np.uint32(synth.checksum(L)),
                        right_crc=np.uint32(synth.checksum(R)))
    print("stereo matches", int((uR >= 0).sum()), "of", len(kL))
    # the order-dependent matcher loops, the vocabulary descent and the distinctive-descriptor choice
    np.savez_compressed(os.path.join(OUT, "matchers.npz"), **matcher_outputs(O))
    # input staging (remap / resize / grey), keypoint undistortion, image bounds and the 64 x 48 grid
    np.savez_compressed(os.path.join(OUT, "staging.npz"), **staging_outputs(O))


def staging_outputs(O):
    c = synth.staging_golden_cases()
    out = {}
    out["remap_bgr"] = O.stage(c["color"], rgb=False, maps=c["maps"])
    out["remap_rgb_small"] = O.stage(c["color"], rgb=True, maps=c["maps_small"])
    out["remap_grey"] = O.stage(c["grey"], maps=c["maps"])
    out["resize_bgr"] = O.stage(c["color"], rgb=False, size=tuple(c["resize"]))
    out["half_grey"] = O.stage(c["grey"], size=tuple(c["half"]))
    out["grey_rgb"] = O.stage(c["color"], rgb=True)
    kps = c["kps"]
    xy = O.undistort_points(np.stack([kps["x"], kps["y"]], 1), c["K"], c["D"])
    b = O.image_bounds(752, 480, c["K"], c["D"])
    un = kps.copy()
    un["x"], un["y"] = xy[:, 0], xy[:, 1]
    start, idx = O.Grid(un, *map(float, b)).csr()
    out.update(undistorted=xy, bounds=b, cell_start=start, cell_idx=idx)
    return out


def matcher_outputs(O):
    """Oracle results on synth.matcher_golden_cases()."""
    c = synth.matcher_golden_cases()
    out = {}
    w = c["window"]
    g = O.Grid(w["kps"], *[float(v) for v in w["bounds"]])
    q = w["q"]
    kpm, taken, n = O.search_window_best(g, w["desc"], w["ur"], q["x"], q["y"], q["xr"], q["radius"], q["min_level"],
                                         q["max_level"], w["qdesc"], 100.0, w["taken"], w["marks"], w["qangle"], 1.0 / 30)
    out.update(window_kp_match=kpm, window_taken=taken, window_n=np.int32(n))
    m, _, n = O.search_by_projection(g, w["desc"], w["ur"], q["x"], q["y"], q["xr"], w["lvl"], q["radius"], w["qdesc"],
                                     w["taken"], 0.8)
    out.update(projection_match=m, projection_n=np.int32(n))
    b = c["bow"]
    m12, n = O.search_by_bow_keyframes(b["d1"], b["a1"], b["v1"], b["i1"], b["o1"], b["d2"], b["a2"], b["v2"], b["i2"],
                                       b["o2"], 0.9, True)
    out.update(bow_kf_matches12=m12, bow_kf_n=np.int32(n))
    mf, n = O.search_by_bow(b["d1"], b["a1"], b["v1"], b["i1"], b["o1"], b["d2"], b["a2"], -1, b["i2"], b["o2"], 0.7, True)
    out.update(bow_frame_match=mf, bow_frame_n=np.int32(n))
    v = c["vocabulary"]
    wo, we, ni = O.bow_transform(v["child_off"], v["child_idx"], v["node_desc"], v["word_id"], v["weight"], int(v["depth"]),
                                 v["features"], 1)
    out.update(voc_word=wo, voc_weight=we, voc_node=ni)
    d = c["distinctive"]
    bi, bm = O.distinctive_descriptors(d["desc"], d["off"])
    out.update(distinct_idx=bi, distinct_median=bm)
    return out

if __name__ == "__main__":
    main()
