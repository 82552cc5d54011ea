"""Per-call latency of the per-frame entry points that tools/host_path_rate.py does not cover (host arrays in and out,
PCIe inclusive): rectified stereo matching of one pair, the fisheye stereo matching, the vocabulary transform of one
frame, distinctive descriptors of a batch of map points, staging of one colour frame.

    python3 tools/latency_table.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import orb_slam3_amd as A
from orb_slam3_amd import synth

A.settle_interpreter()
W, H = 752, 480
MBF, FX = 47.90639384423901, 435.2046959714599


def timed(name, fn, reps=40):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{name:<72s} median {np.median(ts):7.3f} ms per call (min {ts.min():.3f}, max {ts.max():.3f})", flush=True)


L, R = synth.stereo_pair(W, H, 77)
exL = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
exR = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
kL, dL, _ = exL(L, (0, 0))
kR, dR, _ = exR(R, (0, 0))
timed("orbx_extract x 2 (left + right, one after the other)", lambda: (exL(L, (0, 0)), exR(R, (0, 0))))
timed(f"ComputeStereoMatches, one pair ({len(kL)} x {len(kR)} keypoints, pyramids resident)",
      lambda: A.compute_stereo_matches(exL, exR, kL, dL, kR, dR, MBF, MBF / FX))

# fisheye stereo: brute-force kNN over the lapping areas + ratio + triangulation
from orb_slam3_amd.matcher import camera_model, compute_stereo_fisheye_matches
cam = camera_model(1, [190.97, 190.97, 254.93, 256.89, 0.0034, 0.0007, -0.0020, 0.0002])
Rlr = np.eye(3, dtype=np.float32)
tlr = np.array([0.101, 0.0, 0.0], np.float32)
sigma2 = (1.2 ** np.arange(8)).astype(np.float32) ** 2
timed(f"ComputeStereoFishEyeMatches ({len(kL)} x {len(kR)} keypoints, whole frames lapping)",
      lambda: compute_stereo_fisheye_matches(kL, dL, 0, kR, dR, 0, cam, cam, Rlr, tlr, sigma2))

# vocabulary transform of one frame (k = 10, L = 6 as ORBvoc; synthetic vocabulary)
from orb_slam3_amd.vocabulary import Vocabulary
voc = synth.vocabulary(10, 4, 3)
V = Vocabulary(*voc)
timed(f"ComputeBoW: transform of {len(dL)} descriptors (synthetic vocabulary 10^4 words, levelsup 2)", lambda: V.transform(dL, 2))

# distinctive descriptors of 500 map points with 3..12 observations each
rng = np.random.default_rng(9)
counts = rng.integers(3, 13, 500)
off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
desc = synth.random_descriptors(int(off[-1]), 11)
timed(f"ComputeDistinctiveDescriptors, 500 map points, {int(off[-1])} observations", lambda: A.distinctive_descriptors(desc, off))

# plain brute-force searches of per-frame size
q1, t1 = synth.random_descriptors(1000, 5), synth.random_descriptors(1000, 6)
timed("knn_match2 1000 x 1000", lambda: A.knn_match2(q1, t1))
q2, t2 = synth.random_descriptors(2000, 5), synth.random_descriptors(20000, 6)
timed("knn_match2 2000 x 20000", lambda: A.knn_match2(q2, t2))
