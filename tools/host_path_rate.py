"""PCIe-inclusive rate of the host entry points: images in pageable host memory in, keypoints and
descriptors in host memory out (orbx_extract / orbx_extract_batch through the Python mirror).
python3 tools/host_path_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth

W, H = 752, 480
imgs = [synth.scene(W, H, 40 + i) for i in range(8)]
ex1 = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
# CPython's first full garbage collection (35-45 ms over the ~170 000 objects the imports leave behind) would
# otherwise land inside one of the first ~130 calls: run it now (profiles/r03_slow_call_gc_probe.txt).
A.settle_interpreter()
for i in range(20):
    ex1(imgs[i % 8], (0, 0))
t0 = time.perf_counter()
n = 50
for i in range(n):
    ex1(imgs[i % 8], (0, 0))
dt = time.perf_counter() - t0
print(f"orbx_extract, 1 frame per call:        {1e3 * dt / n:7.3f} ms per frame, {n / dt:8.1f} frames/s")
for B in (8, 32, 128, 256):
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B)
    batch = [imgs[i % 8] for i in range(B)]
    for copy in (False, True):
        # copy=False: results are views into arrays the extractor reuses (what a C++ caller with its own buffers sees);
        # copy=True: a fresh copy of the batch's results per call (page faults on the new memory: about half the rate)
        ex.extract_batch(batch, (0, 0), copy=copy)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            ex.extract_batch(batch, (0, 0), copy=copy)
        dt = time.perf_counter() - t0
        print(f"orbx_extract_batch, {B:3d} frames per call, {'fresh result arrays' if copy else 'reused result arrays'}: "
              f"{1e3 * dt / (reps * B):7.3f} ms per frame, {reps * B / dt:8.1f} frames/s")

# matcher entry points (host arrays in and out): per-call latency
rng = np.random.default_rng(3)
k, d, _ = ex1(imgs[0], (0, 0))
nq = 800
src = rng.integers(0, len(k), nq)
q = np.zeros(nq, A.QUERY_DTYPE)
q["x"] = k["x"][src] + rng.normal(0, 3, nq); q["y"] = k["y"][src] + rng.normal(0, 3, nq)
lvl = k["octave"][src]
q["radius"] = 7 * (1.2 ** lvl); q["min_level"] = lvl - 1; q["max_level"] = lvl
qd = d[src].copy()
m = A.ORBmatcher(0.8)
bounds = (0.0, float(W), 0.0, float(H))
frame = A.FrameFeatures(k, d, None, bounds)
taken = np.zeros(len(k), np.uint8)
for name, fn in [("FrameFeatures.assign (upload + grid on the device)", lambda: frame.assign(k, d, None, bounds)),
                 ("SearchByProjectionIn (resident frame, 800 map points)",
                  lambda: (taken.fill(0), m.SearchByProjectionIn(frame, q, qd, taken))),
                 ("SearchWindowBestIn + rotation check (resident frame)",
                  lambda: (taken.fill(0), m.SearchWindowBestIn(frame, q, qd, 100.0, taken, None, k["angle"][src]))),
                 ("SearchByProjection (800 map points, 1000 keypoints)",
                  lambda: m.SearchByProjection(k, d, None, bounds, q, qd, np.zeros(len(k), np.uint8))),
                 ("SearchWindowBest + rotation check", lambda: m.SearchWindowBest(k, d, None, bounds, q, qd, 100.0, np.zeros(len(k), np.uint8), None, k["angle"][src])),
                 ("knn_match2 1000 x 1000", lambda: A.knn_match2(d, d))]:
    fn(); fn()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:55s} median {np.median(ts):7.3f} ms per call (min {min(ts):.3f}, max {max(ts):.3f})")

# the bucketed searches (SearchByBoW between two keyframes of ~1000 keypoints over 300 shared vocabulary nodes): host
# arrays in (both descriptor sets uploaded per call), resident frames, and the same loop with the distances computed
# on one host core (numpy popcount table) for scale
base = synth.random_descriptors(300, 80)
k1, d1 = synth.matcher_frame(rng, 1000, base, 16)
k2, d2 = synth.matcher_frame(rng, 1050, base, 16)
v1 = np.ones(len(k1), np.uint8); v2 = np.ones(len(k2), np.uint8)
fv1, fv2 = synth.feature_vector(rng, len(k1), 300), synth.feature_vector(rng, len(k2), 300)
f1, f2 = A.FrameFeatures(k1, d1, None, bounds), A.FrameFeatures(k2, d2, None, bounds)
mb = A.ORBmatcher(0.9, True)
pop = np.array([bin(i).count("1") for i in range(256)], np.uint8)


def host_pairs():
    tot = 0
    for b in range(300):
        a = fv1[0][fv1[1][b]:fv1[1][b + 1]]; c = fv2[0][fv2[1][b]:fv2[1][b + 1]]
        if len(a) and len(c):
            tot += int(pop[d1[a][:, None, :] ^ d2[c][None, :, :]].sum(2).min())
    return tot


for name, fn in [("SearchByBoWKeyFrames (host arrays, 1000 x 1050 keypoints, 300 nodes)",
                  lambda: mb.SearchByBoWKeyFrames(d1, k1["angle"], v1, fv1, d2, k2["angle"], v2, fv2)),
                 ("SearchByBoWKeyFramesIn (resident frames)", lambda: mb.SearchByBoWKeyFramesIn(f1, v1, fv1, f2, v2, fv2)),
                 ("the same pair distances on one host core (numpy)", host_pairs)]:
    fn(); fn()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:75s} median {np.median(ts):7.3f} ms per call (min {min(ts):.3f}, max {max(ts):.3f})")
