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
for _ in range(3):
    ex1(imgs[0], (0, 0))
t0 = time.perf_counter()
n = 50
for i in range(n):
    ex1(imgs[i % 8], (0, 0))
dt = time.perf_counter() - t0
print(f"orbx_extract, 1 frame per call:        {1e3 * dt / n:7.3f} ms per frame, {n / dt:8.1f} frames/s")
for B in (8, 32, 128):
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B)
    batch = [imgs[i % 8] for i in range(B)]
    ex.extract_batch(batch, (0, 0))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ex.extract_batch(batch, (0, 0))
    dt = time.perf_counter() - t0
    print(f"orbx_extract_batch, {B:3d} frames per call: {1e3 * dt / (reps * B):7.3f} ms per frame, {reps * B / dt:8.1f} frames/s")
