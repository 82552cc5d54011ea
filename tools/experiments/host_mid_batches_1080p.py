"""As host_mid_batches.py, 12 .. 32 frames of 1920x1080 / 5000 features per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth
A.settle_interpreter()
W, H = 1920, 1080
imgs = [synth.scene(W, H, 70 + i, n_prims=1500) for i in range(4)]
for B in (12, 16, 24, 32):
    ex = A.ORBextractor(5000, 1.2, 8, 20, 7, W, H, max_batch=B)
    batch = [imgs[i % 4] for i in range(B)]
    for _ in range(3):
        ex.extract_batch(batch, (0, 0), copy=False)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.extract_batch(batch, (0, 0), copy=False)
    dt = time.perf_counter() - t0
    print(f"{B:3d} frames per call ({B * W * H / 2**20:5.1f} MB): {1e3 * dt / reps:6.3f} ms per call, {reps * B / dt:8.0f} frames/s", flush=True)
    ex.close()
