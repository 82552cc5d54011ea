"""orbx_extract_batch through the host entry point for mid-size batches (16 .. 96 frames of 752x480), to place the
size from which the chunk pipeline beats the single captured sequence (ORBX_HOST_PIPELINE_MIN_MB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth
A.settle_interpreter()
W, H = 752, 480
imgs = [synth.scene(W, H, 40 + i) for i in range(8)]
for B in (12, 16, 24, 32, 48, 64, 96):
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B)
    batch = [imgs[i % 8] for i in range(B)]
    for _ in range(3):
        ex.extract_batch(batch, (0, 0), copy=False)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.extract_batch(batch, (0, 0), copy=False)
    dt = time.perf_counter() - t0
    print(f"{B:3d} frames per call ({B * W * H / 2**20:5.1f} MB): {1e3 * dt / reps:6.3f} ms per call, {reps * B / dt:8.0f} frames/s", flush=True)
    ex.close()
