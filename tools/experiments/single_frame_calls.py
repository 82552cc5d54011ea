"""50 single-frame orbx_extract calls through the host entry point (run under rocprofv3 --hip-trace --kernel-trace
--memory-copy-trace to see where one call's 0.15 ms go)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth
A.settle_interpreter()
W, H = 752, 480
imgs = [synth.scene(W, H, 40 + i) for i in range(4)]
ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
for i in range(30):
    ex(imgs[i % 4], (0, 0))
t0 = time.perf_counter()
for i in range(50):
    ex(imgs[i % 4], (0, 0))
print(f"{1e3 * (time.perf_counter() - t0) / 50:.3f} ms per call")
