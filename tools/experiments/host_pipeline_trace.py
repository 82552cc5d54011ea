"""Stage times of the host chunk pipeline for mid-size batches: run with ORBX_HOST_TRACE=1 (and ORBX_HOST_PIPELINE_MIN_MB=4
so that 24 and 48 frames of 752x480 take the pipeline); the library prints one line per call on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth
A.settle_interpreter()
W, H = 752, 480
imgs = [synth.scene(W, H, 40 + i) for i in range(8)]
for B in (24, 48):
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B)
    batch = [imgs[i % 8] for i in range(B)]
    for _ in range(6):
        ex.extract_batch(batch, (0, 0), copy=False)
    ex.close()
