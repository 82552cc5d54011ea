"""Profiling driver: batched extraction with several extractor handles alternating on as many streams, as the 1080p leg
of bench.py runs it (run under rocprofv3 --kernel-trace; tools/kernel_overlap.py reads the trace).

    python3 tools/prof_pipeline.py [handles] [frames_per_launch] [launches] [width height features]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import orb_slam3_amd as A
from orb_slam3_amd import synth

NH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 32
w, h, nf = (1920, 1080, 5000) if len(sys.argv) <= 4 else (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
dev = torch.device("cuda", 0)
frames = np.stack([synth.scene(w, h, 500000 + i, n_prims=1500 if w > 1000 else 400) for i in range(2 * B)])
dF = [torch.from_numpy(frames[:B]).to(dev), torch.from_numpy(frames[B:]).to(dev)]
exs = [A.ORBextractor(nf, 1.2, 8, 20, 7, w, h, max_batch=B) for _ in range(NH)]
cap = exs[0].max_keypoints
outs = [(torch.empty((B, cap, 28), dtype=torch.uint8, device=dev), torch.empty((B, cap, 32), dtype=torch.uint8, device=dev),
         torch.empty((B, 2), dtype=torch.int32, device=dev)) for _ in range(NH)]
streams = [torch.cuda.Stream(dev) for _ in range(NH)]
for _ in range(2):
    for k in range(NH):
        with torch.cuda.stream(streams[k]):
            exs[k].extract_batch_device(dF[k % 2], (0, 0), out=outs[k])
    torch.cuda.synchronize()
    for e in exs:
        e.status()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(launches):
    k = i % NH
    with torch.cuda.stream(streams[k]):
        exs[k].extract_batch_device(dF[k % 2], (0, 0), out=outs[k])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{NH} handles, {B} frames of {w}x{h} per launch, {launches} launches: {launches * B / dt:.0f} frames/s, "
      f"{1e6 * dt / launches:.1f} us per launch")
