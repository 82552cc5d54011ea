"""Runs the EuRoC stereo step a few times (for rocprofv3): python3 tools/prof_stereo.py [pairs] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_amd as A
from orb_slam3_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 752, 480
dev = torch.device("cuda", 0)
pairs = [synth.stereo_pair(W, H, 1000 + i) for i in range(min(B, 8))]
dL = torch.from_numpy(np.stack([pairs[i % len(pairs)][0] for i in range(B)])).to(dev)
dR = torch.from_numpy(np.stack([pairs[i % len(pairs)][1] for i in range(B)])).to(dev)
exL = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B, device=0)
exR = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B, device=0)
oL = exL.extract_batch_device(dL, (0, 0)); oR = exR.extract_batch_device(dR, (0, 0))
torch.cuda.synchronize()
for _ in range(iters):
    A.compute_stereo_matches_device(exL, exR, oL, oR, 47.90639384423901, 47.90639384423901 / 435.2046959714599)
torch.cuda.synchronize()
