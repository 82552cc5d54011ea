"""Profiling driver: repeated batched extraction on one stream (run under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_amd as A
from orb_slam3_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w, h, nf = (752, 480, 1000) if len(sys.argv) <= 3 else (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
imgs = np.stack([synth.scene(w, h, 1000 + (i % 8)) for i in range(B)])
d = torch.from_numpy(imgs).cuda()
ex = A.ORBextractor(nf, 1.2, 8, 20, 7, w, h, max_batch=B)
out = None
for i in range(iters + 2):
    out = ex.extract_batch_device(d, (0, 0))
    torch.cuda.synchronize()
    ex.status()
print("ok", out[2][:4].cpu().numpy().tolist())
