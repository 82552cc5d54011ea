"""How the extraction kernels' times follow the scene's corner density (752x480, 1000 features, 64 frames per launch).

    python3 tools/scene_density.py [frames_per_launch] [launches]

The benchmark's synthetic scenes are corner-dense (400 primitives per frame); indoor camera frames have more
low-texture cells.  This sweeps the number of primitives and the noise level, and prints per setting: keypoints kept per
frame, FAST candidates of frame 0, and the kernels' times per launch (hipEvents between the kernels, the extractor
alone on one stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import orb_slam3_amd as A
from orb_slam3_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H, NF = 752, 480, 1000

print(f"# {B} frames of {W}x{H} per launch, {NF} features, 8 levels x 1.2, FAST 20/7; ms per launch, mean of {launches} launches")
print(f"# {'primitives':>10} {'noise':>6} {'kp/frame':>9} {'cand/frame':>11} {'fast':>7} {'quadtree':>9} {'describe':>9} "
      f"{'pyramid':>8} {'all':>7}")
for n_prims, sigma in ((400, 2.0), (200, 2.0), (100, 2.0), (50, 1.0), (25, 1.0), (10, 0.5)):
    imgs = np.stack([synth.scene(W, H, 7000 + i, n_prims=n_prims, noise_sigma=sigma) for i in range(B)])
    d = torch.from_numpy(imgs).cuda()
    ex = A.ORBextractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
    for _ in range(3):
        out = ex.extract_batch_device(d, (0, 0))
        torch.cuda.synchronize()
        ex.status()
    ex.set_timing(True)
    for _ in range(launches):
        out = ex.extract_batch_device(d, (0, 0))
    torch.cuda.synchronize()
    t, n = ex.get_timing()
    ex.set_timing(False)
    ms = {k: v / max(n, 1) for k, v in t.items()}
    kp = float(out[2][:, 0].float().mean().item())
    cand = sum(len(ex.debug_level_candidates(l, 0)[0]) for l in range(8))
    print(f"  {n_prims:>10} {sigma:>6} {kp:>9.0f} {cand:>11} {ms['fast']:>7.4f} {ms['quadtree']:>9.4f} {ms['describe']:>9.4f} "
          f"{ms['pyramid']:>8.4f} {sum(ms.values()):>7.4f}", flush=True)
    ex.close()
