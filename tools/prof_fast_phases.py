"""Shader cycles of k_fast's phases (wave 0 of every workgroup), from the diagnostic build of the library:
    make -C orb_slam3_amd/csrc stamps
    ORBX_LIB=build/liborbx_stamps.so python3 tools/prof_fast_phases.py [batch] [w h nfeatures]
Prints the mean cycles a workgroup spends in each phase (with 16 workgroups resident per CU these are
wall-clock shares of a workgroup's lifetime, not issue cycles)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_amd as A
from orb_slam3_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, nf = (752, 480, 1000) if len(sys.argv) <= 2 else (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
L = A.lib()
assert hasattr(L, "orbx_debug_fast_stamps"), "run with ORBX_LIB=build/liborbx_stamps.so"
imgs = np.stack([synth.scene(w, h, 1000 + (i % 8)) for i in range(B)])
d = torch.from_numpy(imgs).cuda()
ex = A.ORBextractor(nf, 1.2, 8, 20, 7, w, h, max_batch=B)
buf = (C.c_ulonglong * 8)()
for i in range(3):
    ex.extract_batch_device(d, (0, 0)); torch.cuda.synchronize(); ex.status()
L.orbx_debug_fast_stamps(buf, 1)
iters = 1     # the stamps of a launch overwrite those of the one before
ex.extract_batch_device(d, (0, 0)); torch.cuda.synchronize()
L.orbx_debug_fast_stamps(buf, 1)
v = list(buf)
n = v[7]
names = ["tile load", "quick reject", "arc strength + compaction", "non-maximum suppression", "round trip + emission"]
tot = sum(v[:5])
print(f"{w}x{h}, {B} frames per launch, {n // iters} workgroups per launch; mean shader cycles per workgroup (wave 0):")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {v[k] / n:9.0f} cycles  {100.0 * v[k] / tot:5.1f} %")
print(f"  {'total':28s} {tot / n:9.0f} cycles")
print(f"  survivors of the quick reject per workgroup: {v[5] / n:.1f}; corners found by wave 0 (of 2): {v[6] / n:.1f}")
L.orbx_debug_desc_stamps(buf, 1)
v = list(buf)
n = v[7]
names = ["patch load", "orientation", "horizontal blur", "vertical blur", "steered tests + output"]
tot = sum(v[:5])
print(f"k_describe, {n} keypoint waves; mean shader cycles per wave:")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {v[k] / n:9.0f} cycles  {100.0 * v[k] / tot:5.1f} %")
print(f"  {'total':28s} {tot / n:9.0f} cycles")
