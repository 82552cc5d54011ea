"""Times the input staging kernel and staged extraction: python3 tools/prof_stage.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import orb_slam3_amd as A
from orb_slam3_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 752, 480
maps = synth.rectify_maps(W, H, 9, strength=0.3)


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for cn, kw, name in [(1, {"maps": maps}, "remap grey"), (3, {"maps": maps}, "remap BGR + grey"), (3, {}, "grey only (BGR)"),
                     (3, {"size": (640, 400)}, "resize BGR 640x400 + grey")]:
    img = synth.scene(W, H, 3) if cn == 1 else synth.color_scene(W, H, 3, cn)
    d = torch.from_numpy(np.stack([img] * B)).cuda()
    st = A.InputStager(W, H, cn, False, **kw)
    out = torch.empty((B, st.height, st.width), dtype=torch.uint8, device="cuda")
    t = timed(lambda: st.stage_batch_device(d, out=out))
    px = B * st.width * st.height
    print(f"{name:28s} {B} frames: {t:8.1f} us  ({px / t / 1e3:.1f} Gpx/s)")
    if kw.get("maps") is not None and cn == 1:
        ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H, max_batch=B)
        bufs = ex.extract_batch_device(out)
        t1 = timed(lambda: (st.stage_batch_device(d, out=out), ex.extract_batch_device(out, out=bufs)))
        t2 = timed(lambda: ex.extract_batch_staged_device(st, d, out=bufs))
        t3 = timed(lambda: ex.extract_batch_device(out, out=bufs))
        print(f"  extract only {t3:.1f} us; stage + extract {t1:.1f} us; staged extract (level 0 written by the stager) {t2:.1f} us")
