"""Batches pipelined by kernel kind instead of by handle: one stream carries the instruction-bound kernels (FAST,
describe) of every batch back to back, one or two other streams the latency-bound ones (level-0 copy, resize chain,
quadtree, layout) beside them (orbx_extract_batch_device_phase; phase 1 of batch i + 1 is enqueued before phase 2 of
batch i).  Compared with the plain pipeline (handles alternating on as many streams).

    python3 tools/experiments/phase_pipeline.py [handles] [frames] [launches] [lat_streams] [w h nfeat]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import orb_slam3_amd as A
from orb_slam3_amd import synth

NH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 32
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 1
w, h, nf = (1920, 1080, 5000) if len(sys.argv) <= 5 else (int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
dev = torch.device("cuda", 0)
L = A.lib()
fn = L.orbx_extract_batch_device_phase
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int,
               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

frames = np.stack([synth.scene(w, h, 500000 + i, n_prims=1500 if w > 1000 else 400) for i in range(2 * B)])
dF = [torch.from_numpy(frames[:B]).to(dev), torch.from_numpy(frames[B:]).to(dev)]
exs = [A.ORBextractor(nf, 1.2, 8, 20, 7, w, h, max_batch=B) for _ in range(NH)]
cap = exs[0].max_keypoints
mk = lambda: (torch.empty((B, cap, 28), dtype=torch.uint8, device=dev), torch.empty((B, cap, 32), dtype=torch.uint8, device=dev),
              torch.empty((B, 2), dtype=torch.int32, device=dev))
outs = [mk() for _ in range(NH)]
ref = [mk() for _ in range(2)]

# streams that really run side by side (two streams can share a hardware queue)
streams = [torch.cuda.Stream(dev)]
while len(streams) < max(NH, 1 + NL):
    streams.append(bench.pick_concurrent_stream(torch, dev, streams)[0])
sH, sLs = streams[0], streams[1:1 + NL]

# reference results and warm-up (buffers may regrow on the first calls)
for _ in range(2):
    for k in range(NH):
        with torch.cuda.stream(streams[k % len(streams)]):
            exs[k].extract_batch_device(dF[k % 2], (0, 0), out=outs[k])
        torch.cuda.synchronize()
        exs[k].status()
for k in range(2):
    exs[k % NH].extract_batch_device(dF[k], (0, 0), out=ref[k])
    torch.cuda.synchronize()


def plain(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % NH
        with torch.cuda.stream(streams[k % len(streams)]):
            exs[k].extract_batch_device(dF[k % 2], (0, 0), out=outs[k])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def call(phase, i):
    k = i % NH
    d = dF[i % 2]
    o = outs[k]
    st = fn(exs[k]._h, phase, d.data_ptr(), B, w, h, d.stride(1), d.stride(0), 0, 0, o[0].data_ptr(), o[1].data_ptr(), cap,
            o[2].data_ptr(), sH.cuda_stream, sLs[i % NL].cuda_stream)
    if st != 0:
        raise RuntimeError(L.orbx_last_error().decode())


def phased(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call(1, 0)
    for i in range(1, n):
        call(1, i)
        call(2, i - 1)
    call(2, n - 1)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


print(f"{NH} handles, {B} frames of {w}x{h} per launch, {launches} launches, {NL} latency stream(s)", flush=True)
for rep in range(2):
    dt = plain(launches)
    print(f"  plain  (handles alternate on {min(NH, len(streams))} streams): {launches * B / dt:9.0f} frames/s", flush=True)
    dt = phased(launches)
    for e in exs:
        e.status()
    print(f"  phased (1 hot + {NL} latency stream(s)):           {launches * B / dt:9.0f} frames/s", flush=True)
# the last two batches against the reference results
ok = True
for i in (launches - 2, launches - 1):
    k = i % NH
    ok = ok and all(torch.equal(outs[k][j], ref[i % 2][j]) for j in (2,))
    n = ref[i % 2][2][:, 0]
    for f in range(B):
        m = int(n[f])
        ok = ok and torch.equal(outs[k][0][f, :m], ref[i % 2][0][f, :m]) and torch.equal(outs[k][1][f, :m], ref[i % 2][1][f, :m])
print("results equal the plain path's:", ok)
