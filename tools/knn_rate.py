"""Times the brute-force 2-NN search (5000 queries x 10 M map descriptors by default) on the current library
(ORBX_LIB selects another build) and checks the result of a reduced case against torch.  python3 tools/knn_rate.py [nq nt iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam3_amd as A

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev, generator=g)
out, ws = A.knn_match2_device(q, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out, ws = A.knn_match2_device(q, t, out=out, workspace=ws)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{nq} x {nt}: {1e3 * dt:.3f} ms per search, {nq * nt / dt / 1e12:.3f} T pairs/s, {nq * nt * 16 / dt / 1e12:.2f} T lane-ops/s  (lib {os.environ.get('ORBX_LIB', 'default')})")
# check on a slice: brute force with torch for 8 queries
lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.uint8, device=dev)
n2 = min(nt, 1_000_000)
o2, _ = A.knn_match2_device(q, t[:n2])
torch.cuda.synchronize()
for qi in (0, 1, nq // 2, nq - 1):
    d = lut[(t[:n2] ^ q[qi]).to(torch.int64)].sum(1, dtype=torch.int32)
    d0 = int(d.min()); i0 = int((d == d0).nonzero()[0])
    d[i0] = 1 << 20
    d1 = int(d.min()); i1 = int((d == d1).nonzero()[0])
    assert [i0, d0, i1, d1] == [int(v) for v in o2[qi].tolist()], (qi, [i0, d0, i1, d1], o2[qi].tolist())
print("checked against torch on", n2, "rows")
