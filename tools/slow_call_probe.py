"""Times every one of the first 400 orbx_extract calls of a fresh process and prints the slow ones with the process
clock at which they start and end, so that a rocprofv3 --hip-trace / --kernel-trace of the same run can be searched for
the API call or queue gap that holds the time.  python3 tools/slow_call_probe.py [n_calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_amd as A
from orb_slam3_amd import synth

W, H = 752, 480
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
imgs = [synth.scene(W, H, 40 + i) for i in range(8)]
ex = A.ORBextractor(1000, 1.2, 8, 20, 7, W, H)
import gc
first_pass = []
_t0 = [0]
def _cb0(phase, info):
    if phase == "start":
        _t0[0] = time.perf_counter_ns()
    else:
        first_pass.append((info["generation"], _t0[0], time.perf_counter_ns()))
gc.callbacks.append(_cb0)
ts = []
for i in range(n):
    t0 = time.perf_counter_ns()
    ex(imgs[i % 8], (0, 0))
    t1 = time.perf_counter_ns()
    ts.append((t0, t1))
gc.callbacks.remove(_cb0)
d = np.array([(b - a) / 1e6 for a, b in ts])
for g, a, b in first_pass:
    inside = [i for i, (s0, s1) in enumerate(ts) if s0 <= a and b <= s1]
    print(f"first pass: collection of generation {g} took {(b - a) / 1e6:.3f} ms, inside call {inside}")
med = float(np.median(d))
print(f"{n} calls: median {med:.3f} ms, max {d.max():.3f} ms at call {int(d.argmax())}")
print("clock_gettime(MONOTONIC) of the first call's start (ns):", ts[0][0], " wall (ns):", time.time_ns())
for i in np.nonzero(d > 3 * med)[0]:
    print(f"slow call {int(i):4d}: {d[i]:8.3f} ms  start_ns {ts[i][0]}  end_ns {ts[i][1]}")
w = [float(100 / (d[k:k + 100].sum() / 1e3)) for k in range(0, n - 99, 100)]
print("frames/s per window of 100 calls:", [round(x) for x in w])

# Second pass: is the slow call the interpreter's cyclic garbage collector?  gc.callbacks times every collection.
import gc
events = []
_t = [0]
def _cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter_ns()
    else:
        events.append((info["generation"], (time.perf_counter_ns() - _t[0]) / 1e6, info["collected"]))
gc.callbacks.append(_cb)
ts2 = []
for i in range(n):
    t0 = time.perf_counter_ns()
    ex(imgs[i % 8], (0, 0))
    ts2.append((time.perf_counter_ns() - t0) / 1e6)
gc.callbacks.remove(_cb)
d2 = np.array(ts2)
print(f"second pass, collector observed: max call {d2.max():.3f} ms at call {int(d2.argmax())}; objects tracked by the collector: {len(gc.get_objects())}")
for g, ms, c in events:
    if ms > 1.0:
        print(f"  collection of generation {g}: {ms:.3f} ms ({c} objects freed)")
print("  collections seen:", {g: sum(1 for e in events if e[0] == g) for g in (0, 1, 2)})
# Third pass: the same calls after gc.freeze() (every object alive now is exempt from later collections)
gc.collect(); gc.freeze()
ts3 = []
for i in range(n):
    t0 = time.perf_counter_ns()
    ex(imgs[i % 8], (0, 0))
    ts3.append((time.perf_counter_ns() - t0) / 1e6)
d3 = np.array(ts3)
print(f"third pass, after gc.freeze(): median {np.median(d3):.3f} ms, max {d3.max():.3f} ms")
