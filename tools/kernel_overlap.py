"""Who runs beside whom: concurrency analysis of a rocprofv3 kernel trace (CSV) of a multi-stream run.

    python3 tools/kernel_overlap.py <..._kernel_trace.csv> [skip_fraction] [end_fraction]

Over the window that starts after the first `skip_fraction` (default 0.4) of the dispatches (and ends at `end_fraction`
of them, default 1): the share of the time with
0, 1, 2, ... kernels in flight, the share with at least one of the issue-bound kernels (k_fast, k_describe) in flight, and
per kernel name the mean duration in this run (stretched by its neighbours) and the share of its time it ran alone."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0] if n.startswith("k_") else n[:28]


def main(path, skip=0.4, end=1.0):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    rows = rows[int(len(rows) * skip):int(len(rows) * end)]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for i, (s, e, n) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    live = set()
    by_depth = defaultdict(int)
    hot = 0
    alone = defaultdict(int)
    prev = t0
    for t, kind, i in ev:
        d = t - prev
        if d > 0:
            by_depth[len(live)] += d
            if any(rows[j][2] in ("k_fast", "k_describe", "k_describe_pre") for j in live):
                hot += d
            if len(live) == 1:
                alone[rows[next(iter(live))][2]] += d
        prev = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    span = t1 - t0
    print(f"window {span / 1e3:.1f} us, {len(rows)} dispatches")
    print("kernels in flight: " + ", ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(by_depth.items())))
    print(f"k_fast or k_describe in flight: {100 * hot / span:.1f} % of the window")
    dur, cnt = defaultdict(int), defaultdict(int)
    for s, e, n in rows:
        dur[n] += e - s; cnt[n] += 1
    print(f"{'kernel':<22} {'calls':>6} {'mean us':>9} {'sum / window':>13} {'alone':>7}")
    for n in sorted(dur, key=dur.get, reverse=True):
        print(f"{n:<22} {cnt[n]:>6} {dur[n] / cnt[n] / 1e3:>9.1f} {dur[n] / span:>13.2f} {100 * alone[n] / dur[n]:>6.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
