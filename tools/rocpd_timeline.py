"""Busy / idle analysis of a rocprofv3 rocpd database: union of kernel intervals, gaps, and a
per-kernel listing of one steady-state window."""
import sqlite3
import sys


def main(path, show=0):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"))
    rows = rows[len(rows) // 3:]          # skip warm-up
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    print(f"span {span / 1e3:.1f} us, busy (union) {busy / 1e3:.1f} us = {100 * busy / span:.1f} %, "
          f"sum of kernel durations {sum(r[1] - r[0] for r in rows) / 1e3:.1f} us, {len(gaps)} gaps, "
          f"mean gap {sum(g[0] for g in gaps) / max(len(gaps), 1) / 1e3:.2f} us")
    if show:
        for s, e, n, q in rows[:show]:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n[:40]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
