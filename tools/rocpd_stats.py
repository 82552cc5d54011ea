"""Per-kernel duration summary of a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace`): count, total, average, min, max in microseconds."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    print(f"{'kernel':48s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for name, n, tot, avg, mn, mx in db.execute(q):
        print(f"{name[:48]:48s} {n:7d} {tot / 1e3:12.1f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
