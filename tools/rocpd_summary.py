"""Summarise a rocprofv3 (rocpd / sqlite) kernel trace into the text table committed under profiles/.
usage: python tools/rocpd_summary.py <results.db> [last_n] [> profiles/<name>.txt]
With last_n, a second table covers only the last `last_n` launches of every kernel (the steady-state window
bench.py's roofline leg measures, as opposed to the whole run including warm-up)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-34s %8s %12s %10s %10s %10s %6s %5s %5s %6s %7s %8s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch", "grid_x", "wg"))
    for r in rows:
        name = r[0].split("(")[0]
        print("%-34s %8d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %6d %7d %8d %5d" % (
            name[:34], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0))


def tail(last_n):
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select distinct name from kernels")]
    print("\nlast %d launches of each kernel" % last_n)
    print("%-34s %8s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
    out = []
    for n in names:
        d = [r[0] for r in cur.execute("select duration from kernels where name=? order by start desc limit ?", (n, last_n))]
        if d:
            out.append((sum(d) / len(d), n.split("(")[0], len(d), min(d), max(d)))
    for avg, n, cnt, mn, mx in sorted(out, reverse=True):
        print("%-34s %8d %10.2f %10.2f %10.2f" % (n[:34], cnt, avg / 1e3, mn / 1e3, mx / 1e3))


if __name__ == "__main__":
    main()
    if len(sys.argv) > 2:
        tail(int(sys.argv[2]))
