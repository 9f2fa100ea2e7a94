"""Summarise a rocprofv3 --pmc pass of SQ counters (rocpd sqlite) per kernel: averages over the last N launches.
usage: python tools/pmc_sq_summary.py <results.db> [last_n]"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    counters = [r[0] for r in cur.execute("select distinct counter_name from counters_collection")]
    kernels = [r[0] for r in cur.execute("select distinct kernel_name from counters_collection")]
    print("%-28s" % "kernel" + "".join("%22s" % c for c in counters))
    for k in kernels:
        row = []
        for c in counters:
            vals = [r[0] for r in cur.execute(
                "select value from counters_collection where counter_name=? and kernel_name=? order by start desc limit ?",
                (c, k, last_n))]
            row.append(sum(vals) / len(vals) if vals else float("nan"))
        print("%-28s" % k.split("(")[0][:28] + "".join("%22.0f" % v for v in row))


if __name__ == "__main__":
    main()
