"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite) into profiles/<name>.json.
usage: python tools/pmc_summary.py <fetch.db> <write.db> [last_n] [workload] > profiles/<name>.json
(`workload` defaults to "bench" = the default `python bench.py` run; bench.py only quotes a summary whose workload is
"bench" and whose kernel_source_sha matches the sources it runs)
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 128-B requests as
64 B for wide (16 B/lane) coalesced streams, i.e. may under-report reads by up to 2x; other widths are uncalibrated.
Both the raw sum and the fetch-doubled upper bound are recorded."""
import json
import sqlite3
import sys


def per_kernel(db_path, counter, last_n):
    cur = sqlite3.connect(db_path).cursor()
    out = {}
    names = [r[0] for r in cur.execute("select distinct kernel_name from counters_collection where counter_name=?", (counter,))]
    for n in names:
        vals = [r[0] for r in cur.execute(
            "select value from counters_collection where counter_name=? and kernel_name=? order by start desc limit ?",
            (counter, n, last_n))]
        if vals:
            out[n.split("(")[0]] = {"launches": len(vals), "avg_kb": sum(vals) / len(vals), "max_kb": max(vals)}
    return out


def main():
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    f = per_kernel(sys.argv[1], "FETCH_SIZE", last_n)
    w = per_kernel(sys.argv[2], "WRITE_SIZE", last_n)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {"window": "last %d launches of each kernel" % last_n, "kernels": {},
           "kernel_source_sha": bench.kernel_source_sha(), "workload": sys.argv[4] if len(sys.argv) > 4 else "bench",
           "note": "FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 --pmc passes; hbm_bytes_fetch_doubled applies the "
                   "gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read)"}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, {}).get("avg_kb", 0.0)
        wk = w.get(k, {}).get("avg_kb", 0.0)
        res["kernels"][k] = {
            "fetch_kb": fk, "write_kb": wk,
            "hbm_bytes_raw": (fk + wk) * 1024.0,
            "hbm_bytes_fetch_doubled": (2 * fk + wk) * 1024.0,
        }
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
