"""One-off sweep (developer tool): reference (under oracle/_ref/libmonotonic_new.so) against the CPU twin with laneChange=true
on seeded irregular networks (tests/test_irregular.py generator).
usage: python tests/tools/lane_change_sweep.py <first_seed> <n_seeds> <steps>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import lane_change_parity as lcp  # noqa: E402
import test_irregular as ti  # noqa: E402
from cityflow_amd import scenarios as scen  # noqa: E402

first, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bad = 0
for seed in range(first, first + n):
    cfg = ti.irregular(scen, "/tmp/cfa_lc_sweep", seed)
    c = json.load(open(cfg))
    c["laneChange"] = True
    json.dump(c, open(cfg, "w"))
    try:
        r = lcp.run("ref", cfg, steps)
    except RuntimeError as e:  # the reference's own asserts are live
        print(json.dumps({"seed": seed, "reference": "aborted", "stderr": str(e)[-200:]}), flush=True)
        continue
    t = lcp.run("twin", cfg, steps)
    diff = lcp.compare(r, t)
    bad += bool(diff)
    print(json.dumps({"seed": seed, "steps": steps, "identical": not diff, "differs": diff, "running": r["count"],
                      "shadows_now": r["count"] - len(r["speed"])}), flush=True)
print("seeds with differences:", bad)
