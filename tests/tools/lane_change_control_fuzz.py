"""One-off fuzz (developer tool): random scripts of control calls between the steps of a lane-change run (signal phases,
pushed vehicles, custom speeds on changing pairs; intervals 1 / 0.5) — the unmodified reference under the ascending-address
allocator, in its own process, against the CPU twin (tests/tools/lane_change_parity.py does the runs).
usage: python tests/tools/lane_change_control_fuzz.py <first_seed> <end_seed>"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import lane_change_parity as lcp
from cityflow_amd import scenarios as scen
wd = tempfile.mkdtemp(prefix="fuzz_lc_")
veh = {"length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0, "usualNegAcc": 4.5,
       "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5}
roads_1x1 = [["road_0_1_0", "road_1_1_0"], ["road_1_0_1", "road_1_1_1"], ["road_2_1_2", "road_1_1_2"], ["road_1_2_3", "road_1_1_3"],
             ["road_0_1_0", "road_1_1_1"], ["road_1_0_1", "road_1_1_2"]]
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    rl = bool(seed % 2)
    steps = int(rng.integers(120, 320))
    script = {}
    for s in range(steps):
        if rl and rng.random() < 0.08:
            script.setdefault(str(s), []).append(["set_tl_phase", "intersection_1_1", int(rng.integers(0, 8))])
        if rng.random() < 0.03:
            v = dict(veh, speed=float(rng.uniform(0, 8)), maxSpeed=float(rng.uniform(9, 17)), length=float(rng.uniform(4, 7)))
            script.setdefault(str(s), []).append(["push_vehicle", v, roads_1x1[int(rng.integers(0, len(roads_1x1)))]])
        if rng.random() < 0.1:
            script.setdefault(str(s), []).append(["slow_changing", int(rng.integers(1, 5)), float(rng.uniform(0, 9))])
        if rng.random() < 0.01:  # (right after a step that may have created shadows: their draws belong to the old stream)
            script.setdefault(str(s), []).append(["set_random_seed", int(rng.integers(0, 1000))])
        if s > 60 and rng.random() < 0.004:
            script.setdefault(str(s), []).append(["reset", bool(rng.integers(0, 2))])
    kw = {"rlTrafficLight": rl, "interval": (1.0, 0.5)[seed % 3 == 0], "seed": int(seed)}
    cfg = scen.materialize("example_1x1", wd, laneChange=True, **kw)
    env = {"CFX_LC_SCRIPT": json.dumps(script)}
    try:
        t = lcp.run("twin", cfg, steps, env=env)
        try:
            r = lcp.run("ref", cfg, steps, env=dict(lcp.reference_env(), **env))
        except RuntimeError as e:  # (e.g. `reset` in the middle of lane changes can crash the reference)
            print("reference failed", seed, kw, steps, str(e)[-200:], flush=True)
            continue
        d = lcp.compare(r, t)
        print(("ok" if not d else "FAIL"), seed, kw, steps, "calls", sum(len(v) for v in script.values()), "shadows", r["count"] - len(r["speed"]), d, flush=True)
        bad += bool(d)
    except Exception as e:
        print("ERROR", seed, str(e)[-400:], flush=True); bad += 1
print("bad", bad)
