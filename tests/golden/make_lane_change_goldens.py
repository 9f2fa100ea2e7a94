"""Generates tests/golden/reference_lane_change.json from the UNMODIFIED reference engine with laneChange=true.

Each checkpoint is one run of oracle/_ref/cityflow_ref in its own process under oracle/_ref/libmonotonic_new.so
(LD_PRELOAD; see oracle/monotonic_new.cpp: the reference's lane-change order is heap-address order, the preload makes
addresses grow with creation order).  Run it where /root/reference exists (oracle/_ref built); the output is committed
so that machines without the reference can still check the twin against reference-produced vectors."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import lane_change_parity as lcp  # noqa: E402

CHECKPOINTS = {"example_1x1": [12, 60, 200, 500, 1000], "grid_6x6": [379, 400, 600, 1000]}


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def record(st):
    """st: lane_change_parity.state() of an engine"""
    return {
        "vehicle_count": st["count"],
        "real_vehicles": len(st["speed"]),
        "lane_count_hash": digest(sorted(st["lane_count"].items())),
        "lane_vehicles_hash": digest(st["lane_vehicles"]),          # includes "<id>_shadow" entries
        "priority_order_hash": digest(st["vehicles"]),              # get_vehicles(): ids in priority order
        "average_travel_time": float(st["average_travel_time"]).hex(),
        "state_hash": digest(sorted((k, float(st["speed"][k]).hex(), float(st["distance"][k]).hex()) for k in st["speed"])),
    }


def main():
    out = {}
    for name, steps in CHECKPOINTS.items():
        cfg = lcp.lane_change_config(name, "/tmp/cfa_lc_goldens")
        out[name] = {str(h): record(lcp.run("ref", cfg, h)) for h in steps}
        print(name, "done", flush=True)
    with open(os.path.join(HERE, "reference_lane_change.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
