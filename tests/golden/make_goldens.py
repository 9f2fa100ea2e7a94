"""Generates tests/golden/*.json from the UNMODIFIED reference engine (oracle/_ref/cityflow_ref*.so, built by
oracle/Makefile from /root/reference/src).  Run it where /root/reference exists; the outputs are committed so
that the GPU box (which has no /root/reference) can still check against reference-produced vectors.

  reference_checkpoints.json   per scenario / checkpoint step: vehicle count, sum and sha256 of the per-lane
                               counts, average travel time (hex float), sha256 of every running vehicle's
                               (id, speed, distance) with exact hex floats
  reference_spawns.json        per scenario: (vehicle id, priority, first lane) of every vehicle created in
                               the first N steps, read from per-step Archive dumps
  roadnet_probe.json           sha256 of oracle/_ref/probe_roadnet output (lane / laneLink lengths, crosses)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

CHECKPOINTS = {
    "example_1x1": [1, 10, 100, 200, 500, 1000],
    "grid_6x6": [250, 500, 750, 1000, 1250, 1500],
    "grid_30x30": [100, 250, 500],
}
SPAWN_STEPS = {"example_1x1": 60, "grid_6x6": 12}


def lane_hash(counts):
    return hashlib.sha256(json.dumps(sorted(counts.items())).encode()).hexdigest()


def state_hash(speed, distance):
    h = hashlib.sha256()
    for k in sorted(speed):
        h.update(("%s %s %s\n" % (k, float(speed[k]).hex(), float(distance[k]).hex())).encode())
    return h.hexdigest()


def checkpoint_record(eng):
    lc = eng.get_lane_vehicle_count()
    return {
        "vehicle_count": eng.get_vehicle_count(),
        "lane_sum": sum(lc.values()),
        "lane_hash": lane_hash(lc),
        "average_travel_time": float(eng.get_average_travel_time()).hex(),
        "state_hash": state_hash(eng.get_vehicle_speed(), eng.get_vehicle_distance()),
    }


def main():
    import cityflow_ref
    from cityflow_amd import scenarios

    work = tempfile.mkdtemp(prefix="goldens_")
    checkpoints, spawns, probes = {}, {}, {}
    for name, steps in CHECKPOINTS.items():
        cfg = scenarios.materialize(name, work)
        eng = cityflow_ref.Engine(cfg, 1)
        rec = {}
        for s in range(1, max(steps) + 1):
            eng.next_step()
            if s in steps:
                rec[str(s)] = checkpoint_record(eng)
        checkpoints[name] = rec
        time.sleep(0.2)  # reference destructor race
        del eng
        print("checkpoints", name, "done", flush=True)
    for name, n in SPAWN_STEPS.items():
        cfg = scenarios.materialize(name, work)
        eng = cityflow_ref.Engine(cfg, 1)
        seen, order = {}, []
        dump = os.path.join(work, "dump.json")
        for s in range(n):
            eng.next_step()
            eng.snapshot().dump(dump)
            with open(dump) as f:
                d = json.load(f)
            for v in d["vehicles"]:
                if v["id"] not in seen:
                    seen[v["id"]] = True
                    order.append([v["id"], v["priority"], v["drivable"], s])
        spawns[name] = sorted(order)
        time.sleep(0.2)
        del eng
        print("spawns", name, len(order), flush=True)
    for name in scenarios.NAMES:
        cfg = scenarios.materialize(name, work)
        roadnet = os.path.join(os.path.dirname(cfg), "roadnet.json")
        out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "probe_roadnet"), roadnet])
        probes[name] = {"sha256": hashlib.sha256(out).hexdigest(), "lines": out.count(b"\n")}
    for fname, obj in (("reference_checkpoints.json", checkpoints), ("reference_spawns.json", spawns),
                       ("roadnet_probe.json", probes)):
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
    print("goldens written")
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
