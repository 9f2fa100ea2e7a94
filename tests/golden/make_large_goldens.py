"""Generates tests/golden/reference_large.json from the UNMODIFIED reference engine (oracle/_ref/cityflow_ref*.so):
BASELINE.json configs[4] — the generator-format 100x100 grid + 33 000 seeded interior flows, bench.py's
`roofline_at_scale` workload, ~1 M running vehicles at t = 300 s — stepped by the reference itself
(Engine::nextStep, /root/reference/src/engine/engine.cpp:566-594) from step 0, with checkpoint records past step 300.

Per checkpoint: vehicle count, sum and sha256 of the per-lane counts (array order = RoadNet::getLanes order, the same
bytes `get_lane_vehicle_count_array().tobytes()` gives on the engine under test), sha256 over every running vehicle's
(id, speed, distance) as exact hex floats, and — read from an Archive dump (src/engine/archive.cpp:250-300) at the last
checkpoint — sha256 over every non-virtual intersection's (id, curPhaseIndex, remainDuration).

Exact-distance ties.  Engine::updateLocation sorts ALL vehicles that change drivable in a step — one global vector, ~20 000
entries at this size — by their new distance with std::sort (engine.cpp:480): libstdc++'s introsort, which is not stable, so
the order of two vehicles that enter a lane with EXACTLY equal distances is a function of the whole vector, of heap addresses
and, with several threads, of which thread pushed first.  Such ties happen a few times in 400 steps at this size (none in the
30x30 goldens).  Two 8-thread runs of the reference agree in every count, every lane, the average travel time and the multiset
of all (speed, distance) pairs — and differ in which vehicle id carries which pair.  include/cityflow_amd.h fixes the order
(ties by vehicle number) and counts the events (cfx_scalars::tie_events).  The order matters physically, not only for the
labels: the pair sits at one position, whichever is listed first is the other's leader at gap -length, and the two have
different routes — so from the first tie on a handful of vehicles around that lane evolve differently in any two engines
(the twin and the reference differ in the id-free multiset of (speed, distance) at step 305 while every count, every lane and
the average travel time still agree at step 420).  So every record carries the id-keyed `state_hash`, the id-free
`kinematics_hash`, and — from the CPU twin, which runs beside the reference — the ties it had counted by then and ITS two
hashes.  A checker asserts counts, lanes, travel time and phases against the reference at every checkpoint; the two
per-vehicle hashes against the reference while no tie has happened, and against the twin's afterwards (the twin is pinned
to the reference, tie-free, on every smaller network: tests/test_oracle.py).

The reference runs with ONE thread and its Vehicle objects at creation-ordered addresses (LD_PRELOAD of
oracle/_ref/libmonotonic_new.so, oracle/monotonic_new.cpp; the script re-executes itself that way): where two vehicles
enter a drivable with EXACTLY equal distances, Engine::updateLocation's unstable std::sort (engine.cpp:480) leaves their
order to heap addresses and, with several threads, to which thread finishes first.  At this size that happens within the
first 150 steps: two 8-thread runs of the reference agree in every count, every lane and the average travel time, and
differ in which vehicle id carries which (speed, distance).  One thread + creation-ordered addresses is the reproducible
reference — and the order include/cityflow_amd.h fixes (ties by vehicle number).

Run it where /root/reference exists (about 25 minutes and 8 GB); the output is committed so that the GPU box, which has
no /root/reference, checks the HIP engine against reference-produced vectors at this size.

  python tests/golden/make_large_goldens.py [--grid 100] [--flows 33000] [--threads 1]
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

CHECKPOINTS = [150, 305, 360, 420]  # bench.py: 305 in the build-up, 360 = end of its timed region, 420 behind its instrumented region


def lane_array_hash(eng, lane_ids):
    import numpy as np
    lc = eng.get_lane_vehicle_count()
    arr = np.array([lc[k] for k in lane_ids], dtype=np.int32)
    return int(arr.sum()), hashlib.sha256(arr.tobytes()).hexdigest()


def state_hash(speed, distance):
    h = hashlib.sha256()
    for k in sorted(speed):
        h.update(("%s %s %s\n" % (k, float(speed[k]).hex(), float(distance[k]).hex())).encode())
    return h.hexdigest()


def kinematics_hash(speed, distance):
    """sha256 over the sorted multiset of every running vehicle's exact (speed, distance) — no ids: what stays equal when
    two vehicles that entered a lane with EXACTLY equal distances are listed in the other order (see the module docstring)."""
    h = hashlib.sha256()
    for pair in sorted((float(speed[k]).hex(), float(distance[k]).hex()) for k in speed):
        h.update(("%s %s\n" % pair).encode())
    return h.hexdigest()


def record(eng, lane_ids):
    lane_sum, lane_sha = lane_array_hash(eng, lane_ids)
    speed, distance = eng.get_vehicle_speed(), eng.get_vehicle_distance()
    return {"vehicle_count": eng.get_vehicle_count(), "lane_sum": lane_sum, "lane_array_sha256": lane_sha,
            "state_hash": state_hash(speed, distance), "kinematics_hash": kinematics_hash(speed, distance),
            "average_travel_time": float(eng.get_average_travel_time()).hex()}


def twin_records(cfg, lane_ids, out_path):
    """The CPU twin (oracle/twin, ties broken by vehicle number) on the same workload, in a process of its own: its records
    and the exact-distance ties it has counted at every checkpoint."""
    from cityflow_amd import _cityflow
    eng = _cityflow.Engine._with_backend(cfg, 1, os.path.join(ROOT, "oracle", "_ref", "libcfx_twin.so"))
    recs = {}
    for s in range(1, max(CHECKPOINTS) + 1):
        eng.next_step()
        if s in CHECKPOINTS:
            recs[str(s)] = dict(record(eng, lane_ids), tie_events=int(eng._scalars()["tie_events"]))
            print("twin checkpoint", s, recs[str(s)], flush=True)
    with open(out_path, "w") as f:
        json.dump(recs, f)


def phase_hash(lights, real):
    h = hashlib.sha256()
    for k in sorted(lights):
        if k in real:
            h.update(("%s %d %s\n" % (k, int(lights[k]["curPhaseIndex"]), float(lights[k]["remainDuration"]).hex())).encode())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=100)
    ap.add_argument("--flows", type=int, default=33000)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(HERE, "reference_large.json"))
    args = ap.parse_args()
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    pre = os.path.join(ref_dir, "libmonotonic_new.so")
    if args.threads == 1 and "CFX_VEHICLE_SIZE" not in os.environ:  # the reproducible reference: see the module docstring
        env = dict(os.environ, LD_PRELOAD=pre, CFX_VEHICLE_SIZE=open(os.path.join(ref_dir, "vehicle_size.txt")).read().strip())
        os.execve(sys.executable, [sys.executable] + sys.argv, env)
    import cityflow_ref
    import bench
    from cityflow_amd import _cityflow

    work = tempfile.mkdtemp(prefix="goldens_large_")
    scen = "gen_%dx%d" % (args.grid, args.grid)
    cfg = bench.build_workload(work, 0, scenario=scen, n_extra=args.flows)
    # lane order of the arrays: the host loader's (pinned to the reference's RoadNet::getLanes order by tests/test_host_loader.py)
    flat = _cityflow._load_roadnet(os.path.join(os.path.dirname(cfg), "roadnet.json"))
    lane_ids = flat["lane_ids"]
    real = {k for k, v in zip(flat["inter_ids"], flat["inter_virtual"]) if not v}  # (virtual intersections have no signal plan)
    twin_out = os.path.join(work, "twin_records.json")
    twin_pid = os.fork()
    if twin_pid == 0:  # (the twin beside the reference: ~4 minutes on one core)
        try:
            os.environ.pop("LD_PRELOAD", None)
            twin_records(cfg, lane_ids, twin_out)
        finally:
            sys.stdout.flush()
            os._exit(0)
    t0 = time.time()
    eng = cityflow_ref.Engine(cfg, args.threads)
    print("reference engine loaded in %.1f s" % (time.time() - t0), flush=True)
    recs = {}
    for s in range(1, max(CHECKPOINTS) + 1):
        eng.next_step()
        if s % 25 == 0:
            print("step %d: %d vehicles, %.0f s" % (s, eng.get_vehicle_count(), time.time() - t0), flush=True)
        if s in CHECKPOINTS:
            recs[str(s)] = record(eng, lane_ids)
            print("checkpoint", s, recs[str(s)], flush=True)
    dump = os.path.join(work, "end.json")
    eng.snapshot().dump(dump)
    with open(dump) as f:
        lights = json.load(f)["trafficLights"]  # (a 1.2 GB file: ~10 GB of Python objects for a minute)
    recs[str(max(CHECKPOINTS))]["phase_hash"] = phase_hash(lights, real)
    os.remove(dump)
    os.waitpid(twin_pid, 0)
    with open(twin_out) as f:
        twin = json.load(f)
    for k, r in recs.items():  # what the twin (ties by vehicle number) has at the same steps
        tw = twin[k]
        r["twin_tie_events"] = tw["tie_events"]
        r["twin_state_hash"] = tw["state_hash"]
        r["twin_kinematics_hash"] = tw["kinematics_hash"]
        same = {f: tw[f] == r[f] for f in ("vehicle_count", "lane_sum", "lane_array_sha256", "average_travel_time",
                                           "kinematics_hash", "state_hash")}
        print("step %s: twin == reference: %r (ties counted by the twin: %d)" % (k, same, tw["tie_events"]), flush=True)
        # counts, lanes and travel times must agree whatever happened; every vehicle's state while no tie has happened
        must = ["vehicle_count", "lane_sum", "lane_array_sha256", "average_travel_time"]
        if tw["tie_events"] == 0:
            must += ["kinematics_hash", "state_hash"]
        if not all(same[f] for f in must):
            raise SystemExit("the twin differs from the reference at step %s beyond what an exact-distance tie explains: %r" % (k, same))
    out = {"workload": "%s (cityflow_amd.scenarios.generate_grid, seed 0) + %d seeded interior flows (bench.build_workload)"
                       % (scen, args.flows),
           "reference_threads": args.threads, "vehicle_addresses": "creation-ordered" if os.environ.get("CFX_VEHICLE_SIZE") else "heap",
           "n_lanes": len(lane_ids), "checkpoints": recs}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("written", args.out, "in %.0f s" % (time.time() - t0), flush=True)
    os._exit(0)  # (reference destructor race)


if __name__ == "__main__":
    main()
