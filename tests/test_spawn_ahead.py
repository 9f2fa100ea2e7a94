"""CPU: the single engine's host runs the spawner of step t+1 right after it has handed step t to the device
(EngineHost::nextStep, Spawner::beginAhead / commitAhead / rollbackAhead; "cfx": {"spawnAhead": false} turns it off).  The
records of a step must not depend on when they were made: an engine that takes every step ahead and one that never does are
driven alike — an RL-style loop (signals, step, counts: the step ahead is consumed), calls that take the step back after every
step (getters by id, push_vehicle, set_vehicle_route, set_random_seed, snapshot / load, reset) — and must agree in every
spawn record, every vehicle and byte for byte in the Archive files they write.  (Against the unmodified reference the same
kinds of sequences run in tests/test_api_sequences.py, with the step ahead on.)  Reference: Engine::nextStep phases 0-1,
/root/reference/src/engine/engine.cpp:566-570, src/flow/flow.cpp:6-22, src/vehicle/vehicle.cpp:38-47."""
import json
import os

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state


def _pair(mod, scen, workdir, **extra):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_ahead.json"), 200, seed=31, interval=3.0,
                            base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow, **extra)
    out = []
    for ahead in (True, False):
        c = json.load(open(cfg))
        c["cfx"] = {"spawnAhead": ahead}
        path = cfg.replace(".json", "_ahead%d.json" % ahead)
        json.dump(c, open(path, "w"))
        out.append(mod.Engine._with_backend(path, 1, TWIN_LIB))
    return out[0], out[1], cfg


def _same(a, b, where):
    assert_same_state(a, b, where)
    assert a.get_vehicles(True) == b.get_vehicles(True), where
    assert a.get_current_time() == b.get_current_time()


def test_rl_loop_consumes_the_step_ahead(mod, scen, workdir):
    a, b, _ = _pair(mod, scen, workdir, rlTrafficLight=True)
    n = len(a.intersection_ids())
    rng = np.random.default_rng(3)
    for s in range(400):
        ph = rng.integers(0, 8, size=n).astype(np.int32)
        for e in (a, b):
            e.set_tl_phases(ph)
            e.next_step()
        assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), s
        assert np.array_equal(a.get_lane_waiting_vehicle_count_array(), b.get_lane_waiting_vehicle_count_array()), s
        assert a.get_vehicle_count() == b.get_vehicle_count()
    _same(a, b, "after the loop")
    assert a.get_vehicle_count() > 500


def test_calls_that_take_the_step_back(mod, scen, workdir, tmp_path):
    a, b, cfg = _pair(mod, scen, workdir)
    with open(os.path.join(os.path.dirname(cfg), "roadnet.json")) as f:
        roads = [r["id"] for r in json.load(f)["roads"]]
    rng = np.random.default_rng(17)
    archives = None
    for round_ in range(60):
        for _ in range(int(rng.integers(1, 6))):
            a.next_step()
            b.next_step()
        op = int(rng.integers(0, 8))
        if op == 0:
            assert a.get_vehicle_speed() == b.get_vehicle_speed()
            assert a.get_lane_vehicles() == b.get_lane_vehicles()
        elif op == 1:
            info = {"length": float(rng.uniform(3.0, 8.0)), "maxSpeed": float(rng.uniform(8.0, 16.0))}
            start = roads[int(rng.integers(0, len(roads)))]
            for e in (a, b):
                e.push_vehicle(info, [start])
        elif op == 2:
            ids = sorted(a.get_vehicles(False))
            if ids:
                vid = ids[int(rng.integers(0, len(ids)))]
                anchors = [roads[int(rng.integers(0, len(roads)))]]
                assert a.set_vehicle_route(vid, anchors) == b.set_vehicle_route(vid, anchors)
                assert a.get_vehicle_info(vid) == b.get_vehicle_info(vid)
        elif op == 3:
            for e in (a, b):
                e.set_random_seed(100 + round_)
        elif op == 4:
            pa, pb = str(tmp_path / ("a%d.json" % round_)), str(tmp_path / ("b%d.json" % round_))
            a.snapshot().dump(pa)
            b.snapshot().dump(pb)
            assert open(pa, "rb").read() == open(pb, "rb").read(), "Archive files differ in round %d" % round_
            archives = (pa, pb)
        elif op == 5 and archives:
            a.load_from_file(archives[0])
            b.load_from_file(archives[1])
        elif op == 6 and round_ % 20 == 19:
            for e in (a, b):
                e.reset(bool(round_ % 40 == 39))
            archives = None
        elif op == 7:
            assert a.get_average_travel_time() == b.get_average_travel_time()
            ids = sorted(a.get_vehicles(True))
            if ids:
                vid = ids[int(rng.integers(0, len(ids)))]
                assert a.get_vehicle_info(vid) == b.get_vehicle_info(vid)
                assert a.get_leader(vid) == b.get_leader(vid)
        _same(a, b, "round %d (op %d)" % (round_, op))


@pytest.mark.parametrize("scenario", ["grid_6x6", "example_1x1"])
def test_spawn_records_do_not_depend_on_when_they_were_made(mod, scen, workdir, scenario):
    """The spawner alone: the stream of records (vehicle, priority, lane, time, template, route, predecessor in the lane's
    queue) with every step taken plainly, taken ahead and consumed, and taken ahead, taken back and taken again."""
    d = os.path.dirname(scen.materialize(scenario, workdir))
    flow = os.path.join(d, "flow.json")
    if scenario == "grid_6x6":
        flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_ahead2.json"), 300, seed=5, interval=2.0, base_flow=flow)
    plain, ahead, back = (mod._spawn_schedule(os.path.join(d, "roadnet.json"), flow, 1.0, 7, 1, 400, m) for m in (0, 1, 2))
    assert plain == ahead
    assert plain == back
    assert sum(len(s) for s in plain) > 300


def _tiled_pair(mod, scen, workdir, **extra):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_ahead.json"), 200, seed=31, interval=3.0,
                            base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow, **extra)
    out = []
    for ahead in (True, False):
        c = json.load(open(cfg))
        c["cfx"] = {"spawnAhead": ahead}
        path = cfg.replace(".json", "_tiled_ahead%d.json" % ahead)
        json.dump(c, open(path, "w"))
        out.append(mod.TiledEngine(path, 2, 2, [], TWIN_LIB))
    return out[0], out[1], cfg


def test_tiled_engine_takes_the_step_ahead_on_a_thread_of_its_own(mod, scen, workdir, tmp_path):
    """TiledEngineHost: the spawner of step t+1 runs on a host thread while step t is submitted (every rank runs the whole
    spawner).  On against off, 2x2 tiles on the twin: an RL-style loop that consumes the batches, then the calls that take a
    prepared step back (id getters, push_vehicle, set_random_seed, snapshot / load, reset with and without a reseed)."""
    a, b, cfg = _tiled_pair(mod, scen, workdir, rlTrafficLight=True)
    with open(os.path.join(os.path.dirname(cfg), "roadnet.json")) as f:
        roads = [r["id"] for r in json.load(f)["roads"]]
    n = len(json.load(open(os.path.join(os.path.dirname(cfg), "roadnet.json")))["intersections"])
    rng = np.random.default_rng(23)

    def same(where):
        assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), where
        sa, sb = a._scalars(), b._scalars()
        for k in ("active_vehicle_count", "finished_vehicle_count", "spawned_vehicle_count", "vehicle_steps", "cumulative_travel_time"):
            assert sa[k] == sb[k], (where, k, sa[k], sb[k])

    for s in range(150):  # the loop an agent runs: the prepared batch is consumed every step
        ph = rng.integers(0, 8, size=n).astype(np.int32)
        for e in (a, b):
            e.set_tl_phases(ph)
            e.next_step()
        same("loop step %d" % s)
    assert a.get_vehicle_count() > 300
    archives = None
    for round_ in range(40):
        for _ in range(int(rng.integers(1, 5))):
            a.next_step()
            b.next_step()
        op = int(rng.integers(0, 6))
        if op == 0:
            assert a.get_vehicle_speed() == b.get_vehicle_speed()
            assert a.get_vehicles(True) == b.get_vehicles(True)
        elif op == 1:
            info = {"length": float(rng.uniform(3.0, 8.0)), "maxSpeed": float(rng.uniform(8.0, 16.0))}
            start = roads[int(rng.integers(0, len(roads)))]
            for e in (a, b):
                e.push_vehicle(info, [start])
        elif op == 2:
            for e in (a, b):
                e.set_random_seed(300 + round_)
        elif op == 3:
            pa, pb = str(tmp_path / ("ta%d.json" % round_)), str(tmp_path / ("tb%d.json" % round_))
            a.snapshot().dump(pa)
            b.snapshot().dump(pb)
            assert open(pa, "rb").read() == open(pb, "rb").read(), "Archive files differ in round %d" % round_
            archives = (pa, pb)
        elif op == 4 and archives:
            a.load_from_file(archives[0])
            b.load_from_file(archives[1])
        elif op == 5 and round_ % 10 == 9:
            for e in (a, b):
                e.reset(bool(round_ % 20 == 19))
            archives = None
        same("round %d (op %d)" % (round_, op))
    assert a.get_vehicle_distance() == b.get_vehicle_distance()
    assert a.get_average_travel_time() == b.get_average_travel_time()
