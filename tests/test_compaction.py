"""Bounded memory: the engine forgets its finished vehicles.  The reference frees a vehicle when it finishes
(src/engine/engine.cpp:296-310); here host and device keep a table row per vehicle NUMBER, and `EngineHost::compactVehicles`
(csrc/host/archive.cpp) — automatic once enough numbers have been handed out, `"cfx": {"compactVehicles": N}` — renumbers the
vehicles that are still waiting or running, in their old order, through the path a load takes.  Nothing a caller can see may
change: the random call sequences of tests/test_api_sequences.py run against the unmodified reference with a compaction every
few vehicles; here: the tables stay bounded, a compacting engine equals one that never compacts, on the twin (CPU) and on the
GPU."""
import json

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_hip_backend


def compacting(cfg, every, **more):
    with open(cfg) as f:
        c = json.load(f)
    c["cfx"] = dict(c.get("cfx", {}), compactVehicles=every, **more)
    path = cfg.replace(".json", "_compact%d.json" % every)
    with open(path, "w") as f:
        json.dump(c, f)
    return path


def unsaturated_grid(scen, workdir):
    cfg = scen.generate_grid(6, 6, workdir, flow_interval=12.0)
    with open(cfg) as f:
        c = json.load(f)
    c["rlTrafficLight"] = True
    path = cfg.replace(".json", "_rl.json")
    with open(path, "w") as f:
        json.dump(c, f)
    return path


def visible(e):
    return {"lanes": e.get_lane_vehicle_count(), "waiting": e.get_lane_waiting_vehicle_count(), "speed": e.get_vehicle_speed(),
            "distance": e.get_vehicle_distance(), "lane_vehicles": e.get_lane_vehicles(), "vehicles": e.get_vehicles(True),
            "count": e.get_vehicle_count(), "time": e.get_current_time(), "travel": e.get_average_travel_time()}


def run_pair(a, b, steps, check_every, rl=True):
    """`a` compacts, `b` never does; both take the same calls."""
    n_inter = len(a.intersection_ids())
    rng = np.random.default_rng(3)
    for s in range(steps):
        if rl and s % 5 == 0:
            ph = rng.integers(0, 4, n_inter).astype(np.int32)
            a.set_tl_phases(ph)
            b.set_tl_phases(ph)
        a.next_step()
        b.next_step()
        if s % 7 == 0:
            assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), s
        if s % check_every == check_every - 1:
            va, vb = visible(a), visible(b)
            for k in va:
                assert va[k] == vb[k], (s, k)
            some = va["vehicles"][:: max(1, len(va["vehicles"]) // 12)]
            for v in some:
                assert a.get_vehicle_info(v) == b.get_vehicle_info(v), (s, v)
                assert a.get_leader(v) == b.get_leader(v), (s, v)
            ha, hb = a._lane_history(), b._lane_history()
            for k in ha:
                assert np.array_equal(ha[k], hb[k]), (s, "lane history", k)


def test_compaction_bounds_the_tables_and_changes_nothing_twin(mod, scen, workdir):
    # (a grid whose demand the network carries: on the stock 6x6 flows most vehicles WAIT in their lanes' buffers for ever — the
    # reference keeps those too — and there is little to forget)
    base = unsaturated_grid(scen, workdir)
    a = mod.Engine._with_backend(compacting(base, 300), 1, TWIN_LIB)
    b = mod.Engine._with_backend(compacting(base, 0), 1, TWIN_LIB)
    peak = 0
    for chunk in range(6):
        run_pair(a, b, 250, 125)
        peak = max(peak, a._vehicle_table()[0])
    held, compactions = a._vehicle_table()
    created = b._vehicle_table()[0]
    assert b._vehicle_table()[1] == 0 and created > 2500
    assert compactions >= 6 and peak <= 300 + len(a.get_vehicles(True)) + 400, (held, compactions, peak)
    # a vehicle that has left is gone on both, whatever its number was
    gone = sorted(set("flow_%d_0" % f for f in range(5)) - set(a.get_vehicles(True)))
    for v in gone:
        for e in (a, b):
            with pytest.raises(RuntimeError, match="not found"):
                e.get_vehicle_info(v)
    # archives of the two still travel in both directions (the numbering is nobody's business)
    arch_a, arch_b = a.snapshot(), b.snapshot()
    run_pair(a, b, 40, 20)
    a.load(arch_b)
    b.load(arch_a)
    run_pair(a, b, 120, 40)
    a._compact_vehicles()  # on request, too
    run_pair(a, b, 60, 30)


def test_compaction_with_lane_change_and_off_by_zero(mod, scen, workdir):
    """Lane change: an id travels along a chain of copies (vehicle, shadow, the shadow's shadow ...): the chains of the vehicles
    alive stay whole, as rows of finished vehicles, and every id still finds who carries it (reference goldens with lane
    change while compacting: tests/test_lane_change.py)."""
    base = scen.materialize("example_1x1", workdir, laneChange=True)
    a = mod.Engine._with_backend(compacting(base, 9), 1, TWIN_LIB)
    b = mod.Engine._with_backend(compacting(base, 0), 1, TWIN_LIB)
    shadows = 0
    for s in range(400):
        a.next_step()
        b.next_step()
        if s % 20 == 19:
            va, vb = visible(a), visible(b)
            for k in va:
                assert va[k] == vb[k], (s, k)
            shadows += sum(v.endswith("_shadow") for vs in va["lane_vehicles"].values() for v in vs)
            for v in va["vehicles"]:
                assert a.get_vehicle_info(v) == b.get_vehicle_info(v), (s, v)
    assert shadows > 0 and a._vehicle_table()[1] > 20 and a._vehicle_table()[0] < b._vehicle_table()[0] // 3
    off = mod.Engine._with_backend(compacting(scen.materialize("example_1x1", workdir), 0), 1, TWIN_LIB)
    for _ in range(400):
        off.next_step()
    assert off._vehicle_table()[1] == 0


@pytest.mark.gpu
def test_compaction_hip_equals_a_twin_that_never_compacts(mod, scen, workdir):
    base = unsaturated_grid(scen, workdir)
    a = mod.Engine(compacting(base, 200), 1)
    assert_hip_backend(a)
    b = mod.Engine._with_backend(compacting(base, 0), 1, TWIN_LIB)
    run_pair(a, b, 1200, 150)
    assert a._vehicle_table()[1] >= 6 and a._vehicle_table()[0] <= len(a.get_vehicles(True)) + 200 < b._vehicle_table()[0], (a._vehicle_table(), b._vehicle_table())
    free0 = a._device_memory()[0]
    run_pair(a, b, 600, 200)
    assert abs(free0 - a._device_memory()[0]) < (4 << 20)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["auto", "dense"])
def test_compaction_on_the_bench_workload(mod, workdir, layout):
    """30x30 with ~97 k vehicles: an engine that compacts on request in the middle of the run against one that does not —
    every visible number equal afterwards, ties included (the renumbering keeps creation order)."""
    import bench
    cfg = bench.build_workload(workdir, 0, scenario="grid_30x30")
    a = mod.Engine(compacting(cfg, 0, layout=layout), 1)
    b = mod.Engine(compacting(cfg, 0, layout=layout), 1)
    assert_hip_backend(a)
    for s in range(bench.BUILD_UP_STEPS):
        a.next_step()
        b.next_step()
    before = a._vehicle_table()[0]
    a._compact_vehicles()
    assert a._vehicle_table()[0] < before and a._vehicle_table()[0] == len(a.get_vehicles(True))
    for s in range(60):
        a.next_step()
        b.next_step()
        if s == 30:
            a._compact_vehicles()
    assert a.get_lane_vehicle_count() == b.get_lane_vehicle_count()
    assert a.get_vehicle_speed() == b.get_vehicle_speed() and a.get_vehicle_distance() == b.get_vehicle_distance()
    assert a.get_average_travel_time() == b.get_average_travel_time() and a.get_vehicles(True) == b.get_vehicles(True)
    sa, sb = a._scalars(), b._scalars()
    for k in ("active_vehicle_count", "finished_vehicle_count", "vehicle_steps", "cumulative_travel_time", "step"):
        assert sa[k] == sb[k], k


@pytest.mark.gpu
def test_compaction_with_lane_change_hip_equals_a_twin_that_never_compacts(mod, scen, workdir):
    base = scen.materialize("example_1x1", workdir, laneChange=True)
    a = mod.Engine(compacting(base, 11), 1)
    assert_hip_backend(a)
    b = mod.Engine._with_backend(compacting(base, 0), 1, TWIN_LIB)
    shadows = 0
    for s in range(500):
        a.next_step()
        b.next_step()
        if s % 25 == 24:
            va, vb = visible(a), visible(b)
            for k in va:
                assert va[k] == vb[k], (s, k)
            shadows += sum(v.endswith("_shadow") for vs in va["lane_vehicles"].values() for v in vs)
    assert shadows > 0 and a._vehicle_table()[1] > 20


# ---- tiles (TiledEngineHost::compactFromParts, csrc/host/tile_engine.cpp): every tile's part of the state, the vehicles alive
#      renumbered, every tile loading its part of the whole — automatic with every tile in one process, a collective over ranks
#      (cityflow_amd/tiled.py: DistributedEngine.compact_vehicles; tests/test_tiling.py::test_two_ranks_compaction_gloo)
def run_tiled_pair(a, b, steps, check_every):
    """`a`: tiles, compacting; `b`: one engine that never compacts; both take the same calls."""
    n_inter = len(b.intersection_ids())
    rng = np.random.default_rng(5)
    for s in range(steps):
        if s % 5 == 0:
            ph = rng.integers(0, 4, n_inter).astype(np.int32)
            a.set_tl_phases(ph)
            b.set_tl_phases(ph)
        a.next_step()
        b.next_step()
        if s % 7 == 0:
            assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), s
        if s % check_every == check_every - 1:
            va, vb = visible(a), visible(b)
            for k in va:
                assert va[k] == vb[k], (s, k)
            some = va["vehicles"][:: max(1, len(va["vehicles"]) // 12)]
            for v in some:
                assert a.get_vehicle_info(v) == b.get_vehicle_info(v), (s, v)
                assert a.get_leader(v) == b.get_leader(v), (s, v)


def _tiled_compaction(mod, scen, workdir, make_tiled, steps):
    base = unsaturated_grid(scen, workdir)
    a = make_tiled(compacting(base, 250))
    b = mod.Engine._with_backend(compacting(base, 0), 1, TWIN_LIB)
    peak = 0
    for chunk in range(steps // 250):
        run_tiled_pair(a, b, 250, 125)
        peak = max(peak, a._vehicle_table()[0])
        if chunk == 1:  # a custom speed for a vehicle still in its lane's waiting buffer rides through the next compaction
            waiting = [v for v in b.get_vehicles(True) if v not in set(b.get_vehicles(False))]
            for v in waiting[:3]:
                a.set_vehicle_speed(v, 3.25)
                b.set_vehicle_speed(v, 3.25)
    held, compactions = a._vehicle_table()
    assert b._vehicle_table()[1] == 0 and b._vehicle_table()[0] > peak and held < b._vehicle_table()[0]
    assert compactions >= steps // 250 - 1 and peak <= 250 + len(a.get_vehicles(True)) + 400, (held, compactions, peak)
    # archives still travel both ways between tiles that compact and an engine that does not
    arch_a, arch_b = a.snapshot(), b.snapshot()
    run_tiled_pair(a, b, 40, 20)
    a.load(arch_b)
    b.load(arch_a)
    run_tiled_pair(a, b, 100, 50)
    a._compact_vehicles()  # on request, too
    run_tiled_pair(a, b, 60, 30)
    a.reset()
    b.reset()
    run_tiled_pair(a, b, 60, 30)


def test_tiled_compaction_bounds_the_tables_and_changes_nothing_twin(mod, scen, workdir):
    _tiled_compaction(mod, scen, workdir, lambda c: mod.TiledEngine(c, 2, 3, [], TWIN_LIB), 1250)


@pytest.mark.gpu
def test_tiled_compaction_hip_equals_a_twin_engine_that_never_compacts(mod, scen, workdir):
    def make(c):
        t = mod.TiledEngine(c, 2, 2)
        t.enable_mailboxes("compact_%d" % __import__("os").getpid())
        return t
    _tiled_compaction(mod, scen, workdir, make, 1000)
