"""CPU: Archive (snapshot / load / dump / load_from_file), set_vehicle_speed, set_vehicle_route on the host + CPU twin,
mirroring the reference's tests/python/test_archive.py and checked live against the unmodified reference engine.
The same scenarios run on the HIP engine in tests/test_hip_api.py (-m gpu)."""
import json
import os
import time

import pytest

from conftest import TWIN_LIB, checkpoint_record


def make(mod, cfg):
    return mod.Engine._with_backend(cfg, 1, TWIN_LIB)


def run(e, n):
    for _ in range(n):
        e.next_step()


def record(e):  # reference tests/python/test_archive.py:25-27
    return e.get_lane_vehicle_count(), e.get_average_travel_time()


def archive_roundtrips(mod, make_engine, cfg, tmp_path):
    period = 100
    # test_save_and_load (test_archive.py:29-49)
    e = make_engine(cfg)
    run(e, period)
    a = e.snapshot()
    run(e, period)
    rec0 = record(e)
    full0 = checkpoint_record(e)
    e.load(a)
    run(e, period)
    assert record(e) == rec0 and checkpoint_record(e) == full0
    # test_multi_save_and_multi_load (test_archive.py:80-96)
    e = make_engine(cfg)
    archives, records = [], []
    for _ in range(4):
        archives.append(mod.Archive(e))
        records.append(record(e))
        run(e, period)
    for j in (2, 0, 1):
        e.load(archives[j])
        run(e, period)
        assert record(e) == records[j + 1]
    # test_save_to_file (test_archive.py:98-107)
    e = make_engine(cfg)
    run(e, period)
    path = str(tmp_path / "save.json")
    e.snapshot().dump(path)
    run(e, period)
    rec = checkpoint_record(e)
    e.load_from_file(path)
    run(e, period)
    assert checkpoint_record(e) == rec


def test_archive_roundtrips_twin(mod, scen, workdir, tmp_path):
    archive_roundtrips(mod, lambda c: make(mod, c), scen.materialize("example_1x1", workdir), tmp_path)


def test_archive_json_is_interchangeable_with_reference(mod, ref_module, scen, workdir, tmp_path):
    """A dump written here loads into the reference engine and vice versa.  Files are read the way the reference reads them —
    rapidjson's default number reader, which is not correctly rounded (csrc/host/json_number.h) — so:
    * a file THIS engine writes (every double as a literal that reader returns exactly) loads into either engine as the very
      state that was saved: both continue exactly like the engine that kept running;
    * a file the REFERENCE writes (near-shortest digits) may come back an ulp off in either engine — the reference's own
      dump -> load_from_file is lossy that way; both engines read the same values from it, the stored gap included
      (cfx_state::r_gap), and continue bit-identically."""
    cfg = scen.materialize("example_1x1", workdir)
    ours, ref = make(mod, cfg), ref_module.Engine(cfg, 1)
    run(ours, 150)
    run(ref, 150)
    p_ours, p_ref = str(tmp_path / "ours.json"), str(tmp_path / "ref.json")
    ours.snapshot().dump(p_ours)
    ref.snapshot().dump(p_ref)
    # continue both to get the expected future
    run(ours, 120)
    run(ref, 120)
    want = checkpoint_record(ref)
    assert checkpoint_record(ours) == want
    # this engine's file, loaded by both
    ours2, ref2 = make(mod, cfg), ref_module.Engine(cfg, 1)
    ours2.load_from_file(p_ours)
    ref2.load_from_file(p_ours)
    assert ours2.get_current_time() == 150.0 and ref2.get_current_time() == 150.0
    assert ours2.get_lane_vehicle_count() == ref2.get_lane_vehicle_count()
    run(ours2, 120)
    run(ref2, 120)
    assert checkpoint_record(ours2) == want
    assert checkpoint_record(ref2) == want
    # the reference's file, loaded by both
    ours3, ref3 = make(mod, cfg), ref_module.Engine(cfg, 1)
    ours3.load_from_file(p_ref)
    ref3.load_from_file(p_ref)
    assert ours3.get_current_time() == 150.0 and ref3.get_current_time() == 150.0
    assert ours3.get_vehicle_distance() == ref3.get_vehicle_distance()
    for _ in range(120):
        ours3.next_step()
        ref3.next_step()
        assert ours3.get_vehicle_distance() == ref3.get_vehicle_distance()
    assert checkpoint_record(ours3) == checkpoint_record(ref3)
    time.sleep(0.1)


def lossy_archive_file(mod, scen, workdir, name="grid_6x6", steps=300, lane_change=False):
    """An Archive file as Python's json writes it: every double as its `repr` — the shortest literal a correctly rounding
    reader returns exactly, which the reference's reader (rapidjson's default number reader, csrc/host/json_number.h) gets an
    ulp or two wrong for about one 16/17-digit literal in six.  So `dis`, `speed` and the stored `gap` of a vehicle come back
    each with its own error: the first step after the load must take its gap from the file (cfx_state::r_gap), not from the
    positions."""
    cfg = scen.materialize(name, workdir, laneChange=True) if lane_change else scen.materialize(name, workdir)
    tw = make(mod, cfg)
    run(tw, steps)
    path = os.path.join(os.path.dirname(cfg), "archive_lossy_%s_%d.json" % (name, int(lane_change)))
    tw.snapshot().dump(path)
    with open(path) as f:
        arc = json.load(f)
    with open(path, "w") as f:
        json.dump(arc, f)
    # (the point of the file: it does NOT come back as it was written)
    moved = sum(mod._parse_json_number(repr(v["dis"]))[0] != v["dis"] for v in arc["vehicles"])
    assert moved > 0
    return cfg, path


def lossy_archive_body(cfg, path, make_a, make_b, steps=40):
    """Both engines load the file and are equal — right after the load (what a dump would show, the gap included) and after
    every one of the following steps, bit for bit."""
    a, b = make_a(cfg), make_b(cfg)
    a.load_from_file(path)
    b.load_from_file(path)

    def same(where):
        assert a.get_lane_vehicles() == b.get_lane_vehicles(), where
        assert a.get_vehicle_speed() == b.get_vehicle_speed(), where
        assert a.get_vehicle_distance() == b.get_vehicle_distance(), where
        sa, sb = a._vehicle_state(), b._vehicle_state()
        has = sa["leader"] >= 0
        assert (sa["leader"] >= 0).tolist() == (sb["leader"] >= 0).tolist(), where
        assert sa["gap"][has].tolist() == sb["gap"][has].tolist(), where + ": gap"

    same("after the load")
    for s in range(steps):
        a.next_step()
        b.next_step()
        same("step %d after the load" % (s + 1))
    return a, b


def test_lossy_archive_file_twin_equals_reference(mod, ref_module, scen, workdir):
    cfg, path = lossy_archive_file(mod, scen, workdir)
    ref, tw = ref_module.Engine(cfg, 1), make(mod, cfg)
    ref.load_from_file(path)
    tw.load_from_file(path)
    assert ref.get_vehicle_distance() == tw.get_vehicle_distance()
    for s in range(40):
        ref.next_step()
        tw.next_step()
        assert ref.get_vehicle_speed() == tw.get_vehicle_speed() and ref.get_vehicle_distance() == tw.get_vehicle_distance(), s
        assert ref.get_lane_vehicles() == tw.get_lane_vehicles(), s
    time.sleep(0.2)


def test_lossy_archive_body_on_the_twin(mod, scen, workdir):
    """(CPU shadow of tests/test_hip_api.py::test_lossy_archive_file_hip_equals_twin: the twin against itself through the
    same body, so that a host-side change that breaks the body shows without a GPU)"""
    cfg, path = lossy_archive_file(mod, scen, workdir)
    lossy_archive_body(cfg, path, lambda c: make(mod, c), lambda c: make(mod, c), steps=5)


def control_script(e, steps=160):
    """Same calls on any engine with the reference API; returns per-step observations."""
    out = []
    for s in range(steps):
        if s == 40:
            ids = sorted(e.get_vehicles())
            for vid in ids[:5]:
                e.set_vehicle_speed(vid, 3.0)
        if s in (41, 42, 60):
            for vid in sorted(e.get_vehicles())[:3]:
                e.set_vehicle_speed(vid, 0.5)
        if s == 80:
            # reroute vehicles that are on road_0_1_0 (lane) towards road_1_1_1 instead of their flow's exit
            changed = []
            for vid in sorted(e.get_vehicles(True)):
                info = e.get_vehicle_info(vid)
                ok = e.set_vehicle_route(vid, ["road_1_1_1"]) if info.get("road") == "road_0_1_0" else None
                changed.append((vid, ok))
            out.append(("route", changed))
        e.next_step()
        out.append((e.get_lane_vehicle_count(), e.get_vehicle_speed() if s % 5 == 0 else None))
    out.append(checkpoint_record(e))
    return out


def test_set_vehicle_speed_and_route_match_reference(mod, ref_module, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir)
    a = control_script(make(mod, cfg))
    ref = ref_module.Engine(cfg, 1)
    b = control_script(ref)
    assert a == b
    time.sleep(0.1)


def test_unknown_vehicle_errors(mod, scen, workdir):
    e = make(mod, scen.materialize("example_1x1", workdir))
    run(e, 5)
    with pytest.raises(RuntimeError, match="not found"):
        e.set_vehicle_speed("flow_99_0", 1.0)
    with pytest.raises(RuntimeError, match="not found"):
        e.get_leader("nope")
    assert e.set_vehicle_route("flow_99_0", ["road_1_1_1"]) is False
    assert e.set_vehicle_route("flow_0_0", ["no_such_road"]) is False


def test_set_tl_phase_call_forms_and_errors(mod, ref_module, scen, workdir):
    """Engine.set_tl_phase is a vectorcall method of its own (an agent calls it once per signal and step): positional and
    by the reference's argument names (src/cityflow.cpp:35), any integer type, the errors of the generic binding; and what it
    sets is what the reference sets."""
    import numpy as np
    cfg = scen.materialize("grid_6x6", workdir, rlTrafficLight=True)
    eng, ref = mod.Engine._with_backend(cfg, 1, TWIN_LIB), ref_module.Engine(cfg, 1)
    net = eng._flat_net()
    ids = eng.intersection_ids()
    real = [i for i, v in zip(ids, net["inter_virtual"]) if not v]
    virtual = [i for i, v in zip(ids, net["inter_virtual"]) if v]
    for s in range(60):
        for k, iid in enumerate(real):
            ph = (s // 7 + k) % 4
            if k % 3 == 0:
                eng.set_tl_phase(iid, ph)
            elif k % 3 == 1:
                eng.set_tl_phase(intersection_id=iid, phase_id=np.int64(ph))
            else:
                eng.set_tl_phase(iid, phase_id=np.int32(ph))
            ref.set_tl_phase(iid, ph)
        eng.next_step()
        ref.next_step()
        assert eng.get_lane_vehicle_count() == ref.get_lane_vehicle_count(), s
    assert eng.get_vehicle_speed() == ref.get_vehicle_speed()
    assert eng.set_tl_phase(real[0], 1) is None
    with pytest.raises(RuntimeError, match="'nope' not found"):
        eng.set_tl_phase("nope", 1)
    for bad in (99, -1):
        with pytest.raises(IndexError, match="out of range for intersection '%s'" % real[0]):
            eng.set_tl_phase(real[0], bad)
    with pytest.raises(IndexError):
        eng.set_tl_phase(virtual[0], 0)
    for args, kw in (((real[0], 1.0), {}), ((real[0],), {}), ((real[0], 1, 2), {}), ((real[0],), {"phase": 1}), ((3, 1), {}),
                     ((real[0], 1), {"phase_id": 2}), ((), {"phase_id": 2}), ((real[0], "1"), {})):
        with pytest.raises(TypeError):
            eng.set_tl_phase(*args, **kw)
    with pytest.raises(TypeError):
        mod.Engine.set_tl_phase(object(), real[0], 1)
    # an engine without rlTrafficLight: the reference's message, nothing set (engine.cpp:719-722)
    plain = mod.Engine._with_backend(scen.materialize("grid_6x6", workdir), 1, TWIN_LIB)
    assert plain.set_tl_phase(real[0], 1) is None
    time.sleep(0.2)
    del ref


def test_set_tl_phases_sends_what_changed_and_forgets_on_reset(mod, ref_module, scen, workdir):
    """Engine.set_tl_phases(array): the call compares with the phases the device is known to hold and sends the difference; a
    reset or a load makes nothing known.  Against the reference driven through set_tl_phase, over a reset and a load."""
    import numpy as np
    cfg = scen.materialize("grid_6x6", workdir, rlTrafficLight=True)
    eng, ref = mod.Engine._with_backend(cfg, 1, TWIN_LIB), ref_module.Engine(cfg, 1)
    ids = eng.intersection_ids()
    virt = np.asarray(eng._flat_net()["inter_virtual"], dtype=bool)
    rng = np.random.default_rng(5)
    phases = np.zeros(len(ids), dtype=np.int32)

    def both(n, change_every):
        for s in range(n):
            if change_every and s % change_every == 0:
                phases[:] = rng.integers(0, 4, len(ids))
            phases[virt] = 77  # (entries of virtual intersections are ignored, whatever they hold)
            eng.set_tl_phases(phases)
            for i, iid in enumerate(ids):
                if not virt[i]:
                    ref.set_tl_phase(iid, int(phases[i]))
            eng.next_step()
            ref.next_step()
            assert eng.get_lane_vehicle_count() == ref.get_lane_vehicle_count(), s

    both(40, 7)
    arch_e, arch_r = eng.snapshot(), ref.snapshot()
    both(20, 1)
    eng.reset(False)
    ref.reset(False)
    both(30, None)  # the very phases of before the reset, never changing: they must reach the device again
    eng.load(arch_e)
    ref.load(arch_r)
    both(30, None)
    assert eng.get_vehicle_speed() == ref.get_vehicle_speed()
    bad = phases.copy()
    bad[np.flatnonzero(~virt)[3]] = 9
    with pytest.raises(IndexError, match="out of range for intersection"):
        eng.set_tl_phases(bad)
    with pytest.raises(RuntimeError, match="one phase per intersection"):
        eng.set_tl_phases(phases[:-1])
    eng.set_tl_phases(phases)  # (the failed call left nothing behind)
    eng.next_step()
    ref.next_step()
    assert eng.get_lane_vehicle_count() == ref.get_lane_vehicle_count()
    time.sleep(0.2)
    del ref


def test_lane_count_dicts_are_fresh_sorted_and_consistent(mod, scen, workdir):
    """get_lane_vehicle_count / get_lane_waiting_vehicle_count: a new dict per call in std::map key order, equal to the
    array getters, unaffected by what the caller does to earlier results (the binding keeps a master copy up to date
    incrementally)."""
    eng = mod.Engine._with_backend(scen.materialize("grid_6x6", workdir), 1, TWIN_LIB)
    prev = None
    for s in range(120):
        eng.next_step()
        d, w = eng.get_lane_vehicle_count(), eng.get_lane_waiting_vehicle_count()
        assert d == dict(zip(eng.lane_ids(), eng.get_lane_vehicle_count_array().tolist())), s
        assert w == dict(zip(eng.lane_ids(), eng.get_lane_waiting_vehicle_count_array().tolist())), s
        assert list(d) == sorted(d)
        if prev is not None:
            assert prev is not d
            prev["not_a_lane"] = 1
            prev[next(iter(prev))] = -5
        prev = d
    eng.reset(False)
    d = eng.get_lane_vehicle_count()
    assert sum(d.values()) == 0 and "not_a_lane" not in d


def test_string_getters_order_and_cache(mod, ref_module, scen, workdir):
    """get_vehicle_speed / get_vehicle_distance / get_lane_vehicles / get_vehicles: same content AND same key / element
    order as the reference (std::map order is string order; the binding sorts by an integer key and reuses id objects),
    across a reset (vehicle numbers are reassigned) and with pushed vehicles mixed in."""
    cfg = scen.materialize("grid_6x6", workdir)
    ours, ref = mod.Engine._with_backend(cfg, 1, TWIN_LIB), ref_module.Engine(cfg, 1)
    for rnd in range(2):
        for s in range(140):
            if s in (3, 60):
                for e in (ours, ref):
                    e.push_vehicle({"maxSpeed": 9.0}, ["road_0_1_0", "road_1_1_0", "road_2_1_0"])
            ours.next_step()
            ref.next_step()
            if s % 35 == 34:
                for name in ("get_vehicle_speed", "get_vehicle_distance", "get_lane_vehicles"):
                    a, b = getattr(ours, name)(), getattr(ref, name)()
                    assert a == b and list(a) == list(b), (name, rnd, s)
                assert ours.get_vehicles() == ref.get_vehicles() and ours.get_vehicles(True) == ref.get_vehicles(True)
        ours.reset(False)
        ref.reset(False)
    # the integer key orders ids exactly like the strings do
    for _ in range(30):
        ours.next_step()
    n = ours._scalars()["spawned_vehicle_count"]
    vids = list(range(n))
    ids = ours._vehicle_ids(__import__("numpy").array(vids, dtype="int32"))
    assert sorted(vids, key=ours._id_sort_key) == sorted(vids, key=lambda v: ids[v])
    time.sleep(0.2)
    del ref


def test_set_vehicle_route_between_a_load_and_the_next_lane(mod, ref_module, scen, workdir):
    """After a load the reference's routers point at their route's FIRST road again (Router copy constructor, router.cpp:11-14)
    until the vehicle enters its next lane: `get_vehicle_info` lists the whole route, and `set_vehicle_route` plans from that
    first road — the new route then begins behind the vehicle.  Same here (where the reference's result is usable at all)."""
    cfg = scen.materialize("grid_6x6", workdir)
    ours, ref = make(mod, cfg), ref_module.Engine(cfg, 1)
    run(ours, 120)
    run(ref, 120)
    ours.load(ours.snapshot())
    ref.load(ref.snapshot())
    rerouted = 0
    for vid in sorted(ref.get_vehicles()):
        info = ref.get_vehicle_info(vid)
        assert info == ours.get_vehicle_info(vid), vid
        roads = info["route"].split()
        if "road" in info and info["road"] in roads[1:]:  # on a lane, and no longer on its route's first road
            a, b = ref.set_vehicle_route(vid, [roads[-1]]), ours.set_vehicle_route(vid, [roads[-1]])
            assert a == b, vid
            got = ours.get_vehicle_info(vid)
            assert ref.get_vehicle_info(vid) == got, vid
            rerouted += int(a and got["route"].split()[0] != info["road"])
        if rerouted >= 10:
            break
    assert rerouted >= 3  # new routes that begin behind the vehicle
    run(ours, 80)
    run(ref, 80)
    assert checkpoint_record(ours) == checkpoint_record(ref)
    time.sleep(0.1)


def test_archive_file_is_streamed_whatever_its_member_order(mod, scen, workdir, tmp_path):
    """load_from_file streams the file (`vehicles` and `drivables` child by child, never a DOM of the whole Archive — a million
    vehicles are 1.1 GB of text).  The result must not depend on the order of the file's members: the same Archive rewritten with
    sorted keys (`drivables` BEFORE `vehicles`, every drivable's members reordered, the drivables themselves in lexicographic, not
    network, order) loads into exactly the same state; a file that lacks one of the network's drivables is refused by name."""
    import json
    from conftest import TWIN_LIB, assert_same_state, dump_json_exact
    cfg = scen.materialize("grid_6x6", workdir)
    eng = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for _ in range(220):
        eng.next_step()
    plain = str(tmp_path / "plain.json")
    eng.snapshot().dump(plain)
    with open(plain) as f:
        doc = json.load(f)
    assert list(doc).index("vehicles") < list(doc).index("drivables") and len(doc["vehicles"]) > 500

    def resorted(x):
        if isinstance(x, dict):
            return {k: resorted(x[k]) for k in sorted(x)}
        if isinstance(x, list):
            return [resorted(v) for v in x]
        return x

    shuffled = str(tmp_path / "sorted.json")
    dump_json_exact(resorted(doc), shuffled)
    a = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    b = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    a.load_from_file(plain)
    b.load_from_file(shuffled)
    assert_same_state(a, b, "right after the loads")
    assert a.get_vehicles(True) == b.get_vehicles(True)
    for _ in range(60):
        a.next_step()
        b.next_step()
    assert_same_state(a, b, "60 steps after the loads")
    assert a.get_vehicle_speed() == b.get_vehicle_speed()
    missing = dict(doc, drivables={k: v for i, (k, v) in enumerate(doc["drivables"].items()) if i != 7})
    gone = list(doc["drivables"])[7]
    broken = str(tmp_path / "missing.json")
    dump_json_exact(missing, broken)
    with pytest.raises(Exception, match=gone):
        a.load_from_file(broken)
