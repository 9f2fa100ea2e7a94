"""CPU: edge cases against the unmodified reference engine (oracle/_ref), on the host + CPU twin:
empty / windowed / invalid flows, non-unit simulation interval, laneLinks without explicit points (generated curves),
getters before the first step, push_vehicle defaults."""
import json
import os
import subprocess
import time

import pytest

from conftest import REF_DIR, TWIN_LIB, checkpoint_record, dump_json_exact


def _variant(scen, workdir, name, tag, flows=None, roadnet_edit=None, **cfg):
    base = scen.materialize(name, workdir)
    d = os.path.dirname(base)
    flow_file = None
    if flows is not None:
        flow_file = os.path.join(d, "flow_%s.json" % tag)
        with open(flow_file, "w") as f:
            json.dump(flows, f)
    path = scen.materialize(name, workdir, flow_file=flow_file, **cfg)
    if roadnet_edit is not None:
        with open(os.path.join(d, "roadnet.json")) as f:
            net = json.load(f)
        roadnet_edit(net)
        rn = "roadnet_%s.json" % tag
        with open(os.path.join(d, rn), "w") as f:
            json.dump(net, f)
        with open(path) as f:
            c = json.load(f)
        c["roadnetFile"] = rn
        path = path.replace(".json", "_%s.json" % tag)
        with open(path, "w") as f:
            json.dump(c, f)
    return path


def _both(mod, ref_module, cfg):
    return mod.Engine._with_backend(cfg, 1, TWIN_LIB), ref_module.Engine(cfg, 1)


def _lockstep(ours, ref, steps, every=1):
    for s in range(steps):
        ours.next_step()
        ref.next_step()
        if s % every == every - 1:
            assert checkpoint_record(ours) == checkpoint_record(ref), "step %d" % (s + 1)
    time.sleep(0.1)


def _example_flows(scen, workdir):
    d = os.path.dirname(scen.materialize("example_1x1", workdir))
    with open(os.path.join(d, "flow.json")) as f:
        return json.load(f)


def test_getters_before_first_step_and_empty_flows(mod, ref_module, scen, workdir):
    cfg = _variant(scen, workdir, "example_1x1", "empty", flows=[])
    ours, ref = _both(mod, ref_module, cfg)
    assert ours.get_vehicle_count() == 0 and ours.get_vehicles(True) == []
    assert ours.get_lane_vehicle_count() == ref.get_lane_vehicle_count()
    assert ours.get_average_travel_time() == ref.get_average_travel_time() == 0
    _lockstep(ours, ref, 30)
    assert ours.get_vehicle_speed() == {} and ours.get_lane_vehicles() == ref.get_lane_vehicles()


def test_flow_windows_and_intervals(mod, ref_module, scen, workdir):
    flows = _example_flows(scen, workdir)
    flows[0].update(startTime=5, endTime=40, interval=3.0)
    flows[1].update(startTime=0, endTime=0, interval=1.0)      # exactly one vehicle
    flows[2].update(startTime=20, endTime=-1, interval=7.5)
    flows[3].update(startTime=10, endTime=10, interval=0.5)    # interval < 1 is legal iff startTime == endTime
    cfg = _variant(scen, workdir, "example_1x1", "windows", flows=flows)
    ours, ref = _both(mod, ref_module, cfg)
    _lockstep(ours, ref, 200)


def test_half_second_steps(mod, ref_module, scen, workdir):
    """interval 0.5: time accumulates in binary fractions; average travel time stays bit-identical"""
    cfg = _variant(scen, workdir, "example_1x1", "half", interval=0.5)
    ours, ref = _both(mod, ref_module, cfg)
    _lockstep(ours, ref, 400, every=10)


def test_invalid_and_single_road_routes_are_dropped(mod, ref_module, scen, workdir, capfd):
    flows = _example_flows(scen, workdir)
    flows[0]["route"] = ["road_0_1_0", "road_1_0_1"]   # not connected through intersection_1_1 -> Dijkstra fails
    flows[1]["route"] = ["road_2_1_2"]                 # route.size() <= 1 -> invalid (router.cpp:239-240)
    cfg = _variant(scen, workdir, "example_1x1", "invalid", flows=flows)
    ours, ref = _both(mod, ref_module, cfg)
    _lockstep(ours, ref, 120)
    assert "Invalid route 'flow_0'" in capfd.readouterr().err
    assert not any(v.startswith("flow_0_") or v.startswith("flow_1_") for v in ours.get_vehicles(True))


def test_generated_lanelink_curves(mod, ref_module, scen, workdir):
    """laneLinks without `points` get the reference's generated curve (roadnet.cpp:212-247); geometry and dynamics agree"""
    def strip(net):
        for inter in net["intersections"]:
            for rl in inter.get("roadLinks", []):
                for ll in rl["laneLinks"]:
                    ll.pop("points", None)
    cfg = _variant(scen, workdir, "example_1x1", "nopoints", roadnet_edit=strip)
    with open(cfg) as f:
        c = json.load(f)
    roadnet = os.path.join(c["dir"], c["roadnetFile"])
    probe = os.path.join(REF_DIR, "probe_roadnet")
    if os.path.exists(probe):
        assert mod._roadnet_probe(roadnet) == subprocess.check_output([probe, roadnet])
    ours, ref = _both(mod, ref_module, cfg)
    _lockstep(ours, ref, 300, every=5)


def test_push_vehicle_defaults_match_reference(mod, ref_module, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir)
    ours, ref = _both(mod, ref_module, cfg)
    for s in range(100):
        if s in (3, 30):
            for e in (ours, ref):
                e.push_vehicle({"maxSpeed": 10.0}, ["road_2_1_2", "road_1_1_3"])   # everything else defaulted
                e.push_vehicle({}, ["road_1_0_1", "road_1_1_0"])
        ours.next_step()
        ref.next_step()
        assert checkpoint_record(ours) == checkpoint_record(ref), s
    assert ours.get_vehicles(True) == ref.get_vehicles(True)
    time.sleep(0.1)


def test_push_vehicle_with_initial_speed(mod, ref_module, scen, workdir, tmp_path):
    """VehicleInfo::speed from push_vehicle (engine.cpp:696): the vehicle enters its first lane already moving; the value
    also survives a snapshot taken while the vehicle is still waiting."""
    cfg = scen.materialize("example_1x1", workdir)
    ours, ref = _both(mod, ref_module, cfg)
    dump = str(tmp_path / "with_waiting.json")
    for s in range(120):
        if s in (2, 3, 50):
            for e in (ours, ref):
                e.push_vehicle({"speed": 9.5, "maxSpeed": 12.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({"speed": 3.0}, ["road_2_1_2", "road_1_1_3"])   # same first road: queues behind
                e.push_vehicle({"speed": 20.0, "length": 4.0}, ["road_1_0_1", "road_1_1_0"])
        if s == 4:  # some of the pushed vehicles are still in a waiting buffer here
            ours.snapshot().dump(dump)
            ours.load_from_file(dump)
        ours.next_step()
        ref.next_step()
        assert checkpoint_record(ours) == checkpoint_record(ref), s
    assert ours.get_vehicle_speed() == ref.get_vehicle_speed()
    time.sleep(0.1)


def test_custom_speed_on_a_vehicle_pushed_since_the_last_step(mod, ref_module, scen, workdir):
    """Engine::setVehicleSpeed (engine.cpp:827-834) finds a vehicle in vehicleMap from the moment push_vehicle created it: the
    speed waits in the vehicle's buffer and caps its FIRST step (getCarFollowSpeed vehicle.cpp:214,220), also when the vehicle
    has to queue behind another one first.  Round 3 raised "not found" here."""
    cfg = scen.materialize("example_1x1", workdir)
    ours, ref = _both(mod, ref_module, cfg)
    for s in range(80):
        if s in (2, 20, 21):
            for e in (ours, ref):
                n0 = len(e.get_vehicles(True))
                e.push_vehicle({"speed": 6.0, "maxSpeed": 14.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({"speed": 2.0}, ["road_2_1_2", "road_1_1_3"])      # same first road: queues behind
                e.push_vehicle({}, ["road_1_0_1", "road_1_1_0"])
                new = sorted(v for v in e.get_vehicles(True) if v.startswith("manually_pushed_"))[-3:]
                assert len(e.get_vehicles(True)) == n0 + 3
                e.set_vehicle_speed(new[0], 1.25)
                e.set_vehicle_speed(new[1], 0.5)
                e.set_vehicle_speed(new[2], 3.0)
                e.set_vehicle_speed(new[2], 2.0)   # the later call wins
        ours.next_step()
        ref.next_step()
        assert checkpoint_record(ours) == checkpoint_record(ref), s
        assert ours.get_vehicle_speed() == ref.get_vehicle_speed(), s
    with pytest.raises(RuntimeError):
        ours.set_vehicle_speed("manually_pushed_999", 1.0)
    time.sleep(0.1)


def out_of_order_archive(mod, scen, workdir):
    """An Archive in which, on several lanes, the SECOND vehicle of the list stands right at the lane's end at speed while
    the first is far behind: the next step removes a vehicle from the middle of a drivable's list while its head stays."""
    cfg = scen.materialize("grid_6x6", workdir)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for _ in range(300):
        tw.next_step()
    path = os.path.join(os.path.dirname(cfg), "archive_out_of_order.json")
    tw.snapshot().dump(path)
    arc = json.load(open(path))
    by_id = {v["id"]: v for v in arc["vehicles"]}
    lens = dict(zip(tw._drivable_ids(), tw._flat_net()["drv_length"]))
    edited = 0
    for name, dr in arc["drivables"].items():
        vs = dr.get("vehicles", [])
        if len(vs) < 3 or "_TO_" in name or edited >= 12:
            continue
        first, second, third = by_id[vs[0]], by_id[vs[1]], by_id[vs[2]]
        if lens[name] - first["dis"] < 20.0:
            continue                                                # (keep the head of the list well away from the end)
        second["dis"], second["speed"] = lens[name] - 0.5, 10.0    # the second: past the first, about to leave the lane
        # the archive carries leader / gap as state (the reference does not recompute them on load): keep them consistent
        second["gap"] = first["dis"] - first["len"] - second["dis"]
        third["gap"] = second["dis"] - second["len"] - third["dis"]
        edited += 1
    assert edited >= 4
    # (floats as literals both kinds of reader return exactly — conftest.dump_json_exact: `dis`, `speed` and `gap` of a vehicle
    # are separate literals, and the reference takes the first step's gap from the file while these engines recompute it)
    dump_json_exact(arc, path)
    return cfg, path


def test_out_of_order_archive_twin_equals_reference(mod, ref_module, scen, workdir):
    """(CPU) the twin handles the edited archive like the reference: this is what pins the device test below."""
    import time
    cfg, path = out_of_order_archive(mod, scen, workdir)
    ref = ref_module.Engine(cfg, 1)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    ref.load_from_file(path)
    tw.load_from_file(path)
    for s in range(40):
        ref.next_step()
        tw.next_step()
        assert ref.get_lane_vehicle_count() == tw.get_lane_vehicle_count(), s
        assert ref.get_vehicle_speed() == tw.get_vehicle_speed() and ref.get_vehicle_distance() == tw.get_vehicle_distance(), s
        assert ref.get_lane_vehicles() == tw.get_lane_vehicles(), s
    time.sleep(0.2)


def test_set_route_on_a_pushed_vehicle(mod, scen, workdir):
    """`set_vehicle_route` on a vehicle pushed since the last step.  The reference dies there — Router::setRoute
    (router.cpp:245-246) asks `vehicle->getCurDrivable()->isLaneLink()` and the drivable is null until planRoute has run at the
    next step (engine.cpp:450-470); run in a process of its own.  This engine answers false and goes on."""
    import sys
    import textwrap
    cfg = scen.materialize("example_1x1", workdir)
    if os.path.exists(os.path.join(REF_DIR, "libcityflow_ref.a")) or any(f.startswith("cityflow_ref") for f in os.listdir(REF_DIR)):
        code = textwrap.dedent("""
            import sys
            sys.path.insert(0, %r)
            import cityflow_ref
            e = cityflow_ref.Engine(%r, 1)
            for _ in range(5):
                e.next_step()
            e.push_vehicle({"speed": 3.0}, ["road_2_1_2", "road_1_1_3"])
            print("pushed", flush=True)
            print("answer", e.set_vehicle_route("manually_pushed_0", ["road_1_1_3"]), flush=True)
        """ % (REF_DIR, cfg))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert "pushed" in r.stdout and "answer" not in r.stdout and r.returncode < 0, (r.returncode, r.stdout, r.stderr[-300:])
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for _ in range(5):
        tw.next_step()
    tw.push_vehicle({"speed": 3.0}, ["road_2_1_2", "road_1_1_3"])
    assert tw.set_vehicle_route("manually_pushed_0", ["road_1_1_3"]) is False
    tw.next_step()
    assert "manually_pushed_0" in tw.get_vehicles(True)
    assert tw.set_vehicle_route("manually_pushed_0", ["road_1_1_3"]) in (True, False)  # (known to the device from here on)
