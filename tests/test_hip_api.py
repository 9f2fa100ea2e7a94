"""GPU (-m gpu): Archive and the control calls on the HIP engine — same scenarios as tests/test_archive_and_control.py."""
import time

import pytest

from conftest import TWIN_LIB, checkpoint_record
from test_archive_and_control import archive_roundtrips, control_script, lossy_archive_body, lossy_archive_file, run

pytestmark = pytest.mark.gpu


def test_archive_roundtrips_hip(mod, scen, workdir, tmp_path):
    archive_roundtrips(mod, lambda c: mod.Engine(c, 1), scen.materialize("example_1x1", workdir), tmp_path)


def test_archive_roundtrip_hip_grid(mod, scen, workdir, tmp_path):
    """snapshot mid-run on the 6x6 grid, restore into the HIP engine and into the twin: all three futures agree."""
    cfg = scen.materialize("grid_6x6", workdir)
    hip = mod.Engine(cfg, 1)
    run(hip, 300)
    a = hip.snapshot()
    path = str(tmp_path / "grid.json")
    a.dump(path)
    run(hip, 200)
    want = checkpoint_record(hip)
    hip.load(a)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    tw.load_from_file(path)
    # load_from_file renumbers vids (dense over the live vehicles), so compare through the id-keyed API
    def same(where):
        assert hip.get_lane_vehicles() == tw.get_lane_vehicles(), where
        assert hip.get_vehicle_speed() == tw.get_vehicle_speed(), where
        assert hip.get_vehicle_distance() == tw.get_vehicle_distance(), where
        assert hip.get_vehicles(True) == tw.get_vehicles(True), where
        assert hip.get_average_travel_time() == tw.get_average_travel_time(), where

    same("after restore")
    for s in range(200):
        hip.next_step()
        tw.next_step()
        if s % 20 == 19:
            same("restored step %d" % (s + 1))
    assert checkpoint_record(hip) == want
    assert checkpoint_record(tw) == want


def test_control_calls_hip_equals_twin_and_reference(mod, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir)
    a = control_script(mod.Engine(cfg, 1))
    b = control_script(mod.Engine._with_backend(cfg, 1, TWIN_LIB))
    assert a == b


def test_control_calls_hip_equals_reference(mod, ref_module, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir)
    a = control_script(mod.Engine(cfg, 1))
    ref = ref_module.Engine(cfg, 1)  # (kept alive over a pause: the reference's destructor races with its worker threads)
    b = control_script(ref)
    time.sleep(0.2)
    del ref
    assert a == b


def test_reference_dump_loads_into_hip(mod, ref_module, scen, workdir, tmp_path):
    """A file the reference wrote, loaded by the HIP engine and by a second reference engine: equal after the load and after
    each of the following steps.  (Not "equal to the reference that kept running": the reference's own reader does not return
    every literal of its own writer exactly — tests/test_archive_and_control.py::test_archive_json_is_interchangeable_...)"""
    cfg = scen.materialize("example_1x1", workdir)
    ref = ref_module.Engine(cfg, 1)
    run(ref, 200)
    path = str(tmp_path / "ref.json")
    ref.snapshot().dump(path)
    hip, ref2 = mod.Engine(cfg, 1), ref_module.Engine(cfg, 1)
    hip.load_from_file(path)
    ref2.load_from_file(path)
    assert hip.get_vehicle_distance() == ref2.get_vehicle_distance()
    for s in range(150):
        hip.next_step()
        ref2.next_step()
        assert hip.get_vehicle_distance() == ref2.get_vehicle_distance(), s
    assert checkpoint_record(hip) == checkpoint_record(ref2)
    time.sleep(0.1)


def test_push_vehicle_with_initial_speed_hip(mod, ref_module, scen, workdir, tmp_path):
    """push_vehicle({"speed": ...}) on the HIP engine vs the reference, with a snapshot taken while pushed vehicles wait."""
    import time
    from conftest import checkpoint_record
    cfg = scen.materialize("example_1x1", workdir)
    hip, ref = mod.Engine(cfg, 1), ref_module.Engine(cfg, 1)
    dump = str(tmp_path / "hip_waiting.json")
    for s in range(120):
        if s in (2, 3, 50):
            for e in (hip, ref):
                e.push_vehicle({"speed": 9.5, "maxSpeed": 12.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({"speed": 3.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({"speed": 20.0, "length": 4.0}, ["road_1_0_1", "road_1_1_0"])
        if s == 4:
            hip.snapshot().dump(dump)
            hip.load_from_file(dump)
        hip.next_step()
        ref.next_step()
        assert checkpoint_record(hip) == checkpoint_record(ref), s
    time.sleep(0.2)
    del ref


@pytest.mark.parametrize("layout", ["ring", "dense"])
def test_custom_speed_on_a_pushed_vehicle_hip_equals_twin(mod, scen, workdir, layout):
    """set_vehicle_speed on vehicles pushed since the last step (engine.cpp:827-834): the device keeps the speed for the
    vehicle number to come and the step that creates the vehicle takes the k_spawn_link path so that the admission sees it
    (tests/test_edge_cases.py pins the host + twin side of this against the reference)."""
    import json
    from conftest import assert_same_state
    base = scen.materialize("example_1x1", workdir)
    c = json.load(open(base))
    c["cfx"] = {"layout": layout}
    cfg = base.replace(".json", "_pushspeed_%s.json" % layout)
    with open(cfg, "w") as f:
        json.dump(c, f)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(base, 1, TWIN_LIB)
    for s in range(80):
        if s in (2, 20, 21):
            for e in (hip, tw):
                e.push_vehicle({"speed": 6.0, "maxSpeed": 14.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({"speed": 2.0}, ["road_2_1_2", "road_1_1_3"])
                e.push_vehicle({}, ["road_1_0_1", "road_1_1_0"])
                new = sorted(v for v in e.get_vehicles(True) if v.startswith("manually_pushed_"))[-3:]
                e.set_vehicle_speed(new[0], 1.25)
                e.set_vehicle_speed(new[1], 0.5)
                e.set_vehicle_speed(new[2], 2.0)
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "pushed + custom speed (%s) step %d" % (layout, s + 1))
    assert hip.get_vehicle_speed() == tw.get_vehicle_speed()


def test_repeated_set_tl_phase_before_a_step_last_call_wins(mod, scen, workdir):
    """Several set_tl_phase calls on one intersection between two steps: TrafficLight::setPhase (trafficlight.cpp:39-41) leaves
    the LAST one.  The host hands the collected calls to the device in one batch; the device must not let two threads race
    for one intersection's phase."""
    import json
    base = scen.materialize("grid_6x6", workdir)
    c = json.load(open(base))
    c["rlTrafficLight"] = True
    cfg = base.replace(".json", "_rl_repeat.json")
    with open(cfg, "w") as f:
        json.dump(c, f)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    ids = hip.intersection_ids()
    virt = hip._flat_net()["inter_virtual"]
    real = [iid for i, iid in enumerate(ids) if not virt[i]]
    for s in range(120):
        if s % 3 == 0:
            for e in (hip, tw):
                for rep in range(4):  # four calls per intersection, the last one decides
                    for j, iid in enumerate(real):
                        e.set_tl_phase(iid, (s + 3 * rep + j) % 8)
        hip.next_step()
        tw.next_step()
        assert hip._tl_state()[0].tolist() == tw._tl_state()[0].tolist(), s
        if s % 10 == 9:
            from conftest import assert_same_state
            assert_same_state(hip, tw, "repeated set_tl_phase step %d" % (s + 1))


@pytest.mark.parametrize("layout", ["dense", "ring"])
def test_lossy_archive_file_hip_equals_twin(mod, scen, workdir, layout):
    """An Archive file whose doubles do not survive the reference's JSON reader (Python's `repr` literals): the HIP engine takes
    the first step's gaps from the loaded state like the twin — which tests/test_archive_and_control.py pins to the reference
    on this very file — on both vehicle layouts; equal right after the load and after each of 40 steps."""
    import json
    cfg, path = lossy_archive_file(mod, scen, workdir)
    c = json.load(open(cfg))
    c["cfx"] = {"layout": layout}
    cfg_l = cfg.replace(".json", "_%s.json" % layout)
    json.dump(c, open(cfg_l, "w"))
    lossy_archive_body(cfg, path, lambda _c: mod.Engine(cfg_l, 1), lambda c_: mod.Engine._with_backend(c_, 1, TWIN_LIB))


def test_lossy_archive_file_lane_change_hip_equals_twin(mod, scen, workdir):
    cfg, path = lossy_archive_file(mod, scen, workdir, lane_change=True)
    lossy_archive_body(cfg, path, lambda c: mod.Engine(c, 1), lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), steps=30)
