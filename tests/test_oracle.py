"""CPU: pins the oracle.
 * the unmodified reference build (oracle/_ref/cityflow_ref) reproduces the committed golden vectors and the
   self-consistency properties the reference's own tests assert (tests/cpp/basic_test.cpp:37-53 reset,
   tests/python/test_archive.py:25-121 snapshot/restore);
 * the CPU twin (oracle/twin, a restatement of engine.cpp/vehicle.cpp/roadnet.cpp/router.cpp on the flat
   network) matches the golden vectors and, live, the reference engine vehicle by vehicle."""
import time

import pytest

from conftest import TWIN_LIB, checkpoint_record


def _run_to(eng, steps, want):
    out = {}
    for s in range(1, max(steps) + 1):
        eng.next_step()
        if s in steps:
            out[str(s)] = checkpoint_record(eng)
    assert out == {k: want[k] for k in out}


def test_reference_build_reproduces_goldens(ref_module, scen, workdir, golden):
    g = golden["reference_checkpoints"]["example_1x1"]
    eng = ref_module.Engine(scen.materialize("example_1x1", workdir), 1)
    _run_to(eng, [1, 10, 100, 200, 500], g)
    time.sleep(0.1)  # reference destructor race, SURVEY.md §5.2


def test_reference_reset_property(ref_module, scen, workdir):
    """reference tests/cpp/basic_test.cpp:37-53"""
    eng = ref_module.Engine(scen.materialize("example_1x1", workdir), 1)
    for _ in range(200):
        eng.next_step()
    a = (eng.get_current_time(), eng.get_vehicle_count(), eng.get_lane_vehicle_count())
    eng.reset(True)
    for _ in range(200):
        eng.next_step()
    assert a == (eng.get_current_time(), eng.get_vehicle_count(), eng.get_lane_vehicle_count())
    time.sleep(0.1)


@pytest.mark.parametrize("name,steps", [("example_1x1", [1, 10, 100, 200, 500, 1000]),
                                        ("grid_6x6", [250, 500, 750]),
                                        ("grid_30x30", [100, 250])])
def test_twin_matches_reference_goldens(mod, scen, workdir, golden, name, steps):
    eng = mod.Engine._with_backend(scen.materialize(name, workdir), 1, TWIN_LIB)
    assert eng.backend_name() == "cpu-twin"
    _run_to(eng, steps, golden["reference_checkpoints"][name])


@pytest.mark.parametrize("name,steps", [("example_1x1", 300), ("grid_6x6", 300)])
def test_twin_matches_reference_live(mod, ref_module, scen, workdir, name, steps):
    cfg = scen.materialize(name, workdir)
    ref = ref_module.Engine(cfg, 1)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(steps):
        ref.next_step()
        tw.next_step()
        assert ref.get_lane_vehicle_count() == tw.get_lane_vehicle_count(), s
        if s % 10 == 9:
            assert ref.get_vehicle_speed() == tw.get_vehicle_speed(), s
            assert ref.get_vehicle_distance() == tw.get_vehicle_distance(), s
            assert ref.get_lane_waiting_vehicle_count() == tw.get_lane_waiting_vehicle_count(), s
            assert ref.get_lane_vehicles() == tw.get_lane_vehicles(), s
            assert ref.get_vehicles(True) == tw.get_vehicles(True), s
            assert ref.get_average_travel_time() == tw.get_average_travel_time(), s
    # get_leader for every running vehicle
    for vid in ref.get_vehicles():
        assert ref.get_leader(vid) == tw.get_leader(vid)
    time.sleep(0.1)


def test_twin_reset_and_seed(mod, scen, workdir):
    """reset(True) replays identically; reset() keeps the RNG running (engine.cpp:744-760)."""
    eng = mod.Engine._with_backend(scen.materialize("example_1x1", workdir), 1, TWIN_LIB)
    for _ in range(200):
        eng.next_step()
    a = checkpoint_record(eng)
    eng.reset(True)
    assert eng.get_vehicle_count() == 0 and eng.get_current_time() == 0
    for _ in range(200):
        eng.next_step()
    assert checkpoint_record(eng) == a
    eng.reset(False)
    for _ in range(200):
        eng.next_step()
    b = checkpoint_record(eng)
    assert b["vehicle_count"] == a["vehicle_count"] and b["lane_hash"] != a["lane_hash"]
