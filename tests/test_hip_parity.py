"""GPU (-m gpu): the HIP/gfx950 engine, called through the C ABI by the host, against
  (a) the CPU twin, every vehicle field, bit-exact (integer fields and FP64 positions / speeds / gaps alike;
      BASELINE.json asks for 1e-6 relative on positions and speeds — the bar here is 0 ulp);
  (b) golden vectors produced by the unmodified reference engine;
  (c) the unmodified reference engine itself (oracle/_ref, shipped prebuilt), live, through the Python API;
  (d) size-independent conservation properties at benchmark scale."""
import os
import time

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state, checkpoint_record, assert_hip_backend

pytestmark = pytest.mark.gpu


def _pair(mod, cfg):
    hip = mod.Engine(cfg, 1)
    assert_hip_backend(hip)
    return hip, mod.Engine._with_backend(cfg, 1, TWIN_LIB)


@pytest.mark.parametrize("name,steps,every", [("example_1x1", 400, 1), ("grid_6x6", 300, 1), ("grid_30x30", 300, 10)])
def test_hip_equals_twin_every_field(mod, scen, workdir, name, steps, every):
    hip, tw = _pair(mod, scen.materialize(name, workdir))
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % every == every - 1:
            assert_same_state(hip, tw, "%s step %d" % (name, s + 1))


@pytest.mark.parametrize("name,steps", [("example_1x1", [1, 10, 100, 200, 500, 1000]),
                                        ("grid_6x6", [250, 500, 750, 1000, 1250, 1500]),
                                        ("grid_30x30", [100, 250, 500])])
def test_hip_matches_reference_goldens(mod, scen, workdir, golden, name, steps):
    eng = mod.Engine(scen.materialize(name, workdir), 1)
    want = golden["reference_checkpoints"][name]
    for s in range(1, max(steps) + 1):
        eng.next_step()
        if s in steps:
            assert checkpoint_record(eng) == want[str(s)], "%s step %d" % (name, s)


def test_hip_matches_reference_engine_live(mod, ref_module, scen, workdir):
    """Mirrors the reference's tests/python/test_api.py loop with the reference engine beside it."""
    cfg = scen.materialize("example_1x1", workdir)
    ref = ref_module.Engine(cfg, 1)
    eng = mod.Engine(cfg, 1)
    for s in range(300):
        ref.next_step()
        eng.next_step()
        assert ref.get_lane_vehicle_count() == eng.get_lane_vehicle_count(), s
        assert ref.get_vehicle_count() == eng.get_vehicle_count(), s
        if s % 10 == 9:
            assert ref.get_vehicle_speed() == eng.get_vehicle_speed(), s
            assert ref.get_vehicle_distance() == eng.get_vehicle_distance(), s
            assert ref.get_lane_waiting_vehicle_count() == eng.get_lane_waiting_vehicle_count(), s
            assert ref.get_lane_vehicles() == eng.get_lane_vehicles(), s
            assert ref.get_vehicles() == eng.get_vehicles(), s
            assert ref.get_vehicles(True) == eng.get_vehicles(include_waiting=True), s
            assert ref.get_current_time() == eng.get_current_time()
            assert ref.get_average_travel_time() == eng.get_average_travel_time(), s
    for vid in ref.get_vehicles():
        assert ref.get_leader(vid) == eng.get_leader(vid)
        a, b = ref.get_vehicle_info(vid), eng.get_vehicle_info(vid)
        assert a == b, (vid, a, b)
    time.sleep(0.1)


def test_reset_reproduces_run(mod, scen, workdir):
    """reference tests/cpp/basic_test.cpp:37-53, strengthened to the full checkpoint record"""
    eng = mod.Engine(scen.materialize("example_1x1", workdir), 1)
    for _ in range(200):
        eng.next_step()
    a = checkpoint_record(eng)
    eng.reset(True)
    assert eng.get_vehicle_count() == 0 and eng.get_current_time() == 0.0
    for _ in range(200):
        eng.next_step()
    assert checkpoint_record(eng) == a


def test_rl_traffic_light_control(mod, scen, workdir):
    """rlTrafficLight: lights move only through set_tl_phase (engine.cpp:583-587,719-725)."""
    cfg = scen.materialize("grid_6x6", workdir, rlTrafficLight=True)
    hip, tw = _pair(mod, cfg)
    inters = [i for i in hip.intersection_ids()]
    net = hip._flat_net()
    real = [i for i, v in enumerate(net["inter_virtual"]) if not v]
    for s in range(240):
        if s % 10 == 0:
            ph = (s // 10) % 8
            for i in real:
                hip.set_tl_phase(inters[i], ph)
                tw.set_tl_phase(inters[i], ph)
        hip.next_step()
        tw.next_step()
        if s % 5 == 4:
            assert_same_state(hip, tw, "rl step %d" % (s + 1))
    with pytest.raises((IndexError, ValueError, RuntimeError)):
        hip.set_tl_phase(inters[real[0]], 99)


def test_set_tl_phase_ignored_without_rl(mod, scen, workdir, capfd):
    eng = mod.Engine(scen.materialize("example_1x1", workdir), 1)
    before = eng._tl_state()[0].copy()
    eng.set_tl_phase("intersection_1_1", 3)  # message on stderr, no effect (engine.cpp:720-723)
    assert np.array_equal(before, eng._tl_state()[0])
    assert "rlTrafficLight" in capfd.readouterr().err


def test_push_vehicle(mod, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir)
    hip, tw = _pair(mod, cfg)
    info = {"length": 4.0, "width": 2.0, "maxPosAcc": 2.5, "maxNegAcc": 5.0, "usualPosAcc": 2.0, "usualNegAcc": 4.0,
            "minGap": 2.0, "maxSpeed": 12.0, "headwayTime": 1.2}
    for s in range(120):
        if s in (5, 6, 40):
            for e in (hip, tw):
                e.push_vehicle(info, ["road_0_1_0", "road_1_1_1"])
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "push step %d" % (s + 1))
    assert any(v.startswith("manually_pushed_") for v in hip.get_vehicles(True) + list(hip.get_vehicle_speed()))  \
        or hip._scalars()["finished_vehicle_count"] > 0


def test_congested_dense_grid_equals_twin(mod, scen, workdir):
    """Interior-origin flows until lanes jam: exercises blocked admissions, canEnter, yielding and deadlock walks."""
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, flow_file=flow))
    for s in range(400):
        hip.next_step()
        tw.next_step()
        if s % 4 == 3:
            assert_same_state(hip, tw, "dense step %d" % (s + 1))
    assert hip.get_vehicle_count() > 3000


def test_more_spawn_records_than_the_arguments_hold_equals_twin(mod, scen, workdir):
    """1 500 flows at interval 1 on the 6x6 grid: > 1 024 spawn records in EVERY step, more than the admission kernel's
    arguments hold (kAdmitRecsBig).  On the ring layout they travel in pinned memory, sorted by lane, each block of lanes
    reading its own (SpawnBatchMem, kr_admit; Flow::nextStep flow.cpp:6-22 + Lane::pushWaitingVehicle roadnet.h:365-367 +
    Engine::handleWaiting engine.cpp:497-520 are what it restates) — the commit keeps riding with the admission.  Every
    field against the twin, waiting buffers included; the kernel that ran is asked for by name."""
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_many.json"), 1500, seed=31,
                            interval=1.0, base_flow=os.path.join(d, "flow.json"), end_time=40)
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, flow_file=flow))
    for s in range(120):
        hip.next_step()
        tw.next_step()
        if s < 6 or s % 5 == 4:
            assert_same_state(hip, tw, "many records, step %d" % (s + 1))
            (wa, la), (wb, lb) = hip._waiting(), tw._waiting()
            assert np.array_equal(wa, wb) and np.array_equal(la, lb), s
        if s == 20:
            assert hip._scalars()["spawned_vehicle_count"] > 21 * 1024
            syms = hip._profile_symbols()  # (empty where the test's body runs on the CPU twin: tests/test_gpu_shadow.py)
            assert not syms or "SpawnBatchMem" in syms["k_admit"], syms
    assert hip._scalars()["finished_vehicle_count"] > 0


def test_conservation_at_benchmark_scale(mod, scen, workdir):
    """Size-independent properties on the 30x30 benchmark workload (no oracle needed at this size)."""
    base = scen.materialize("grid_30x30", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_bench.json"), 6000, seed=12345,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    eng = mod.Engine(scen.materialize("grid_30x30", workdir, flow_file=flow), 1)
    L = len(eng.lane_ids())
    for s in range(300):
        eng.next_step()
        if s % 50 == 49:
            sc = eng._scalars()
            st = eng._vehicle_state()
            wv, wl = eng._waiting()
            n = len(st["vid"])
            assert n == sc["active_vehicle_count"]
            assert sc["spawned_vehicle_count"] == n + len(wv) + sc["finished_vehicle_count"]
            assert len(np.unique(st["vid"])) == n and len(np.intersect1d(st["vid"], wv)) == 0
            counts = eng.get_lane_vehicle_count_array()
            assert counts.sum() == int((st["drivable"] < L).sum())
            assert np.array_equal(np.bincount(st["drivable"][st["drivable"] < L], minlength=L), counts)
            # the device keeps Drivable::vehicles order: grouped by drivable, ascending
            assert np.all(np.diff(st["drivable"]) >= 0)
            assert np.all(st["speed"] >= 0) and np.all(st["speed"] <= 16.67 + 1e-9)
            assert np.all(st["dis"] >= 0)
    assert eng.get_vehicle_count() > 20000


def test_edge_case_variants_hip_equals_twin(mod, scen, workdir):
    """The scenarios of tests/test_edge_cases.py (checked against the reference on CPU) on the HIP engine."""
    from test_edge_cases import _example_flows, _variant

    flows = _example_flows(scen, workdir)
    flows[0].update(startTime=5, endTime=40, interval=3.0)
    flows[1].update(startTime=0, endTime=0, interval=1.0)
    flows[2].update(startTime=20, endTime=-1, interval=7.5)
    flows[3]["route"] = ["road_0_1_0", "road_1_0_1"]  # invalid: dropped

    def strip(net):
        for inter in net["intersections"]:
            for rl in inter.get("roadLinks", []):
                for ll in rl["laneLinks"]:
                    ll.pop("points", None)

    for cfg in (_variant(scen, workdir, "example_1x1", "gpu_windows", flows=flows),
                _variant(scen, workdir, "example_1x1", "gpu_half", interval=0.5),
                _variant(scen, workdir, "example_1x1", "gpu_nopoints", roadnet_edit=strip),
                _variant(scen, workdir, "example_1x1", "gpu_empty", flows=[])):
        hip, tw = _pair(mod, cfg)
        for s in range(150):
            hip.next_step()
            tw.next_step()
            if s % 3 == 2:
                assert_same_state(hip, tw, "%s step %d" % (os.path.basename(cfg), s + 1))
        assert hip.get_average_travel_time() == tw.get_average_travel_time()
