"""The grid generator of cityflow_amd.scenarios against the reference generator's committed outputs, and a generated
non-square grid through the reference engine and the CPU twin."""
import gzip
import json
import os
import time

import numpy as np

from conftest import GOLDEN, TWIN_LIB


def _norm(net):
    for it in net["intersections"]:
        for ph in it["trafficLight"]["lightphases"]:
            ph["availableRoadLinks"] = sorted(ph["availableRoadLinks"])  # a set in the reference tool: order is free
    return net


def test_generator_reproduces_reference_fixtures(scen):
    for n in (6, 30):
        d = os.path.join(scen.SCENARIO_DIR, "grid_%dx%d" % (n, n))
        with gzip.open(os.path.join(d, "roadnet.json.gz")) as f:
            fixture = _norm(json.load(f))
        assert fixture == _norm(scen.grid_roadnet(n, n)), "roadnet %dx%d" % (n, n)  # every float bit-identical
        with gzip.open(os.path.join(d, "flow.json.gz")) as f:
            assert json.load(f) == scen.grid_flows(n, n, 1.0)


def test_generated_rectangular_grid_loads_like_a_fixture(mod, scen, workdir):
    """Same flattened network whether the 6x6 grid comes from the fixture or from the generator."""
    a = mod.Engine._with_backend(scen.materialize("grid_6x6", workdir), 1, TWIN_LIB)._flat_net()
    b = mod.Engine._with_backend(scen.generate_grid(6, 6, workdir), 1, TWIN_LIB)._flat_net()
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_generated_3x5_grid_reference_vs_twin(mod, scen, workdir, ref_module):
    cfg = scen.generate_grid(3, 5, workdir, flow_interval=2.0)
    ref = ref_module.Engine(cfg, 1)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(400):
        ref.next_step()
        tw.next_step()
        if s % 40 == 39:
            assert ref.get_lane_vehicle_count() == tw.get_lane_vehicle_count(), "step %d" % s
            rs, ts = ref.get_vehicle_speed(), tw.get_vehicle_speed()
            assert rs == ts, "step %d" % s
            assert ref.get_vehicle_distance() == tw.get_vehicle_distance(), "step %d" % s
    assert ref.get_vehicle_count() > 50
    time.sleep(0.2)  # reference destructor race
    del ref


def test_bench_roofline_keeps_plain_averages_and_reports_the_chunk_median_beside_them():
    """bench.chunk_medians: an instrumented run read in parts.  `avg_launch_us`, `achieved` and `frac` always mean the plain
    average over every instrumented launch; the figures priced with the median of the parts' averages stand beside them under
    their own names, and one launch that the box stretched to milliseconds is named, not averaged away."""
    import bench
    veh = 87890.0
    chunks = [{"k_action": (0.3125, 25), "k_cross": (0.4, 25)} for _ in range(4)]      # 12.5 us per launch
    chunks[2] = {"k_action": (76.3, 25), "k_cross": (0.4, 25)}                           # one 76 ms launch among them
    prof = {"k_action": (sum(c["k_action"][0] for c in chunks), 100), "k_cross": (1.6, 100)}
    roof = bench.roofline_from_profile(prof, veh * 100, "no-such-workload", "test", with_traffic=False,
                                       symbols={"k_action": "kr_action<256>"})
    assert roof["kernel"] == "kr_action<256>" and roof["profile_slot"] == "k_action"
    plain, plain_frac = roof["avg_launch_us"], roof["frac"]
    assert plain > 700
    bench.chunk_medians(roof, chunks)
    assert roof["avg_launch_us"] == plain and roof["frac"] == plain_frac  # unchanged meaning
    assert abs(roof["avg_launch_us_median_of_chunk_averages"] - 12.5) < 1e-9
    assert abs(roof["frac_median_priced"] - 48 * veh / 12.5e-6 / 1e9 / 8000.0) < 1e-12
    assert "outlier" in roof["outlier_note"]
    # a run without an outlier: the two agree and nothing is flagged
    calm = [{"k_action": (0.3125 + 0.001 * i, 25)} for i in range(4)]
    prof = {"k_action": (sum(c["k_action"][0] for c in calm), 100)}
    roof = bench.roofline_from_profile(prof, veh * 100, "no-such-workload", "test", with_traffic=False)
    assert roof["kernel"] == "k_action"  # (no symbol known: the slot's name)
    bench.chunk_medians(roof, calm)
    assert "outlier_note" not in roof and abs(roof["frac_median_priced"] - roof["frac"]) < 0.01 * roof["frac"]


def test_scale_record_and_compare_rules(mod, scen, workdir):
    """bench.scale_record / scale_compare (the 1 M-vehicle checkpoints of tests/golden/reference_large.json) on a small run of
    the twin: a record equals itself; per-vehicle differences from the reference are tolerated only once an exact-distance tie
    has been counted, and then only if they are the twin's; differences in counts, lanes or travel time never are."""
    import bench
    from conftest import TWIN_LIB
    eng = mod.Engine._with_backend(scen.materialize("grid_6x6", workdir), 1, TWIN_LIB)
    for _ in range(120):
        eng.next_step()
    net = eng._flat_net()
    real = {k for k, v in zip(eng.intersection_ids(), net["inter_virtual"]) if not v}
    rec = bench.scale_record(eng, real)
    assert rec["vehicle_count"] == eng.get_vehicle_count() > 100 and rec["lane_sum"] <= rec["vehicle_count"]
    want = dict(rec, twin_tie_events=0, twin_state_hash=rec["state_hash"], twin_kinematics_hash=rec["kinematics_hash"])
    assert bench.scale_compare(rec, want, 0)["equal"]
    other = dict(want, state_hash="0" * 64, kinematics_hash="0" * 64)  # the reference broke a tie the other way
    assert not bench.scale_compare(rec, other, 0)["equal"]          # ... but no tie was counted: a real difference
    assert bench.scale_compare(rec, dict(other, twin_tie_events=1), 1)["equal"]  # after a tie: the twin's record is the bar
    assert not bench.scale_compare(rec, dict(other, twin_tie_events=1, twin_state_hash="1" * 64), 1)["equal"]
    assert not bench.scale_compare(rec, dict(other, twin_tie_events=1, twin_kinematics_hash="1" * 64), 1)["equal"]
    assert not bench.scale_compare(rec, dict(other, twin_tie_events=2), 1)["equal"]  # another number of ties than the twin
    assert not bench.scale_compare(rec, dict(want, lane_array_sha256="0" * 64), 0)["equal"]
    assert not bench.scale_compare(rec, dict(other, twin_tie_events=1, average_travel_time="0x0p+0"), 1)["equal"]
