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


def test_bench_roofline_is_priced_with_the_chunk_median_when_one_launch_is_an_outlier():
    """bench.chunk_medians: an instrumented run read in parts; one launch that the box stretched to milliseconds must not
    price the roofline (both figures stay in the line, and the choice is named)."""
    import bench
    veh = 87890.0
    chunks = [{"k_action": (0.3125, 25), "k_cross": (0.4, 25)} for _ in range(4)]      # 12.5 us per launch
    chunks[2] = {"k_action": (76.3, 25), "k_cross": (0.4, 25)}                           # one 76 ms launch among them
    prof = {"k_action": (sum(c["k_action"][0] for c in chunks), 100), "k_cross": (1.6, 100)}
    roof = bench.roofline_from_profile(prof, veh * 100, "no-such-workload", "test", with_traffic=False)
    assert roof["avg_launch_us"] > 700
    bench.chunk_medians(roof, chunks)
    assert abs(roof["avg_launch_us"] - 12.5) < 1e-9 and roof["avg_launch_us_all_launches"] > 700
    assert roof["duration_estimator"].startswith("median of 4 chunk averages")
    assert abs(roof["frac"] - 48 * veh / 12.5e-6 / 1e9 / 8000.0) < 1e-12
    # a run without an outlier keeps the plain average
    calm = [{"k_action": (0.3125 + 0.001 * i, 25)} for i in range(4)]
    prof = {"k_action": (sum(c["k_action"][0] for c in calm), 100)}
    roof = bench.roofline_from_profile(prof, veh * 100, "no-such-workload", "test", with_traffic=False)
    before = roof["avg_launch_us"]
    bench.chunk_medians(roof, calm)
    assert roof["avg_launch_us"] == before and roof["duration_estimator"] == "average over all instrumented launches"
