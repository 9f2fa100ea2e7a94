"""GPU (-m gpu): oracle pins for the kernel variants and workloads the published numbers rest on.

  * the benchmark workload itself (bench.build_workload: 30x30 + 3000 seeded flows, laneChange false — the
    `k_action<false>` / `k_cross<false>` instantiations bench.py times), HIP == CPU twin on every vehicle field from
    step 0 through the demand build-up and 100 steps beyond;
  * every implementation choice of the engine that must not change results (config "cfx": crossMode latency /
    throughput = k_cross / k_cross2, layout dense / ring), forced on networks where it would not be picked by size:
    the 1x1 example (up to 118 crosses per laneLink), the congested 6x6, irregular networks, the bench workload;
  * one HIP == twin checkpoint beyond 300 k vehicles (60x60) and one at BASELINE.json configs[4] size (100x100, ~1 M
    vehicles): the state the HIP engine built up is injected into the twin through an Archive, both take the same steps.

The twin is pinned to the unmodified reference (tests/test_oracle.py); reference semantics of the cross walk:
/root/reference/src/roadnet/roadnet.cpp:603-676, of the step: src/engine/engine.cpp:566-594."""
import json
import os

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state, assert_hip_backend

pytestmark = pytest.mark.gpu

# every (crossMode, layout) the HIP engine can be forced into
CHOICES = [("latency", "dense"), ("throughput", "dense"), ("latency", "ring"), ("throughput", "ring")]


def _with_cfx(path, **cfx):
    c = json.load(open(path))
    c["cfx"] = cfx
    out = path.replace(".json", "_" + "_".join("%s-%s" % kv for kv in sorted(cfx.items())) + ".json")
    with open(out, "w") as f:
        json.dump(c, f)
    return out


def _hip(mod, cfg):
    hip = mod.Engine(cfg, 1)
    assert_hip_backend(hip)
    return hip


def _twin_device(mod, cfg):
    """The CPU shadows' "device": the same host on the twin library (the "cfx" object of the config is ignored there)."""
    return mod.Engine._with_backend(cfg, 1, TWIN_LIB)


def _pair(mod, cfg, device=_hip, **cfx):
    return device(mod, _with_cfx(cfg, **cfx) if cfx else cfg), mod.Engine._with_backend(cfg, 1, TWIN_LIB)


def _bench_cfg(workdir):
    import bench
    return bench.build_workload(workdir, 0, scenario="grid_30x30")


@pytest.mark.parametrize("cross,layout", [("auto", "auto"), ("throughput", "dense"), ("latency", "dense"), ("throughput", "ring")])
def test_bench_workload_equals_twin_from_step_0(mod, workdir, cross, layout):
    import bench
    hip, tw = _pair(mod, _bench_cfg(workdir), crossMode=cross, layout=layout)
    steps = bench.BUILD_UP_STEPS + 100
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % 10 == 9:
            assert_same_state(hip, tw, "bench workload (%s, %s) step %d" % (cross, layout, s + 1))
    assert hip.get_vehicle_count() > 80000
    assert hip.get_average_travel_time() == tw.get_average_travel_time()


@pytest.mark.parametrize("cross,layout", CHOICES)
@pytest.mark.parametrize("name,steps", [("example_1x1", 500), ("grid_6x6", 300)])
def test_forced_choices_equal_twin_every_step(mod, scen, workdir, name, steps, cross, layout):
    hip, tw = _pair(mod, scen.materialize(name, workdir), crossMode=cross, layout=layout)
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "%s (%s, %s) step %d" % (name, cross, layout, s + 1))
    assert hip.get_vehicle_count() > 50


@pytest.mark.parametrize("cross,layout", CHOICES)
def test_forced_choices_congested_grid(mod, scen, workdir, cross, layout):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, flow_file=flow), crossMode=cross, layout=layout)
    for s in range(400):
        hip.next_step()
        tw.next_step()
        if s % 2 == 1:
            assert_same_state(hip, tw, "dense (%s, %s) step %d" % (cross, layout, s + 1))
    assert hip.get_vehicle_count() > 3000


@pytest.mark.parametrize("form", [10000, 20000, 30000, 60000])
def test_ring_step_forms_equal_twin(mod, scen, workdir, form):
    """The forms of the ring layout's action kernel that `auto` picks by size, forced (cfx.ringLanesPerWave / 10000:
    1 = wave-granular kw_action, 2 = block form kr_action, 3 = vehicle list kr_index + kl_action, 6 = the same with the list's tiles handed out by ticket) on the 1x1 example (up to 118 crosses per laneLink, an
    intersection with more than 64 laneLinks) and the congested 6x6."""
    hip, tw = _pair(mod, scen.materialize("example_1x1", workdir), layout="ring", ringLanesPerWave=form)
    for s in range(400):
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "1x1 form %d step %d" % (form, s + 1))
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, flow_file=flow), layout="ring", ringLanesPerWave=form + 6)
    for s in range(300):
        hip.next_step()
        tw.next_step()
        if s % 2 == 1:
            assert_same_state(hip, tw, "dense 6x6 form %d step %d" % (form, s + 1))
    assert hip.get_vehicle_count() > 2500


def test_ring_list_form_free_running_equals_dense_layout(mod, workdir):
    """The list form of the ring layout's action phase sizes its vehicle list and its launch from a host-side bound of the
    running vehicles while the host runs steps ahead of the device: 3000 steps of the bench workload from an empty network
    without a single synchronisation (the list grows from nothing), then every field equal to the dense layout's — an
    engine that shares neither the layout nor the action kernel with it — and 40 more steps compared one by one."""
    cfg = _bench_cfg(workdir)
    lst = _hip(mod, _with_cfx(cfg, layout="ring", ringLanesPerWave=30000))
    dense = _hip(mod, _with_cfx(cfg, layout="dense"))
    for _ in range(3000):
        lst.next_step()
    for _ in range(3000):
        dense.next_step()
    assert_same_state(lst, dense, "after 3000 free-running steps")
    assert lst.get_vehicle_count() > 50000
    for s in range(40):
        lst.next_step()
        dense.next_step()
        assert_same_state(lst, dense, "step %d after the free run" % (s + 1))


@pytest.mark.parametrize("cross,layout", CHOICES)
@pytest.mark.parametrize("seed", [11, 14])
def test_forced_choices_irregular_networks(mod, scen, workdir, seed, cross, layout):
    from test_irregular import irregular
    hip, tw = _pair(mod, irregular(scen, workdir, seed), crossMode=cross, layout=layout)
    for s in range(500):
        hip.next_step()
        tw.next_step()
        if s % 5 == 4:
            assert_same_state(hip, tw, "irregular %d (%s, %s) step %d" % (seed, cross, layout, s + 1))
    assert hip.get_vehicle_count() > 150


def large_checkpoint_body(mod, scen, workdir, device, n, flows_per_100, min_running, twin_steps, layout, rl, build_steps=300):
    """Body shared by the GPU test below (device = the HIP engine) and its CPU shadow in tests/test_pin_shadows.py (device =
    the same host on the twin library), so that a host-side change that makes it stale shows without a GPU."""
    base = scen.generate_grid(n, n, workdir)
    d = os.path.dirname(base)
    n_extra = n * n * flows_per_100 // 100
    flow = os.path.join(d, "flow_pin_%d.json" % n_extra)
    if not os.path.exists(flow):
        scen.dense_flows(os.path.join(d, "roadnet.json"), flow, n_extra, seed=4242, interval=6.0,
                         base_flow=os.path.join(d, "flow.json"), end_time=240)
    cfg = os.path.join(d, "config_pin.json")
    with open(cfg, "w") as f:
        json.dump(dict(json.load(open(base)), flowFile=os.path.basename(flow), rlTrafficLight=rl), f)
    hip = device(_with_cfx(cfg, layout=layout))
    rng = np.random.default_rng(99)
    n_inter = len(hip.intersection_ids())
    for s in range(build_steps):
        if rl and s % 15 == 0:
            hip.set_tl_phases(rng.integers(0, 8, size=n_inter).astype(np.int32))
        hip.next_step()
    assert hip.get_vehicle_count() >= min_running
    # One Archive goes into all three engines: a load restarts every route cursor at the route's first road (the reference's
    # Router copy constructor, router.cpp:11-14; archive.cpp here), so an engine that loaded is compared with engines that
    # loaded, field for field.  `live` keeps running without the load: it must take the very same steps (nextOf searches
    # forward from the cursor, cfx_device.h / router.cpp:49-58), everything but the cursor itself equal.
    snap = hip.snapshot()
    live = hip
    hip = device(_with_cfx(cfg, layout=layout))
    hip.load(snap)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    tw.load(snap)
    assert_same_state(hip, tw, "%dx%d after the transfer" % (n, n))
    assert_same_state(hip, live, "%dx%d loaded copy vs the live engine" % (n, n), skip=("route_pos",))
    for s in range(twin_steps):
        if rl and s % 2 == 0:  # the agent's action
            ph = rng.integers(0, 8, size=n_inter).astype(np.int32)
            for e in (hip, tw, live):
                e.set_tl_phases(ph)
        for e in (hip, tw, live):
            e.next_step()
        if rl:  # the agent's observation
            assert np.array_equal(hip.get_lane_vehicle_count_array(), tw.get_lane_vehicle_count_array()), s
            assert np.array_equal(hip.get_lane_vehicle_count_array(), live.get_lane_vehicle_count_array()), s
        if s % 4 == 3 or s == twin_steps - 1:
            assert_same_state(hip, tw, "%dx%d %s step %d after the transfer" % (n, n, layout, s + 1))
            assert_same_state(hip, live, "%dx%d %s step %d: loaded copy vs live" % (n, n, layout, s + 1), skip=("route_pos",))


@pytest.mark.parametrize("n,flows_per_100,min_running,twin_steps,layout,rl", [
    (60, 333, 300000, 20, "dense", False), (60, 333, 300000, 20, "ring", False), (100, 333, 900000, 60, "auto", True)])
def test_large_checkpoint_equals_twin(mod, scen, workdir, n, flows_per_100, min_running, twin_steps, layout, rl):
    """Sizes where the engine switches to its throughput kernels by itself (k_cross2 above 240 k slots).  The 100x100 case is
    BASELINE.json configs[4] as an RL agent drives it: rlTrafficLight, a new phase for every signal through set_tl_phases
    (during the build-up and between the compared steps) and the lane-count observation read every step — HIP == twin."""
    large_checkpoint_body(mod, scen, workdir, lambda c: _hip(mod, c), n, flows_per_100, min_running, twin_steps, layout, rl)


@pytest.mark.parametrize("layout", ["dense", "ring"])
@pytest.mark.parametrize("interval,steps", [(0.3, 2000), (0.7, 900)])
def test_travel_time_sum_keeps_the_reference_order(mod, scen, workdir, interval, steps, layout):
    """cumulativeTravelTime is a running FP64 sum in removal order (engine.cpp:296-310).  With interval 1.0 / 0.5 every
    partial sum is exact and the device adds in parallel; an interval that is not a multiple of 2^-10 sends it down the
    strictly ordered path — both must give the twin's bits."""
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, interval=interval), layout=layout)
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % 50 == 49:
            assert_same_state(hip, tw, "interval %s (%s) step %d" % (interval, layout, s + 1))
    assert hip._scalars()["finished_vehicle_count"] > 100
    assert hip.get_average_travel_time() == tw.get_average_travel_time()


def test_ring_growth_path(mod, scen, workdir):
    ring_growth_body(mod, scen, workdir, _hip)


def ring_growth_body(mod, scen, workdir, device, steps=400):
    """Rings start at a third of their bumper-to-bumper capacity (config "cfx": ringCapacityPercent): lanes fill up, the
    commit raises its near-full flag, the next step doubles every capacity and carries the vehicles over (gather ->
    re-allocate -> scatter) — several times during the run, with the state equal to the twin's throughout."""
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow)
    hip, tw = _pair(mod, cfg, device, layout="ring", ringCapacityPercent=30)
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % 4 == 3:
            assert_same_state(hip, tw, "growing rings step %d" % (s + 1))
    if device is not _hip:
        return
    assert hip.get_vehicle_count() > 3000
    slots, scale = hip._ring_info()
    assert hip._layout() == "ring" and scale >= 2 and slots > 0, (slots, scale)


@pytest.mark.parametrize("layout", ["dense", "ring"])
def test_leavers_that_are_not_a_prefix(mod, scen, workdir, layout):
    """Both layouts assume "leavers are a prefix of the list" on their fast path and take an exact general path otherwise
    (in-ring compaction / counted ranks).  Traffic never produces the general case without lane change; a hand-edited
    Archive does."""
    from test_edge_cases import out_of_order_archive
    cfg, path = out_of_order_archive(mod, scen, workdir)
    hip, tw = _pair(mod, cfg, layout=layout)
    hip.load_from_file(path)
    tw.load_from_file(path)
    for s in range(60):
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "out-of-order archive (%s) step %d" % (layout, s + 1))


def test_many_spawns_per_lane_in_one_step(mod, scen, workdir):
    many_spawns_body(mod, scen, workdir, _hip)


def many_spawns_body(mod, scen, workdir, device, layouts=("ring", "dense"), **cfx):
    """The ring step links a step's spawn records inside kr_admit (they travel in its kernel arguments, sorted by lane): 60
    flows that all start on the same three lanes put ~20 records on a lane in one step — chains inside the batch, heads where
    the queue had drained, appends behind vehicles still waiting — and a second phase with more records than the arguments
    hold takes the k_spawn_link path on the same queues.  Lane::pushWaitingVehicle roadnet.h:365-367, engine.cpp:502-516."""
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flows = json.load(open(os.path.join(d, "flow.json")))
    proto = [f for f in flows if len(f["route"]) >= 3][:3]
    out = []
    for i in range(60):      # 60 records per step on 3 roads (9 lanes), every step
        f = json.loads(json.dumps(proto[i % 3]))
        f.update(interval=1.0, startTime=0, endTime=60)
        out.append(f)
    for i in range(200):     # later: 260 records per step (> kAdmitRecs) for a while, then back to few
        f = json.loads(json.dumps(flows[i % len(flows)]))
        f.update(interval=1.0, startTime=30, endTime=45)
        out.append(f)
    for i in range(20):
        f = json.loads(json.dumps(flows[(7 * i) % len(flows)]))
        f.update(interval=3.0, startTime=0, endTime=-1)
        out.append(f)
    flow_file = os.path.join(d, "flow_same_lanes.json")
    with open(flow_file, "w") as fh:
        json.dump(out, fh)
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow_file)
    for layout in layouts:
        hip, tw = _pair(mod, cfg, device, layout=layout, **cfx)
        for s in range(220):
            hip.next_step()
            tw.next_step()
            if s < 70 or s % 10 == 9:
                assert_same_state(hip, tw, "same-lane spawns (%s) step %d" % (layout, s + 1))
                assert np.array_equal(hip.get_lane_waiting_vehicle_count_array(), tw.get_lane_waiting_vehicle_count_array()), (layout, s)
        assert hip._scalars()["spawned_vehicle_count"] > 6000
        assert hip.get_average_travel_time() == tw.get_average_travel_time()
