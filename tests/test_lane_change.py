"""Lane change (reference src/vehicle/lanechange.cpp): the CPU twin against the reference.

The reference walks lane-change candidates in heap-address order, so a reference run is only reproducible under
oracle/_ref/libmonotonic_new.so (see oracle/monotonic_new.cpp and tests/tools/lane_change_parity.py); the committed
vectors in tests/golden/reference_lane_change.json were produced that way.  Every run below happens in its own process
."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from conftest import REF_DIR, TWIN_LIB, assert_hip_backend

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import lane_change_parity as lcp  # noqa: E402
from make_lane_change_goldens import record  # noqa: E402


@pytest.fixture(scope="module")
def lc_golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_lane_change.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name,steps", [("example_1x1", [12, 60, 200, 500]), ("grid_6x6", [379, 400, 600])])
def test_twin_lane_change_matches_reference_goldens(scen, workdir, lc_golden, name, steps):
    """Vehicle count (with shadows), per-lane counts, every lane's vehicle list (shadows by id), the priority order of
    the real vehicles, average travel time and every real vehicle's speed / distance, all exact."""
    cfg = scen.materialize(name, workdir, laneChange=True)
    for h in steps:
        got = record(lcp.run("twin", cfg, h))
        assert got == lc_golden[name][str(h)], (name, h)
    assert lc_golden[name][str(steps[0])]["vehicle_count"] > lc_golden[name][str(steps[0])]["real_vehicles"]  # shadows alive


@pytest.mark.parametrize("name,steps,every", [("example_1x1", [60, 200, 500], 7), ("grid_6x6", [400, 600], 150)])
def test_twin_lane_change_matches_reference_goldens_while_it_forgets_finished_vehicles(scen, workdir, lc_golden, name, steps, every):
    """... with `"cfx": {"compactVehicles": N}`: the finished vehicles forgotten and the others renumbered every N vehicles
    (EngineHost::compactVehicles; an id's chain of copies stays whole) — the same goldens: ids with their "_shadow", the
    priority order, every speed and distance."""
    cfg = scen.materialize(name, workdir, laneChange=True)
    with open(cfg) as f:
        c = json.load(f)
    c["cfx"] = dict(c.get("cfx", {}), compactVehicles=every)
    cfg = cfg.replace(".json", "_compact.json")
    with open(cfg, "w") as f:
        json.dump(c, f)
    for h in steps:
        got = record(lcp.run("twin", cfg, h))
        assert got == lc_golden[name][str(h)], (name, h)


def test_twin_lane_change_matches_reference_live(scen, workdir):
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    for h in (35, 150):
        r, t = lcp.run("ref", cfg, h), lcp.run("twin", cfg, h)
        assert lcp.compare(r, t) == [], h


def test_walk_order_permutation_is_std_sort(workdir):
    """engine.cpp:793-794 sorts the candidates by urgency with std::sort; all urgencies are equal, and libstdc++'s introsort
    still permutes more than 16 of them.  The device computes that permutation in closed form (lcSortedPosition,
    cfx_lc_kernels.h); here the same procedure is checked against std::sort itself for every count up to 3000."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "cityflow_amd", "csrc", "hip", "cfx_lc_kernels.h")).read()
    body = hdr[hdr.index("__device__ inline int lcSortedPosition"):]
    body = body[:body.index("\n}\n") + 3].replace("__device__ inline", "static")
    src = os.path.join(workdir, "sorted_position.cpp")
    with open(src, "w") as f:
        f.write("#include <algorithm>\n#include <cstdio>\n#include <vector>\n" + body + """
int main() {
    for (int n = 0; n <= 3000; ++n) {
        std::vector<int> v(n);
        for (int i = 0; i < n; ++i) v[i] = i;
        std::sort(v.begin(), v.end(), [](int, int) { return false; });
        for (int p = 0; p < n; ++p)
            if (lcSortedPosition(v[p], n) != p) { printf("mismatch n=%d\\n", n); return 1; }
    }
    printf("ok\\n");
    return 0;
}
""")
    exe = os.path.join(workdir, "sorted_position")
    subprocess.check_call(["g++", "-O2", "-std=c++11", src, "-o", exe])
    assert subprocess.check_output([exe]).decode().strip() == "ok"


def test_lane_change_api_surface(mod, scen, workdir):
    """Shadows: hidden from get_vehicles / get_vehicle_speed / get_vehicle_distance (Engine::getRunningVehicles,
    engine.cpp:780-790), counted by get_vehicle_count, listed by id in get_lane_vehicles; '<id>_shadow' resolves."""
    eng = mod.Engine._with_backend(scen.materialize("example_1x1", workdir, laneChange=True), 1, TWIN_LIB)
    for _ in range(12):
        eng.next_step()
    speeds = eng.get_vehicle_speed()
    shadows = [v for lane in eng.get_lane_vehicles().values() for v in lane if v.endswith("_shadow")]
    assert shadows and eng.get_vehicle_count() == len(speeds) + len(shadows)
    assert not any(k.endswith("_shadow") for k in speeds) and set(eng.get_vehicles()) == set(speeds)
    assert set(eng.get_vehicle_distance()) == set(speeds)
    sh = shadows[0]
    parent = sh[:-len("_shadow")]
    info_s, info_p = eng.get_vehicle_info(sh), eng.get_vehicle_info(parent)
    assert info_s["running"] == "1" and info_s["distance"] == info_p["distance"] and info_s["drivable"] != info_p["drivable"]
    assert eng.get_leader(sh) == eng.get_leader(parent)  # engine.cpp:842-845: a shadow answers for its partner
    # the change completes (LaneChange::finishChanging): the shadow takes over the id, "<id>_shadow" is gone
    gone = False
    for _ in range(8):
        eng.next_step()
        lanes = eng.get_lane_vehicles()
        if not any(sh in v for v in lanes.values()):
            gone = True
            with pytest.raises(RuntimeError):
                eng.get_vehicle_info(sh)
            assert eng.get_vehicle_info(parent)["drivable"] == info_s["drivable"]  # the id now lives in the target lane
            break
    assert gone
    # reset clears lane-change state too
    eng.reset(True)
    for _ in range(12):
        eng.next_step()
    assert sorted(v for lane in eng.get_lane_vehicles().values() for v in lane if v.endswith("_shadow")) == sorted(shadows)


def test_tiled_engine_refuses_lane_change(mod, scen, workdir):
    """Lane change is built for the single engine and for batched environments (tests/test_vector_engine.py); the tiled host
    must say that it is not, and not ignore the flag (the schedule walk is ONE order over the candidates of all tiles)."""
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    with pytest.raises(RuntimeError, match="laneChange"):
        mod.TiledEngine(cfg, 1, 2, [], TWIN_LIB)


def _state(e):
    s = e._vehicle_state()
    order = np.argsort(s["vid"], kind="stable")
    return {k: v[order] for k, v in s.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [("example_1x1", 400), ("grid_6x6", 460)])
def test_hip_lane_change_equals_twin_every_step(mod, scen, workdir, name, steps):
    """The HIP engine against the twin (which is pinned against the reference, above), laneChange=true: after EVERY step
    every running vehicle's drivable, distance, speed, leader, stored gap, blocker, route cursor and lane-change state
    (partner, shadow / parent / changing flags, lateral offset, direction, target lane, cooling timer), the per-lane counts
    and the scalars are equal, bit for bit.  1x1: dozens of interacting changes from step 6 on (signals, yielding, aborts);
    6x6: every vehicle changes lane on its last road from step 378 on."""
    cfg = scen.materialize(name, workdir, laneChange=True)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    assert_hip_backend(hip)
    shadows = 0
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        a, b = _state(hip), _state(tw)
        assert a.keys() == b.keys() and "lc_flags" in a
        for k in a:
            assert np.array_equal(a[k], b[k]), (s, k)
        assert np.array_equal(hip.get_lane_vehicle_count_array(), tw.get_lane_vehicle_count_array()), s
        sa, sb = hip._scalars(), tw._scalars()
        for k in ("active_vehicle_count", "finished_vehicle_count", "spawned_vehicle_count", "cumulative_travel_time", "vehicle_steps"):
            assert sa[k] == sb[k], (s, k, sa, sb)
        shadows = max(shadows, int((a["lc_flags"] & 1).sum()))
    assert shadows >= 5
    assert hip.get_lane_vehicles() == tw.get_lane_vehicles() and hip.get_vehicles() == tw.get_vehicles()
    assert hip.get_average_travel_time() == tw.get_average_travel_time()


@pytest.mark.gpu
def test_hip_lane_change_on_the_bench_workload(mod, scen, workdir):
    """30x30 with the bench.py demand (~90 k running vehicles, ~70 new shadows per step, thousands of pairs in flight):
    HIP == twin on every vehicle field, every 10th step."""
    import bench
    base = bench.build_workload(workdir, 0, scenario="grid_30x30")
    c = json.load(open(base))
    c["laneChange"] = True
    cfg = base.replace(".json", "_lanechange.json")
    with open(cfg, "w") as f:
        json.dump(c, f)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(250):
        hip.next_step()
        tw.next_step()
        if s % 10 == 9:
            a, b = _state(hip), _state(tw)
            for k in a:
                assert np.array_equal(a[k], b[k]), (s, k)
    assert int((a["lc_flags"] & 1).sum()) > 20 and len(a["vid"]) > 50000


@pytest.mark.gpu
def test_hip_lane_change_matches_reference_goldens(mod, scen, workdir, lc_golden):
    """... and against the vectors the reference itself produced (1x1: more than 16 candidates in many steps, i.e. the walk
    order is the std::sort permutation)."""
    for name, horizons in (("example_1x1", (12, 60, 200, 500)), ("grid_6x6", (379, 400, 600))):
        eng = mod.Engine(scen.materialize(name, workdir, laneChange=True), 1)
        done = 0
        for h in horizons:
            for _ in range(h - done):
                eng.next_step()
            done = h
            assert record(lcp.state(eng)) == lc_golden[name][str(h)], (name, h)


def test_replay_log_with_lane_change_matches_reference(scen, workdir):
    """Engine::updateLog with lane change: positions blended towards the target lane by the lateral offset
    (Vehicle::getPoint vehicle.cpp:81-105), the laneChangeDir column, shadows left out."""
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    from test_replay import _cfg, _parse_line
    steps = 60
    cfg_r, _, log_r = _cfg(scen, workdir, "example_1x1", "lc_ref", laneChange=True)
    cfg_m, _, log_m = _cfg(scen, workdir, "example_1x1", "lc_mine", laneChange=True)
    lcp.run("ref", cfg_r, steps)
    lcp.run("twin", cfg_m, steps)
    la, lb = open(log_r).read().splitlines(), open(log_m).read().splitlines()
    assert len(la) == len(lb) == steps
    dirs = set()
    for i, (x, y) in enumerate(zip(la, lb)):
        va, ga = _parse_line(x)
        vb, gb = _parse_line(y)
        assert ga == gb and va == vb, "step %d differs" % i
        dirs.update(v[4] for v in va)
    assert dirs == {-1, 0, 1}  # both directions of lane change were logged


@pytest.mark.gpu
def test_snapshot_and_load_with_lane_change_hip(mod, scen, workdir):
    """the same on the HIP engine: cfx_load_state restores the vid-indexed lane-change tables"""
    _snapshot_roundtrip(mod, scen, workdir, lambda cfg: mod.Engine(cfg, 1))


def test_snapshot_and_load_with_lane_change(mod, scen, workdir):
    """Engine.snapshot() / load() in memory (reference engine.h:176-177) carry the lane-change state that outlives a step:
    partner links, lateral offset, the signal of a change in progress, cooling timers, the stored gap, the id chains and
    the generator.  A run resumed from a snapshot taken in the middle of several lane changes equals the uninterrupted
    one (the JSON form: test_archive_json_with_lane_change_both_directions)."""
    _snapshot_roundtrip(mod, scen, workdir, lambda cfg: mod.Engine._with_backend(cfg, 1, TWIN_LIB))


def _snapshot_roundtrip(mod, scen, workdir, make):
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    eng = make(cfg)
    for _ in range(31):
        eng.next_step()
    assert eng.get_vehicle_count() > len(eng.get_vehicle_speed())  # shadows alive: changes in progress
    arch = eng.snapshot()

    def advance(e, n):
        for _ in range(n):
            e.next_step()
        return (e.get_vehicle_count(), e.get_vehicle_speed(), e.get_vehicle_distance(), e.get_lane_vehicles(), e.get_vehicles(True),
                e.get_average_travel_time())

    want = advance(eng, 80)
    eng.load(arch)
    assert advance(eng, 80) == want
    other = make(cfg)  # a fresh engine: nothing but the archive
    other.load(arch)
    assert advance(other, 80) == want


def _irregular_lane_change(scen, workdir, seed, bends=True):
    from test_irregular import irregular
    cfg = irregular(scen, os.path.join(workdir, "lc"), seed, bends=bends)
    c = json.load(open(cfg))
    c["laneChange"] = True
    with open(cfg, "w") as f:
        json.dump(c, f)
    return cfg


@pytest.mark.parametrize("seed", [11, 13])
def test_irregular_lane_change_reference_vs_twin(scen, workdir, seed):
    """Jittered networks (bent multi-point roads, random lane widths, speed limits, signal plans, vehicle templates,
    tests/test_irregular.py) with laneChange=true: thousands of vehicles, changes on every multi-lane road."""
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    cfg = _irregular_lane_change(scen, workdir, seed)
    r, t = lcp.run("ref", cfg, 200), lcp.run("twin", cfg, 200)
    assert lcp.compare(r, t) == []
    assert r["count"] > 2000 and r["count"] > len(r["speed"])  # shadows alive at the checkpoint


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_irregular_lane_change_hip_vs_twin(mod, scen, workdir, seed):
    """Jittered networks with bent roads: the lanes of such a road differ in length, hence in their segment boundaries,
    and the reference looks a vehicle's neighbours up with the segment number it has on its OWN lane (lanechange.cpp:30,52;
    roadnet.cpp:877-898) — near a boundary it skips the true neighbour and inserts the shadow out of distance order.  The
    device engine walks the segment lists the same way (k_lc_segments, k_lc_schedule)."""
    cfg = _irregular_lane_change(scen, workdir, seed)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(400):
        hip.next_step()
        tw.next_step()
        if s % 10 == 9:
            a, b = _state(hip), _state(tw)
            for k in a:
                assert np.array_equal(a[k], b[k]), (seed, s, k)
    assert len(a["vid"]) > 2000


@pytest.mark.gpu
def test_hip_lane_change_with_control_calls(mod, scen, workdir):
    """Lane change together with the control API: RL-driven signals, push_vehicle (with an initial speed), set_vehicle_speed
    on real vehicles of changing pairs, reset(seed) in the middle — HIP == twin after every step."""
    cfg = scen.materialize("example_1x1", workdir, laneChange=True, rlTrafficLight=True)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    inter = [i for i in hip.intersection_ids() if not i.endswith("_0_1") or True]
    roads = sorted({l.rsplit("_", 1)[0] for l in hip.lane_ids()})
    for s in range(260):
        if s % 15 == 0:
            for e in (hip, tw):
                for iid in inter:
                    try:
                        e.set_tl_phase(iid, (s // 15) % 8)
                    except (IndexError, RuntimeError):
                        pass  # virtual intersections
        if s in (20, 21, 90):
            for e in (hip, tw):
                e.push_vehicle({"length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0,
                                "usualNegAcc": 4.5, "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5, "speed": 3.0},
                               ["road_0_1_0", "road_1_1_0"])
        if s % 7 == 3:  # slow down whoever is changing lane right now
            st = _state(tw)
            for v in st["vid"][(st["lc_flags"] & 2) != 0][:3]:
                vid = tw._vehicle_id(int(v))
                for e in (hip, tw):
                    e.set_vehicle_speed(vid, 4.0)
        if s == 130:
            for e in (hip, tw):
                e.reset(True)
        hip.next_step()
        tw.next_step()
        a, b = _state(hip), _state(tw)
        for k in a:
            assert np.array_equal(a[k], b[k]), (s, k)
    assert roads and len(a["vid"]) > 100
    assert hip.get_vehicle_speed() == tw.get_vehicle_speed() and hip.get_lane_vehicles() == tw.get_lane_vehicles()


def test_archive_json_with_lane_change_both_directions(scen, workdir):
    """Archive.dump / load_from_file with lane-change state (reference archive.cpp:229-246,407-474): a dump taken by this
    engine in the middle of several lane changes resumes identically in the reference and here, and so does a dump taken by
    the reference.  (After a load both sides number / allocate the vehicles in the file's order.)"""
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    for writer in ("twin", "ref"):
        path = os.path.join(workdir, "lc_archive_%s.json" % writer)
        st = lcp.run(writer, cfg, 31, save=path)
        assert st["count"] > len(st["speed"])  # shadows alive in the dump
        assert any(v.get("partnerType") == 2 for v in json.load(open(path))["vehicles"])
        r, t = lcp.run("ref", cfg, 70, load=path), lcp.run("twin", cfg, 70, load=path)
        assert lcp.compare(r, t) == [], writer


def test_lane_change_with_control_calls_reference_vs_twin(scen, workdir):
    """laneChange with the control API on both sides: RL-set signal phases, push_vehicle with an initial speed,
    set_vehicle_speed on the real vehicles of changing pairs (the shadow copies the pending custom speed), half-second
    steps, another seed."""
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    veh = {"length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0, "usualNegAcc": 4.5,
           "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5, "speed": 3.0}
    script = {}
    for s in range(0, 200, 15):
        script.setdefault(str(s), []).append(["set_tl_phase", "intersection_1_1", (s // 15) % 8])
    for s in (20, 21, 90):
        script.setdefault(str(s), []).append(["push_vehicle", veh, ["road_0_1_0", "road_1_1_0"]])
    for s in range(10, 200, 7):
        script.setdefault(str(s), []).append(["slow_changing", 3, 4.0])
    for kw in ({"rlTrafficLight": True}, {"interval": 0.5, "seed": 7}):
        cfg = scen.materialize("example_1x1", workdir, laneChange=True, **kw)
        calls = script if kw.get("rlTrafficLight") else {k: [c for c in v if c[0] != "set_tl_phase"] for k, v in script.items()}
        env = {"CFX_LC_SCRIPT": json.dumps(calls)}
        r = lcp.run("ref", cfg, 200, env=dict(lcp.reference_env(), **env))
        t = lcp.run("twin", cfg, 200, env=env)
        assert lcp.compare(r, t) == [], kw
        assert r["count"] > 100


@pytest.mark.parametrize("seed", [0, 3])
def test_lane_change_random_control_scripts_reference_vs_twin(scen, workdir, seed):
    """Random scripts of control calls between the steps of a lane-change run — signal phases, vehicles pushed with an initial
    speed (also right after a step that created shadows: the new vehicle's priority is drawn AFTER those shadows' — found by
    the fuzz run of tests/tools/lane_change_control_fuzz.py), custom speeds on the real vehicles of changing pairs — on the reference
    (ascending `new Vehicle` addresses, own process) and on the twin: every getter's result equal at the end."""
    if not os.path.exists(os.path.join(REF_DIR, "libmonotonic_new.so")):
        pytest.skip("oracle/_ref reference build not present")
    import numpy as np
    veh = {"length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0, "usualNegAcc": 4.5,
           "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5}
    routes = [["road_0_1_0", "road_1_1_0"], ["road_1_0_1", "road_1_1_1"], ["road_2_1_2", "road_1_1_2"], ["road_1_2_3", "road_1_1_3"],
              ["road_0_1_0", "road_1_1_1"], ["road_1_0_1", "road_1_1_2"]]
    rng = np.random.default_rng(seed)
    rl = bool(seed % 2)
    steps = int(rng.integers(120, 320))
    script = {}
    for s in range(steps):
        if rl and rng.random() < 0.08:
            script.setdefault(str(s), []).append(["set_tl_phase", "intersection_1_1", int(rng.integers(0, 8))])
        if rng.random() < 0.03:
            v = dict(veh, speed=float(rng.uniform(0, 8)), maxSpeed=float(rng.uniform(9, 17)), length=float(rng.uniform(4, 7)))
            script.setdefault(str(s), []).append(["push_vehicle", v, routes[int(rng.integers(0, len(routes)))]])
        if rng.random() < 0.1:
            script.setdefault(str(s), []).append(["slow_changing", int(rng.integers(1, 5)), float(rng.uniform(0, 9))])
    cfg = scen.materialize("example_1x1", workdir, laneChange=True, rlTrafficLight=rl, interval=(1.0, 0.5)[seed % 3 == 0],
                           seed=int(seed))
    env = {"CFX_LC_SCRIPT": json.dumps(script)}
    r = lcp.run("ref", cfg, steps, env=dict(lcp.reference_env(), **env))
    t = lcp.run("twin", cfg, steps, env=env)
    assert lcp.compare(r, t) == [], seed
    assert r["count"] > 100 and any(c[0] == "push_vehicle" for calls in script.values() for c in calls)


def test_shadow_priority_peek_exact_loop(scen, workdir, lc_golden):
    """The host offers the step the priorities its generator would hand out next.  Normally that is n plain draws (fast
    path); a draw that meets a live priority or repeats sends it through the exact redraw loop of the Vehicle constructor
    (vehicle.cpp:33).  Forced through that loop, the run still reproduces the reference's vectors."""
    cfg = scen.materialize("example_1x1", workdir, laneChange=True, cfx={"exactShadowPeek": True})
    got = record(lcp.run("twin", cfg, 200))
    assert got == lc_golden["example_1x1"]["200"]
