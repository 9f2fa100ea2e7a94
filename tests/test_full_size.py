"""BASELINE.json configs[4] — 100x100 grid, ~1M running vehicles, 8 tiles, per-step set_tl_phase / get_lane_vehicle_count
RL calls — through the size-independent property the tiling offers: the tiled network (2x4 tiles, halo through GPU
mailboxes; all on the one GPU of the test box) and the single engine, driven by the same random signal plan and read
back through the RL getters every step, must agree exactly — and, from a checkpoint of that run on, both must agree with
the CPU twin (the oracle) step by step under the same agent.  The same loop runs small on the CPU twin."""
import os
import time

import numpy as np
import pytest

from conftest import TWIN_LIB


def build(scen, workdir, n, flows_per_100_inters, **config):
    base = scen.generate_grid(n, n, workdir)
    d = os.path.dirname(base)
    n_extra = n * n * flows_per_100_inters // 100
    flow = os.path.join(d, "flow_rl_%d.json" % n_extra)
    if not os.path.exists(flow):
        scen.dense_flows(os.path.join(d, "roadnet.json"), flow, n_extra, seed=4242, interval=6.0,
                         base_flow=os.path.join(d, "flow.json"), end_time=240)
    import json
    cfg = dict(json.load(open(base)), flowFile=os.path.basename(flow), rlTrafficLight=True, **config)
    path = os.path.join(d, "config_rl.json")
    with open(path, "w") as f:
        json.dump(cfg, f)
    return path


def rl_loop(mod, cfg, rows, cols, lib, warmup, steps, min_running, twin_steps=0):
    single = mod.Engine._with_backend(cfg, 1, lib)
    tiled = mod.TiledEngine(cfg, rows, cols, [], lib)
    tiled.enable_mailboxes("full_%d_%d" % (os.getpid(), rows * cols))
    rng = np.random.default_rng(2024)
    n_inter = len(single.intersection_ids())
    for s in range(warmup + steps):
        if s % 15 == 0:  # the agent's action: a new phase for every signal
            ph = rng.integers(0, 8, size=n_inter).astype(np.int32)
            single.set_tl_phases(ph)
            tiled.set_tl_phases(ph)
        single.next_step()
        tiled.next_step()
        if s >= warmup or s % 50 == 49:  # the agent's observation
            a, b = single.get_lane_vehicle_count_array(), tiled.get_lane_vehicle_count_array()
            assert np.array_equal(a, b), "step %d: lane counts differ on %d lanes" % (s, int((a != b).sum()))
    assert np.array_equal(single.get_lane_waiting_vehicle_count_array(), tiled.get_lane_waiting_vehicle_count_array())
    sa, sb = single._scalars(), tiled._scalars()
    for k in ("active_vehicle_count", "finished_vehicle_count", "vehicle_steps", "cumulative_travel_time"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    assert sa["active_vehicle_count"] >= min_running, sa["active_vehicle_count"]
    va, vb = single._vehicle_state(), tiled._vehicle_state()
    for k in ("vid", "drivable", "dis", "speed", "leader", "blocker"):
        if not np.array_equal(va[k], vb[k]):
            bad = np.nonzero(va[k] != vb[k])[0]
            raise AssertionError("field %s differs for %d vehicles; first: vid %s drivable %s single %s tiled %s" % (
                k, bad.size, va["vid"][bad[:5]], va["drivable"][bad[:5]], va[k][bad[:5]], vb[k][bad[:5]]))
    if twin_steps:
        # ... and against the ORACLE at this checkpoint: the state goes into the CPU twin (and into a fresh engine of `lib`: a
        # load restarts every route cursor, so loaded engines are compared with loaded engines) through an Archive; the three
        # take the same agent-driven steps — the tiles included, which never loaded — every vehicle field compared
        from conftest import assert_same_state
        snap = single.snapshot()
        loaded = mod.Engine._with_backend(cfg, 1, lib)
        loaded.load(snap)
        tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
        tw.load(snap)
        del snap
        assert_same_state(loaded, tw, "%dx%d tiles' checkpoint in the twin" % (rows, cols))
        for s in range(twin_steps):
            if s % 3 == 0:
                ph = rng.integers(0, 8, size=n_inter).astype(np.int32)
                for e in (loaded, tw, tiled):
                    e.set_tl_phases(ph)
            for e in (loaded, tw, tiled):
                e.next_step()
            a = loaded.get_lane_vehicle_count_array()
            assert np.array_equal(a, tw.get_lane_vehicle_count_array()), "twin step %d" % s
            assert np.array_equal(a, tiled.get_lane_vehicle_count_array()), "tiles, twin step %d" % s
        assert_same_state(loaded, tw, "after %d agent-driven steps from the checkpoint" % twin_steps)
    return sa["active_vehicle_count"]


def test_rl_loop_12x12_twin(mod, scen, workdir):
    cfg = build(scen, workdir, 12, 330)
    rl_loop(mod, cfg, 2, 2, TWIN_LIB, warmup=150, steps=30, min_running=5000, twin_steps=6)


@pytest.mark.gpu
def test_config5_100x100_one_million_vehicles(mod, scen, workdir):
    t0 = time.time()
    cfg = build(scen, workdir, 100, 333)
    running = rl_loop(mod, cfg, 2, 4, mod._default_backend_path(), warmup=300, steps=40, min_running=900000, twin_steps=12)
    print("100x100: %d running vehicles, %.0f s" % (running, time.time() - t0))


# ---- BASELINE.json configs[4] against the REFERENCE ITSELF -------------------------------------------------------------
# tests/golden/reference_large.json: the unmodified reference engine (one thread, Vehicle objects at creation-ordered
# addresses — the reproducible reference, see tests/golden/make_large_goldens.py) stepped from step 0 on bench.py's
# `roofline_at_scale` workload (gen_100x100 + 33 000 seeded flows, ~1.04 M running vehicles past step 300).
def _large_golden():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_large.json")
    with open(path) as f:
        return json.load(f)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["auto", "dense"])
def test_config5_one_million_vehicles_matches_reference_goldens(mod, workdir, layout):
    """HIP engine from step 0 on the 100x100 / 1 M vehicle workload == the reference's own records at steps 150, 305, 360,
    420 (bench.py's scale leg passes the same steps): vehicle count, every lane's count, average travel time, the multiset of
    all (speed, distance) bits, signal phases; which vehicle carries which pair: see bench.scale_compare.
    `auto` is the ring layout's list form at this size (kr_index + kl_action + k_cross2), `dense` the scan / scatter layout."""
    import bench
    from test_parity_pins import _with_cfx
    t0 = time.time()
    gold = _large_golden()
    cfg = bench.build_workload(workdir, 0, scenario="gen_%dx%d" % (bench.SCALE_GRID, bench.SCALE_GRID), n_extra=bench.SCALE_FLOWS)
    eng = mod.Engine(_with_cfx(cfg, layout=layout) if layout != "auto" else cfg, 1)
    assert len(eng.lane_ids()) == gold["n_lanes"]
    net = eng._flat_net()
    real = {k for k, v in zip(eng.intersection_ids(), net["inter_virtual"]) if not v}
    want = {int(k): v for k, v in gold["checkpoints"].items()}
    for s in range(1, max(want) + 1):
        eng.next_step()
        if s in want:
            got = bench.scale_record(eng, real if "phase_hash" in want[s] else None)
            verdict = bench.scale_compare(got, want[s], int(eng._scalars()["tie_events"]))
            assert verdict["equal"], "step %d (%s layout): %r" % (s, layout, verdict)
    assert eng.get_vehicle_count() > 1000000
    print("100x100 vs reference goldens (%s, %s): %.0f s" % (layout, eng._layout(), time.time() - t0))
