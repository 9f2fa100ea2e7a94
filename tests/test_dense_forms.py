"""GPU (-m gpu): the organisations of the dense layout's step that cfx_config::dense_form selects (config "cfx": denseForm =
256 + bits; include/cityflow_amd.h) against the CPU twin — bit 1: the admission kernel over the lanes only (kd_admit<true>: a
laneLink's gate record rewritten only when its intersection's phase has changed, laneLink tails read from the committed
records); bit 2: up to 1024 spawn records of a step in the admission kernel's arguments.  256 = both off (kd_admit over all
drivables, k_spawn_link beyond 128 records); 0 (the default) = both on.  Each form is forced on the networks where the step's
corner cases live (the 1x1 example with up to 118 crosses per laneLink, the congested 6x6, irregular networks, the bench workload
through its demand build-up, an RL-driven run that changes phases every other step, resets and loads in between) with both forms
of the cross phase, and must give the twin's bits on every vehicle field.  Reference semantics: src/engine/engine.cpp:502-516
(handleWaiting), 317-372 (threadNotifyCross), src/roadnet/roadnet.cpp:603-676 (Cross::canPass), roadnet.h:429-431 (isAvailable)."""
import json
import os

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state
from test_parity_pins import _pair, _bench_cfg, _hip, many_spawns_body

pytestmark = pytest.mark.gpu

FORMS = [256, 258, 260, 262]


def _cross_for(form):
    """Small networks run the throughput form of the cross phase (k_cross2) only when told to: half of the forms do."""
    return "throughput" if form & 2 else "auto"


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("cross", ["latency", "throughput"])
def test_dense_forms_example_every_step(mod, scen, workdir, form, cross):
    hip, tw = _pair(mod, scen.materialize("example_1x1", workdir), layout="dense", crossMode=cross, denseForm=form)
    for s in range(400):
        hip.next_step()
        tw.next_step()
        assert_same_state(hip, tw, "1x1 denseForm %d (%s) step %d" % (form, cross, s + 1))
    assert hip.get_vehicle_count() > 50


@pytest.mark.parametrize("form", FORMS)
def test_dense_forms_congested_grid(mod, scen, workdir, form):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    hip, tw = _pair(mod, scen.materialize("grid_6x6", workdir, flow_file=flow), layout="dense", denseForm=form, crossMode=_cross_for(form))
    for s in range(400):
        hip.next_step()
        tw.next_step()
        if s % 2 == 1:
            assert_same_state(hip, tw, "dense 6x6 denseForm %d step %d" % (form, s + 1))
    assert hip.get_vehicle_count() > 3000
    assert hip.get_average_travel_time() == tw.get_average_travel_time()


@pytest.mark.parametrize("form", FORMS[1:])
@pytest.mark.parametrize("seed", [11, 14])
def test_dense_forms_irregular_networks(mod, scen, workdir, seed, form):
    from test_irregular import irregular
    hip, tw = _pair(mod, irregular(scen, workdir, seed), layout="dense", denseForm=form, crossMode=_cross_for(form))
    for s in range(500):
        hip.next_step()
        tw.next_step()
        if s % 5 == 4:
            assert_same_state(hip, tw, "irregular %d denseForm %d step %d" % (seed, form, s + 1))
    assert hip.get_vehicle_count() > 150


@pytest.mark.parametrize("form", FORMS[1:])
def test_dense_forms_rl_control_and_reset(mod, scen, workdir, form):
    """An agent's loop on the 6x6 grid (rlTrafficLight): new phases for a random third of the signals every other step through
    set_tl_phases, single set_tl_phase calls in between, the lane counts read every step; then reset() and the same again from
    step 0, then snapshot / load into a fresh engine — the gate records of the laneLinks must follow every one of these."""
    base = scen.materialize("grid_6x6", workdir)
    cfg = base.replace(".json", "_rl.json")
    with open(cfg, "w") as f:
        json.dump(dict(json.load(open(base)), rlTrafficLight=True), f)
    hip, tw = _pair(mod, cfg, layout="dense", denseForm=form, crossMode=_cross_for(form))
    ids = hip.intersection_ids()
    rng = np.random.default_rng(5)

    def drive(a, b, steps, tag):
        for s in range(steps):
            if s % 2 == 0:
                ph = rng.integers(0, 8, size=len(ids)).astype(np.int32)
                keep = rng.random(len(ids)) < 0.33
                cur = a._tl_state()[0]
                ph = np.where(keep, ph, cur).astype(np.int32)
                a.set_tl_phases(ph)
                b.set_tl_phases(ph)
            elif s % 7 == 3:
                i = int(rng.integers(0, len(ids)))
                p = int(rng.integers(0, 8))
                try:
                    a.set_tl_phase(ids[i], p)
                    b.set_tl_phase(ids[i], p)
                except (RuntimeError, IndexError):
                    pass  # (a virtual intersection has no phases to set)
            a.next_step()
            b.next_step()
            assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), (tag, s)
            if s % 3 == 2:
                assert_same_state(a, b, "%s denseForm %d step %d" % (tag, form, s + 1))

    drive(hip, tw, 150, "rl")
    hip.reset()
    tw.reset()
    drive(hip, tw, 120, "rl after reset")
    snap = hip.snapshot()
    hip2, tw2 = _pair(mod, cfg, layout="dense", denseForm=form, crossMode=_cross_for(form))
    hip2.load(snap)
    tw2.load(snap)
    drive(hip2, tw2, 60, "rl after load")


@pytest.mark.parametrize("form", [256, 262])
def test_dense_forms_bench_workload(mod, workdir, form):
    import bench
    hip, tw = _pair(mod, _bench_cfg(workdir), layout="dense", crossMode="throughput", denseForm=form)
    steps = bench.BUILD_UP_STEPS + 60
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % 10 == 9:
            assert_same_state(hip, tw, "bench workload denseForm %d step %d" % (form, s + 1))
    assert hip.get_vehicle_count() > 80000
    assert hip.get_average_travel_time() == tw.get_average_travel_time()


@pytest.mark.parametrize("form", [260, 262])
def test_dense_forms_many_spawns_per_lane(mod, scen, workdir, form):
    """Hundreds of spawn records in one step, ~20 of them on one lane, travel in kd_admit's arguments (bit 2: up to 1024; the
    body is tests/test_parity_pins.py's: chains inside the batch, heads where the queue had drained, appends behind vehicles
    still waiting); Lane::pushWaitingVehicle roadnet.h:365-367, engine.cpp:502-516."""
    many_spawns_body(mod, scen, workdir, _hip, layouts=("dense",), denseForm=form)
