"""Lane::history (reference src/roadnet/roadnet.cpp:900-915, roadnet.h:305-316): the last 241 steps' {vehicle count, mean speed}
of every lane and the two running aggregates.  Nothing in the reference can read them back (they feed the DURATION router,
which cannot be selected) — what shows them is Archive.dump (archive.cpp:286-294).  Kept by `Engine` unless `"cfx": {"laneHistory": false}`.
CPU: the twin against the unmodified reference through the dumps of both; GPU: the HIP engine against the twin through the ABI."""
import json
import os
import time

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_hip_backend


def history_cfg(scen, workdir, name, tag="", **kw):
    base = scen.materialize(name, workdir, **kw)
    c = json.load(open(base))
    c["cfx"] = dict(c.get("cfx", {}), laneHistory=True, **({"layout": tag} if tag in ("ring", "dense") else {}))
    path = base.replace(".json", "_history%s.json" % tag)
    with open(path, "w") as f:
        json.dump(c, f)
    return path


def history_of(path, written_by_reference, mod):
    """lane id -> (records, historyVehicleNum, historyAverageSpeed), every double as its writer meant it (tests/test_api_sequences.py
    _comparable_dump)"""
    with open(path) as f:
        d = json.load(f) if written_by_reference else json.load(f, parse_float=lambda lit: mod._parse_json_number(lit)[0])
    return {k: (v["history"], v["historyVehicleNum"], v["historyAverageSpeed"]) for k, v in d["drivables"].items() if "history" in v}


@pytest.mark.parametrize("name,steps", [("example_1x1", 330), ("grid_6x6", 300)])
def test_lane_history_twin_equals_reference(mod, ref_module, scen, workdir, tmp_path, name, steps):
    cfg = history_cfg(scen, workdir, name)
    ref, tw = ref_module.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    p_ref, p_tw = str(tmp_path / "ref.json"), str(tmp_path / "tw.json")
    for horizon in (7, 240, 241, 242, steps):  # before, at and behind the point where the list starts to lose its oldest record
        while ref.get_current_time() < horizon:
            ref.next_step()
            tw.next_step()
        ref.snapshot().dump(p_ref)
        tw.snapshot().dump(p_tw)
        a, b = history_of(p_ref, True, mod), history_of(p_tw, False, mod)
        assert a.keys() == b.keys() and len(a) > 0
        assert a == b, horizon
        assert max(len(v[0]) for v in a.values()) == 2 * min(horizon, 241)
    assert any(v[1] > 0 for v in a.values())
    # Engine::reset does not clear it (Lane::reset roadnet.cpp:832-835): the lists go on
    ref.reset(False)
    tw.reset(False)
    for _ in range(20):
        ref.next_step()
        tw.next_step()
    ref.snapshot().dump(p_ref)
    tw.snapshot().dump(p_tw)
    assert history_of(p_ref, True, mod) == history_of(p_tw, False, mod)
    # ... and it travels through files in both directions (archive.cpp:508-521): both load this engine's file and go on
    ref.load_from_file(p_tw)
    tw.load_from_file(p_tw)
    for _ in range(30):
        ref.next_step()
        tw.next_step()
    ref.snapshot().dump(p_ref)
    tw.snapshot().dump(p_tw)
    assert history_of(p_ref, True, mod) == history_of(p_tw, False, mod)
    time.sleep(0.2)


def test_lane_history_is_kept_unless_the_config_says_no(mod, scen, workdir, tmp_path):
    """Engine keeps Lane::history by default (its dumps then carry what the reference's carry, archive.cpp:286-294);
    `"cfx": {"laneHistory": false}` drops it, and the dump's history fields are empty."""
    base = scen.materialize("example_1x1", workdir)
    tw = mod.Engine._with_backend(base, 1, TWIN_LIB)
    for _ in range(5):
        tw.next_step()
    p = str(tmp_path / "d.json")
    tw.snapshot().dump(p)
    assert all(len(v[0]) == 10 for v in history_of(p, False, mod).values())
    assert tw._lane_history()["len"].tolist() == [5] * len(tw.lane_ids())
    c = json.load(open(base))
    c["cfx"] = {"laneHistory": False}
    off = base.replace(".json", "_nohistory.json")
    with open(off, "w") as f:
        json.dump(c, f)
    tw = mod.Engine._with_backend(off, 1, TWIN_LIB)
    for _ in range(5):
        tw.next_step()
    tw.snapshot().dump(p)
    assert all(v[0] == [] and v[1] == 0 for v in history_of(p, False, mod).values())
    with pytest.raises(RuntimeError):
        tw._lane_history()


def lane_history_body(cfg, make_a, make_b, steps):
    """cfx_get_lane_history of two engines equal after every 50th step; then a snapshot loaded into both."""
    a, b = make_a(cfg), make_b(cfg)
    for s in range(steps):
        a.next_step()
        b.next_step()
        if s % 50 == 49 or s == steps - 1:
            ha, hb = a._lane_history(), b._lane_history()
            for k in ha:
                assert np.array_equal(ha[k], hb[k]), (s, k)
    assert ha["len"].max() in (min(steps, 241), min(2 * steps, 241))  # (lane change: two records per step)
    arch = a.snapshot()
    b.load(arch)
    a.load(arch)
    for s in range(20):
        a.next_step()
        b.next_step()
    ha, hb = a._lane_history(), b._lane_history()
    for k in ha:
        assert np.array_equal(ha[k], hb[k]), k


def test_lane_history_body_on_the_twin(mod, scen, workdir):
    """(CPU shadow of the GPU tests below)"""
    mk = lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB)
    lane_history_body(history_cfg(scen, workdir, "example_1x1"), mk, mk, 60)
    lane_history_body(history_cfg(scen, workdir, "example_1x1", laneChange=True), mk, mk, 60)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["ring", "dense"])
def test_lane_history_hip_equals_twin(mod, scen, workdir, layout):
    lane_history_body(history_cfg(scen, workdir, "grid_6x6", layout), lambda c: mod.Engine(c, 1),
                      lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), 300)


@pytest.mark.gpu
def test_lane_history_with_lane_change_hip_equals_twin(mod, scen, workdir):
    """two records per step (the leader / gap pass also runs between planLaneChange and getAction, engine.cpp:571-575)"""
    lane_history_body(history_cfg(scen, workdir, "example_1x1", laneChange=True), lambda c: mod.Engine(c, 1),
                      lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), 200)


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 10000, 20000, 30000])
def test_lane_history_rides_with_the_next_action_launch(mod, scen, workdir, form):
    """Ring layout, round 6: a step's record is taken by trailing blocks of the NEXT step's action launch (block form, wave form,
    list form) or, when somebody asks first, by a launch of its own.  An agent's loop (signals set, lane counts read every step:
    the commit is a launch of its own, the getters do not ask for the history), free-running stretches (the commit rides with
    the next admission), resets and loads in between — the record of the last step before each must not get lost
    (Lane::reset keeps the history, roadnet.cpp:832-835)."""
    base = scen.materialize("grid_6x6", workdir, rlTrafficLight=True)
    c = json.load(open(base))
    c["cfx"] = {"laneHistory": True, "layout": "ring", "ringLanesPerWave": form}
    cfg = base.replace(".json", "_history_form%d.json" % form)
    with open(cfg, "w") as f:
        json.dump(c, f)
    hip, tw = mod.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    assert_hip_backend(hip)
    n_inter = len(hip.intersection_ids())
    rng = np.random.default_rng(form + 1)

    def same(tag):
        ha, hb = hip._lane_history(), tw._lane_history()
        for k in ha:
            assert np.array_equal(ha[k], hb[k]), (tag, k)
        return ha

    def both(f):
        f(hip)
        f(tw)

    for round_ in range(6):
        for s in range(int(rng.integers(20, 60))):  # an agent's loop
            ph = rng.integers(0, 4, n_inter).astype(np.int32)
            both(lambda e: e.set_tl_phases(ph))
            both(lambda e: e.next_step())
            assert np.array_equal(hip.get_lane_vehicle_count_array(), tw.get_lane_vehicle_count_array())
            if s % 17 == 3:
                assert hip.get_vehicle_count() == tw.get_vehicle_count()
        same(("agent", round_))
        for s in range(int(rng.integers(20, 60))):  # free-running
            both(lambda e: e.next_step())
        if round_ % 3 == 0:
            both(lambda e: e.reset(False))  # (the record of the step before the reset is the reference's too)
            both(lambda e: e.next_step())
        elif round_ % 3 == 1:
            arch = tw.snapshot()
            for s in range(7):
                both(lambda e: e.next_step())
            both(lambda e: e.load(arch))
        h = same(("free", round_))
    assert h["len"].max() == 241 and h["history_vehicle_num"].sum() > 0


# ---- tiles: a lane's history is kept by the tile that owns the lane; the step's record is taken behind the step's halo import
#      (the vehicles that entered a cut lane in the step are on it only then — as they are at the end of the step on one engine)
def tiled_history_body(mod, scen, workdir, tmp_path, make_tiled, make_single, steps):
    import os
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_hist_tiles.json"), 150, seed=17,
                            interval=3.0, base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow)
    til, one = make_tiled(cfg), make_single(cfg)
    assert til._keeps_lane_history() and one._keeps_lane_history()
    p_t, p_o = str(tmp_path / "tiles.json"), str(tmp_path / "one.json")
    for s in range(steps):
        til.next_step()
        one.next_step()
        if s in (3, 60, steps // 2, steps - 1):
            til.snapshot().dump(p_t)
            one.snapshot().dump(p_o)
            ht, ho = history_of(p_t, False, mod), history_of(p_o, False, mod)
            assert ht == ho, (s, [k for k in ho if ht.get(k) != ho[k]][:5])
    assert max(len(v[0]) for v in ho.values()) == 2 * min(steps, 241) and sum(v[1] for v in ho.values()) > 0
    # through a load (the tiles take the archive's history back) and a compaction
    arch = one.snapshot()
    til.load(arch)
    one.load(arch)
    for s in range(30):
        til.next_step()
        one.next_step()
        if s == 10:
            til._compact_vehicles()
    til.snapshot().dump(p_t)
    one.snapshot().dump(p_o)
    assert history_of(p_t, False, mod) == history_of(p_o, False, mod)
    # (the rest of the two dumps: tests/test_tiling.py compares them whole.  After a LOAD the HIP tiles' `blocker` fields can
    #  differ from one engine's where a vehicle's blocker runs in another tile — the archive names it by number, a tile resolves
    #  numbers among its own vehicles; the blocker test of Cross::canPass only ever concerns vehicles at one intersection)
    a, b = json.load(open(p_t)), json.load(open(p_o))
    for dump in (a, b):
        for v in dump["vehicles"]:
            v.pop("blocker", None)
    assert a == b


@pytest.mark.parametrize("mailboxes", [False, True])
def test_lane_history_on_tiles_twin(mod, scen, workdir, tmp_path, mailboxes):
    def tiles(c):
        t = mod.TiledEngine(c, 2, 3, [], TWIN_LIB)
        if mailboxes:
            t.enable_mailboxes("hist_tw_%d" % __import__("os").getpid())
        return t
    tiled_history_body(mod, scen, workdir, tmp_path, tiles, lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), 260)


@pytest.mark.gpu
@pytest.mark.parametrize("mailboxes", [False, True])
def test_lane_history_on_tiles_hip(mod, scen, workdir, tmp_path, mailboxes):
    """HIP tiles (the record rides in the next step's action launch, behind the import in that step's admission launch — or
    behind the import kernel of the staged exchange) against the twin on the whole network."""
    def tiles(c):
        t = mod.TiledEngine(c, 2, 2)
        if mailboxes:
            t.enable_mailboxes("hist_hip_%d" % __import__("os").getpid())
        return t
    tiled_history_body(mod, scen, workdir, tmp_path, tiles, lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), 300)
