"""Lane::history (reference src/roadnet/roadnet.cpp:900-915, roadnet.h:305-316): the last 241 steps' {vehicle count, mean speed}
of every lane and the two running aggregates.  Nothing in the reference can read them back (they feed the DURATION router,
which cannot be selected) — what shows them is Archive.dump (archive.cpp:286-294).  Kept with `"cfx": {"laneHistory": true}`.
CPU: the twin against the unmodified reference through the dumps of both; GPU: the HIP engine against the twin through the ABI."""
import json
import os
import time

import numpy as np
import pytest

from conftest import TWIN_LIB


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


def test_lane_history_is_off_by_default(mod, scen, workdir, tmp_path):
    tw = mod.Engine._with_backend(scen.materialize("example_1x1", workdir), 1, TWIN_LIB)
    for _ in range(5):
        tw.next_step()
    p = str(tmp_path / "d.json")
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
