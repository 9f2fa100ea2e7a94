"""Replay logging (reference Engine::updateLog engine.cpp:518-554, RoadNet::convertToJson roadnet.cpp:327-394) against the
files the unmodified reference engine writes for the same run: same structure and ids, every number parses to the same
double (the two printers differ in style — "5.0" vs "5" — but both round-trip), same light string."""
import json
import os
import time

import pytest

from conftest import TWIN_LIB


def _cfg(scen, workdir, name, tag, **extra):
    base = scen.materialize(name, workdir, saveReplay=True)
    c = json.load(open(base))
    c.update(roadnetLogFile="roadnet_log_%s.json" % tag, replayLogFile="replay_%s.txt" % tag, **extra)
    path = os.path.join(os.path.dirname(base), "config_replay_%s.json" % tag)
    with open(path, "w") as f:
        json.dump(c, f)
    return path, c["dir"] + c["roadnetLogFile"], c["dir"] + c["replayLogFile"]


def _parse_line(line):
    veh, lights = line.rstrip("\n").split(";")
    out = []
    for v in veh.split(","):
        if not v:
            continue
        x, y, ang, vid, lc, ln, w = v.split(" ")
        out.append((float(x), float(y), float(ang), vid, int(lc), float(ln), float(w)))
    return out, lights


@pytest.mark.parametrize("name,steps", [("example_1x1", 150), ("grid_6x6", 120)])
def test_replay_files_match_reference(mod, scen, workdir, ref_module, name, steps):
    cfg_r, net_r, log_r = _cfg(scen, workdir, name, "ref")
    cfg_m, net_m, log_m = _cfg(scen, workdir, name, "mine")
    ref = ref_module.Engine(cfg_r, 1)
    mine = mod.Engine._with_backend(cfg_m, 1, TWIN_LIB)
    for _ in range(steps):
        ref.next_step()
        mine.next_step()
    time.sleep(0.2)  # reference destructor race
    del ref, mine    # closes both logs
    # static roadnet log: equal as JSON (floats exactly)
    a, b = json.load(open(net_r)), json.load(open(net_m))
    assert a == b
    assert len(a["static"]["nodes"]) > 0 and all(len(n["outline"]) >= 6 for n in a["static"]["nodes"] if not n["virtual"])
    # per-step log
    la, lb = open(log_r).read().splitlines(), open(log_m).read().splitlines()
    assert len(la) == len(lb) == steps
    seen = 0
    for i, (x, y) in enumerate(zip(la, lb)):
        va, ga = _parse_line(x)
        vb, gb = _parse_line(y)
        assert ga == gb, "step %d: light states differ" % i
        assert va == vb, "step %d: vehicles differ" % i
        seen = max(seen, len(va))
    assert seen > 10


def test_replay_switches(mod, scen, workdir, capfd):
    # without saveReplay in the config both calls only print the reference's message (engine.cpp:727-742)
    eng = mod.Engine._with_backend(scen.materialize("example_1x1", workdir), 1, TWIN_LIB)
    eng.set_replay_file("x.txt")
    eng.set_save_replay(True)
    assert capfd.readouterr().err.count("saveReplay is not set to true in config file!") == 2
    # with it: the file can be switched and logging paused
    cfg, _net, log = _cfg(scen, workdir, "example_1x1", "switch")
    eng = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for _ in range(5):
        eng.next_step()
    eng.set_save_replay(False)
    for _ in range(5):
        eng.next_step()
    eng.set_save_replay(True)
    eng.set_replay_file("replay_second.txt")
    for _ in range(3):
        eng.next_step()
    del eng
    assert len(open(log).read().splitlines()) == 5
    assert len(open(os.path.join(os.path.dirname(log), "replay_second.txt")).read().splitlines()) == 3


@pytest.mark.gpu
def test_replay_hip_equals_twin(mod, scen, workdir):
    cfg_a, _n, log_a = _cfg(scen, workdir, "grid_6x6", "hip")
    cfg_b, _n, log_b = _cfg(scen, workdir, "grid_6x6", "twin")
    a = mod.Engine(cfg_a, 1)
    b = mod.Engine._with_backend(cfg_b, 1, TWIN_LIB)
    for _ in range(100):
        a.next_step()
        b.next_step()
    del a, b
    assert open(log_a).read() == open(log_b).read()


@pytest.mark.parametrize("seed", [3, 11, 14, 27])
def test_roadnet_log_outlines_match_reference_on_irregular_networks(mod, scen, workdir, ref_module, seed):
    """Intersection::getOutline (roadnet.cpp:750-818) on jittered intersections, bent roads and T / L junctions: hull pops,
    collinear candidates and equal polar angles occur there, which the axis-aligned grids never produce — every outline point
    equal to the reference's as a double."""
    from test_irregular import irregular
    base = irregular(scen, workdir, seed)
    c = json.load(open(base))
    logs = []
    for tag in ("ref", "mine"):
        cc = dict(c, saveReplay=True, roadnetLogFile="roadnet_log_irr%d_%s.json" % (seed, tag), replayLogFile="replay_irr%d_%s.txt" % (seed, tag))
        path = os.path.join(os.path.dirname(base), "config_replay_irr%d_%s.json" % (seed, tag))
        with open(path, "w") as f:
            json.dump(cc, f)
        logs.append((path, cc["dir"] + cc["roadnetLogFile"]))
    ref = ref_module.Engine(logs[0][0], 1)
    mine = mod.Engine._with_backend(logs[1][0], 1, TWIN_LIB)
    ref.next_step()
    mine.next_step()
    time.sleep(0.2)
    del ref, mine
    a, b = json.load(open(logs[0][1])), json.load(open(logs[1][1]))
    assert a == b
    assert sum(len(n["outline"]) for n in a["static"]["nodes"]) > 100
