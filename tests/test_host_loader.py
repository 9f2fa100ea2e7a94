"""CPU: the from-scratch host loader and spawner against vectors produced by the UNMODIFIED reference
(tests/golden/make_goldens.py): load-time geometry (reference roadnet.cpp:42-325,456-576) and the mt19937
spawn stream (flow.cpp:6-22, vehicle.cpp:38-47, engine.cpp:450-470,605-613, router.cpp:23-37,96-100)."""
import hashlib
import os

import pytest


@pytest.mark.parametrize("name", ["example_1x1", "grid_6x6", "grid_30x30"])
def test_geometry_matches_reference_probe(mod, scen, workdir, golden, name):
    cfg = scen.materialize(name, workdir)
    text = mod._roadnet_probe(os.path.join(os.path.dirname(cfg), "roadnet.json"))
    g = golden["roadnet_probe"][name]
    assert text.count(b"\n") == g["lines"]
    assert hashlib.sha256(text).hexdigest() == g["sha256"], "lane/laneLink lengths or crosses differ from the reference"


@pytest.mark.parametrize("name", ["example_1x1", "grid_6x6"])
def test_spawn_stream_matches_reference(mod, scen, workdir, golden, name):
    cfg = scen.materialize(name, workdir)
    d = os.path.dirname(cfg)
    ref = golden["reference_spawns"][name]
    steps = max(r[3] for r in ref) + 1
    sched = mod._spawn_schedule(os.path.join(d, "roadnet.json"), os.path.join(d, "flow.json"), 1.0, 0, 1, steps)
    mine = sorted([vid, prio, lane, s] for s, recs in enumerate(sched) for (vid, prio, lane, *_rest) in recs)
    assert mine == [list(r) for r in ref]


def test_spawn_stream_is_thread_count_invariant(mod, scen, workdir):
    """thread_num only changes `rnd() % threadNum`, whose value is discarded (engine.cpp:606)."""
    cfg = scen.materialize("example_1x1", workdir)
    d = os.path.dirname(cfg)
    a = mod._spawn_schedule(os.path.join(d, "roadnet.json"), os.path.join(d, "flow.json"), 1.0, 0, 1, 40)
    b = mod._spawn_schedule(os.path.join(d, "roadnet.json"), os.path.join(d, "flow.json"), 1.0, 0, 8, 40)
    assert a == b


def test_flat_net_shape(mod, scen, workdir):
    cfg = scen.materialize("grid_6x6", workdir)
    net = mod._load_roadnet(os.path.join(os.path.dirname(cfg), "roadnet.json"))
    # SURVEY.md §8 size table, 6x6 row
    assert (net["n_lanes"], net["n_lanelinks"], net["n_xentries"]) == (504, 1296, 2 * 9072)
    import numpy as np
    peer = net["x_peer"]
    assert np.array_equal(peer[peer], np.arange(len(peer)))  # peer is an involution
    # entries of a laneLink are sorted by distance (roadnet.cpp:568-575)
    xs, dist = net["ll_x_start"], net["x_dist"]
    for k in range(net["n_lanelinks"]):
        seg = dist[xs[k]:xs[k + 1]]
        assert np.all(seg[1:] >= seg[:-1])


def test_malformed_config_raises(mod, tmp_path):
    p = tmp_path / "bad.json"
    p.write_text('{"interval": 1.0}')
    with pytest.raises(RuntimeError):
        mod.Engine(str(p), 1)
