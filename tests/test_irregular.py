"""Parity on IRREGULAR networks: generated grids are axis-aligned and uniform, so they exercise a thin slice of the load-time
geometry (lane offsets, generated laneLink curves, cross detection and distances) and of the dynamics (speed limits,
vehicle parameters).  Here a grid is jittered into a seeded irregular network — moved intersections, bent roads, roads
removed (T- and L-junctions), random lane widths / speed limits / intersection widths / signal plans, laneLink curves left
to the engine, random vehicle templates and random-walk routes — and the reference engine, the CPU twin and the HIP engine
must still agree exactly."""
import json
import os
import random
import time

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state, checkpoint_record


def irregular(scen, workdir, seed, n=5, bends=True):
    rng = random.Random(seed)
    net = scen.grid_roadnet(n, n)
    inters = {i["id"]: i for i in net["intersections"]}
    for it in net["intersections"]:
        amp = 25.0 if it["virtual"] else 70.0
        it["point"]["x"] += rng.uniform(-amp, amp)
        it["point"]["y"] += rng.uniform(-amp, amp)
        if not it["virtual"]:
            it["width"] = rng.choice([20, 25, 30, 35])
    # drop a few interior roads (one direction each)
    interior = [r for r in net["roads"] if not inters[r["startIntersection"]]["virtual"] and not inters[r["endIntersection"]]["virtual"]]
    dropped = {r["id"] for r in rng.sample(interior, max(2, len(interior) // 9))}
    net["roads"] = [r for r in net["roads"] if r["id"] not in dropped]
    for r in net["roads"]:
        a, b = inters[r["startIntersection"]]["point"], inters[r["endIntersection"]]["point"]
        pts = [dict(a)]
        if rng.random() < 0.4 and bends:  # a bend (the random draws are the same either way)
            mx, my = (a["x"] + b["x"]) / 2, (a["y"] + b["y"]) / 2
            pts.append({"x": mx + rng.uniform(-30, 30), "y": my + rng.uniform(-30, 30)})
        pts.append(dict(b))
        r["points"] = pts
        speed = rng.choice([8.33, 11.11, 13.89, 16.67])
        r["lanes"] = [{"width": rng.choice([3.0, 3.5, 4.0]), "maxSpeed": speed} for _ in r["lanes"]]
    for it in net["intersections"]:
        it["roads"] = [r for r in it["roads"] if r not in dropped]
        keep = [i for i, rl in enumerate(it["roadLinks"]) if rl["startRoad"] not in dropped and rl["endRoad"] not in dropped]
        remap = {old: new for new, old in enumerate(keep)}
        it["roadLinks"] = [it["roadLinks"][i] for i in keep]
        for rl in it["roadLinks"]:
            for ll in rl["laneLinks"]:
                ll.pop("points", None)  # the engine generates the curve (roadnet.cpp:212-247)
        tl = it["trafficLight"]
        tl["roadLinkIndices"] = list(range(len(keep)))
        for ph in tl["lightphases"]:
            ph["availableRoadLinks"] = [remap[i] for i in ph["availableRoadLinks"] if i in remap]
            ph["time"] = rng.choice([5, 10, 20, 30]) if ph["time"] == 30 else rng.choice([3, 5])
    d = os.path.join(workdir, "irregular_%d%s" % (seed, "" if bends else "_straight"))
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "roadnet.json"), "w") as f:
        json.dump(net, f)
    # random-walk routes over the surviving roadLinks, random vehicle templates
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow.json"), 160, seed=seed, interval=4.0,
                            min_len=3, max_len=8)
    flows = json.load(open(flow))
    for f in flows:
        f["interval"] = rng.choice([2.0, 3.0, 5.0])
        f["startTime"] = rng.choice([0, 0, 10])
        f["vehicle"] = {"length": rng.choice([4.0, 5.0, 6.5]), "width": 2.0, "maxPosAcc": rng.choice([2.0, 3.0]),
                        "maxNegAcc": rng.choice([4.0, 4.5, 6.0]), "usualPosAcc": rng.choice([1.5, 2.0]),
                        "usualNegAcc": rng.choice([2.5, 3.5, 4.5]), "minGap": rng.choice([2.0, 2.5, 3.0]),
                        "maxSpeed": rng.choice([9.0, 13.0, 16.67, 20.0]), "headwayTime": rng.choice([1.0, 1.5, 2.0])}
    with open(flow, "w") as f:
        json.dump(flows, f)
    cfg = {"interval": 1.0, "seed": seed, "dir": d + "/", "roadnetFile": "roadnet.json", "flowFile": "flow.json",
           "rlTrafficLight": False, "laneChange": False, "saveReplay": False}
    path = os.path.join(d, "config.json")
    with open(path, "w") as f:
        json.dump(cfg, f)
    return path


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_irregular_reference_vs_twin(mod, scen, workdir, ref_module, seed):
    cfg = irregular(scen, workdir, seed)
    ref = ref_module.Engine(cfg, 1)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    finished = 0
    for s in range(500):
        ref.next_step()
        tw.next_step()
        if s % 10 == 9:
            assert checkpoint_record(tw) == checkpoint_record(ref), "seed %d step %d" % (seed, s + 1)
    finished = tw._scalars()["finished_vehicle_count"]
    assert tw.get_vehicle_count() > 150 and finished > 50
    assert ref.get_average_travel_time() == tw.get_average_travel_time()
    time.sleep(0.2)  # reference destructor race
    del ref


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_irregular_hip_vs_twin(mod, scen, workdir, seed):
    cfg = irregular(scen, workdir, seed)
    hip = mod.Engine(cfg, 1)
    tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(500):
        hip.next_step()
        tw.next_step()
        if s % 25 == 24:
            assert_same_state(hip, tw, "seed %d step %d" % (seed, s + 1))
    assert hip.get_vehicle_count() > 150


@pytest.mark.gpu
def test_irregular_tiled_hip(mod, scen, workdir):
    """Tiling by coordinate blocks on a jittered network (cut roads are no longer axis-aligned or equally long)."""
    cfg = irregular(scen, workdir, 21, n=6)
    single = mod.Engine(cfg, 1)
    tiled = mod.TiledEngine(cfg, 2, 2)
    tiled.enable_mailboxes("irr_%d" % os.getpid())
    for s in range(400):
        single.next_step()
        tiled.next_step()
        if s % 20 == 19:
            assert np.array_equal(single.get_lane_vehicle_count_array(), tiled.get_lane_vehicle_count_array()), s
    a, b = single._vehicle_state(), tiled._vehicle_state()
    for k in ("vid", "drivable", "dis", "speed", "leader", "blocker"):
        assert np.array_equal(a[k], b[k]), k
