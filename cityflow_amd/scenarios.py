"""Scenario fixtures for tests and bench.py.

The road networks / flows under tests/golden/scenarios/ were produced by the reference's own generator
and examples (tests/golden/make_scenarios.py); they are stored gzip-compressed and materialised into a
work directory together with a config.json here.  Nothing in this module touches /root/reference.

`dense_flows` adds seeded interior-origin flows on top of a generated grid: the stock generator's demand
saturates far below the vehicle counts BASELINE.json names, so the benchmark scenarios
need more origins.  Routes are random walks over the roadnet's own roadLinks (any road may start a
route: the reference Router only needs >= 2 connected roads).
"""
import gzip
import json
import os
import random
import shutil
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENARIO_DIR = os.path.join(ROOT, "tests", "golden", "scenarios")

NAMES = ("example_1x1", "grid_6x6", "grid_30x30")

# vehicle template of the reference generator (tools/generator/generate_grid_scenario.py:16-24 defaults)
GRID_VEHICLE = {
    "length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0, "usualNegAcc": 4.5,
    "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5,
}


def _gunzip(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as out:
        shutil.copyfileobj(f, out)


def materialize(name, workdir=None, flow_file=None, **config):
    """Unpack scenario `name` into `workdir/<name>/` and write a config.json; returns the config path.

    Keyword arguments override config keys (interval, seed, rlTrafficLight, laneChange, saveReplay).
    `flow_file`, if given, is a path to a flow JSON used instead of the scenario's own.
    """
    if name not in NAMES:
        raise ValueError("unknown scenario %r (have %s)" % (name, ", ".join(NAMES)))
    if workdir is None:
        workdir = tempfile.mkdtemp(prefix="cityflow_amd_")
    d = os.path.join(workdir, name)
    os.makedirs(d, exist_ok=True)
    roadnet = os.path.join(d, "roadnet.json")
    flow = os.path.join(d, "flow.json")
    if not os.path.exists(roadnet):
        _gunzip(os.path.join(SCENARIO_DIR, name, "roadnet.json.gz"), roadnet)
    if not os.path.exists(flow):
        _gunzip(os.path.join(SCENARIO_DIR, name, "flow.json.gz"), flow)
    flow_name = "flow.json"
    if flow_file is not None:
        flow_name = os.path.basename(flow_file)
        if os.path.abspath(os.path.dirname(flow_file)) != os.path.abspath(d):
            shutil.copyfile(flow_file, os.path.join(d, flow_name))
    cfg = {
        "interval": 1.0, "seed": 0, "dir": d + "/", "roadnetFile": "roadnet.json", "flowFile": flow_name,
        "rlTrafficLight": False, "laneChange": False, "saveReplay": False,
    }
    cfg.update(config)
    tag = "_".join("%s-%s" % (k, config[k]) for k in sorted(config)) if config else "default"
    path = os.path.join(d, "config_%s_%s.json" % (flow_name.replace(".json", ""), tag))
    with open(path, "w") as f:
        json.dump(cfg, f)
    return path


def dense_flows(roadnet_path, out_path, n_extra, seed=12345, interval=2.0, min_len=3, max_len=7, base_flow=None,
                end_time=-1):
    """Write a flow file = `base_flow` (optional, kept first and unchanged) + `n_extra` seeded random-walk flows."""
    with open(roadnet_path) as f:
        net = json.load(f)
    succ = {}
    for inter in net["intersections"]:
        if inter.get("virtual"):
            continue
        for rl in inter["roadLinks"]:
            if rl["laneLinks"]:
                succ.setdefault(rl["startRoad"], [])
                if rl["endRoad"] not in succ[rl["startRoad"]]:
                    succ[rl["startRoad"]].append(rl["endRoad"])
    starts = sorted(succ)
    rng = random.Random(seed)
    flows = []
    if base_flow is not None:
        with open(base_flow) as f:
            flows = json.load(f)
    made = 0
    while made < n_extra:
        route = [rng.choice(starts)]
        want = rng.randint(min_len, max_len)
        while len(route) < want and route[-1] in succ:
            nxt = [r for r in succ[route[-1]] if r not in route]
            if not nxt:
                break
            route.append(rng.choice(nxt))
        if len(route) < 2:
            continue
        flows.append({"vehicle": dict(GRID_VEHICLE), "route": route, "interval": interval,
                      "startTime": 0, "endTime": end_time})
        made += 1
    with open(out_path, "w") as f:
        json.dump(flows, f)
    return out_path
