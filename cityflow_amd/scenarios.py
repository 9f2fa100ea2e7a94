"""Scenario fixtures for tests and bench.py.

The road networks / flows under cityflow_amd/data/scenarios/ were produced by the reference's own generator
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

SCENARIO_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scenarios")

NAMES = ("example_1x1", "grid_6x6", "grid_30x30")

# vehicle template of the reference generator (tools/generator/generate_grid_scenario.py:16-24 defaults)
GRID_VEHICLE = {
    "length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0, "usualNegAcc": 4.5,
    "minGap": 2.5, "maxSpeed": 16.67, "headwayTime": 1.5,
}


def _write_json_atomic(path, obj):
    """Write-then-rename, so a reader (or a later run) never sees a half-written file."""
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)


def _gunzip(src, dst):
    tmp = "%s.tmp%d" % (dst, os.getpid())  # (write-then-rename: several ranks may materialize the same scenario at once)
    with gzip.open(src, "rb") as f, open(tmp, "wb") as out:
        shutil.copyfileobj(f, out)
    os.replace(tmp, dst)


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
            tmp = os.path.join(d, "%s.tmp%d" % (flow_name, os.getpid()))
            shutil.copyfile(flow_file, tmp)
            os.replace(tmp, os.path.join(d, flow_name))
    cfg = {
        "interval": 1.0, "seed": 0, "dir": d + "/", "roadnetFile": "roadnet.json", "flowFile": flow_name,
        "rlTrafficLight": False, "laneChange": False, "saveReplay": False,
    }
    cfg.update(config)
    def _val(v):  # nested values ("cfx": {...}) become key=value lists; nothing but [A-Za-z0-9_.=-] reaches the file name
        if isinstance(v, dict):
            return ".".join("%s=%s" % (k, _val(v[k])) for k in sorted(v))
        return "".join(ch if (ch.isalnum() or ch in "._-") else "-" for ch in str(v))

    tag = "_".join("%s-%s" % (k, _val(config[k])) for k in sorted(config)) if config else "default"
    path = os.path.join(d, "config_%s_%s.json" % (flow_name.replace(".json", ""), tag))
    _write_json_atomic(path, cfg)  # (another rank may be reading it: a rewrite must never show a truncated file)
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
    _write_json_atomic(out_path, flows)
    return out_path


# ----------------------------------------------------------------------------------------------- grid generator
# Grids of any size in the format (and with the float arithmetic) of the reference's generator
# (tools/generator/generate_grid_scenario.py + generate_json_from_grid.py; `--tlPlan`, 3 lanes per road,
# 300 m spacing, 30 m intersections are its defaults / the flags tests/golden/make_scenarios.py used).
# tests/test_scenarios.py checks that grid_roadnet(6, 6) and (30, 30) reproduce the committed fixtures value for
# value, so bigger grids (weak-scaling runs, the 100x100 configuration) need no fixture.
_DX = (1, 0, -1, 0)
_DY = (0, 1, 0, -1)


def _unit(road):
    (x0, y0), (x1, y1) = road["_p0"], road["_p1"]
    ex, ey = x1 - x0, y1 - y0
    n = (ex * ex + ey * ey) ** 0.5
    return ex / n, ey / n


def _lane_shift(lane_index, lane_width):
    shift = 0.0
    for _ in range(lane_index):
        shift += lane_width
    return shift + lane_width * .5


def _hermite(road_a, lane_a, road_b, lane_b, width, lane_width, mid_points):
    """laneLink geometry: Hermite spline from the out-point of lane_a to the in-point of lane_b (findPath)."""
    ax, ay = _unit(road_a)
    bx, by = _unit(road_b)
    sa, sb = _lane_shift(lane_a, lane_width), _lane_shift(lane_b, lane_width)
    pxa, pya = road_a["_p1"][0] - ax * width, road_a["_p1"][1] - ay * width
    pxa, pya = pxa + ay * sa, pya - ax * sa
    pxb, pyb = road_b["_p0"][0] + bx * width, road_b["_p0"][1] + by * width
    pxb, pyb = pxb + by * sb, pyb - bx * sb
    tax, tay, tbx, tby = ax * width, ay * width, bx * width, by * width
    pts = []
    for i in range(mid_points + 1):
        t = i / mid_points
        t3, t2 = t * t * t, t * t
        k1, k2, k3, k4 = 2 * t3 - 3 * t2 + 1, t3 - 2 * t2 + t, -2 * t3 + 3 * t2, t3 - t2
        pts.append({"x": k1 * pxa + k2 * tax + k3 * pxb + k4 * tbx, "y": k1 * pya + k2 * tay + k3 * pyb + k4 * tby})
    return pts


def grid_roadnet(rows, cols, distance=300, inter_width=30, lane_width=4, lane_max_speed=16.67, mid_points=10):
    """Roadnet dict of a rows x cols signalised grid (3 lanes per road: left / straight / right; `--tlPlan` phases)."""
    R, C = rows + 2, cols + 2  # with the ring of virtual border intersections
    inner = lambda i, j: 0 < i < R - 1 and 0 < j < C - 1  # noqa: E731
    inside = lambda i, j: 0 <= i < R and 0 <= j < C  # noqa: E731
    xs = [-distance + j * distance for j in range(C)]
    ys = [-distance + i * distance for i in range(R)]
    n_lanes = 3

    road = {}
    for i in range(R):
        for j in range(C):
            for k in range(4):
                ni, nj = i + _DY[k], j + _DX[k]
                if inside(ni, nj) and (inner(i, j) or inner(ni, nj)):
                    road[i, j, k] = {
                        "id": "road_%d_%d_%d" % (j, i, k), "_dir": k, "_p0": (xs[j], ys[i]), "_p1": (xs[nj], ys[ni]),
                        "points": [{"x": xs[j], "y": ys[i]}, {"x": xs[nj], "y": ys[ni]}],
                        "lanes": [{"width": lane_width, "maxSpeed": lane_max_speed} for _ in range(n_lanes)],
                        "startIntersection": "intersection_%d_%d" % (j, i),
                        "endIntersection": "intersection_%d_%d" % (nj, ni),
                    }

    def turn_type(a, b):
        da, db = a["_dir"], b["_dir"]
        if (da + 1) % 4 == db:
            return "turn_left"
        if (db + 1) % 4 == da:
            return "turn_right"
        return "go_straight" if da == db else None

    lanes_of = {"turn_left": (0,), "go_straight": (1,), "turn_right": (2,)}
    inters = []
    for i in range(R):
        for j in range(C):
            if (i in (0, R - 1)) and (j in (0, C - 1)):
                continue  # corners of the border ring are dropped
            width = inter_width if inner(i, j) else 0
            outs = [road[i, j, k] for k in range(4) if (i, j, k) in road]
            ins = [road[i - _DY[k], j - _DX[k], k] for k in range(4) if (i - _DY[k], j - _DX[k], k) in road]
            links = []
            for a in ins:
                for b in outs:
                    t = turn_type(a, b)
                    if t is None:
                        continue
                    lls = [{"startLaneIndex": c, "endLaneIndex": d,
                            "points": _hermite(a, c, b, d, width, lane_width, mid_points)}
                           for c in lanes_of[t] for d in range(n_lanes)]
                    links.append({"type": t, "startRoad": a["id"], "endRoad": b["id"], "direction": a["_dir"],
                                  "laneLinks": lls})
            idx = range(len(links))
            of_type = lambda t: {x for x in idx if links[x]["type"] == t}  # noqa: E731
            of_dir = lambda d: {x for x in idx if links[x]["direction"] == d}  # noqa: E731
            right, left, straight = of_type("turn_right"), of_type("turn_left"), of_type("go_straight")
            we_ew, ns_sn = of_dir(0) | of_dir(2), of_dir(1) | of_dir(3)
            phases = []
            for green in ((we_ew & straight), (we_ew & left), (ns_sn & straight), (ns_sn & left)):
                phases.append({"time": 30, "availableRoadLinks": sorted(green | right)})
                phases.append({"time": 5, "availableRoadLinks": sorted(right)})
            inters.append({
                "id": "intersection_%d_%d" % (j, i), "point": {"x": xs[j], "y": ys[i]}, "width": width,
                "roads": [r["id"] for r in ins + outs], "roadLinks": links,
                "trafficLight": {"roadLinkIndices": list(idx), "lightphases": phases},
                "virtual": not inner(i, j),
            })
    roads = []
    for i in range(R):
        for j in range(C):
            for k in range(4):
                if (i, j, k) in road:
                    roads.append({key: v for key, v in road[i, j, k].items() if not key.startswith("_")})
    return {"intersections": inters, "roads": roads}


def grid_flows(rows, cols, interval=1.0):
    """The generator's own demand: one straight-through flow per row and column and direction (generate_route)."""
    def straight(x, y, direction, n):
        route = []
        for _ in range(n):
            route.append("road_%d_%d_%d" % (x, y, direction))
            x, y = x + _DX[direction], y + _DY[direction]
        return route

    routes = []
    for i in range(1, rows + 1):
        routes.append(straight(0, i, 0, cols + 1))
        routes.append(straight(cols + 1, i, 2, cols + 1))
    for i in range(1, cols + 1):
        routes.append(straight(i, 0, 1, rows + 1))
        routes.append(straight(i, rows + 1, 3, rows + 1))
    return [{"vehicle": dict(GRID_VEHICLE), "route": r, "interval": interval, "startTime": 0, "endTime": -1} for r in routes]


def generate_grid(rows, cols, workdir=None, flow_interval=1.0, **config):
    """Write roadnet / flow / config for a generated rows x cols grid; returns the config path."""
    if workdir is None:
        workdir = tempfile.mkdtemp(prefix="cityflow_amd_")
    d = os.path.join(workdir, "gen_%dx%d" % (rows, cols))
    os.makedirs(d, exist_ok=True)
    roadnet = os.path.join(d, "roadnet.json")
    if not os.path.exists(roadnet):
        _write_json_atomic(roadnet, grid_roadnet(rows, cols))
    _write_json_atomic(os.path.join(d, "flow.json"), grid_flows(rows, cols, flow_interval))
    cfg = {"interval": 1.0, "seed": 0, "dir": d + "/", "roadnetFile": "roadnet.json", "flowFile": "flow.json",
           "rlTrafficLight": False, "laneChange": False, "saveReplay": False}
    cfg.update(config)
    def _val(v):  # nested values ("cfx": {...}) become key=value lists; nothing but [A-Za-z0-9_.=-] reaches the file name
        if isinstance(v, dict):
            return ".".join("%s=%s" % (k, _val(v[k])) for k in sorted(v))
        return "".join(ch if (ch.isalnum() or ch in "._-") else "-" for ch in str(v))

    tag = "_".join("%s-%s" % (k, _val(config[k])) for k in sorted(config)) if config else "default"
    path = os.path.join(d, "config_flow_%s.json" % tag)
    _write_json_atomic(path, cfg)
    return path
