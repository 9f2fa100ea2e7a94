"""Checker (developer tool, test infrastructure): lane change of the CPU twin against the unmodified reference.

With laneChange=true the reference walks its lane-change candidates in `std::set<Vehicle*>` order, i.e. by heap address
(SURVEY.md App. C-6), so its results depend on everything the process has allocated so far — even on the getters that
were called between steps, or on another engine living in the same process.  The C ABI fixes the order to creation order
(include/cityflow_amd.h "Lane change"), which is address order when addresses only grow.  So each reference run happens in
its own process under oracle/_ref/libmonotonic_new.so (LD_PRELOAD: `new Vehicle` gets ascending addresses, nothing else
about the reference changes), steps H times and reports its state; the twin does the same.  Both must agree exactly.

usage: python tests/tools/lane_change_parity.py <scenario> H [H ...]     (scenario: example_1x1 | grid_6x6 | gen_RxC)
       python tests/tools/lane_change_parity.py --dump ref|twin <config.json> H      (internal: one run, JSON on stdout)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def state(eng):
    return {"count": eng.get_vehicle_count(), "speed": eng.get_vehicle_speed(), "distance": eng.get_vehicle_distance(),
            "lane_vehicles": eng.get_lane_vehicles(), "vehicles": eng.get_vehicles(),
            "average_travel_time": eng.get_average_travel_time(), "lane_count": eng.get_lane_vehicle_count()}


def dump(which, cfg, steps, load=None, save=None):
    """load: an Archive JSON to start from (load_from_file); save: where to Archive.dump the state after `steps` steps"""
    if which == "ref":
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
        import cityflow_ref
        eng = cityflow_ref.Engine(cfg, 1)
    else:
        from conftest import TWIN_LIB
        from cityflow_amd import _cityflow
        eng = _cityflow.Engine._with_backend(cfg, 1, TWIN_LIB)
    if load:
        eng.load_from_file(load)
    script = json.loads(os.environ.get("CFX_LC_SCRIPT", "{}"))  # {"<step>": [[method, args...], ...]}: control calls before a step
    for s in range(steps):
        for call in script.get(str(s), []):
            if call[0] == "slow_changing":  # set_vehicle_speed on the real vehicles of the first changing pairs
                lanes = eng.get_lane_vehicles()
                ids = sorted(v[:-len("_shadow")] for lane in lanes.values() for v in lane if v.endswith("_shadow"))[:call[1]]
                for vid in ids:
                    eng.set_vehicle_speed(vid, call[2])
            else:
                getattr(eng, call[0])(*call[1:])
        eng.next_step()
    if save:
        eng.snapshot().dump(save)
    print(json.dumps(state(eng)))
    time.sleep(0.1)  # the reference's worker threads must be parked before the engine goes away


def reference_env():
    """LD_PRELOAD that gives the reference's Vehicle objects ascending addresses (oracle/monotonic_new.cpp)."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    return {"LD_PRELOAD": os.path.join(ref_dir, "libmonotonic_new.so"),
            "CFX_VEHICLE_SIZE": open(os.path.join(ref_dir, "vehicle_size.txt")).read().strip()}


def run(which, cfg, steps, env=None, load=None, save=None):
    if which == "ref" and env is None:
        env = reference_env()
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", which, cfg, str(steps), load or "-", save or "-"],
                         capture_output=True,
                         text=True, timeout=1800, env=dict(os.environ, **(env or {})))
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def lane_change_config(scenario, workdir):
    from cityflow_amd import scenarios
    if scenario.startswith("gen_"):  # generated RxC grid of the bench workloads
        import bench
        cfg = bench.build_workload(workdir, 0, scenario=scenario, n_extra=0)
        c = json.load(open(cfg))
        c["laneChange"] = True
        path = cfg.replace(".json", "_lanechange.json")
        with open(path, "w") as f:
            json.dump(c, f)
        return path
    return scenarios.materialize(scenario, workdir, laneChange=True)


def compare(a, b):
    """-> list of the keys that differ"""
    return [k for k in a if a[k] != b[k]]


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        opt = [None if a == "-" else a for a in sys.argv[5:7]] + [None, None]
        dump(sys.argv[2], sys.argv[3], int(sys.argv[4]), load=opt[0], save=opt[1])
        sys.exit(0)
    cfg = lane_change_config(sys.argv[1], "/tmp/cfa_lc_parity")
    for H in [int(x) for x in sys.argv[2:]]:
        r, t = run("ref", cfg, H), run("twin", cfg, H)
        shadows = r["count"] - len(r["speed"])
        diff = compare(r, t)
        print(json.dumps({"steps": H, "identical": not diff, "differs": diff, "running": r["count"], "shadows_now": shadows}), flush=True)
