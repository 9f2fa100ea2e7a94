"""Developer tool: A/B of implementation choices (config "cfx") with the measurement rounds INTERLEAVED — every engine is
built once, then round after round each engine in turn reloads the same Archive, takes a few instrumented steps and a timed
run — so clock and thermal drift of the box hit every choice alike.  Prints the median per-kernel time and wall time per step.
After the last round every engine has taken the same steps from the same Archive: their states are compared field by field with the
first engine's (a choice that changes a result shows here before any pinned test is asked).
usage: python tools/ab_bench.py [scenario] [rounds=N] 'layout=ring' 'layout=ring,ringLanesPerWave=40000' 'layout=dense,lib=gpurun_exp/libx.so' ..."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
scenario, rounds = "grid_30x30", 6
if args and "=" not in args[0]:
    scenario = args.pop(0)
if args and args[0].startswith("rounds="):
    rounds = int(args.pop(0).split("=")[1])
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=scenario, n_extra=int(os.environ.get("CFX_EXP_EXTRA", bench.N_EXTRA_FLOWS)))
if os.environ.get("CFX_AB_LANE_CHANGE"):  # the lane-change step (the state is built with lane change too)
    _c = json.load(open(cfg)); _c["laneChange"] = True
    cfg = cfg.replace(".json", "_lc.json"); json.dump(_c, open(cfg, "w"))
base = _cityflow.Engine(cfg, 1)
for _ in range(int(os.environ.get("CFX_EXP_BUILD", 300))):
    base.next_step()
arch = base.snapshot()
running = base.get_vehicle_count()
del base
engines = []
for i, spec in enumerate(args):
    cfx, lib = {}, None
    for kv in spec.split(","):
        k, v = kv.split("=")
        if k == "lib":  # a differently built device library (tools/README.md)
            lib = os.path.abspath(v)
        else:
            cfx[k] = int(v) if v.lstrip("-").isdigit() else {"true": True, "false": False}.get(v, v)
    c = json.load(open(cfg)); c["cfx"] = cfx
    path = cfg.replace(".json", "_ab%d.json" % i)
    json.dump(c, open(path, "w"))
    engines.append((spec, _cityflow.Engine._with_backend(path, 1, lib) if lib else _cityflow.Engine(path, 1), {}, []))
steps_timed = int(os.environ.get("CFX_AB_STEPS", 200))
for r in range(rounds):
    for spec, eng, res, wall in engines:
        eng.load(arch)
        for _ in range(12):  # (lets the engine's adaptive choices settle)
            eng.next_step()
        eng.sync()
        eng._profile_enable(True)
        for _ in range(10):
            eng.next_step()
        prof = eng._profile_read()
        eng._profile_enable(False)
        steps = max(n for _ms, n in prof.values())
        for k, (ms, n) in prof.items():
            if n:
                res.setdefault(k, []).append(ms / steps * 1e3)  # per STEP (a kernel that runs every other step counts half)
        eng.load(arch)
        for _ in range(12):
            eng.next_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps_timed):
            eng.next_step()
        eng.sync()
        wall.append((time.perf_counter() - t0) / steps_timed * 1e6)
print("scenario", scenario, "running", running, "rounds", rounds)
sys.path.insert(0, os.path.join(ROOT, "tests"))
try:
    from conftest import assert_same_state
    for spec, eng, _res, _wall in engines[1:]:
        assert_same_state(engines[0][1], eng, "%s vs %s" % (engines[0][0], spec))
    print("states equal after the last round:", len(engines), "engines")
except AssertionError as ex:
    print("STATE MISMATCH:", ex)
for spec, eng, res, wall in engines:
    sc = eng._scalars()
    print("   diag", spec, {k: v for k, v in sc.items() if k.startswith("diag_")}, "running", sc["active_vehicle_count"])
    med = {k: round(statistics.median(v), 2) for k, v in res.items()}
    print("%-46s %s sum %.1f | wall us/step median %.1f min %.1f" % (spec, med, sum(med.values()), statistics.median(wall), min(wall)), flush=True)
