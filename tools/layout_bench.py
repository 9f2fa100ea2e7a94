"""Developer tool: per-kernel times of the engine's implementation choices (config "cfx": layout, crossMode,
ringLanesPerWave) from the SAME warm state, transferred through an in-memory Archive.
usage: python tools/layout_bench.py [scenario] 'layout=ring,ringLanesPerWave=1' 'layout=dense' ..."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
scenario = "grid_30x30"
if args and "=" not in args[0]:
    scenario = args.pop(0)
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=scenario, n_extra=int(os.environ.get("CFX_EXP_EXTRA", bench.N_EXTRA_FLOWS)))
base = _cityflow.Engine(cfg, 1)
for _ in range(int(os.environ.get("CFX_EXP_BUILD", 300))):
    base.next_step()
arch = base.snapshot()
del base
for i, spec in enumerate(args or ["layout=dense", "layout=ring"]):
    cfx = {}
    for kv in spec.split(","):
        k, v = kv.split("=")
        cfx[k] = int(v) if v.lstrip("-").isdigit() else v
    c = json.load(open(cfg)); c["cfx"] = cfx
    path = cfg.replace(".json", "_exp%d.json" % i)
    json.dump(c, open(path, "w"))
    eng = _cityflow.Engine(path, 1)
    res = {}
    for rep in range(3):
        eng.load(arch)
        eng.next_step(); eng.next_step()
        eng.sync()
        eng._profile_enable(True)
        for _ in range(8):
            eng.next_step()
        prof = eng._profile_read()
        eng._profile_enable(False)
        for k, (ms, n) in prof.items():
            if n:
                res.setdefault(k, []).append(ms / n * 1e3)
    wall = []
    for rep in range(3):
        eng.load(arch)
        for _ in range(5):
            eng.next_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.next_step()
        eng.sync()
        wall.append((time.perf_counter() - t0) / 200 * 1e6)
    print(spec, {k: round(min(v), 1) for k, v in res.items()}, "running", eng.get_vehicle_count(),
          "wall us/step", [round(w, 1) for w in wall], flush=True)
    del eng
