import sys, json
sys.path.insert(0, '.')
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from cityflow_amd import _cityflow
name, steps = sys.argv[1], int(sys.argv[2])
cfg = bench.build_workload("/tmp/cfa_lcdbg", 0, scenario=name, n_extra=bench.N_EXTRA_FLOWS if name == "grid_30x30" else 0)
c = json.load(open(cfg)); c["laneChange"] = True
path = cfg.replace(".json", "_lc.json"); json.dump(c, open(path, "w"))
eng = _cityflow.Engine(path, 1)
for s in range(steps):
    eng.next_step()
    if s % 25 == 24:
        eng.sync()
        print(s + 1, eng._scalars()["active_vehicle_count"], flush=True)
print("done", flush=True)
