import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.argv=[sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_snap", 0, scenario="grid_30x30")
for hist in (True, False):
    c = json.load(open(cfg)); c["cfx"] = {"laneHistory": hist}
    p = cfg.replace(".json", "_h%d.json" % hist); json.dump(c, open(p, "w"))
    e = _cityflow.Engine(p, 1)
    for _ in range(300): e.next_step()
    e.sync()
    for rep in range(3):
        t = time.perf_counter(); a = e.snapshot(); t1 = time.perf_counter() - t
        t = time.perf_counter(); e.load(a); e.sync(); t2 = time.perf_counter() - t
        print("laneHistory=%s: snapshot %.1f ms, load %.1f ms (%d vehicles)" % (hist, t1 * 1e3, t2 * 1e3, e.get_vehicle_count()), flush=True)
