import sys, os, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench
from cityflow_amd import _cityflow as m
base = bench.build_workload("/tmp/cfa_dev", 0, scenario="grid_30x30")
c = json.load(open(base)); c["cfx"] = {"debugSync": True}
cfg = base.replace(".json", "_dbg.json"); json.dump(c, open(cfg, "w"))
hip = m.Engine(cfg, 1)
for s in range(330):
    hip.next_step()
    if s % 10 == 9:
        st = hip._vehicle_state()
        print("step", s + 1, "vehicles", len(st["vid"]), file=sys.stderr, flush=True)
print("done")
