"""The lane-change step alone on the bench.py workload (30x30, laneChange true): 300 steps of build-up, 200 timed.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table committed as profiles/r03_kernel_trace_lane_change_30x30.txt
(summarised by tools/rocpd_summary.py <db> 100)."""
import sys, json, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_lconly", 0, scenario="grid_30x30")
c = json.load(open(cfg)); c["laneChange"] = True
path = cfg.replace(".json", "_lc.json"); json.dump(c, open(path, "w"))
eng = _cityflow.Engine(path, 1)
for _ in range(300): eng.next_step()
eng.sync()
t0 = time.perf_counter()
for _ in range(200): eng.next_step()
eng.sync()
print("us/step", (time.perf_counter() - t0) / 200 * 1e6)
