"""Developer probe: per-step wall time right after a load + getter (finds one-off stalls)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario="grid_30x30")
e = _cityflow.Engine(cfg, 1)
for _ in range(300): e.next_step()
dump = "/tmp/cfa_exp/probe_state.json"
e.snapshot().dump(dump)
del e
for pre in (25, 55):
    e = _cityflow.Engine(cfg, 1)
    e.load_from_file(dump)
    for _ in range(pre): e.next_step()
    e.sync()
    s = e._scalars()
    x = e.get_lane_vehicle_count_array()
    slow = []
    t_all = time.perf_counter()
    for i in range(200):
        t0 = time.perf_counter(); e.next_step(); dt = (time.perf_counter() - t0) * 1e6
        if dt > 150: slow.append((i, round(dt)))
    e.sync()
    print("pre", pre, "avg %.1f us/step" % ((time.perf_counter() - t_all) / 200 * 1e6), "host-side slow calls (step, us):", slow[:20], flush=True)
    print("  layout", e._layout(), "ring", e._ring_info(), flush=True)
    del e
