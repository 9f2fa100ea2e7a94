"""Developer probe: the host's own time per next_step() call on the 100x100 / 1 M-vehicle workload (is the free-running step
host-bound?) — median / mean call, us per step, the slowest next_step's parts (Engine._host_stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario="gen_100x100", n_extra=33000)
if os.environ.get("CFX_CFX"):
    cfg = bench.with_config(cfg, "cfx", cfx={k: {"true": True, "false": False}.get(v, v) for k, v in (kv.split("=") for kv in os.environ["CFX_CFX"].split(","))})
e = _cityflow.Engine(cfg, 1)
for _ in range(310):
    e.next_step()
e.sync()
for rep in range(3):
    calls = []
    e._host_stats(True)
    t0 = time.perf_counter()
    t1 = t0
    for _ in range(200):
        e.next_step()
        t2 = time.perf_counter()
        calls.append(t2 - t1)
        t1 = t2
    e.sync()
    dt = time.perf_counter() - t0
    calls.sort()
    hs = e._host_stats(True)
    print("us/step %.1f  median call %.1f  mean call %.1f  p90 %.1f  cfx_step mean %.1f  slowest parts %s" % (
        dt / 200 * 1e6, calls[100] * 1e6, sum(calls) / 200 * 1e6, calls[180] * 1e6, hs["step_call_us_mean"],
        [round(x, 1) for x in hs["slowest_next_step"][2]]), flush=True)
