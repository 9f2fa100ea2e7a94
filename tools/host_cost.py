"""Developer tool: host-side cost of a step (launch submission) — tiny network, so the device is never the bottleneck."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cityflow_amd import _cityflow as m, scenarios
cfg = scenarios.materialize("grid_6x6", "/tmp/cfa_hc")
def rate(eng, n=3000):
    for _ in range(200): eng.next_step()
    eng.sync(); t0 = time.perf_counter()
    for _ in range(n): eng.next_step()
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("single engine: host %.1f us/step enqueue, %.1f us/step incl. drain" % rate(m.Engine(cfg, 1)))
for kind in ("device", "host"):
    t = m.TiledEngine(cfg, 1, 2)
    (t.enable_device_mailboxes if kind == "device" else t.enable_mailboxes)("hc_%s_%d" % (kind, os.getpid()))
    a, b = rate(t)
    hs = t._host_seconds()
    print("tiled 1x2 in one process (%s mailboxes): host %.1f us/step enqueue, %.1f incl. drain; spawner/submit s %s" % (kind, a, b, hs))
