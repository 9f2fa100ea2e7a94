"""Developer check: a long run (default 50 000 steps of the 6x6 grid) — step time and host memory must stay flat, the
device tables must keep growing correctly (vehicle ids are never reused), counts must stay consistent."""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cityflow_amd import _cityflow as m, scenarios  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
eng = m.Engine(scenarios.materialize("grid_6x6", "/tmp/cfa_long"), 1)
t0 = time.time()
last = t0
for s in range(steps):
    eng.next_step()
    if s % 10000 == 9999:
        sc = eng._scalars()
        now = time.time()
        assert sc["spawned_vehicle_count"] == sc["active_vehicle_count"] + sc["finished_vehicle_count"] + len(eng._waiting()[0])
        print("step %6d: %.1f us/step over the last 10k, %d spawned, %d running, %d finished, rss %.0f MB" % (
            s + 1, (now - last) / 10000 * 1e6, sc["spawned_vehicle_count"], sc["active_vehicle_count"],
            sc["finished_vehicle_count"], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
        last = now
print("avg travel time %.2f" % eng.get_average_travel_time())
