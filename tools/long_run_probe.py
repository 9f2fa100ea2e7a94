"""Developer probe (GPU): ONE engine free-running for a long time without reset(): time per 20 000 steps, vehicles created, device
memory used and host resident set — does a step get slower, what does a created vehicle cost the host.
usage: python tools/long_run_probe.py [scenario] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scenario = sys.argv[1] if len(sys.argv) > 1 else "grid_6x6"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
from cityflow_amd import _cityflow as m, scenarios as scen


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0


work = "/tmp/cfa_long_run"
os.makedirs(work, exist_ok=True)
e = m.Engine(scen.materialize(scenario, work), 1)
free0, r0, c0 = e._device_memory()[0], None, 0
t = time.perf_counter()
for s in range(steps + 1):
    e.next_step()
    if s % 20000 == 0:
        e.sync()
        sc = e._scalars()
        now = time.perf_counter()
        if r0 is None:
            r0, c0 = rss_mb(), sc["spawned_vehicle_count"]
        print("step %-8d created %-9d running %-6d  %.1f us/step  device +%.1f MB  host RSS %.1f MB (%.0f B per vehicle created)" % (
            s, sc["spawned_vehicle_count"], sc["active_vehicle_count"], (now - t) / 20000 * 1e6, (free0 - e._device_memory()[0]) / 1e6,
            rss_mb(), (rss_mb() - r0) * 1048576 / max(1, sc["spawned_vehicle_count"] - c0)), flush=True)
        t = time.perf_counter()
