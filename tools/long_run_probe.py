"""Developer probe (GPU): ONE engine free-running for a long time without reset(): time per 20 000 steps, vehicles created, device
memory used and host resident set — does a step get slower, what does a created vehicle cost the host.
usage: python tools/long_run_probe.py [scenario | unsaturated] [steps] [compactVehicles]
`unsaturated`: a generated 6x6 grid whose demand the network carries (vehicles finish; on the stock 6x6 flows most of them wait
in their lanes' buffers for ever, which the reference keeps too); compactVehicles: the "cfx" key (0 never, default automatic)."""
import json
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
cfg = scen.generate_grid(6, 6, work, flow_interval=12.0) if scenario == "unsaturated" else scen.materialize(scenario, work)
if len(sys.argv) > 3:
    c = json.load(open(cfg))
    c["cfx"] = {"compactVehicles": int(sys.argv[3])}
    cfg = cfg.replace(".json", "_compact.json")
    json.dump(c, open(cfg, "w"))
e = m.Engine(cfg, 1)
free0, r0, c0 = e._device_memory()[0], None, 0
every = max(1, steps // 10)
t = time.perf_counter()
for s in range(steps + 1):
    e.next_step()
    if s % every == 0:
        e.sync()
        sc = e._scalars()
        now = time.perf_counter()
        if r0 is None:
            r0, c0 = rss_mb(), sc["spawned_vehicle_count"]
        print("step %-8d finished %-9d running %-6d numbers held %-8d compactions %-3d %.1f us/step  device +%.1f MB  host RSS %.1f MB" % (
            s, sc["finished_vehicle_count"], sc["active_vehicle_count"], e._vehicle_table()[0], e._vehicle_table()[1],
            (now - t) / every * 1e6, (free0 - e._device_memory()[0]) / 1e6, rss_mb()), flush=True)
        t = time.perf_counter()
