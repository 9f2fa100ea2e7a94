"""Developer probe (GPU): a long free run of ring tiles that forget their finished vehicles — a 6x6 grid whose demand the network
carries (flow interval 12 s), cut rows x cols, `"cfx": {"compactVehicles": N}` — with the vehicle numbers held, the compactions,
device memory and host RSS sampled along the way, and the tiles compared with ONE engine that never forgets (every visible
number, Lane::history included) at the end.  usage: python tools/tile_long_run.py ROWS COLS STEPS [N]"""
import json, os, resource, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows, cols, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
from cityflow_amd import _cityflow as m, scenarios
wd = "/tmp/cfa_tile_long"
os.makedirs(wd, exist_ok=True)
base = scenarios.generate_grid(6, 6, wd, flow_interval=12.0)


def variant(tag, **cfx):
    c = json.load(open(base))
    c["cfx"] = cfx
    p = base.replace(".json", "_%s.json" % tag)
    json.dump(c, open(p, "w"))
    return p


til = m.TiledEngine(variant("tiles", compactVehicles=every), rows, cols)
til.enable_mailboxes("tlong_%d" % os.getpid())
one = m.Engine(variant("one", compactVehicles=0), 1)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
t0 = time.time()
for s in range(steps):
    til.next_step()
    one.next_step()
    if s % (steps // 10) == steps // 10 - 1:
        til.sync()
        held, times = til._vehicle_table()
        print("step %7d: tiles hold %6d vehicle numbers (%d compactions), the engine that never forgets %7d; alive %5d; device free %d MB; "
              "host RSS %d MB; %.1f us per step (both engines)" % (s + 1, held, times, one._vehicle_table()[0], len(one.get_vehicles(True)),
                                                                 one._device_memory()[0] >> 20, rss(), (time.time() - t0) / (s + 1) * 1e6), flush=True)
assert til.get_lane_vehicle_count() == one.get_lane_vehicle_count()
assert til.get_vehicle_speed() == one.get_vehicle_speed() and til.get_vehicle_distance() == one.get_vehicle_distance()
assert til.get_average_travel_time() == one.get_average_travel_time() and til.get_vehicles(True) == one.get_vehicles(True)
til.snapshot().dump(wd + "/t.json")
one.snapshot().dump(wd + "/o.json")
a, b = json.load(open(wd + "/t.json")), json.load(open(wd + "/o.json"))
assert a["drivables"] == b["drivables"], "Lane::history differs"
print("tiles %dx%d == one engine after %d steps: lane counts, speeds, distances, travel time, vehicle lists, Lane::history" % (rows, cols, steps))
