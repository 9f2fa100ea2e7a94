"""Developer soak (GPU): a long run of ONE engine without reset() under an agent's calls — signals set, lane counts read, the
reference's dict getters now and then, snapshots and loads — against the CPU twin taking the same calls, with the device's
free memory and the process's resident set noted along the way (tables grow with the vehicles CREATED
since the last reset, 46 B per vehicle on the device).  usage: python tools/long_soak.py [scenario] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scenario = sys.argv[1] if len(sys.argv) > 1 else "grid_6x6"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sys.argv = [sys.argv[0]]
from cityflow_amd import _cityflow as m, scenarios as scen

work = "/tmp/cfa_long_soak"
os.makedirs(work, exist_ok=True)
cfg = scen.materialize(scenario, work, rlTrafficLight=True)
hip = m.Engine(cfg, 1)
assert hip.backend_name() == "hip-gfx950"
twin = m.Engine._with_backend(cfg, 1, os.path.join(ROOT, "oracle", "_ref", "libcfx_twin.so"))
n_inter = len(hip.intersection_ids())
rng = np.random.default_rng(11)
keys = ["vid", "drivable", "prev_drivable", "leader", "blocker", "enter_ll_time", "route_pos", "dis", "speed"]


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return 0.0


def both(f):
    f(hip)
    f(twin)


free0 = hip._device_memory()[0]
t0 = time.time()
phases = np.zeros(n_inter, dtype=np.int32)
snap = None
checks = 0
print("step      created   running   device MB used   host RSS MB   (both engines in this process)", flush=True)
for s in range(steps):
    if s % 10 == 0:
        phases = rng.integers(0, 4, n_inter).astype(np.int32)
    both(lambda e: e.set_tl_phases(phases))
    both(lambda e: e.next_step())
    r = int(rng.integers(0, 1000))
    if r < 300:
        a, b = hip.get_lane_vehicle_count_array(), twin.get_lane_vehicle_count_array()
        assert np.array_equal(a, b), "step %d: lane counts" % s
        checks += 1
    if r == 500:
        assert hip.get_lane_vehicle_count() == twin.get_lane_vehicle_count() and hip.get_vehicle_speed() == twin.get_vehicle_speed(), s
        checks += 1
    if r == 501 or s == steps - 1:
        va, vb = hip._vehicle_state(), twin._vehicle_state()
        for k in keys:
            assert np.array_equal(va[k], vb[k]), "step %d: %s differs" % (s, k)
        assert hip.get_average_travel_time() == twin.get_average_travel_time(), s
        ha, hb = hip._lane_history(), twin._lane_history()
        for k in ha:
            assert np.array_equal(ha[k], hb[k]), "step %d: lane history %s" % (s, k)
        checks += 1
    if r == 502:
        snap = twin.snapshot()
    if r == 503 and snap is not None:
        both(lambda e: e.load(snap))
        snap = None
    if s % (steps // 10) == 0 or s == steps - 1:
        sc = hip._scalars()
        print("%-9d %-9d %-9d %-16.1f %-13.1f" % (s, sc["spawned_vehicle_count"], sc["active_vehicle_count"],
                                                  (free0 - hip._device_memory()[0]) / 1e6, rss_mb()), flush=True)
print("%s: %d steps, %d comparisons with the twin, %.0f s: equal" % (scenario, steps, checks, time.time() - t0))
