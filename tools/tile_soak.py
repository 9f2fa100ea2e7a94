"""Developer soak (GPU): the bench workload (30x30 + 3000 seeded flows, ~97 k vehicles) cut into rows x cols ring tiles against
the single engine for N steps, with getters BETWEEN steps at irregular intervals (so that the import runs both inside the next
admission and as a kernel of its own), resets and Archive loads.  usage: python tools/tile_soak.py ROWS COLS STEPS"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows, cols, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow as m
cfg = bench.build_workload("/tmp/cfa_tsoak", 0)
single = m.Engine(cfg, 1)
tiled = m.TiledEngine(cfg, rows, cols, [], "")
tiled.enable_mailboxes("tsoak_%d" % os.getpid())
rng = np.random.default_rng(7)
keys = ["vid", "drivable", "prev_drivable", "leader", "blocker", "enter_ll_time", "route_pos", "dis", "speed"]
t0 = time.time()
checks = 0
snap = None
for s in range(steps):
    single.next_step(); tiled.next_step()
    r = rng.integers(0, 100)
    if r < 12:  # the agent's observation (count getter: settles a pending import as a kernel)
        a, b = single.get_lane_vehicle_count_array(), tiled.get_lane_vehicle_count_array()
        assert np.array_equal(a, b), "step %d: lane counts differ on %d lanes" % (s, int((a != b).sum()))
        checks += 1
    if r == 50 or s == steps - 1:
        va, vb = single._vehicle_state(), tiled._vehicle_state()
        for k in keys:
            assert np.array_equal(va[k], vb[k]), "step %d: %s differs" % (s, k)
        sa, sb = single._scalars(), tiled._scalars()
        for k in ("active_vehicle_count", "finished_vehicle_count", "vehicle_steps", "cumulative_travel_time"):
            assert sa[k] == sb[k], (s, k, sa[k], sb[k])
        checks += 1
    if s == steps // 3:
        snap = single.snapshot()
    if s == 2 * steps // 3 and snap is not None:  # back to an earlier state, on both
        single.load(snap); tiled.load(snap)
# Lane::history (kept by ring tiles like one engine; the getters above made its record come by both ways: with the next action
# launch and as a launch of its own): the two Archives' drivables, every record of every lane
import json
single.snapshot().dump("/tmp/cfa_tsoak_one.json")
tiled.snapshot().dump("/tmp/cfa_tsoak_tiles.json")
da, db = json.load(open("/tmp/cfa_tsoak_one.json"))["drivables"], json.load(open("/tmp/cfa_tsoak_tiles.json"))["drivables"]
assert da == db, "Lane::history differs on %d drivables" % sum(da[k] != db.get(k) for k in da)
assert not tiled._keeps_lane_history() or sum(len(v.get("history", [])) for v in db.values()) > 0
print("tiles %dx%d (%s): %d steps, %d comparisons, %d running vehicles, %.0f s: equal (Lane::history of %d lanes included)" % (
    rows, cols, tiled._layout(), steps, checks, single.get_vehicle_count(), time.time() - t0, sum("history" in v for v in db.values())))
