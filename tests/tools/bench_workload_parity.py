"""Developer check of the exact workload bench.py runs at N GPUs (generated (30*rows)x(30*cols) grid + dense flows), with all
tiles on ONE GPU: tiled (mailbox halo) vs single engine, lane counts every `every` steps and the full vehicle state at the
end.   python tests/tools/bench_workload_parity.py ROWS COLS [STEPS] [EVERY]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rows, cols = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
every = int(sys.argv[4]) if len(sys.argv) > 4 else 20
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from cityflow_amd import _cityflow as m  # noqa: E402

t0 = time.time()
cfg = bench.build_tiled_workload(tempfile.mkdtemp(prefix="bwp_"), rows, cols, 30, bench.N_EXTRA_FLOWS)
single = m.Engine(cfg, 1)
tiled = m.TiledEngine(cfg, rows, cols)
tiled.enable_mailboxes("bwp_%d" % os.getpid())
print("setup %.0f s" % (time.time() - t0), flush=True)
for s in range(steps):
    single.next_step()
    tiled.next_step()
    if s % every == every - 1:
        a, b = single.get_lane_vehicle_count_array(), tiled.get_lane_vehicle_count_array()
        assert np.array_equal(a, b), "step %d: %d lanes differ" % (s + 1, int((a != b).sum()))
va, vb = single._vehicle_state(), tiled._vehicle_state()
for k in ("vid", "drivable", "dis", "speed", "leader", "blocker", "route_pos", "enter_ll_time"):
    assert np.array_equal(va[k], vb[k]), k
sa, sb = single._scalars(), tiled._scalars()
for k in ("active_vehicle_count", "finished_vehicle_count", "vehicle_steps", "cumulative_travel_time"):
    assert sa[k] == sb[k], k
print("OK: %dx%d tiles, %d steps, %d running vehicles, %.0f s" % (rows, cols, steps, sa["active_vehicle_count"], time.time() - t0))
