"""Developer tool: per-kernel times of ONE tile of the bench workload cut rows x cols (all tiles in this process).
usage: python tools/tile_kernels.py ROWS COLS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows, cols = int(sys.argv[1]), int(sys.argv[2])
scale = len(sys.argv) > 3 and sys.argv[3] == "scale"  # the 100x100 / 1 M-vehicle workload of bench.py's scale leg
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow as m
if scale:
    cfg = bench.build_workload("/tmp/cfa_tilek_scale", 0, scenario="gen_%dx%d" % (bench.SCALE_GRID, bench.SCALE_GRID),
                               n_extra=bench.SCALE_FLOWS)
else:
    cfg = bench.build_workload("/tmp/cfa_tilek", 0)
eng = m.TiledEngine(cfg, rows, cols, [], "")
eng.enable_mailboxes("tilek_%d" % os.getpid())
for _ in range(320):
    eng.next_step()
eng.sync()
tiles = sorted({0, (rows * cols) // 2 + (1 if cols > 2 else 0)} & set(range(rows * cols)))
for t in tiles:
    eng._profile_enable(t, True)
for _ in range(10):
    eng.next_step()
for t in tiles:
    eng._profile_read(t)
import time
t0 = time.perf_counter()
for _ in range(100):
    eng.next_step()
eng.sync()
dt = time.perf_counter() - t0
for t in tiles:
    prof = eng._profile_read(t)
    eng._profile_enable(t, False)
    n = max(c for _ms, c in prof.values())
    out = {k: round(ms / n * 1e3, 2) for k, (ms, c) in prof.items() if c}
    print("tiles %dx%d%s, tile %d:" % (rows, cols, ", 100x100 scale workload" if scale else "", t), out,
          "sum %.1f us per tile-step" % sum(out.values()), flush=True)
print("all %d tiles in this process on one GPU: %.1f us per step (instrumented), running vehicles %d" %
      (rows * cols, dt / 100 * 1e6, eng._scalars()["active_vehicle_count"]), flush=True)
