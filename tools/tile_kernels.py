"""Developer tool: per-kernel times of ONE tile of the bench workload cut rows x cols (all tiles in this process).
usage: python tools/tile_kernels.py ROWS COLS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows, cols = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow as m
cfg = bench.build_workload("/tmp/cfa_tilek", 0)
eng = m.TiledEngine(cfg, rows, cols, [], "")
eng.enable_mailboxes("tilek_%d" % os.getpid())
for _ in range(320):
    eng.next_step()
eng.sync()
eng._profile_enable(0, True)
for _ in range(10):
    eng.next_step()
eng._profile_read(0)
for _ in range(100):
    eng.next_step()
prof = eng._profile_read(0)
eng._profile_enable(0, False)
n = max(c for _ms, c in prof.values())
out = {k: round(ms / n * 1e3, 2) for k, (ms, c) in prof.items() if c}
print("tiles %dx%d, tile 0:" % (rows, cols), out, "sum %.1f us per tile-step" % sum(out.values()), flush=True)
