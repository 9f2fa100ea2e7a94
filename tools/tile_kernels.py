"""Developer tool: per-kernel times of ONE tile of the bench workload cut rows x cols (all tiles in this process).
usage: python tools/tile_kernels.py ROWS COLS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows, cols = int(sys.argv[1]), int(sys.argv[2])
scale = len(sys.argv) > 3 and sys.argv[3] == "scale"  # the 100x100 / 1 M-vehicle workload of bench.py's scale leg
weak = len(sys.argv) > 3 and sys.argv[3] == "weak"    # every tile a 100x100 network of its own size (1 M vehicles per tile);
#                                                       the single engine on the whole of it is timed first, for comparison
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow as m
if weak:
    import time
    cfg = bench.build_tiled_workload("/tmp/cfa_tilek_weak", rows, cols, bench.SCALE_GRID, bench.SCALE_FLOWS)
    one = m.Engine(cfg, 1)
    for _ in range(320):
        one.next_step()
    one.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        one.next_step()
    one.sync()
    dt = time.perf_counter() - t0
    one._profile_enable(True)
    for _ in range(50):
        one.next_step()
    prof = one._profile_read()
    one._profile_enable(False)
    print("single engine on the whole %dx%d network: %.1f us per step, kernels %.1f us per step %s, running vehicles %d" % (
        bench.SCALE_GRID * rows, bench.SCALE_GRID * cols, dt / 100 * 1e6, sum(ms for ms, n in prof.values()) / 50 * 1e3,
        {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in prof.items() if n}, one._scalars()["active_vehicle_count"]), flush=True)
    del one
elif scale:
    cfg = bench.build_workload("/tmp/cfa_tilek_scale", 0, scenario="gen_%dx%d" % (bench.SCALE_GRID, bench.SCALE_GRID),
                               n_extra=bench.SCALE_FLOWS)
else:
    cfg = bench.build_workload("/tmp/cfa_tilek", 0)
if os.environ.get("CFX_TILE_CFX"):  # implementation choices, "key=value,key=value".  debugSync=1: every kernel of every tile runs
    # ALONE on the GPU (a host synchronisation behind each launch; the tiles are stepped one after the other), so the times
    # below are a tile's own, as on a GPU of its own — without it the tiles' kernels share the device
    def _val(v):
        return {"true": True, "false": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    cfg = bench.with_config(cfg, "cfx", cfx={k: _val(v) for k, v in (kv.split("=") for kv in os.environ["CFX_TILE_CFX"].split(","))})
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
    print("tiles %dx%d%s, tile %d:" % (rows, cols, ", 100x100 scale workload" if scale else (", a 100x100 network per tile" if weak else ""), t), out,
          "sum %.1f us per tile-step" % sum(out.values()), flush=True)
print("all %d tiles in this process on one GPU: %.1f us per step (instrumented), running vehicles %d" %
      (rows * cols, dt / 100 * 1e6, eng._scalars()["active_vehicle_count"]), flush=True)
