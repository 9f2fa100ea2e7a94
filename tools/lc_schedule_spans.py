"""Developer tool: where a candidate's turn in k_lc_schedule's walk goes (device library built with -DCFX_TRACE
-DCFX_TRACE_KERNEL=9: python tools/build_variants.py tr9="-DCFX_TRACE -DCFX_TRACE_KERNEL=9"), on the bench workload with
laneChange true.  Per road with candidates: staging, ranks, and the walk split into own fields / segment searches /
laneLink search / signals / insertion (us)."""
import ctypes, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_lcspan", 0, scenario="grid_30x30")
c = json.load(open(cfg)); c["laneChange"] = True
path = cfg.replace(".json", "_lc.json"); json.dump(c, open(path, "w"))
lib = os.path.join(ROOT, "gpurun_exp", "lib_tr9.so")
dll = ctypes.CDLL(lib)
eng = _cityflow.Engine._with_backend(path, 1, lib)
dll.cfx_trace_dump(b"/tmp/x", 0)   # arm (the traced kernel writes its stamps from its first launch on)
for _ in range(400):
    eng.next_step()
eng.sync()
rows = []
for rep in range(5):
    dll.cfx_trace_dump(b"/tmp/x", 0)   # arm
    for _ in range(3):
        eng.next_step()
    eng.sync()
    dll.cfx_trace_dump(b"/tmp/x", -1)  # clear
    eng.next_step()
    eng.sync()
    dll.cfx_trace_dump(b"/tmp/lcspan.bin", 65536)
    a = np.fromfile("/tmp/lcspan.bin", dtype=np.int64).reshape(-1, 8)
    a = a[(a[:, 0] > 0) & (a[:, 4] > 0)]
    rows.append(a)
a = np.concatenate(rows)
n = a[:, 5]
tA = (a[:, 3] & 0xFFFFFFFF) / 100.0
tB = (a[:, 3] >> 32) / 100.0
tC = (a[:, 6] & 0xFFFFF) / 100.0
tD = ((a[:, 6] >> 20) & 0xFFFFF) / 100.0
tE = ((a[:, 6] >> 40) & 0xFFFFF) / 100.0
print("roads with candidates (5 steps): %d, candidates %d (per road avg %.2f max %d)" % (len(a), n.sum(), n.mean(), n.max()))
print("per road, us: staging %.2f  ranks %.2f  walk %.2f (p90 %.2f max %.2f)  whole block %.2f (max %.2f)" % (
    ((a[:, 1] - a[:, 0]) / 100.0).mean(), ((a[:, 2] - a[:, 1]) / 100.0).mean(), ((a[:, 4] - a[:, 2]) / 100.0).mean(),
    np.percentile((a[:, 4] - a[:, 2]) / 100.0, 90), ((a[:, 4] - a[:, 2]) / 100.0).max(),
    ((a[:, 4] - a[:, 0]) / 100.0).mean(), ((a[:, 4] - a[:, 0]) / 100.0).max()))
tot = n.sum()
print("per candidate, us: own fields %.2f | segment searches %.2f | laneLink search %.2f | signals %.2f | insertion %.2f | sum %.2f" % (
    tA.sum() / tot, tB.sum() / tot, tC.sum() / tot, tD.sum() / tot, tE.sum() / tot, (tA + tB + tC + tD + tE).sum() / tot))
