"""Developer tool: time the per-step kernels of alternative builds of the C-ABI library from the SAME warm state
(warmed by the product library, transferred through an in-memory Archive), so experiments that change the logic
are still compared on identical inputs for the first few steps.
usage: python tools/exp_bench.py lib1.so [lib2.so ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = sys.argv[1:]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=os.environ.get("CFX_EXP_SCENARIO", "grid_30x30"),
                           n_extra=int(os.environ.get("CFX_EXP_EXTRA", bench.N_EXTRA_FLOWS)))
base = _cityflow.Engine(cfg, 1)
for _ in range(300):
    base.next_step()
arch = base.snapshot()
for lib in [_cityflow._default_backend_path()] + libs:
    eng = _cityflow.Engine._with_backend(cfg, 1, os.path.abspath(lib))
    res = {}
    for rep in range(3):
        eng.load(arch)
        eng.next_step(); eng.next_step()   # warm caches / clocks
        eng.sync()
        eng._profile_enable(True)
        for _ in range(8):
            eng.next_step()
        prof = eng._profile_read()
        eng._profile_enable(False)
        for k, (ms, n) in prof.items():
            res.setdefault(k, []).append(ms / max(n, 1) * 1e3)
    import time
    wall = []
    for rep in range(3):  # un-instrumented wall clock per step from the same state
        eng.load(arch)
        for _ in range(5):
            eng.next_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.next_step()
        eng.sync()
        wall.append((time.perf_counter() - t0) / 200 * 1e6)
    print(os.path.basename(lib), {k: round(min(v), 1) for k, v in res.items()}, "running", eng.get_vehicle_count(),
          "wall us/step", [round(w, 1) for w in wall], flush=True)
    del eng
