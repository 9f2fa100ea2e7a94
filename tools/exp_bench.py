"""Developer tool: time the per-step kernels of alternative builds of the C-ABI library (A/B experiments).
usage: python tools/exp_bench.py lib1.so [lib2.so ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = sys.argv[1:]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0)
for lib in libs:
    eng = _cityflow.Engine._with_backend(cfg, 1, os.path.abspath(lib))
    for _ in range(300):
        eng.next_step()
    eng.sync()
    eng._profile_enable(True)
    for _ in range(60):
        eng.next_step()
    prof = eng._profile_read()
    eng._profile_enable(False)
    print(os.path.basename(lib), {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in prof.items()},
          "running", eng.get_vehicle_count(), flush=True)
    del eng
