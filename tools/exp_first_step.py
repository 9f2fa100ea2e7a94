"""Developer tool: per-kernel time of the FIRST step after a load of the 1 M-vehicle state, for differently built device
libraries — for timing-only experiments whose results are wrong (a variant that skips part of the work), where a second
step would already run on a corrupted state.  usage: python tools/exp_first_step.py lib1.so lib2.so ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = sys.argv[1:]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario="gen_100x100", n_extra=33000)
base = _cityflow.Engine(cfg, 1)
for _ in range(300): base.next_step()
arch = base.snapshot()
del base
for lib in [_cityflow._default_backend_path()] + libs:
    eng = _cityflow.Engine._with_backend(cfg, 1, os.path.abspath(lib))
    res = {}
    for rep in range(6):
        eng.load(arch); eng.sync()
        eng._profile_enable(True)
        eng.next_step()
        prof = eng._profile_read(); eng._profile_enable(False)
        for k, (ms, n) in prof.items():
            if n: res.setdefault(k, []).append(ms / n * 1e3)
    print(os.path.basename(lib), {k: (round(min(v), 1), round(sorted(v)[len(v) // 2], 1)) for k, v in res.items()}, flush=True)
    del eng
