"""Developer soak run: the bench.py workload (30x30, ~100k vehicles) on the HIP engine vs the CPU twin over a long horizon,
every per-vehicle field every `every` steps.   python tests/tools/soak_parity.py STEPS [EVERY]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
steps = int(sys.argv[1])
every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from cityflow_amd import _cityflow as m  # noqa: E402
from conftest import TWIN_LIB, assert_same_state  # noqa: E402

cfg = bench.build_workload("/tmp/cfa_soak", 0)
hip, tw = m.Engine(cfg, 1), m.Engine._with_backend(cfg, 1, TWIN_LIB)
t0 = time.time()
for s in range(steps):
    hip.next_step()
    tw.next_step()
    if s % every == every - 1:
        assert_same_state(hip, tw, "step %d" % (s + 1))
        print("step", s + 1, "ok:", hip.get_vehicle_count(), "running,", hip._scalars()["finished_vehicle_count"], "finished, %.0f s" % (time.time() - t0), flush=True)
