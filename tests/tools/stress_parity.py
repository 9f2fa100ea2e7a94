"""Developer stress run on the GPU box: many seeded irregular networks (tests/test_irregular.py generator) HIP vs CPU twin,
every per-vehicle field.   python tests/tools/stress_parity.py FIRST_SEED N_SEEDS [STEPS]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cityflow_amd import _cityflow as m, scenarios  # noqa: E402
from conftest import TWIN_LIB, assert_same_state  # noqa: E402
from test_irregular import irregular  # noqa: E402

first, n = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
wd = tempfile.mkdtemp(prefix="stress_")
for seed in range(first, first + n):
    cfg = irregular(scenarios, wd, seed, n=4 + seed % 4)
    hip, tw = m.Engine(cfg, 1), m.Engine._with_backend(cfg, 1, TWIN_LIB)
    for s in range(steps):
        hip.next_step()
        tw.next_step()
        if s % 20 == 19:
            assert_same_state(hip, tw, "seed %d step %d" % (seed, s + 1))
    print("seed", seed, "ok:", hip.get_vehicle_count(), "running,", hip._scalars()["finished_vehicle_count"], "finished", flush=True)
