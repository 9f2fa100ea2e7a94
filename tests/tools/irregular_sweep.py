"""One-off sweep (developer tool): the unmodified reference (oracle/_ref) against the CPU twin WITHOUT lane change on seeded
irregular networks (tests/test_irregular.py generator): per-lane counts, every vehicle's speed and distance, the average
travel time.   usage: python tests/tools/irregular_sweep.py <first_seed> <n_seeds> <steps> [every]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as cf  # noqa: E402
import test_irregular as ti  # noqa: E402
from cityflow_amd import _cityflow as m, scenarios as scen  # noqa: E402

sys.path.insert(0, cf.REF_DIR)
import cityflow_ref  # noqa: E402

first, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
bad = 0
for seed in range(first, first + n):
    cfg = ti.irregular(scen, "/tmp/cfa_irr_sweep", seed)
    ref, tw = cityflow_ref.Engine(cfg, 1), m.Engine._with_backend(cfg, 1, cf.TWIN_LIB)
    diff = None
    for s in range(steps):
        ref.next_step()
        tw.next_step()
        if s % every == every - 1 and cf.checkpoint_record(tw) != cf.checkpoint_record(ref):
            diff = s + 1
            break
    same_att = ref.get_average_travel_time() == tw.get_average_travel_time()
    print({"seed": seed, "steps": steps, "first_difference": diff, "average_travel_time_equal": same_att,
           "running": tw.get_vehicle_count(), "finished": tw._scalars()["finished_vehicle_count"]}, flush=True)
    bad += diff is not None or not same_att
    time.sleep(0.2)  # reference destructor race
    del ref
print("seeds with differences:", bad)
