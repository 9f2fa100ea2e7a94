"""RL-loop throughput on the bench workload with rlTrafficLight: per step set every signal, step, read per-lane counts.
Compares the array API with the reference-style dict API on this engine (and, with --ref, times the reference engine)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
want_ref = "--ref" in sys.argv
sys.argv = [sys.argv[0]]
import json
import numpy as np
import bench
from cityflow_amd import _cityflow
cfg0 = bench.build_workload("/tmp/cfa_rl", 0)
c = json.load(open(cfg0)); c["rlTrafficLight"] = True
cfg = cfg0.replace(".json", "_rl.json"); json.dump(c, open(cfg, "w"))
eng = (_cityflow.Engine._with_backend(cfg, 1, os.path.abspath(os.environ["CFX_BACKEND_LIB"]))
       if os.environ.get("CFX_BACKEND_LIB") else _cityflow.Engine(cfg, 1))
ids = eng.intersection_ids(); I = len(ids)
virt = eng._flat_net()["inter_virtual"]
for s in range(300):
    if s % 10 == 0: eng.set_tl_phases(np.full(I, (s // 10) % 8, dtype=np.int32))
    eng.next_step()
eng.sync()
def loop(n, mode):
    t0 = time.perf_counter()
    for s in range(n):
        ph = (s // 10) % 8
        if mode == "array":
            eng.set_tl_phases(np.full(I, ph, dtype=np.int32)); eng.next_step(); obs = eng.get_lane_vehicle_count_array()
        else:
            for i, iid in enumerate(ids):
                if not virt[i]: eng.set_tl_phase(iid, ph)
            eng.next_step(); obs = eng.get_lane_vehicle_count()
    eng.sync()
    return n / (time.perf_counter() - t0)
print(json.dumps({"running": eng.get_vehicle_count(), "array_api_steps_per_sec": loop(200, "array"),
                  "dict_api_steps_per_sec": loop(20, "dict")}), flush=True)
if want_ref:
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import cityflow_ref
    dump = "/tmp/cfa_rl/state.json"; eng.snapshot().dump(dump)
    ref = cityflow_ref.Engine(cfg, 8); ref.load_from_file(dump)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 10:
        for i, iid in enumerate(ids):
            if not virt[i]: ref.set_tl_phase(iid, (n // 10) % 8)
        ref.next_step(); obs = ref.get_lane_vehicle_count(); n += 1
    print(json.dumps({"reference_8_threads_dict_api_steps_per_sec": n / (time.perf_counter() - t0)}), flush=True)
    time.sleep(0.2)
