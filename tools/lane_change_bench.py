"""Step time with laneChange=true next to laneChange=false on the bench.py workload (and the stock 6x6 grid).
usage: python tools/lane_change_bench.py [scenario ...]        -> one JSON line per scenario"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
names = sys.argv[1:] or ["grid_6x6", "grid_30x30"]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow

for name in names:
    out = {"scenario": name}
    for lc in (False, True):
        cfg = bench.build_workload("/tmp/cfa_lcbench", 0, scenario=name,
                                   n_extra=bench.N_EXTRA_FLOWS if name == "grid_30x30" else 0)
        c = json.load(open(cfg))
        c["laneChange"] = lc
        path = cfg.replace(".json", "_lc%d.json" % lc)
        json.dump(c, open(path, "w"))
        eng = _cityflow.Engine(path, 1)
        warm = 300 if name == "grid_30x30" else 450  # 6x6: lane changes start at step 378
        for _ in range(warm):
            eng.next_step()
        eng.sync()
        K = 200
        s0 = eng._scalars()
        t0 = time.perf_counter()
        for _ in range(K):
            eng.next_step()
        eng.sync()
        dt = time.perf_counter() - t0
        s1 = eng._scalars()
        key = "lane_change" if lc else "plain"
        eng._profile_enable(True)  # (per-kernel times of 50 more steps, us per step)
        for _ in range(50):
            eng.next_step()
        prof = eng._profile_read()
        eng._profile_enable(False)
        syms = eng._profile_symbols()
        out[key] = {"us_per_step": dt / K * 1e6, "running": s1["active_vehicle_count"],
                    "vehicle_steps_per_sec": (s1["vehicle_steps"] - s0["vehicle_steps"]) / dt,
                    "vehicles_created": s1["spawned_vehicle_count"] - s0["spawned_vehicle_count"]}
        out[key]["kernel_us_per_step"] = {syms.get(k, k): round(ms / 50 * 1e3, 2) for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]) if n}
        out[key]["lane_history_kept"] = eng._keeps_lane_history()
        if lc:
            st = eng._vehicle_state()
            out[key]["shadows_now"] = int((st["lc_flags"] & 1).sum())
        del eng
    print(json.dumps(out), flush=True)
