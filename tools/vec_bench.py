"""Batched-replica throughput: R copies of the bench.py workload advanced by one device engine.
usage: python tools/vec_bench.py R [R ...]   -> one JSON line per R"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rs = [int(x) for x in sys.argv[1:]] or [1, 4, 8]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
# default: the bench.py workload; CFX_VEC_SCENARIO=grid_6x6 CFX_VEC_EXTRA=0 gives the stock small grid RL work uses
cfg = bench.build_workload("/tmp/cfa_vec", 0, scenario=os.environ.get("CFX_VEC_SCENARIO", "grid_30x30"),
                           n_extra=int(os.environ.get("CFX_VEC_EXTRA", bench.N_EXTRA_FLOWS)))
if os.environ.get("CFX_VEC_LC"):  # the same with laneChange: true (every environment its own lane-change schedule)
    c = json.load(open(cfg))
    c["laneChange"] = True
    cfg = cfg.replace(".json", "_lc.json")
    json.dump(c, open(cfg, "w"))
if os.environ.get("CFX_VEC_CFX"):  # implementation choices, "key=value,key=value" (e.g. spawnAhead=false)
    def _val(v):
        return {"true": True, "false": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    cfg = bench.with_config(cfg, "cfx", cfx={k: _val(v) for k, v in (kv.split("=") for kv in os.environ["CFX_VEC_CFX"].split(","))})
for R in rs:
    t0 = time.perf_counter()
    lib = os.environ.get("CFX_VEC_LIB")  # a differently built device library
    eng = _cityflow.VectorEngine._with_backend(cfg, R, 1, os.path.abspath(lib)) if lib else _cityflow.VectorEngine(cfg, R, 1)
    t_load = time.perf_counter() - t0
    for _ in range(300):
        eng.next_step()
    eng.sync()
    s0 = eng._scalars()
    K = 100
    h0 = eng._host_seconds()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.next_step()
    eng.sync()
    dt = time.perf_counter() - t0
    h1 = eng._host_seconds()
    s1 = eng._scalars()
    eng._profile_enable(True)
    for _ in range(50):
        eng.next_step()
    prof = eng._profile_read()
    eng._profile_enable(False)
    s2 = eng._scalars()
    vs = s1["vehicle_steps"] - s0["vehicle_steps"]
    act_ms, act_n = prof["k_action"]
    vpl = (s2["vehicle_steps"] - s1["vehicle_steps"]) / max(act_n, 1)
    gbs = 48.0 * vpl / (act_ms / act_n / 1e3) / 1e9
    print(json.dumps({"envs": R, "running_vehicles": s1["active_vehicle_count"], "ms_per_step": dt / K * 1e3,
                      "env_steps_per_sec": K * R / dt, "vehicle_steps_per_sec": vs / dt,
                      "k_action_us": act_ms / act_n * 1e3, "k_action_GBps": gbs, "k_action_frac_of_8TBps": gbs / 8000.0,
                      "kernel_us": {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in prof.items()},
                      "kernel_launches_in_50_steps": {k: n for k, (ms, n) in prof.items() if n},
                      "kernel_us_per_step": round(sum(ms for ms, n in prof.values()) / 50 * 1e3, 1),
                      "host_us_per_step": {k: round((b - a) / K * 1e6, 1) for k, a, b in zip(("spawn", "translate", "submit", "ahead_thread"), h0, h1)},
                      "cfx": os.environ.get("CFX_VEC_CFX"),
                      "load_s": round(t_load, 1)}), flush=True)
    del eng
