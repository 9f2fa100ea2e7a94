"""Developer probe: where an RL iteration's time goes on the headline network (rlTrafficLight): steps alone, steps with a
sync each, steps with the lane counts read, phases set with and without the read.  us per iteration, 400 iterations each."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
scenario = os.environ.get("CFX_RL_SCENARIO", "grid_30x30")  # gen_100x100 with CFX_RL_EXTRA=33000: the 1 M-vehicle network
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=scenario, n_extra=int(os.environ.get("CFX_RL_EXTRA", bench.N_EXTRA_FLOWS)))
if os.environ.get("CFX_RL_SPAWN_AHEAD") == "0":  # the host's step ahead off ("cfx": {"spawnAhead": false})
    import json
    c = json.load(open(cfg)); c["cfx"] = {"spawnAhead": False}
    cfg = cfg.replace(".json", "_noahead.json"); json.dump(c, open(cfg, "w"))
rl = bench.with_config(cfg, "rl", rlTrafficLight=True)
e = _cityflow.Engine(rl, 1)
for _ in range(300):
    e.next_step()
e.sync()
n_inter = len(e.intersection_ids())
N = int(os.environ.get("CFX_RL_ITERS", 400))


def run(name, body):
    for s in range(20):
        body(s)
    e.sync()
    t0 = time.perf_counter()
    for s in range(N):
        body(s)
    e.sync()
    print("%-58s %7.1f us" % (name, (time.perf_counter() - t0) / N * 1e6), flush=True)


run("next_step", lambda s: e.next_step())
run("next_step + sync", lambda s: (e.next_step(), e.sync()))
run("next_step + lane counts", lambda s: (e.next_step(), e.get_lane_vehicle_count_array()))
run("set_tl_phases (changing every 10) + next_step", lambda s: (e.set_tl_phases(np.full(n_inter, (s // 10) % 8, dtype=np.int32)), e.next_step()))
run("set_tl_phases (changing every step) + next_step", lambda s: (e.set_tl_phases(np.full(n_inter, s % 8, dtype=np.int32)), e.next_step()))
run("set_tl_phases (every 10) + next_step + lane counts", lambda s: (e.set_tl_phases(np.full(n_inter, (s // 10) % 8, dtype=np.int32)), e.next_step(), e.get_lane_vehicle_count_array()))
run("lane counts alone (no step between)", lambda s: e.get_lane_vehicle_count_array())
run("np.full alone", lambda s: np.full(n_inter, (s // 10) % 8, dtype=np.int32))
_same = np.full(n_inter, 3, dtype=np.int32)
e.set_tl_phases(_same)
run("set_tl_phases alone, nothing changes (host filter only)", lambda s: e.set_tl_phases(_same))
run("set_tl_phases alone, every signal changes (no step between)", lambda s: e.set_tl_phases(np.full(n_inter, s % 8, dtype=np.int32)))
run("set_tl_phases (never changing) + next_step + lane counts", lambda s: (e.set_tl_phases(_same), e.next_step(), e.get_lane_vehicle_count_array()))
run("set_tl_phases (changing every step) + next_step + lane counts", lambda s: (e.set_tl_phases(np.full(n_inter, s % 8, dtype=np.int32)), e.next_step(), e.get_lane_vehicle_count_array()))
_ids = e.intersection_ids()
_virt = e._flat_net()["inter_virtual"]
_real = [iid for i, iid in enumerate(_ids) if not _virt[i]]


def _dict_iteration(s):
    ph = (s // 10) % 8
    for iid in _real:
        e.set_tl_phase(iid, ph)
    e.next_step()
    return e.get_lane_vehicle_count()


run("the reference's calls: %d x set_tl_phase alone" % len(_real), lambda s: [e.set_tl_phase(iid, 3) for iid in _real])
run("the reference's calls: get_lane_vehicle_count alone (dict)", lambda s: e.get_lane_vehicle_count())
run("the reference's calls: set_tl_phase x N + next_step + dict", _dict_iteration)
