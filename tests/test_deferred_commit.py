"""GPU (-m gpu): the ring layout defers a step's commit into the next step's admission (kr_admit<true>); every
other entry point of the ABI must launch a pending commit first.  Random sequences of API calls — steps in runs of varying
length (so that merged and standalone commits alternate), getters, signal plans, custom speeds, routes, snapshot / load,
reset — are applied to the HIP engine and to the CPU twin alike; whatever either returns, and the whole state at the end of
every round, must be equal.  Reference semantics of the calls: /root/reference/src/engine/engine.cpp:566-594 (nextStep),
628-634 (getLaneVehicleCount), 719-725 (setTrafficLightPhase), 827-834 (setVehicleSpeed), 744-760 (reset),
src/engine/archive.cpp:9-126 (snapshot / load)."""
import json
import os

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state, assert_hip_backend

pytestmark = pytest.mark.gpu


def _engines(mod, scen, workdir, rl, form=0):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_mix.json"), 250, seed=23, interval=3.0,
                            base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow, rlTrafficLight=rl)
    c = json.load(open(cfg))
    c["cfx"] = {"layout": "ring", "ringLanesPerWave": form}  # (30000: the list form of the action phase)
    ring = cfg.replace(".json", "_ring.json")
    json.dump(c, open(ring, "w"))
    hip = mod.Engine(ring, 1)
    assert_hip_backend(hip)
    assert hip._layout() == "ring"
    return hip, mod.Engine._with_backend(cfg, 1, TWIN_LIB)


@pytest.mark.parametrize("rl,seed,form", [(True, 1, 0), (True, 2, 0), (False, 3, 0), (True, 4, 30000), (False, 5, 30000)])
def test_random_call_sequences_equal_twin(mod, scen, workdir, rl, seed, form):
    hip, tw = _engines(mod, scen, workdir, rl, form)
    rng = np.random.default_rng(seed)
    n_inter = len(hip.intersection_ids())
    archives = None
    steps = 0
    for round_ in range(40):
        for _ in range(int(rng.integers(1, 9))):
            op = int(rng.integers(0, 10))
            if op <= 3:  # a run of steps: its inner commits ride with the next admission
                for _ in range(int(rng.integers(1, 12))):
                    hip.next_step()
                    tw.next_step()
                    steps += 1
            elif op == 4:
                assert np.array_equal(hip.get_lane_vehicle_count_array(), tw.get_lane_vehicle_count_array())
            elif op == 5 and rl:
                ph = rng.integers(0, 8, size=n_inter).astype(np.int32)
                hip.set_tl_phases(ph)
                tw.set_tl_phases(ph)
            elif op == 6:
                speeds = hip.get_vehicle_speed()
                assert speeds == tw.get_vehicle_speed()
                if speeds:
                    vid = sorted(speeds)[int(rng.integers(0, len(speeds)))]
                    v = float(rng.uniform(0.0, 12.0))
                    hip.set_vehicle_speed(vid, v)
                    tw.set_vehicle_speed(vid, v)
            elif op == 7:
                assert hip.get_vehicle_count() == tw.get_vehicle_count()
                assert hip.get_lane_waiting_vehicle_count() == tw.get_lane_waiting_vehicle_count()
            elif op == 8:
                if archives is None or rng.random() < 0.5:
                    archives = (hip.snapshot(), tw.snapshot(), steps)
                else:  # back to an earlier state: a pending commit must neither run after the load nor be lost before it
                    hip.load(archives[0])
                    tw.load(archives[1])
                    steps = archives[2]
            elif op == 9 and round_ % 13 == 12:
                hip.reset(False)
                tw.reset(False)
                archives = None
                steps = 0
        assert_same_state(hip, tw, "random calls (rl %s, seed %d) round %d, step %d" % (rl, seed, round_, steps))
    assert hip.get_average_travel_time() == tw.get_average_travel_time()
