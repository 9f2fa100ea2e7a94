"""CPU: random sequences of `cityflow.Engine` calls applied to the unmodified reference (oracle/_ref) and to this host on the
CPU twin alike — steps in runs of varying length, the dict getters, signal phases, custom speeds, snapshot / load, reset —
whatever either returns must be equal, and so must the state at the end of every round.  (The same kind of sequence pins the
HIP engine against the twin: tests/test_deferred_commit.py.)  Reference semantics: /root/reference/src/engine/engine.cpp
:566-594 (nextStep), 615-691 (getters), 719-725 (setTrafficLightPhase), 827-834 (setVehicleSpeed), 744-760 (reset),
src/engine/archive.cpp:9-126 (snapshot / load)."""
import json
import os
import time

import numpy as np
import pytest

from conftest import TWIN_LIB, checkpoint_record


def _config(scen, workdir, rl):
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_seq.json"), 180, seed=29, interval=4.0,
                            base_flow=os.path.join(d, "flow.json"))
    return scen.materialize("grid_6x6", workdir, flow_file=flow, rlTrafficLight=rl)


@pytest.mark.parametrize("rl,seed", [(True, 5), (True, 6), (False, 7)])
def test_random_call_sequences_equal_reference(mod, ref_module, scen, workdir, rl, seed):
    cfg = _config(scen, workdir, rl)
    ref, tw = ref_module.Engine(cfg, 1), mod.Engine._with_backend(cfg, 1, TWIN_LIB)
    rng = np.random.default_rng(seed)
    with open(os.path.join(os.path.dirname(cfg), "roadnet.json")) as f:
        inters = [i["id"] for i in json.load(f)["intersections"] if not i["virtual"]]
    archives = None
    for round_ in range(30):
        for _ in range(int(rng.integers(1, 7))):
            op = int(rng.integers(0, 9))
            if op <= 3:
                for _ in range(int(rng.integers(1, 15))):
                    ref.next_step()
                    tw.next_step()
            elif op == 4:
                assert ref.get_lane_vehicle_count() == tw.get_lane_vehicle_count()
                assert ref.get_lane_waiting_vehicle_count() == tw.get_lane_waiting_vehicle_count()
            elif op == 5 and rl:
                for i in rng.choice(len(inters), size=5, replace=False):
                    ph = int(rng.integers(0, 8))
                    ref.set_tl_phase(inters[int(i)], ph)
                    tw.set_tl_phase(inters[int(i)], ph)
            elif op == 6:
                speeds = tw.get_vehicle_speed()
                assert speeds == ref.get_vehicle_speed()
                if speeds:
                    vid = sorted(speeds)[int(rng.integers(0, len(speeds)))]
                    v = float(rng.uniform(0.0, 12.0))
                    ref.set_vehicle_speed(vid, v)
                    tw.set_vehicle_speed(vid, v)
                    assert ref.get_vehicle_info(vid) == tw.get_vehicle_info(vid)
            elif op == 7:
                assert ref.get_vehicle_count() == tw.get_vehicle_count()
                assert ref.get_vehicles(True) == tw.get_vehicles(True)
                assert ref.get_lane_vehicles() == tw.get_lane_vehicles()
            elif op == 8:
                if archives is None or rng.random() < 0.5:
                    archives = (ref.snapshot(), tw.snapshot())
                else:
                    ref.load(archives[0])
                    tw.load(archives[1])
        if round_ == 19:
            ref.reset(False)
            tw.reset(False)
            archives = None
        assert checkpoint_record(tw) == checkpoint_record(ref), "rl %s seed %d round %d" % (rl, seed, round_)
        assert ref.get_current_time() == tw.get_current_time()
    time.sleep(0.2)  # reference destructor race
    del ref
