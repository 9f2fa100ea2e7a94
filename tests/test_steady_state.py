"""GPU (-m gpu): a long run without reset() neither stalls nor grows.  The reference frees a vehicle when it finishes
(src/engine/engine.cpp:296-310) and allocates one when it is created; here the device tables are sized once (cfx_create: slots
for half the network's bumper-to-bumper capacity, vehicle numbers for 4 M vehicles between two resets) and a step of an
ordinary run allocates nothing — round 4's driver run showed a 70 ms next_step() where the vehicle table doubled from 64 k."""
import time

import numpy as np
import pytest

from conftest import TWIN_LIB, assert_same_state, assert_hip_backend

pytestmark = pytest.mark.gpu


def test_ten_thousand_steps_without_a_stall_or_growth(mod, workdir):
    import bench
    cfg = bench.build_workload(workdir, 0, scenario="grid_30x30")
    eng = mod.Engine(cfg, 1)
    assert_hip_backend(eng)
    for _ in range(bench.BUILD_UP_STEPS):
        eng.next_step()
    eng.sync()
    free0 = eng._device_memory()[0]
    slow, t_all = [], time.perf_counter()
    for s in range(10000):
        t0 = time.perf_counter()
        eng.next_step()
        dt = time.perf_counter() - t0
        if dt >= 2e-3:
            slow.append((s, dt))
        if s % 32 == 31:
            eng.sync()  # (a caller that never waits runs ahead until the HIP queue is full and then blocks inside a launch for
                        # milliseconds at a time — time the device is busy in, not a stall; an RL loop waits every step)
    eng.sync()
    total = time.perf_counter() - t_all
    free1 = eng._device_memory()[0]
    sc = eng._scalars()
    assert sc["spawned_vehicle_count"] > 300000 and sc["active_vehicle_count"] > 20000, sc
    print("next_step() calls of 10 000 above 2 ms: %r; %.1f us per step" % (slow, total / 10000 * 1e6))
    # (measured: worst 0.5-0.7 ms, a launch that waits for queue room.  ONE slow call in 10 000 is left to the machine — another
    # tenant's burst, a scheduler hiccup; a stall that belongs to the engine — table growth, a ring regrow — comes back)
    assert len(slow) <= 1 and all(dt < 20e-3 for _, dt in slow), "slow next_step() calls (step, seconds): %r" % slow
    assert free0 - free1 < 8 << 20, "device memory grew by %d MiB over 10 000 steps" % ((free0 - free1) >> 20)
    assert total / 10000 < 200e-6, "%.1f us per step over the long run" % (total / 10000 * 1e6)


def test_bench_sequence_free_running_without_a_stall(mod, workdir, tmp_path):
    """bench.py's own sequence — demand build-up, Archive dump, load_from_file, warm-up — and then 1 000 FREE-RUNNING steps (no
    sync inside, as bench.py's sustained windows): no next_step() call above 1.5 ms.  Rounds 4-6 saw one 45-80 ms call per
    driver-shaped run there; it was the container's CPU quota throttled by BLAS workers a lazy `import numpy` started (DESIGN.md
    section 6) — numpy is imported with the package now, and what a free-running caller can still meet is the spawner's
    priority-collision query waiting for the queued steps, which cfx_step bounds to 16 (0.65 ms)."""
    import bench
    cfg = bench.build_workload(workdir, 0, scenario="grid_30x30")
    eng = mod.Engine(cfg, 1)
    assert_hip_backend(eng)
    for _ in range(bench.BUILD_UP_STEPS):
        eng.next_step()
    dump = str(tmp_path / "state.json")
    eng.snapshot().dump(dump)
    eng.load_from_file(dump)
    for _ in range(25):
        eng.next_step()
    eng.sync()
    # Two windows of 1 000 steps, as bench.py takes five of 200: the stall of rounds 4-6 was there in EVERY run; a single slow
    # call that does not come back in the second window is the machine's (this test shares its box with whatever else runs
    # there), and the engine's own causes are asserted on separately through the device library's statistics.
    seen = []
    for attempt in range(2):
        eng._host_stats(True)
        calls = []
        t1 = t_all = time.perf_counter()
        for _ in range(1000):
            eng.next_step()
            t2 = time.perf_counter()
            calls.append(t2 - t1)
            t1 = t2
        eng.sync()
        total = time.perf_counter() - t_all
        worst = max(range(len(calls)), key=lambda i: calls[i])
        hs = eng._host_stats(True)
        print("free-running window %d: %.1f us per step, worst call %.0f us at %d, device library: %r"
              % (attempt, total / 1000 * 1e6, calls[worst] * 1e6, worst, hs))
        assert hs["ring_regrows_total"] == 0 and hs["worst_step_call_cause"] == 0, hs
        seen.append((calls[worst], worst, total, hs))
        if calls[worst] < 1.5e-3 and total / 1000 < 80e-6:
            break
    else:
        raise AssertionError("both windows: (worst call s, its index, window s, host stats) = %r" % (seen,))


def test_vehicle_tables_grow_without_draining_the_stream(mod, scen, workdir):
    """The growth path itself (config "cfx": ringCapacityPercent below 100 starts the vehicle tables at 4 k numbers): they double
    several times during the run — new arrays, a copy ordered on the stream, the old ones freed at the next sync — with the state
    equal to the twin's throughout, on both layouts."""
    import json
    import os
    base = scen.materialize("grid_6x6", workdir)
    d = os.path.dirname(base)
    flow = scen.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 400, seed=7,
                            interval=2.0, base_flow=os.path.join(d, "flow.json"))
    cfg = scen.materialize("grid_6x6", workdir, flow_file=flow)
    for layout in ("ring", "dense"):
        c = json.load(open(cfg))
        c["cfx"] = {"layout": layout, "ringCapacityPercent": 60}
        path = cfg.replace(".json", "_smallvid_%s.json" % layout)
        with open(path, "w") as f:
            json.dump(c, f)
        hip = mod.Engine(path, 1)
        tw = mod.Engine._with_backend(cfg, 1, TWIN_LIB)
        for s in range(500):
            hip.next_step()
            tw.next_step()
            if s % 25 == 24:
                assert_same_state(hip, tw, "growing vehicle tables (%s) step %d" % (layout, s + 1))
        assert hip._scalars()["spawned_vehicle_count"] > 30000  # (4 k -> 8 k -> 16 k -> 32 k -> 64 k)
