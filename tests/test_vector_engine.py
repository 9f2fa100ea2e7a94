"""VectorEngine: every environment of a batched engine evolves exactly like a standalone Engine with seed + env index.
CPU (twin backend) here; the same check on the HIP engine carries the gpu marker."""
import numpy as np
import pytest

from conftest import TWIN_LIB


def _check(mod, scen, workdir, make_vec, make_single, name="example_1x1", envs=3, steps=150, rl=False, lane_change=False,
           every=10):
    kw = {"rlTrafficLight": True} if rl else {}
    if lane_change:
        kw["laneChange"] = True
    vec = make_vec(scen.materialize(name, workdir, **kw), envs)
    singles = [make_single(scen.materialize(name, workdir, seed=e, **kw)) for e in range(envs)]
    inter_ids = vec.intersection_ids()
    assert vec.lane_ids() == singles[0].lane_ids()
    for s in range(steps):
        if rl and s % 10 == 0:
            ph = np.zeros((envs, len(inter_ids)), dtype=np.int32)
            for e in range(envs):
                ph[e, :] = (s // 10 + e) % 8
                for i, iid in enumerate(inter_ids):
                    try:
                        singles[e].set_tl_phase(iid, int(ph[e, i]))
                    except (IndexError, RuntimeError):
                        pass  # virtual intersections
            vec.set_tl_phases(ph)
        vec.next_step()
        for e in singles:
            e.next_step()
        if s % every == every - 1:
            counts = vec.get_lane_vehicle_count_array()
            waits = vec.get_lane_waiting_vehicle_count_array()
            assert counts.shape == (envs, len(vec.lane_ids()))
            for e in range(envs):
                assert np.array_equal(counts[e], singles[e].get_lane_vehicle_count_array()), (s, e)
                assert np.array_equal(waits[e], singles[e].get_lane_waiting_vehicle_count_array()), (s, e)
                assert vec.get_vehicle_speed(e) == singles[e].get_vehicle_speed(), (s, e)
    assert vec.get_vehicle_count() == sum(e.get_vehicle_count() for e in singles)
    if name == "example_1x1":  # seeds pick different first lanes there; on the grids every route has one candidate
        c = vec.get_lane_vehicle_count_array()
        assert not np.array_equal(c[0], c[1])


def test_vector_engine_twin(mod, scen, workdir):
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine._with_backend(c, n, 1, TWIN_LIB),
           lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB))


def test_vector_engine_twin_rl(mod, scen, workdir):
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine._with_backend(c, n, 1, TWIN_LIB),
           lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), name="grid_6x6", envs=2, steps=60, rl=True)


def test_vector_engine_threaded_spawners_twin(mod, scen, workdir):
    """40 environments: spawners and record translation run on the host thread pool (used from 32 environments on); every
    environment still equals its standalone engine."""
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine._with_backend(c, n, 1, TWIN_LIB),
           lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), envs=40, steps=100)


def test_vector_engine_explicit_host_threads_twin(mod, scen, workdir):
    """`"cfx": {"hostThreads": 3}` puts the spawners of even a handful of environments on the thread pool (automatic only from
    32 environments on); results do not depend on it."""
    import json
    base = scen.materialize("example_1x1", workdir)
    c = json.load(open(base))
    c["cfx"] = {"hostThreads": 3}
    threaded = base.replace(".json", "_ht3.json")
    with open(threaded, "w") as f:
        json.dump(c, f)
    vec = mod.VectorEngine._with_backend(threaded, 6, 1, TWIN_LIB)
    ser = mod.VectorEngine._with_backend(base, 6, 1, TWIN_LIB)
    for s in range(120):
        vec.next_step()
        ser.next_step()
        if s % 20 == 19:
            assert np.array_equal(vec.get_lane_vehicle_count_array(), ser.get_lane_vehicle_count_array()), s
    for e in range(6):
        assert vec.get_vehicle_speed(e) == ser.get_vehicle_speed(e)


def test_vector_engine_lane_change_twin(mod, scen, workdir):
    """laneChange: true in batched environments: every environment's lane-change schedule walk (candidates in creation order
    through std::sort's permutation of ITS candidate count) and shadow priorities (ITS generator) are its own
    (cfx_config::n_envs; reference engine.cpp:374-400,792-820 per Engine) — each environment equals the standalone engine with
    its seed after every step, shadows alive and changes completing all along."""
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine._with_backend(c, n, 1, TWIN_LIB),
           lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), envs=3, steps=200, lane_change=True, every=1)


def test_vector_engine_lane_change_grid_twin(mod, scen, workdir):
    """(the 6x6 grid, where lanes fill up from step ~370 on and vehicles start to change)"""
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine._with_backend(c, n, 1, TWIN_LIB),
           lambda c: mod.Engine._with_backend(c, 1, TWIN_LIB), name="grid_6x6", envs=2, steps=470, lane_change=True, every=5)


def test_vector_engine_lane_change_threaded_spawners_twin(mod, scen, workdir):
    """... with the environments' spawners, shadow-priority peeks and record translation on the host thread pool
    (`"cfx": {"hostThreads": 3}`) and with the exact redraw loop of the peek forced (`exactShadowPeek`): same results."""
    import json
    base = scen.materialize("example_1x1", workdir, laneChange=True)
    c = json.load(open(base))
    c["cfx"] = {"hostThreads": 3, "exactShadowPeek": True}
    threaded = base.replace(".json", "_lc_ht3.json")
    with open(threaded, "w") as f:
        json.dump(c, f)
    vec = mod.VectorEngine._with_backend(threaded, 6, 1, TWIN_LIB)
    ser = mod.VectorEngine._with_backend(base, 6, 1, TWIN_LIB)
    for s in range(150):
        vec.next_step()
        ser.next_step()
        assert np.array_equal(vec.get_lane_vehicle_count_array(), ser.get_lane_vehicle_count_array()), s
    for e in range(6):
        assert vec.get_vehicle_speed(e) == ser.get_vehicle_speed(e)
    assert vec.get_vehicle_count() > sum(len(vec.get_vehicle_speed(e)) for e in range(6))  # shadows alive


@pytest.mark.parametrize("seed", [1, 2])
def test_vector_engine_lane_change_random_control_twin(mod, scen, workdir, seed):
    """Batched environments with lane change under random control — every signal set per environment, resets with and without
    a new seed in the middle of lane changes: every environment still equals its standalone engine after every step."""
    rng = np.random.default_rng(seed)
    envs = 3
    cfg = scen.materialize("example_1x1", workdir, laneChange=True, rlTrafficLight=True)
    vec = mod.VectorEngine._with_backend(cfg, envs, 1, TWIN_LIB)
    singles = [mod.Engine._with_backend(scen.materialize("example_1x1", workdir, laneChange=True, rlTrafficLight=True, seed=e), 1, TWIN_LIB)
               for e in range(envs)]
    inter_ids = vec.intersection_ids()
    shadows = 0
    for s in range(260):
        r = rng.random()
        if r < 0.15:
            ph = rng.integers(0, 8, size=(envs, len(inter_ids))).astype(np.int32)
            for e in range(envs):
                for i, iid in enumerate(inter_ids):
                    try:
                        singles[e].set_tl_phase(iid, int(ph[e, i]))
                    except (IndexError, RuntimeError):
                        pass  # virtual intersections
            vec.set_tl_phases(ph)
        elif r < 0.17 and s > 30:
            reseed = bool(rng.integers(0, 2))
            vec.reset(reseed)
            for e in singles:
                e.reset(reseed)
        vec.next_step()
        for e in singles:
            e.next_step()
        counts = vec.get_lane_vehicle_count_array()
        for e in range(envs):
            assert np.array_equal(counts[e], singles[e].get_lane_vehicle_count_array()), (s, e)
        if s % 7 == 0:
            for e in range(envs):
                assert vec.get_vehicle_speed(e) == singles[e].get_vehicle_speed(), (s, e)
            shadows = max(shadows, vec.get_vehicle_count() - sum(len(vec.get_vehicle_speed(e)) for e in range(envs)))
    assert shadows > 0


def test_vector_engine_lane_change_reset_twin(mod, scen, workdir):
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    vec = mod.VectorEngine._with_backend(cfg, 2, 1, TWIN_LIB)
    for _ in range(80):
        vec.next_step()
    a = vec.get_lane_vehicle_count_array().copy()
    sp = [vec.get_vehicle_speed(e) for e in range(2)]
    vec.reset(True)
    assert vec.get_vehicle_count() == 0
    for _ in range(80):
        vec.next_step()
    assert np.array_equal(a, vec.get_lane_vehicle_count_array())
    assert sp == [vec.get_vehicle_speed(e) for e in range(2)]


def test_vector_engine_reset(mod, scen, workdir):
    vec = mod.VectorEngine._with_backend(scen.materialize("example_1x1", workdir), 2, 1, TWIN_LIB)
    for _ in range(60):
        vec.next_step()
    a = vec.get_lane_vehicle_count_array().copy()
    vec.reset(True)
    assert vec.get_vehicle_count() == 0
    for _ in range(60):
        vec.next_step()
    assert np.array_equal(a, vec.get_lane_vehicle_count_array())


@pytest.mark.gpu
def test_vector_engine_hip(mod, scen, workdir):
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine(c, n, 1), lambda c: mod.Engine(c, 1), name="grid_6x6",
           envs=4, steps=200)


@pytest.mark.gpu
def test_vector_engine_hip_rl(mod, scen, workdir):
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine(c, n, 1), lambda c: mod.Engine(c, 1), name="grid_6x6",
           envs=3, steps=80, rl=True)


@pytest.mark.gpu
def test_vector_engine_lane_change_hip(mod, scen, workdir):
    """laneChange: true, 4 batched 6x6 environments on the HIP engine == 4 standalone HIP engines (which
    tests/test_lane_change.py pins to the twin and the reference) after every 5th step, and == the batched twin."""
    _check(mod, scen, workdir, lambda c, n: mod.VectorEngine(c, n, 1), lambda c: mod.Engine(c, 1), name="grid_6x6",
           envs=3, steps=470, lane_change=True, every=5)
    cfg = scen.materialize("example_1x1", workdir, laneChange=True)
    hip, twin = mod.VectorEngine(cfg, 5, 1), mod.VectorEngine._with_backend(cfg, 5, 1, TWIN_LIB)
    for s in range(200):
        hip.next_step()
        twin.next_step()
        assert np.array_equal(hip.get_lane_vehicle_count_array(), twin.get_lane_vehicle_count_array()), s
    for e in range(5):
        assert hip.get_vehicle_speed(e) == twin.get_vehicle_speed(e)


@pytest.mark.gpu
def test_vector_engine_hip_many_finishers(mod, scen, workdir):
    """96 environments of the 6x6 grid (identical flows, so their vehicles finish in the same steps): > 1024 vehicles
    finish per step from step 391 on, so the step's finish statistics (travel times
    added in the reference's order, engine.cpp:296-310) run as several blocks; the sum stays bit-identical to the twin."""
    cfg = scen.materialize("grid_6x6", workdir)
    hip, twin = mod.VectorEngine(cfg, 96, 1), mod.VectorEngine._with_backend(cfg, 96, 1, TWIN_LIB)
    prev, biggest = 0, 0
    for s in range(402):
        hip.next_step()
        twin.next_step()
        if s % 20 == 19 or s >= 388:
            a, b = hip._scalars(), twin._scalars()
            assert a == b, (s, a, b)
            if s > 388:  # compared every step there: finishers of ONE step
                biggest = max(biggest, a["finished_vehicle_count"] - prev)
            prev = a["finished_vehicle_count"]
    assert biggest > 1024
    assert np.array_equal(hip.get_lane_vehicle_count_array(), twin.get_lane_vehicle_count_array())


def _cfx_variant(path, suffix, **cfx):
    import json
    c = json.load(open(path))
    c["cfx"] = cfx
    out = path.replace(".json", "_%s.json" % suffix)
    with open(out, "w") as f:
        json.dump(c, f)
    return out


def test_vector_engine_batch_a_step_ahead_twin(mod, scen, workdir):
    """The batch of step t+1 is prepared on a host thread of its own while step t is submitted and runs (Flow::nextStep and
    planRoute, reference src/flow/flow.cpp:6-22, src/engine/engine.cpp:450-470, depend on nothing the step computes).  On against
    off (`"cfx": {"spawnAhead": false}`): every count, every speed of every environment, through resets that do and do not
    reseed — a reset that keeps the generators must find them where the last step that was TAKEN left them, not behind the
    step that was prepared."""
    base = scen.materialize("example_1x1", workdir)
    on = mod.VectorEngine._with_backend(base, 4, 1, TWIN_LIB)
    off = mod.VectorEngine._with_backend(_cfx_variant(base, "noahead", spawnAhead=False), 4, 1, TWIN_LIB)
    plan = [(70, None), (45, False), (60, True), (30, False)]  # (steps, reset(reseed) afterwards)
    for steps, reseed in plan:
        for s in range(steps):
            on.next_step()
            off.next_step()
            if s % 9 == 8:  # getters between steps must not disturb the prepared batch
                assert np.array_equal(on.get_lane_vehicle_count_array(), off.get_lane_vehicle_count_array()), s
                assert np.array_equal(on.get_lane_waiting_vehicle_count_array(), off.get_lane_waiting_vehicle_count_array()), s
        for e in range(4):
            assert on.get_vehicle_speed(e) == off.get_vehicle_speed(e), e
        assert on.get_vehicle_count() == off.get_vehicle_count() > 0
        h = on._host_seconds()
        assert len(h) == 4 and h[3] > 0.0  # the ahead thread did the spawners' work ...
        assert off._host_seconds()[3] == 0.0  # ... and only where it is on
        if reseed is not None:
            on.reset(reseed)
            off.reset(reseed)
            assert on.get_vehicle_count() == 0 and on.get_current_time() == 0.0
