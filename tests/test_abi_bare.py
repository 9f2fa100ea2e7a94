"""The C ABI of include/cityflow_amd.h driven with nothing but ctypes: a road network written out by hand (no JSON, no C++
host, no pybind11), `cfx_create -> cfx_add_templates -> cfx_add_routes -> cfx_step -> cfx_get_lane_counts /
cfx_get_vehicles / cfx_get_scalars -> cfx_reset -> cfx_destroy`.  This is the call sequence INTEGRATION.md shows for a
maintainer binding the reference's Engine::nextStep (src/engine/engine.cpp:566-594) to the library.

CPU: the twin (oracle) behind the same ABI.  GPU (-m gpu): the HIP library, every step compared with the twin bit for bit.

The network: road 0 (one lane, 100 m) -> intersection 1 (one go-straight roadLink, one 20 m laneLink, one always-green
phase) -> road 1 (one lane, 100 m); intersections 0 and 2 are virtual.  One vehicle template (the generator's), one route.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, TWIN_LIB

c_i32p, c_f64p, c_u8p = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)


class CfxNet(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_roads", "n_lanes", "n_lanelinks", "n_inters", "n_xentries", "n_phases", "n_avail")] + [
        ("drv_length", c_f64p), ("drv_max_speed", c_f64p),
        ("lane_road", c_i32p), ("lane_index", c_i32p), ("lane_ll_start", c_i32p), ("lane_ll", c_i32p),
        ("road_lane_start", c_i32p),
        ("ll_start_lane", c_i32p), ("ll_end_lane", c_i32p), ("ll_inter", c_i32p), ("ll_roadlink", c_i32p), ("ll_type", c_i32p),
        ("ll_x_start", c_i32p),
        ("x_dist", c_f64p), ("x_peer", c_i32p), ("x_ll", c_i32p),
        ("inter_virtual", c_i32p), ("inter_n_roadlinks", c_i32p), ("inter_phase_start", c_i32p), ("inter_avail_start", c_i32p),
        ("phase_time", c_f64p), ("phase_avail", c_u8p),
        ("lane_width", c_f64p), ("lane_n_segments", c_i32p)]


class CfxConfig(C.Structure):
    _fields_ = [("interval", C.c_double)] + [(n, C.c_int32) for n in (
        "rl_traffic_light", "lane_change", "device", "cross_mode", "layout", "debug_sync", "ring_lanes_per_wave",
        "ring_capacity_percent", "lane_history", "n_envs", "dense_form")]


class CfxTemplate(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("len", "width", "max_pos_acc", "max_neg_acc", "usual_pos_acc", "usual_neg_acc", "min_gap",
                                          "max_speed", "headway_time", "yield_distance", "turn_speed", "approach_dist", "initial_speed")]


class CfxSpawn(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vid", "priority", "templ", "route", "lane", "prev_wait")] + [("enter_time", C.c_double)]


class CfxScalars(C.Structure):
    _fields_ = [("step", C.c_int64), ("active_vehicle_count", C.c_int64), ("finished_vehicle_count", C.c_int64),
                ("spawned_vehicle_count", C.c_int64), ("cumulative_travel_time", C.c_double), ("live_enter_time_sum", C.c_double),
                ("vehicle_steps", C.c_int64), ("tie_events", C.c_int64), ("tie_drivables", C.c_int32 * 8),
                ("diag_cross_jobs", C.c_int32), ("dropped_future_speeds", C.c_int32)]


class CfxVehicleView(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("count", C.c_int32)] + [(n, c_i32p) for n in (
        "vid", "drivable", "prev_drivable", "leader_vid", "blocker_vid", "enter_ll_time", "route_pos")] + [
        ("dis", c_f64p), ("speed", c_f64p), ("gap", c_f64p),
        ("lc_partner_vid", c_i32p), ("lc_flags", c_u8p), ("lc_offset", c_f64p), ("lc_last_dir", c_i32p), ("lc_target_lane", c_i32p),
        ("lc_direction", c_i32p), ("lc_last_change_time", c_f64p), ("lc_waiting_time", c_f64p)]


def _i32(*v):
    return np.array(v, dtype=np.int32)


def _f64(*v):
    return np.array(v, dtype=np.float64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Corridor:
    """The hand-written network and one engine on it, ctypes only."""

    def __init__(self, path):
        self.dll = d = C.CDLL(path)
        d.cfx_last_error.restype = C.c_char_p
        d.cfx_last_error.argtypes = [C.c_void_p]
        d.cfx_create.argtypes = [C.POINTER(CfxNet), C.POINTER(CfxConfig), C.POINTER(C.c_void_p)]
        d.cfx_destroy.argtypes = [C.c_void_p]
        d.cfx_destroy.restype = None
        d.cfx_add_templates.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CfxTemplate)]
        d.cfx_add_routes.argtypes = [C.c_void_p, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p]
        d.cfx_step.argtypes = [C.c_void_p, C.POINTER(CfxSpawn), C.c_int32]
        d.cfx_sync.argtypes = [C.c_void_p]
        d.cfx_reset.argtypes = [C.c_void_p]
        d.cfx_get_scalars.argtypes = [C.c_void_p, C.POINTER(CfxScalars)]
        d.cfx_get_lane_counts.argtypes = [C.c_void_p, c_i32p]
        d.cfx_get_vehicles.argtypes = [C.c_void_p, C.POINTER(CfxVehicleView)]
        d.cfx_get_tl_state.argtypes = [C.c_void_p, c_i32p, c_f64p]
        # the arrays must outlive cfx_create only (it copies them); kept on self anyway
        self.keep = k = {
            "drv_length": _f64(100.0, 100.0, 20.0), "drv_max_speed": _f64(16.67, 16.67, 10000.0),
            "lane_road": _i32(0, 1), "lane_index": _i32(0, 0), "lane_ll_start": _i32(0, 1, 1), "lane_ll": _i32(0),
            "road_lane_start": _i32(0, 1, 2),
            "ll_start_lane": _i32(0), "ll_end_lane": _i32(1), "ll_inter": _i32(1), "ll_roadlink": _i32(0), "ll_type": _i32(3),
            "ll_x_start": _i32(0, 0),
            "x_dist": _f64(0.0), "x_peer": _i32(0), "x_ll": _i32(0),  # (no crosses: n_xentries = 0; the pointers stay valid)
            "inter_virtual": _i32(1, 0, 1), "inter_n_roadlinks": _i32(0, 1, 0), "inter_phase_start": _i32(0, 0, 1, 1),
            "inter_avail_start": _i32(0, 0, 1), "phase_time": _f64(30.0), "phase_avail": np.array([1], dtype=np.uint8),
            "lane_width": _f64(4.0, 4.0), "lane_n_segments": _i32(1, 1)}
        net = CfxNet(n_roads=2, n_lanes=2, n_lanelinks=1, n_inters=3, n_xentries=0, n_phases=1, n_avail=1)
        for name, typ in CfxNet._fields_[7:]:
            setattr(net, name, _ptr(k[name], typ))
        cfg = CfxConfig(interval=1.0)
        self.h = C.c_void_p()
        rc = d.cfx_create(C.byref(net), C.byref(cfg), C.byref(self.h))
        assert rc == 0 and self.h, (rc, d.cfx_last_error(None))
        max_speed, usual_neg = 16.67, 4.5
        t = CfxTemplate(len=5.0, width=2.0, max_pos_acc=2.0, max_neg_acc=4.5, usual_pos_acc=2.0, usual_neg_acc=usual_neg, min_gap=2.5,
                        max_speed=max_speed, headway_time=1.5, yield_distance=5.0, turn_speed=8.3333,
                        approach_dist=max_speed * max_speed / usual_neg / 2 + max_speed * 1.0 * 2, initial_speed=0.0)
        self.ok(d.cfx_add_templates(self.h, 1, C.byref(t)))
        rs, roads, ns, nll = _i32(0, 2), _i32(0, 1), _i32(0, 1, 2), _i32(0, -1)
        self.ok(d.cfx_add_routes(self.h, 1, _ptr(rs, c_i32p), _ptr(roads, c_i32p), _ptr(ns, c_i32p), _ptr(nll, c_i32p)))
        self.spawned = 0
        self.last_on_lane0 = -1

    def ok(self, rc):
        assert rc == 0, (rc, self.dll.cfx_last_error(self.h))

    def step(self, spawn_priority=None, time=0.0):
        if spawn_priority is None:
            self.ok(self.dll.cfx_step(self.h, None, 0))
            return
        rec = CfxSpawn(vid=self.spawned, priority=spawn_priority, templ=0, route=0, lane=0, prev_wait=self.last_on_lane0,
                       enter_time=time)
        self.ok(self.dll.cfx_step(self.h, C.byref(rec), 1))
        self.last_on_lane0 = self.spawned
        self.spawned += 1

    def lane_counts(self):
        out = np.zeros(2, dtype=np.int32)
        self.ok(self.dll.cfx_get_lane_counts(self.h, _ptr(out, c_i32p)))
        return out.tolist()

    def scalars(self):
        s = CfxScalars()
        self.ok(self.dll.cfx_get_scalars(self.h, C.byref(s)))
        return s

    def vehicles(self):
        cap = 16
        a = {n: np.zeros(cap, dtype=np.int32) for n in ("vid", "drivable", "route_pos")}
        f = {n: np.zeros(cap, dtype=np.float64) for n in ("dis", "speed")}
        v = CfxVehicleView(capacity=cap)
        for n, arr in a.items():
            setattr(v, n, _ptr(arr, c_i32p))
        for n, arr in f.items():
            setattr(v, n, _ptr(arr, c_f64p))
        self.ok(self.dll.cfx_get_vehicles(self.h, C.byref(v)))
        n = v.count
        return {k: x[:n].copy() for k, x in list(a.items()) + list(f.items())}

    def close(self):
        if self.h:
            self.dll.cfx_destroy(self.h)
            self.h = None


def _drive(engines, steps=70):
    """Three vehicles enter at steps 0, 3 and 6; everything is compared between the engines after every step."""
    trace = []
    for s in range(steps):
        for e in engines:
            e.step(spawn_priority=1000 + s if s in (0, 3, 6) else None, time=float(s))
        recs = [(e.lane_counts(), e.vehicles(), e.scalars()) for e in engines]
        lc0, v0, s0 = recs[0]
        for lc, v, sc in recs[1:]:
            assert lc == lc0, s
            for k in v0:
                assert np.array_equal(v[k], v0[k]), (s, k, v[k], v0[k])  # bit for bit, doubles included
            assert (sc.step, sc.active_vehicle_count, sc.finished_vehicle_count, sc.vehicle_steps) == \
                   (s0.step, s0.active_vehicle_count, s0.finished_vehicle_count, s0.vehicle_steps)
            assert sc.cumulative_travel_time == s0.cumulative_travel_time
        trace.append((lc0, v0, s0.active_vehicle_count, s0.finished_vehicle_count))
    return trace


def _check_corridor_facts(trace):
    """What anybody can work out by hand for this network (Engine::nextStep on an empty, always-green corridor)."""
    lc, v, active, finished = trace[0]
    assert lc == [1, 0] and active == 1 and finished == 0  # admitted in the step that created it (engine.cpp:317-330)
    assert v["vid"].tolist() == [0] and v["drivable"].tolist() == [0]
    # Vehicle starts at rest at distance 0; its first step accelerates by maxPosAcc * interval: v = 2, moved (0 + 2) / 2 = 1 m
    assert v["speed"][0] == 2.0 and v["dis"][0] == 1.0
    seen_on = {0: False, 1: False, 2: False}  # lane 0, lane 1, the laneLink (drivable 2 = n_lanes + 0)
    last_dis = {}
    for lc, v, active, finished in trace:
        for vid, drv, dis in zip(v["vid"].tolist(), v["drivable"].tolist(), v["dis"].tolist()):
            seen_on[drv] = True
            key = (vid, drv)
            assert dis >= last_dis.get(key, 0.0)  # nobody moves backwards
            last_dis[key] = dis
        assert lc[0] + lc[1] <= active  # (vehicles on the laneLink are counted on no lane)
        on = v["drivable"].tolist()
        assert on == sorted(on)  # the view is ordered by drivable, front to back inside one
    assert all(seen_on.values())
    assert trace[-1][2] == 0 and trace[-1][3] == 3  # 220 m at <= 16.67 m/s: all three are through after 70 s


def test_bare_abi_on_the_cpu_twin():
    e = Corridor(TWIN_LIB)
    try:
        trace = _drive([e])
        _check_corridor_facts(trace)
        e.ok(e.dll.cfx_reset(e.h))
        s = e.scalars()
        assert (s.step, s.active_vehicle_count, s.finished_vehicle_count) == (0, 0, 0)
        assert e.lane_counts() == [0, 0]
    finally:
        e.close()


@pytest.mark.gpu
def test_bare_abi_on_the_hip_library_equals_twin(mod):
    path = mod._default_backend_path()
    assert os.path.exists(path), path
    hip, twin = Corridor(path), Corridor(TWIN_LIB)
    try:
        hip.dll.cfx_backend_name.restype = C.c_char_p
        assert hip.dll.cfx_backend_name() == b"hip-gfx950"
        trace = _drive([hip, twin])
        _check_corridor_facts(trace)
        for e in (hip, twin):  # reset, and the same again: Engine::reset (engine.cpp:744-760)
            e.ok(e.dll.cfx_reset(e.h))
            e.spawned, e.last_on_lane0 = 0, -1
        again = _drive([hip, twin])
        for (lc_a, v_a, *_), (lc_b, v_b, *_) in zip(trace, again):
            assert lc_a == lc_b and all(np.array_equal(v_a[k], v_b[k]) for k in v_a)
    finally:
        hip.close()
        twin.close()
