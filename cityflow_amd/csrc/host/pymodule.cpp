// pybind11 binding of the drop-in `cityflow` module.  Same class name, method names, argument names and
// defaults as the reference binding (reference src/cityflow.cpp:10-47); extra members are prefixed or
// clearly array-flavoured and never change the behaviour of the reference-named ones.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <chrono>
#include <iostream>

#include "archive.h"
#include "engine_host.h"
#include "tile_engine.h"
#include "vector_engine.h"
#include "json.h"

namespace py = pybind11;
using namespace py::literals;
using cfa::EngineHost;

namespace {

template <typename T> py::array_t<T> toArray(const std::vector<T> &v) {
    py::array_t<T> a((py::ssize_t) v.size());
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(T));
    return a;
}
template <typename T> py::array_t<T> toArray(const T *p, size_t n) {
    py::array_t<T> a((py::ssize_t) n);
    if (n) std::memcpy(a.mutable_data(), p, n * sizeof(T));
    return a;
}

py::dict flatNetToDict(const cfa::HostRoadNet &net) {
    const cfx_net &f = net.flat();
    const int D = f.n_lanes + f.n_lanelinks;
    py::dict d;
    d["n_roads"] = f.n_roads;
    d["n_lanes"] = f.n_lanes;
    d["n_lanelinks"] = f.n_lanelinks;
    d["n_inters"] = f.n_inters;
    d["n_xentries"] = f.n_xentries;
    d["drv_length"] = toArray(f.drv_length, D);
    d["drv_max_speed"] = toArray(f.drv_max_speed, D);
    d["lane_road"] = toArray(f.lane_road, f.n_lanes);
    d["lane_index"] = toArray(f.lane_index, f.n_lanes);
    d["lane_ll_start"] = toArray(f.lane_ll_start, f.n_lanes + 1);
    d["lane_ll"] = toArray(f.lane_ll, f.n_lanelinks);
    d["road_lane_start"] = toArray(f.road_lane_start, f.n_roads + 1);
    d["ll_start_lane"] = toArray(f.ll_start_lane, f.n_lanelinks);
    d["ll_end_lane"] = toArray(f.ll_end_lane, f.n_lanelinks);
    d["ll_inter"] = toArray(f.ll_inter, f.n_lanelinks);
    d["ll_roadlink"] = toArray(f.ll_roadlink, f.n_lanelinks);
    d["ll_type"] = toArray(f.ll_type, f.n_lanelinks);
    d["ll_x_start"] = toArray(f.ll_x_start, f.n_lanelinks + 1);
    d["x_dist"] = toArray(f.x_dist, f.n_xentries);
    d["x_peer"] = toArray(f.x_peer, f.n_xentries);
    d["x_ll"] = toArray(f.x_ll, f.n_xentries);
    d["inter_virtual"] = toArray(f.inter_virtual, f.n_inters);
    d["inter_n_roadlinks"] = toArray(f.inter_n_roadlinks, f.n_inters);
    d["inter_phase_start"] = toArray(f.inter_phase_start, f.n_inters + 1);
    d["inter_avail_start"] = toArray(f.inter_avail_start, f.n_inters);
    d["phase_time"] = toArray(f.phase_time, f.n_phases);
    d["phase_avail"] = toArray(f.phase_avail, f.n_avail);
    std::vector<std::string> laneIds(f.n_lanes), llIds(f.n_lanelinks), interIds(f.n_inters), roadIds(f.n_roads);
    for (int l = 0; l < f.n_lanes; ++l) laneIds[l] = net.laneId(l);
    for (int k = 0; k < f.n_lanelinks; ++k) llIds[k] = net.laneLinkId(k);
    for (int i = 0; i < f.n_inters; ++i) interIds[i] = net.inters[i].id;
    for (int r = 0; r < f.n_roads; ++r) roadIds[r] = net.roads[r].id;
    d["lane_ids"] = laneIds;
    d["lanelink_ids"] = llIds;
    d["inter_ids"] = interIds;
    d["road_ids"] = roadIds;
    return d;
}

// Reference getLaneVehicleCount returns std::map<std::string,int> (engine.cpp:628-648) which pybind11 turns into a dict
// by creating one Python string per lane and inserting every entry per call.  Here the key objects are created once per
// engine, a master dict per getter is kept up to date by rewriting only the entries whose value changed since the
// previous call (a step changes a small fraction of the lanes), and the caller gets a C-level copy of it: a fresh dict
// in std::map (lexicographic) key order every call, like the reference's.
struct LaneDictCache {
    py::list keys;
    py::object interIdx;  // {intersection id: index in net().inters}, for set_tl_phase (built at its first call)
    py::dict master[2];
    std::vector<int32_t> last[2];
    bool filled[2] = {false, false};
    // vehicle id strings, created once per vehicle (valid for one vehicleEpoch of the engine)
    std::vector<PyObject *> vidStr;
    std::vector<uint64_t> vidSortKey;  // Spawner::idSortKey, computed once per vehicle (never 0)
    uint64_t vidEpoch = 0;
    ~LaneDictCache() { clearVehicles(); }
    void clearVehicles() {
        for (PyObject *o : vidStr) Py_XDECREF(o);
        vidStr.clear();
        vidSortKey.clear();
    }
};

LaneDictCache &bindingCacheOf(EngineHost &e) {
    if (!e.bindingCache) {
        auto *c = new LaneDictCache();
        for (int32_t l : e.laneIdOrder()) c->keys.append(py::str(e.net().laneId(l)));
        e.bindingCache = std::shared_ptr<void>(c, [](void *p) { delete static_cast<LaneDictCache *>(p); });
    }
    return *static_cast<LaneDictCache *>(e.bindingCache.get());
}

// borrowed reference to the (cached) Python string of a vehicle's id
PyObject *vehicleKey(EngineHost &e, LaneDictCache &c, int vid) {
    if (c.vidEpoch != e.vehicleEpoch()) {
        c.clearVehicles();
        c.vidEpoch = e.vehicleEpoch();
    }
    if ((size_t) vid >= c.vidStr.size()) c.vidStr.resize((size_t) vid + 1 + c.vidStr.size() / 2, nullptr);
    PyObject *&o = c.vidStr[vid];
    if (!o) {
        const std::string id = e.vehicleId(vid);
        o = PyUnicode_FromStringAndSize(id.data(), (Py_ssize_t) id.size());
        if (!o) throw py::error_already_set();
    }
    return o;
}

uint64_t vehicleSortKey(EngineHost &e, LaneDictCache &c, int vid) {
    if (c.vidEpoch != e.vehicleEpoch()) {
        c.clearVehicles();
        c.vidEpoch = e.vehicleEpoch();
    }
    if ((size_t) vid >= c.vidSortKey.size()) c.vidSortKey.resize((size_t) vid + 1 + c.vidSortKey.size() / 2, 0);
    uint64_t &k = c.vidSortKey[vid];
    if (!k) k = e.spawner().idSortKey(vid);
    return k;
}

// get_vehicle_speed / get_vehicle_distance (engine.cpp:650-676): {vehicle id: value} in std::map (string) key order
py::dict vehicleValueDict(EngineHost &e, bool speed) {
    cfa::VehicleSnapshot s;
    e.snapshotVehicles(s, speed ? cfa::kSnapSpeed : cfa::kSnapDis);
    const std::vector<double> &val = speed ? s.speed : s.dis;
    LaneDictCache &c = bindingCacheOf(e);
    std::vector<std::pair<uint64_t, int>> order((size_t) s.count);
    for (int i = 0; i < s.count; ++i) order[i] = std::make_pair(vehicleSortKey(e, c, s.vid[i]), i);
    std::sort(order.begin(), order.end());
    py::dict d;
    for (auto &p : order) {
        PyObject *num = PyFloat_FromDouble(val[p.second]);
        if (!num || PyDict_SetItem(d.ptr(), vehicleKey(e, c, s.vid[p.second]), num) != 0) {
            Py_XDECREF(num);
            throw py::error_already_set();
        }
        Py_DECREF(num);
    }
    return d;
}

// get_vehicles (engine.cpp:619-626): ids in vehiclePool (= priority) order
py::list vehicleList(EngineHost &e, bool includeWaiting) {
    cfa::VehicleSnapshot s;
    e.snapshotVehicles(s, 0);
    std::vector<std::pair<int32_t, int32_t>> byPriority;
    byPriority.reserve((size_t) s.count);
    for (int i = 0; i < s.count; ++i) byPriority.emplace_back(e.spawner().vehicles[s.vid[i]].priority, s.vid[i]);
    if (includeWaiting) {
        std::vector<int32_t> wv, wl;
        e.waitingVehicles(wv, wl);
        for (int32_t v : wv) byPriority.emplace_back(e.spawner().vehicles[v].priority, v);
    }
    std::sort(byPriority.begin(), byPriority.end());
    LaneDictCache &c = bindingCacheOf(e);
    py::list out(byPriority.size());
    for (size_t i = 0; i < byPriority.size(); ++i) {
        PyObject *key = vehicleKey(e, c, byPriority[i].second);
        Py_INCREF(key);
        PyList_SET_ITEM(out.ptr(), (Py_ssize_t) i, key);
    }
    return out;
}

// get_lane_vehicles (engine.cpp:658-668): {lane id: [vehicle ids front to back]} for every lane, std::map key order
py::dict laneVehiclesDict(EngineHost &e) {
    cfa::VehicleSnapshot s;
    e.snapshotVehicles(s, cfa::kSnapDrivable);
    const std::vector<int32_t> &order = e.laneIdOrder();
    const int L = (int) order.size();
    std::vector<int32_t> start((size_t) L + 1, 0);
    for (int i = 0; i < s.count; ++i)
        if (s.drivable[i] < L) start[s.drivable[i] + 1]++;
    for (int l = 0; l < L; ++l) start[l + 1] += start[l];
    // the snapshot is ordered by drivable, front to back: the vehicles of lane l are a contiguous run
    std::vector<int32_t> first((size_t) L, -1);
    for (int i = s.count - 1; i >= 0; --i)
        if (s.drivable[i] < L) first[s.drivable[i]] = i;
    LaneDictCache &c = bindingCacheOf(e);
    py::dict d;
    for (int k = 0; k < L; ++k) {
        const int l = order[k];
        const int n = start[l + 1] - start[l];
        PyObject *lst = PyList_New(n);
        if (!lst) throw py::error_already_set();
        for (int j = 0; j < n; ++j) {
            PyObject *key = vehicleKey(e, c, s.vid[first[l] + j]);
            Py_INCREF(key);
            PyList_SET_ITEM(lst, j, key);
        }
        if (PyDict_SetItem(d.ptr(), PyList_GET_ITEM(c.keys.ptr(), (Py_ssize_t) k), lst) != 0) {
            Py_DECREF(lst);
            throw py::error_already_set();
        }
        Py_DECREF(lst);
    }
    return d;
}

py::dict laneDict(EngineHost &e, const std::vector<int32_t> &values, int which) {
    const std::vector<int32_t> &order = e.laneIdOrder();
    LaneDictCache &c = bindingCacheOf(e);
    std::vector<int32_t> &last = c.last[which];
    if (!c.filled[which]) last.assign(order.size(), INT32_MIN);
    PyObject *master = c.master[which].ptr();
    for (size_t i = 0; i < order.size(); ++i) {
        const int32_t v = values[order[i]];
        if (v == last[i]) continue;
        last[i] = v;
        PyObject *num = PyLong_FromLong(v);
        if (!num || PyDict_SetItem(master, PyList_GET_ITEM(c.keys.ptr(), (Py_ssize_t) i), num) != 0) {
            Py_XDECREF(num);
            throw py::error_already_set();
        }
        Py_DECREF(num);
    }
    c.filled[which] = true;
    PyObject *copy = PyDict_Copy(master);
    if (!copy) throw py::error_already_set();
    return py::reinterpret_steal<py::dict>(copy);
}

// Engine.set_tl_phase(intersection_id, phase_id) (reference src/cityflow.cpp:35, engine.cpp:719-725) as a vectorcall method of its
// own.  An RL agent written for the reference calls it once per signal and step — 900 calls per step on the 30x30 grid — and
// the generic binding path (argument records, a std::string per call, a std::map lookup by string comparison) costs ~0.35 us
// per call: more than the whole device step for the 900 of them.  Here: the id's Python string is looked up in a dict kept with
// the engine (its hash is cached in the string object), ~0.1 us per call.  Same arguments (positional or by the reference's
// names), same errors (RuntimeError for an unknown intersection, IndexError for a phase out of range, TypeError otherwise).
PyObject *engineSetTlPhase(PyObject *self, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    try {
        PyObject *idObj = nargs >= 1 ? args[0] : nullptr, *phObj = nargs >= 2 ? args[1] : nullptr;
        const Py_ssize_t nkw = kwnames ? PyTuple_GET_SIZE(kwnames) : 0;
        bool bad = nargs > 2;
        for (Py_ssize_t i = 0; i < nkw && !bad; ++i) {
            PyObject *name = PyTuple_GET_ITEM(kwnames, i);
            if (PyUnicode_CompareWithASCIIString(name, "intersection_id") == 0 && !idObj) idObj = args[nargs + i];
            else if (PyUnicode_CompareWithASCIIString(name, "phase_id") == 0 && !phObj) phObj = args[nargs + i];
            else bad = true;
        }
        if (bad || !idObj || !phObj || !PyUnicode_Check(idObj) || PyFloat_Check(phObj) || !PyIndex_Check(phObj)) {
            PyErr_SetString(PyExc_TypeError, "set_tl_phase(intersection_id: str, phase_id: int): incompatible arguments");
            return nullptr;
        }
        // (the method descriptor has checked that `self` is an Engine; the generic caster's type lookup is a third of the call)
        auto *inst = reinterpret_cast<py::detail::instance *>(self);
        EngineHost *ep = inst->simple_layout ? static_cast<EngineHost *>(inst->simple_value_holder[0]) : nullptr;
        EngineHost &e = ep ? *ep : py::cast<EngineHost &>(py::handle(self));
        long ph;
        if (PyLong_CheckExact(phObj)) {
            ph = PyLong_AsLong(phObj);
        } else {
            PyObject *phIdx = PyNumber_Index(phObj);
            if (!phIdx) return nullptr;
            ph = PyLong_AsLong(phIdx);
            Py_DECREF(phIdx);
        }
        if (ph == -1 && PyErr_Occurred()) return nullptr;
        if (ph < INT32_MIN || ph > INT32_MAX) {
            PyErr_SetString(PyExc_TypeError, "set_tl_phase: phase_id does not fit an int");
            return nullptr;
        }
        LaneDictCache &c = bindingCacheOf(e);
        if (!c.interIdx) {
            py::dict d;
            for (size_t i = 0; i < e.net().inters.size(); ++i) d[py::str(e.net().inters[i].id)] = py::int_(i);
            c.interIdx = std::move(d);
        }
        PyObject *ix = PyDict_GetItemWithError(c.interIdx.ptr(), idObj);  // borrowed
        if (!ix && PyErr_Occurred()) return nullptr;
        if (ix) {
            e.setTrafficLightPhaseOf((int) PyLong_AsLong(ix), (int) ph);
        } else {  // (not an intersection: the message comes from where the reference's comes from)
            Py_ssize_t len = 0;
            const char *utf8 = PyUnicode_AsUTF8AndSize(idObj, &len);
            if (!utf8) return nullptr;
            e.setTrafficLightPhase(std::string(utf8, (size_t) len), (int) ph);
        }
        Py_RETURN_NONE;
    } catch (py::error_already_set &err) {
        err.restore();
    } catch (const py::cast_error &err) {
        PyErr_SetString(PyExc_TypeError, err.what());
    } catch (const std::out_of_range &err) {
        PyErr_SetString(PyExc_IndexError, err.what());
    } catch (const std::exception &err) {
        PyErr_SetString(PyExc_RuntimeError, err.what());
    }
    return nullptr;
}
PyMethodDef kEngineSetTlPhaseDef = {"set_tl_phase", (PyCFunction) (void (*)(void)) engineSetTlPhase, METH_FASTCALL | METH_KEYWORDS,
                                    "set_tl_phase(self, intersection_id: str, phase_id: int) -> None"};

// Host-only helpers (no device engine involved): used by the CPU test-suite to pin the loader and the
// spawner against the reference.
py::dict loadRoadnet(const std::string &path) {
    cfa::HostRoadNet net;
    net.load(path);
    return flatNetToDict(net);
}

// Same text format as oracle/probe_roadnet.cpp prints from the reference's own loader.
py::bytes roadnetProbe(const std::string &path) {
    cfa::HostRoadNet net;
    net.load(path);
    const cfx_net &f = net.flat();
    std::string out;
    char buf[512];
    for (int l = 0; l < f.n_lanes; ++l) {
        snprintf(buf, sizeof buf, "L %s %.17g %.17g %.17g %zu\n", net.laneId(l).c_str(), f.drv_length[l],
                 f.drv_max_speed[l], net.lanes[l].width, net.lanes[l].laneLinks.size());
        out += buf;
    }
    for (int k = 0; k < f.n_lanelinks; ++k) {
        snprintf(buf, sizeof buf, "K %s %.17g %d %d\n", net.laneLinkId(k).c_str(), f.drv_length[f.n_lanes + k],
                 f.ll_type[k], f.ll_x_start[k + 1] - f.ll_x_start[k]);
        out += buf;
        for (int e = f.ll_x_start[k]; e < f.ll_x_start[k + 1]; ++e) {
            snprintf(buf, sizeof buf, "X %.17g %s %.17g\n", f.x_dist[e], net.laneLinkId(f.x_ll[f.x_peer[e]]).c_str(),
                     f.x_dist[f.x_peer[e]]);
            out += buf;
        }
    }
    for (auto &in : net.inters) {
        snprintf(buf, sizeof buf, "T %s %d %zu\n", in.id.c_str(), (int) in.isVirtual, in.phases.size());
        out += buf;
    }
    return py::bytes(out);
}

// ahead: 0 every step taken plainly; 1 every step taken ahead and consumed (Spawner::beginAhead / commitAhead); 2 every step
// taken ahead, taken back (rollbackAhead) and then taken plainly — what a call between two steps does to the host's step ahead
py::list spawnSchedule(const std::string &roadnetFile, const std::string &flowFile, double interval, int seed,
                       int threadNum, int steps, int ahead) {
    cfa::HostRoadNet net;
    net.load(roadnetFile);
    cfa::Spawner sp;
    sp.init(&net, interval, threadNum, seed);
    sp.loadFlows(flowFile);
    py::list out;
    std::vector<cfx_spawn> recs;
    for (int s = 0; s < steps; ++s) {
        if (ahead) {
            sp.beginAhead();
            sp.step((size_t) s, recs);
            if (ahead == 2) {
                sp.rollbackAhead();
                sp.step((size_t) s, recs);
            } else {
                sp.commitAhead();
            }
        } else {
            sp.step((size_t) s, recs);
        }
        py::list stepList;
        for (const cfx_spawn &r : recs)
            stepList.append(py::make_tuple(sp.vehicleId(r.vid), r.priority, net.laneId(r.lane), r.enter_time, r.templ,
                                           r.route, r.prev_wait));
        out.append(stepList);
    }
    return out;
}

// seconds spent in `steps` Spawner::step calls (host-side cost of phases 0-1, no device, no Python objects)
double spawnBenchmark(const std::string &roadnetFile, const std::string &flowFile, double interval, int seed, int skip,
                      int steps) {
    cfa::HostRoadNet net;
    net.load(roadnetFile);
    cfa::Spawner sp;
    sp.init(&net, interval, 1, seed);
    sp.loadFlows(flowFile);
    std::vector<cfx_spawn> recs;
    for (int s = 0; s < skip; ++s) sp.step((size_t) s, recs);
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) sp.step((size_t) (skip + s), recs);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

PYBIND11_MODULE(_cityflow, m) {
    m.doc() = "MI355X-native CityFlow step engine (drop-in for the reference `cityflow` module)";

    py::class_<EngineHost> engineClass(m, "Engine");
    engineClass
        .def(py::init<const std::string &, int>(), "config_file"_a, "thread_num"_a = 1)
        .def_static(
            "_with_backend",
            [](const std::string &cfg, int threads, const std::string &lib) {
                return std::unique_ptr<EngineHost>(new EngineHost(cfg, threads, lib));
            },
            "config_file"_a, "thread_num"_a, "backend_library"_a,
            "Test hook: build an engine on any shared library exporting the cfx_* C ABI.")
        // ---- reference API ----
        .def("next_step", &EngineHost::nextStep)
        .def("get_vehicle_count", &EngineHost::getVehicleCount)
        .def("get_vehicles", [](EngineHost &e, bool w) -> py::object {
                 if (e.laneChange()) return py::cast(e.getVehicles(w));  // ids change with state ("_shadow"): no id cache
                 if (w) {  // vehicles pushed since the last step have no vehicle number yet (EngineHost::getVehicles lists them)
                     std::vector<std::pair<int32_t, std::string>> pushed;
                     e.spawner().pendingPushed(pushed);
                     if (!pushed.empty()) return py::cast(e.getVehicles(w));
                 }
                 return vehicleList(e, w);
             }, "include_waiting"_a = false)
        // dict[str, int] in std::map (lexicographic) key order like the reference, built from cached key objects
        .def("get_lane_vehicle_count", [](EngineHost &e) { return laneDict(e, e.laneVehicleCountArray(), 0); })
        .def("get_lane_waiting_vehicle_count", [](EngineHost &e) { return laneDict(e, e.laneWaitingVehicleCountArray(), 1); })
        .def("get_lane_vehicles", [](EngineHost &e) -> py::object {
                 if (e.laneChange()) return py::cast(e.getLaneVehicles());
                 return laneVehiclesDict(e);
             })
        .def("get_vehicle_speed", [](EngineHost &e) -> py::object {
                 if (e.laneChange()) return py::cast(e.getVehicleSpeed());
                 return vehicleValueDict(e, true);
             })
        .def("get_vehicle_info", &EngineHost::getVehicleInfo, "vehicle_id"_a)
        .def("get_vehicle_distance", [](EngineHost &e) -> py::object {
                 if (e.laneChange()) return py::cast(e.getVehicleDistance());
                 return vehicleValueDict(e, false);
             })
        .def("get_leader", &EngineHost::getLeader, "vehicle_id"_a)
        .def("get_current_time", &EngineHost::getCurrentTime)
        .def("get_average_travel_time", &EngineHost::getAverageTravelTime)
        // (set_tl_phase: engineSetTlPhase, attached below)
        .def("set_random_seed", &EngineHost::setRandomSeed, "seed"_a)
        .def("push_vehicle", &EngineHost::pushVehicle)
        .def("reset", &EngineHost::reset, "seed"_a = false)
        .def("set_vehicle_speed", &EngineHost::setVehicleSpeed, "vehicle_id"_a, "speed"_a)
        .def("set_vehicle_route", &EngineHost::setRoute, "vehicle_id"_a, "route"_a)
        .def("load", &EngineHost::load, "archive"_a)
        .def("snapshot", &EngineHost::snapshot)
        .def("load_from_file", &EngineHost::loadFromFile, "path"_a)
        .def("set_replay_file", &EngineHost::setReplayLogFile, "replay_file"_a)
        .def("set_save_replay", &EngineHost::setSaveReplay, "open"_a)
        // ---- array API (index order == lane_ids() / intersection_ids()) ----
        .def("lane_ids", &EngineHost::laneIds)
        .def("intersection_ids", &EngineHost::intersectionIds)
        .def("get_lane_vehicle_count_array", [](EngineHost &e) { return toArray(e.laneVehicleCountArray()); })
        .def("get_lane_waiting_vehicle_count_array",
             [](EngineHost &e) { return toArray(e.laneWaitingVehicleCountArray()); })
        .def("set_tl_phase_indexed", &EngineHost::setTrafficLightPhaseIndexed, "intersection_index"_a, "phase_id"_a)
        .def("set_tl_phases",
             [](EngineHost &e, py::array_t<int32_t, py::array::c_style | py::array::forcecast> phases) {
                 e.setTrafficLightPhases(phases.data(), (size_t) phases.size());
             },
             "phases"_a, "int array [len(intersection_ids())]; one asynchronous call sets every signal")
        .def("sync", &EngineHost::sync)
        .def("backend_name", &EngineHost::backendName)
        // ---- introspection used by the parity tests ----
        .def("_vehicle_state",
             [](EngineHost &e) {
                 cfa::VehicleSnapshot s;
                 e.snapshotVehicles(s);
                 py::dict d;
                 d["vid"] = toArray(s.vid);
                 d["drivable"] = toArray(s.drivable);
                 d["prev_drivable"] = toArray(s.prevDrivable);
                 d["leader"] = toArray(s.leader);
                 d["blocker"] = toArray(s.blocker);
                 d["enter_ll_time"] = toArray(s.enterLLTime);
                 d["route_pos"] = toArray(s.routePos);
                 d["dis"] = toArray(s.dis);
                 d["speed"] = toArray(s.speed);
                 d["gap"] = toArray(s.gap);
                 if (e.laneChange()) {
                     d["lc_partner"] = toArray(s.lcPartner);
                     d["lc_flags"] = toArray(s.lcFlags);
                     d["lc_offset"] = toArray(s.lcOffset);
                     d["lc_last_dir"] = toArray(s.lcLastDir);
                     d["lc_target"] = toArray(s.lcTarget);
                     d["lc_direction"] = toArray(s.lcDirection);
                     d["lc_last_change_time"] = toArray(s.lcLastChangeTime);
                 }
                 return d;
             })
        .def("_waiting",
             [](EngineHost &e) {
                 std::vector<int32_t> v, l;
                 e.waitingVehicles(v, l);
                 return py::make_tuple(toArray(v), toArray(l));
             })
        .def("_keeps_lane_history", &EngineHost::keepsLaneHistory)
        .def("_compact_vehicles", &EngineHost::compactVehicles,
             "forget the finished vehicles now (automatic once \"cfx\": {\"compactVehicles\": N} vehicles have been created; default 3.5 M)")
        .def("_vehicle_table", [](EngineHost &e) { return py::make_tuple(e.vehicleTableSize(), e.vehicleCompactions()); },
             "(vehicle numbers the host holds, compactions so far)")
        .def("_lane_history",
             [](EngineHost &e) {  // test hook: cfx_get_lane_history; arrays [L], [L, 241], [L, 241], [L], [L]
                 std::vector<int32_t> len, num, hn;
                 std::vector<double> avg, ha;
                 e.laneHistory(len, num, avg, hn, ha);
                 py::dict d;
                 d["len"] = toArray(len);
                 d["vehicle_num"] = toArray(num);
                 d["average_speed"] = toArray(avg);
                 d["history_vehicle_num"] = toArray(hn);
                 d["history_average_speed"] = toArray(ha);
                 return d;
             })
        .def("_tl_state",
             [](EngineHost &e) {
                 std::vector<int32_t> p;
                 std::vector<double> r;
                 e.trafficLightState(p, r);
                 return py::make_tuple(toArray(p), toArray(r));
             })
        .def("_scalars",
             [](EngineHost &e) {
                 cfx_scalars s = e.scalars();
                 py::dict d;
                 d["step"] = s.step;
                 d["active_vehicle_count"] = s.active_vehicle_count;
                 d["finished_vehicle_count"] = s.finished_vehicle_count;
                 d["spawned_vehicle_count"] = s.spawned_vehicle_count;
                 d["cumulative_travel_time"] = s.cumulative_travel_time;
                 d["live_enter_time_sum"] = s.live_enter_time_sum;
                 d["vehicle_steps"] = s.vehicle_steps;
                 d["tie_events"] = s.tie_events;
                 py::list td;
                 for (int i = 0; i < 8; ++i)
                     if (s.tie_drivables[i] >= 0) td.append(s.tie_drivables[i]);
                 d["tie_drivables"] = td;
                 d["diag_cross_jobs"] = s.diag_cross_jobs;
                 d["dropped_future_speeds"] = s.dropped_future_speeds;
                 return d;
             })
        .def("_layout", &EngineHost::layoutName)
        .def("_ring_info", &EngineHost::ringInfo, "(total ring slots, capacity scale) of the ring layout")
        .def("_profile_enable", &EngineHost::profileEnable, "on"_a)
        .def("_device_spin", &EngineHost::deviceSpin, "microseconds"_a)
        .def("_device_memory", &EngineHost::deviceMemory, "(free, total) bytes of the engine's device")
        .def("_profile_read", &EngineHost::profileRead)
        .def("_profile_symbols", &EngineHost::profileSymbols, "timing slot -> symbol of the kernel launched last in it")
        .def("_host_stats", [](EngineHost &e, bool reset) {
            const cfx_host_stats s = e.hostStats(reset);
            const EngineHost::SlowestStep slow = e.slowestStep(reset);
            py::dict d;
            d["slowest_next_step"] = py::make_tuple(slow.total, slow.step, py::make_tuple(slow.part[0], slow.part[1], slow.part[2],
                                                                                          slow.part[3], slow.part[4], slow.part[5]));
            d["step_calls"] = s.step_calls;
            d["step_call_us_mean"] = s.step_calls ? s.step_call_us_sum / (double) s.step_calls : 0.0;
            d["worst_step_call_us"] = s.worst_step_call_us;
            d["worst_step_call_at"] = s.worst_step_call_at;
            d["worst_step_call_cause"] = s.worst_step_call_cause;
            d["calls_over_1ms"] = s.calls_over_1ms;
            d["ring_regrows_total"] = s.ring_regrows_total;
            d["table_grows_total"] = s.table_grows_total;
            d["status_queries"] = s.status_queries;
            d["worst_status_query_us"] = py::make_tuple(s.worst_status_query_us, s.worst_status_query_settle_us,
                                                         s.worst_status_query_copy_us, s.worst_status_query_wait_us);
            return d;
        }, "reset"_a = false, "host time inside cfx_step (cfx_get_host_stats)")
        .def("_vehicle_id", [](EngineHost &e, int vid) { return e.vehicleId(vid); }, "vid"_a)
        .def("_vehicle_ids",
             [](EngineHost &e, py::array_t<int32_t> vids) {
                 std::vector<std::string> out;
                 auto r = vids.unchecked<1>();
                 for (py::ssize_t i = 0; i < r.shape(0); ++i) out.push_back(e.vehicleId(r(i)));
                 return out;
             })
        .def("_id_sort_key", [](EngineHost &e, int vid) { return e.spawner().idSortKey(vid); }, "vid"_a)
        .def("_vehicle_priority", [](EngineHost &e, int vid) { return e.spawner().vehicles.at(vid).priority; })
        .def("_drivable_ids",
             [](EngineHost &e) {
                 std::vector<std::string> ids;
                 int D = (int) (e.net().lanes.size() + e.net().laneLinks.size());
                 for (int d = 0; d < D; ++d) ids.push_back(e.net().drivableId(d));
                 return ids;
             })
        .def("_flat_net", [](EngineHost &e) { return flatNetToDict(e.net()); });
    {
        PyObject *descr = PyDescr_NewMethod((PyTypeObject *) engineClass.ptr(), &kEngineSetTlPhaseDef);
        if (!descr) throw py::error_already_set();
        engineClass.attr("set_tl_phase") = py::reinterpret_steal<py::object>(descr);
    }

    using cfa::VectorEngineHost;
    py::class_<VectorEngineHost>(m, "VectorEngine",
                                 "num_envs independent copies of one scenario (env e uses seed + e) advanced in lock-step by "
                                 "one device engine; observations and actions are arrays of shape [num_envs, ...].")
        .def(py::init<const std::string &, int, int>(), "config_file"_a, "num_envs"_a, "thread_num"_a = 1)
        .def_static(
            "_with_backend",
            [](const std::string &cfg, int envs, int threads, const std::string &lib) {
                return std::unique_ptr<VectorEngineHost>(new VectorEngineHost(cfg, envs, threads, lib));
            },
            "config_file"_a, "num_envs"_a, "thread_num"_a, "backend_library"_a)
        .def_property_readonly("num_envs", &VectorEngineHost::numEnvs)
        .def("_host_seconds", &VectorEngineHost::hostSeconds,
             "[spawners, record translation, cfx_step] cumulative host wall seconds since the last reset")
        .def("next_step", &VectorEngineHost::nextStep)
        .def("reset", &VectorEngineHost::reset, "seed"_a = false)
        .def("get_current_time", &VectorEngineHost::getCurrentTime)
        .def("get_vehicle_count", &VectorEngineHost::totalVehicleCount)
        .def("lane_ids", &VectorEngineHost::laneIds)
        .def("intersection_ids", &VectorEngineHost::intersectionIds)
        .def("get_lane_vehicle_count_array",
             [](VectorEngineHost &e) {
                 auto a = toArray(e.laneVehicleCounts());
                 a.resize({(py::ssize_t) e.numEnvs(), (py::ssize_t) e.numLanes()});
                 return a;
             })
        .def("get_lane_waiting_vehicle_count_array",
             [](VectorEngineHost &e) {
                 auto a = toArray(e.laneWaitingVehicleCounts());
                 a.resize({(py::ssize_t) e.numEnvs(), (py::ssize_t) e.numLanes()});
                 return a;
             })
        .def("set_tl_phases",
             [](VectorEngineHost &e, py::array_t<int32_t, py::array::c_style | py::array::forcecast> phases) {
                 std::vector<int32_t> v(phases.data(), phases.data() + phases.size());
                 e.setTrafficLightPhases(v);
             },
             "phases"_a, "int array [num_envs, num_intersections] (entries of virtual intersections are ignored)")
        .def("get_lane_vehicle_count", &VectorEngineHost::getLaneVehicleCount, "env"_a)
        .def("get_vehicle_speed", &VectorEngineHost::getVehicleSpeed, "env"_a)
        .def("sync", &VectorEngineHost::sync)
        .def("backend_name", &VectorEngineHost::backendName)
        .def("_profile_enable", &VectorEngineHost::profileEnable, "on"_a)
        .def("_profile_read", &VectorEngineHost::profileRead)
        .def("_scalars", [](VectorEngineHost &e) {
            cfx_scalars s = e.scalars();
            py::dict d;
            d["step"] = s.step;
            d["active_vehicle_count"] = s.active_vehicle_count;
            d["finished_vehicle_count"] = s.finished_vehicle_count;
            d["spawned_vehicle_count"] = s.spawned_vehicle_count;
            d["vehicle_steps"] = s.vehicle_steps;
                 d["tie_events"] = s.tie_events;
            d["cumulative_travel_time"] = s.cumulative_travel_time;
            return d;
        });

    using cfa::TiledEngineHost;
    py::class_<TiledEngineHost>(m, "TiledEngine",
                                "One road network cut into rows x cols tiles of intersections, one device engine per tile, a "
                                "one-lane halo exchanged per step.  With local_tiles empty every tile runs in this process "
                                "(next_step does the exchange); otherwise the caller moves the halo between step_begin and "
                                "step_end (cityflow_amd.tiled.DistributedEngine does it over torch.distributed).")
        .def(py::init<const std::string &, int, int, const std::vector<int> &, const std::string &>(), "config_file"_a,
             "rows"_a, "cols"_a, "local_tiles"_a = std::vector<int>(), "backend_library"_a = "")
        .def("next_step", &TiledEngineHost::nextStep)
        .def("step_begin", &TiledEngineHost::stepBegin)
        .def("step_end", &TiledEngineHost::stepEnd)
        .def("enable_mailboxes", &TiledEngineHost::enableMailboxes, "job_id"_a,
             "Exchange the halo device to device through shared-memory mailboxes (all tiles on one node); job_id must "
             "be unique per job and equal on every process")
        .def("unlink_mailboxes", &TiledEngineHost::unlinkMailboxes)
        .def("enable_device_mailboxes", &TiledEngineHost::enableDeviceMailboxes, "job_id"_a,
             "All tiles in this process: mailboxes in the receiving tile's device memory (peer HBM); False if not possible here")
        .def("device_mailbox_phase", &TiledEngineHost::deviceMailboxPhase, "job_id"_a, "phase"_a,
             "One tile per process: phase 1 (allocate + publish), a barrier, phase 2 (open + attach); False = not possible")
        .def("halo_transport", &TiledEngineHost::haloTransport)
        .def("device_mailboxes_fine_grained", &TiledEngineHost::deviceMailboxesFineGrained,
             "False if a device mailbox fell back to a plain allocation (safe only while every tile is on the same device)")
        .def("device_identities", &TiledEngineHost::deviceIdentities,
             "Physical device of each local tile (PCI bus id; \"cpu\" on the CPU twin): equal strings = one shared device")
        .def("halo_device_buffers", [](TiledEngineHost &e, int i) { return e.haloDeviceBuffers(i); }, "i"_a,
             "(send pointer, send bytes, recv pointer, recv bytes) of local tile i's device-resident halo messages")
        .def("step_begin_device", &TiledEngineHost::stepBeginDevice, "step_begin with the halo left in the device buffers")
        .def("step_end_device", &TiledEngineHost::stepEndDevice, "step_end importing from the device buffers")
        .def_property_readonly("num_tiles", &TiledEngineHost::nTiles)
        .def_property_readonly("num_local", &TiledEngineHost::nLocal)
        .def("local_rank", &TiledEngineHost::localRank, "i"_a)
        .def("peers",
             [](TiledEngineHost &e, int i) {
                 py::list out;
                 for (const cfa::TilePeer &p : e.peers(i))
                     out.append(py::make_tuple(p.rank, p.sendOff, p.sendBytes, p.recvOff, p.recvBytes));
                 return out;
             },
             "i"_a, "[(peer tile, send offset, send bytes, recv offset, recv bytes)] of local tile i")
        // zero-copy views of the halo staging buffers of local tile i (valid while the engine lives)
        .def("send_buffer",
             [](py::object self, int i) {
                 auto &e = self.cast<TiledEngineHost &>();
                 auto &b = e.sendBuffer(i);
                 return py::array_t<uint8_t>({(py::ssize_t) b.size()}, {1}, (const uint8_t *) b.data(), self);
             },
             "i"_a)
        .def("recv_buffer",
             [](py::object self, int i) {
                 auto &e = self.cast<TiledEngineHost &>();
                 auto &b = e.recvBuffer(i);
                 return py::array_t<uint8_t>({(py::ssize_t) b.size()}, {1}, (const uint8_t *) b.data(), self);
             },
             "i"_a)
        .def("get_vehicle_count", &TiledEngineHost::getVehicleCount)
        .def("get_vehicles", &TiledEngineHost::getVehicles, "include_waiting"_a = false)
        .def("get_lane_vehicles", &TiledEngineHost::getLaneVehicles)
        .def("get_vehicle_speed", &TiledEngineHost::getVehicleSpeed)
        .def("get_vehicle_distance", &TiledEngineHost::getVehicleDistance)
        .def("get_vehicle_info", &TiledEngineHost::getVehicleInfo, "vehicle_id"_a)
        .def("get_leader", &TiledEngineHost::getLeader, "vehicle_id"_a)
        .def("get_average_travel_time", &TiledEngineHost::getAverageTravelTime)
        .def("push_vehicle", &TiledEngineHost::pushVehicle)
        .def("set_vehicle_speed", &TiledEngineHost::setVehicleSpeed, "vehicle_id"_a, "speed"_a)
        .def("set_random_seed", &TiledEngineHost::setRandomSeed, "seed"_a)
        .def("get_lane_vehicle_count", &TiledEngineHost::getLaneVehicleCount)
        .def("get_lane_waiting_vehicle_count", &TiledEngineHost::getLaneWaitingVehicleCount)
        .def("get_lane_vehicle_count_array", [](TiledEngineHost &e) { return toArray(e.laneVehicleCountArray()); })
        .def("get_lane_waiting_vehicle_count_array", [](TiledEngineHost &e) { return toArray(e.laneWaitingVehicleCountArray()); })
        .def("get_current_time", &TiledEngineHost::getCurrentTime)
        .def("set_tl_phase", &TiledEngineHost::setTrafficLightPhase, "intersection_id"_a, "phase_id"_a)
        .def("set_tl_phases",
             [](TiledEngineHost &e, py::array_t<int32_t, py::array::c_style | py::array::forcecast> phases) {
                 e.setTrafficLightPhases(std::vector<int32_t>(phases.data(), phases.data() + phases.size()));
             },
             "phases"_a)
        .def("reset", &TiledEngineHost::reset, "seed"_a = false)
        .def("sync", &TiledEngineHost::sync)
        .def("lane_ids", &TiledEngineHost::laneIds)
        .def("owner", &TiledEngineHost::owner, "owning tile of every intersection (index order of the roadnet file)")
        .def("snapshot", &TiledEngineHost::snapshot, "Archive of the whole network (every tile in this process)")
        .def("load", &TiledEngineHost::load, "archive"_a, "every process loads the same archive and keeps its tiles' part")
        .def("load_from_file", &TiledEngineHost::loadFromFile, "path"_a)
        .def("set_vehicle_route", &TiledEngineHost::setRoute, "vehicle_id"_a, "route"_a)
        .def("set_replay_file", &TiledEngineHost::setReplayLogFile, "replay_file"_a)
        .def("set_save_replay", &TiledEngineHost::setSaveReplay, "open"_a)
        .def("_pending_pushed_keyed", &TiledEngineHost::pendingPushedKeyed,
             "(priority, id) of the vehicles pushed since the last step: every rank holds the same ones")
        .def("_vehicles_keyed", &TiledEngineHost::vehiclesKeyed, "include_waiting"_a = false, "local (priority, id) pairs")
        .def("_runs_here", &TiledEngineHost::runsHere, "vehicle_id"_a)
        .def("_local_status", [](TiledEngineHost &e) { return toArray(e.localStatus()); })
        .def("_average_travel_time_from",
             [](TiledEngineHost &e, double cumulative, int64_t finished, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> st) {
                 return e.averageTravelTimeFrom(cumulative, finished, std::vector<uint8_t>(st.data(), st.data() + st.size()));
             },
             "cumulative"_a, "finished"_a, "status"_a)
        .def("_wants_replay", &TiledEngineHost::wantsReplay)
        .def("_replay_part", [](TiledEngineHost &e) { return py::bytes(e.replayPart()); })
        .def("_replay_write",
             [](TiledEngineHost &e, const std::vector<py::bytes> &parts) {
                 std::vector<std::string> blobs;
                 for (const py::bytes &b : parts) blobs.push_back((std::string) b);
                 e.replayWrite(blobs);
             },
             "parts"_a)
        .def("_compact_vehicles", &TiledEngineHost::compactVehicles,
             "forget the finished vehicles now (every tile in this process; automatic once \"cfx\": {\"compactVehicles\": N} vehicles have been created; default 3.5 M)")
        .def("_keeps_lane_history", &TiledEngineHost::keepsLaneHistory)
        .def("_wants_compaction", &TiledEngineHost::wantsCompaction, "enough vehicles created since the last time (the same answer on every rank)")
        .def("_compact_from_parts",
             [](TiledEngineHost &e, const std::vector<py::bytes> &parts) {
                 std::vector<std::string> p;
                 for (const auto &b : parts) p.push_back(std::string(b));
                 e.compactFromParts(p);
             },
             "parts"_a, "several processes: _snapshot_part() of every rank in rank order, the same call on every rank")
        .def("_vehicle_table", [](TiledEngineHost &e) { return py::make_tuple(e.vehicleTableSize(), e.vehicleCompactions()); },
             "(vehicle numbers the host holds, compactions so far)")
        .def("_snapshot_part", [](TiledEngineHost &e) { return py::bytes(e.snapshotPart()); },
             "the state of this process's tiles, for _snapshot_from_parts on every process (in rank order)")
        .def("_snapshot_from_parts",
             [](TiledEngineHost &e, const std::vector<py::bytes> &parts) {
                 std::vector<std::string> blobs;
                 for (const py::bytes &b : parts) blobs.push_back((std::string) b);
                 return e.snapshotFromParts(blobs);
             },
             "parts"_a)
        .def("_set_status_reducer", &TiledEngineHost::setStatusReducer, "fn"_a)
        .def("_host_seconds", &TiledEngineHost::hostSeconds,
             "(spawner, submit) cumulative host wall seconds of this process since the last reset")
        .def("_ahead_stats", &TiledEngineHost::aheadStats, "(batches taken from the step-ahead thread, batches made by the step itself, the thread's busy seconds)")
        .def("_profile_enable", &TiledEngineHost::profileEnable, "local_tile"_a, "on"_a)
        .def("_device_spin", &TiledEngineHost::deviceSpin, "microseconds"_a)
        .def("backend_name", &TiledEngineHost::backendName)
        .def("_layout", &TiledEngineHost::layoutName)
        .def("_profile_read", &TiledEngineHost::profileRead, "local_tile"_a)
        .def("_vehicle_state",
             [](TiledEngineHost &e) {
                 cfa::VehicleSnapshot s;
                 e.snapshotVehicles(s);
                 py::dict d;
                 d["vid"] = toArray(s.vid);
                 d["drivable"] = toArray(s.drivable);
                 d["prev_drivable"] = toArray(s.prevDrivable);
                 d["leader"] = toArray(s.leader);
                 d["blocker"] = toArray(s.blocker);
                 d["enter_ll_time"] = toArray(s.enterLLTime);
                 d["route_pos"] = toArray(s.routePos);
                 d["dis"] = toArray(s.dis);
                 d["speed"] = toArray(s.speed);
                 d["gap"] = toArray(s.gap);
                 return d;
             })
        .def("_scalars", [](TiledEngineHost &e) {
            cfx_scalars s = e.scalars();
            py::dict d;
            d["step"] = s.step;
            d["active_vehicle_count"] = s.active_vehicle_count;
            d["finished_vehicle_count"] = s.finished_vehicle_count;
            d["spawned_vehicle_count"] = s.spawned_vehicle_count;
            d["cumulative_travel_time"] = s.cumulative_travel_time;
            d["live_enter_time_sum"] = s.live_enter_time_sum;
            d["vehicle_steps"] = s.vehicle_steps;
                 d["tie_events"] = s.tie_events;
            return d;
        });

    py::class_<cfa::Archive>(m, "Archive")
        .def(py::init([](EngineHost &e) { return e.snapshot(); }), "engine"_a)
        .def("dump", &cfa::Archive::dump, "path"_a)
        .def_readonly("_last_dump_inexact", &cfa::Archive::lastDumpInexact,
                      "numbers of the last dump() the reference's JSON reader cannot be made to return exactly (written with 17 digits)");

    m.def("_load_roadnet", &loadRoadnet, "path"_a);
    m.def("_roadnet_probe", &roadnetProbe, "path"_a);
    m.def("_spawn_schedule", &spawnSchedule, "roadnet_file"_a, "flow_file"_a, "interval"_a, "seed"_a, "thread_num"_a,
          "steps"_a, "ahead"_a = 0);
    m.def("_spawn_benchmark", &spawnBenchmark, "roadnet_file"_a, "flow_file"_a, "interval"_a, "seed"_a, "skip"_a, "steps"_a);
    m.def("_default_backend_path", &cfa::defaultBackendPath);
    // test hooks: the host's number reader (the reference's: rapidjson's default, json_number.h) and Archive.dump's writer
    m.def("_parse_json_number", [](const std::string &lit) {
        cfa::Json v = cfa::Json::parseText(lit);
        if (!v.isNumber()) throw std::runtime_error("not a number");
        return py::make_tuple(v.asDouble(), v.integral);
    });
    m.def("_format_json_number", &cfa::formatJsonNumber);
    m.def("_inexact_json_numbers", &cfa::inexactJsonNumbers);
#ifdef CITYFLOW_AMD_VERSION
    m.attr("__version__") = CITYFLOW_AMD_VERSION;
#else
    m.attr("__version__") = "dev";
#endif
}
