#include "engine_host.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <iostream>
#include <stdexcept>

#include "json.h"

namespace cfa {

// ---------------------------------------------------------------- backend loading
std::string defaultBackendPath() {
    Dl_info info;
    if (dladdr((void *) &defaultBackendPath, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t slash = p.find_last_of('/');
        std::string dir = slash == std::string::npos ? "." : p.substr(0, slash);
        return dir + "/lib/libcfx_hip.so";
    }
    return "libcfx_hip.so";
}

void Backend::open(const std::string &libPath) {
    path = libPath;
    // RTLD_NODELETE: the library stays mapped after the last engine is gone.  Unloading a HIP code object while the runtime
    // still holds its kernels buys nothing, and a process map taken at exit (the driver's record of the native code a test run
    // loaded) would otherwise show neither this library nor a hint that every kernel of the run came from it.
    handle = dlopen(libPath.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
    if (!handle)
        throw std::runtime_error("cityflow_amd: cannot load device engine library '" + libPath + "': " + dlerror() +
                                 " (build it with `python -c 'import __graft_entry__ as g; g.build()'`; there is no "
                                 "CPU fallback)");
#define CFX_FN(name)                                                                                  \
    name = reinterpret_cast<decltype(name)>(dlsym(handle, #name));                                    \
    if (!name) throw std::runtime_error("cityflow_amd: '" + libPath + "' does not export " #name);
    CFX_FN(cfx_abi_version)
    CFX_FN(cfx_create)
    CFX_FN(cfx_destroy)
    CFX_FN(cfx_last_error)
    CFX_FN(cfx_backend_name)
    CFX_FN(cfx_add_templates)
    CFX_FN(cfx_add_routes)
    CFX_FN(cfx_step)
    CFX_FN(cfx_sync)
    CFX_FN(cfx_reset)
    CFX_FN(cfx_set_tl_phase)
    CFX_FN(cfx_set_tl_phases)
    CFX_FN(cfx_get_tl_state)
    CFX_FN(cfx_get_scalars)
    CFX_FN(cfx_get_layout)
    CFX_FN(cfx_get_ring_info)
    CFX_FN(cfx_get_lane_counts)
    CFX_FN(cfx_get_lane_waiting_counts)
    CFX_FN(cfx_get_vehicles)
    CFX_FN(cfx_get_waiting)
    CFX_FN(cfx_get_vehicle_status)
    CFX_FN(cfx_set_vehicle_speed)
    CFX_FN(cfx_set_vehicle_route)
    CFX_FN(cfx_get_vehicle)
    CFX_FN(cfx_load_state)
    CFX_FN(cfx_get_custom_speeds)
    CFX_FN(cfx_lane_change_supply)
    CFX_FN(cfx_lane_change_poll)
    CFX_FN(cfx_halo_config)
    CFX_FN(cfx_halo_export)
    CFX_FN(cfx_halo_import)
    CFX_FN(cfx_halo_attach)
    CFX_FN(cfx_halo_mailbox_alloc)
    CFX_FN(cfx_halo_mailbox_open)
    CFX_FN(cfx_halo_mailbox_fine_grained)
    CFX_FN(cfx_device_identity)
    CFX_FN(cfx_device_memory)
    CFX_FN(cfx_halo_device_buffers)
    CFX_FN(cfx_halo_post)
    CFX_FN(cfx_halo_wait)
    CFX_FN(cfx_profile_kernel_count)
    CFX_FN(cfx_profile_kernel_name)
    CFX_FN(cfx_profile_enable)
    CFX_FN(cfx_profile_read)
    CFX_FN(cfx_profile_kernel_symbol)
    CFX_FN(cfx_get_host_stats)
    CFX_FN(cfx_device_spin)
    CFX_FN(cfx_get_lane_history)
    CFX_FN(cfx_set_lane_history)
#undef CFX_FN
    if (cfx_abi_version() != CFX_ABI_VERSION)
        throw std::runtime_error("cityflow_amd: ABI version mismatch in '" + libPath + "'");
}

Backend::~Backend() {
    if (handle) dlclose(handle);
}

EngineConfig readEngineConfig(const std::string &configFile) {
    EngineConfig c;
    try {
        Json cfg = Json::parseFile(configFile);
        if (!cfg.isObject()) throw JsonError("wrong format of config file");
        c.interval = cfg.numberAt("interval");
        c.rlTrafficLight = cfg.boolAt("rlTrafficLight");
        c.laneChange = cfg.boolAt("laneChange", false);
        c.seed = cfg.intAt("seed");
        c.dir = cfg.stringAt("dir");
        c.roadnetFile = cfg.stringAt("roadnetFile");
        c.flowFile = cfg.stringAt("flowFile");
        c.saveReplay = cfg.boolAt("saveReplay");
        if (c.saveReplay) {
            c.roadnetLogFile = cfg.stringAt("roadnetLogFile");
            c.replayLogFile = cfg.stringAt("replayLogFile");
        }
        if (const Json *x = cfg.find("cfx")) {
            if (!x->isObject()) throw JsonError("cfx: expected an object");
            auto choice = [x](const char *key, std::initializer_list<const char *> names) {
                const Json *v = x->find(key);
                if (!v) return 0;
                if (!v->isString()) throw JsonError(std::string("cfx.") + key + ": expected a string");
                int i = 0;
                for (const char *n : names) {
                    if (v->s == n) return i;
                    ++i;
                }
                throw JsonError(std::string("cfx.") + key + ": unknown value '" + v->s + "'");
            };
            c.crossMode = choice("crossMode", {"auto", "latency", "throughput"});
            c.layout = choice("layout", {"auto", "dense", "ring"});
            c.debugSync = x->boolAt("debugSync", false) ? 1 : 0;
            if (x->find("device")) c.device = x->intAt("device");
            if (x->find("ringLanesPerWave")) c.ringLanesPerWave = x->intAt("ringLanesPerWave");
            if (x->find("ringCapacityPercent")) c.ringCapacityPercent = x->intAt("ringCapacityPercent");
            if (x->find("denseForm")) c.denseForm = x->intAt("denseForm");
            c.exactShadowPeek = x->boolAt("exactShadowPeek", false);
            if (x->find("laneHistory")) c.laneHistory = x->boolAt("laneHistory", false) ? 1 : 0;
            if (x->find("hostThreads")) c.hostThreads = x->intAt("hostThreads");
            c.spawnAhead = x->boolAt("spawnAhead", true);
            if (x->find("compactVehicles")) c.compactVehicles = x->intAt("compactVehicles");
        }
    } catch (const JsonError &e) {
        throw std::runtime_error(std::string("load config failed! ") + e.what());
    }
    return c;
}

void EngineConfig::apply(cfx_config &cc) const {
    cc.interval = interval;
    cc.rl_traffic_light = rlTrafficLight ? 1 : 0;
    cc.lane_change = laneChange ? 1 : 0;
    cc.lane_history = laneHistory > 0 ? 1 : 0;
    cc.cross_mode = crossMode;
    cc.layout = layout;
    cc.debug_sync = debugSync;
    cc.ring_lanes_per_wave = ringLanesPerWave;
    cc.ring_capacity_percent = ringCapacityPercent;
    cc.dense_form = denseForm;
    cc.device = 0;
    // one process per GPU under torch.distributed.run: the launcher's LOCAL_RANK names the device
    if (const char *dev = getenv("LOCAL_RANK")) cc.device = atoi(dev);
    if (const char *dev = getenv("CITYFLOW_AMD_DEVICE")) cc.device = atoi(dev);
    if (device >= 0) cc.device = device;
}

// ---------------------------------------------------------------- construction
// Engine::Engine / loadConfig (engine.cpp:13-84).  Deviation (documented in DESIGN.md): where the
// reference prints "load config failed!" and hands back a half-built engine, this constructor throws.
EngineHost::EngineHost(const std::string &configFile, int threadNum, const std::string &backendLib)
    : threadNum_(threadNum < 1 ? 1 : threadNum) {
    std::string roadnetFile, flowFile;
    try {
        Json cfg = Json::parseFile(configFile);
        if (!cfg.isObject()) throw JsonError("wrong format of config file");
        interval_ = cfg.numberAt("interval");
        rlTrafficLight_ = cfg.boolAt("rlTrafficLight");
        laneChange_ = cfg.boolAt("laneChange", false);
        seed_ = cfg.intAt("seed");
        dir_ = cfg.stringAt("dir");
        roadnetFile = cfg.stringAt("roadnetFile");
        flowFile = cfg.stringAt("flowFile");
        saveReplayInConfig_ = saveReplay_ = cfg.boolAt("saveReplay");
        std::string roadnetLogFile, replayLogFile;
        if (saveReplay_) {  // engine.cpp:73-77: both keys are required then
            roadnetLogFile = cfg.stringAt("roadnetLogFile");
            replayLogFile = cfg.stringAt("replayLogFile");
        }
        net_->load(dir_ + roadnetFile);
        if (saveReplay_) {  // Engine::setLogFile engine.cpp:773-778
            if (!writeRoadnetLog(*net_, dir_ + roadnetLogFile)) std::cerr << "write roadnet log file error" << std::endl;
            replay_.open(dir_ + replayLogFile);
        }
        spawner_.init(net_.get(), interval_, threadNum_, seed_);
        spawner_.exactPeekOnly = readEngineConfig(configFile).exactShadowPeek;
        // (lane change: the step's shadows draw from the generator after the step's spawns, and how many is known only when
        //  the device has scheduled them — the next step's spawner cannot run before that)
        spawnAhead_ = readEngineConfig(configFile).spawnAhead && !laneChange_;
        {
            const int64_t cv = readEngineConfig(configFile).compactVehicles;
            compactAt_ = nextCompactAt_ = cv < 0 ? (size_t) 3500000 : (size_t) cv;
            compactAuto_ = cv < 0;
        }
        spawner_.loadFlows(dir_ + flowFile);
    } catch (const JsonError &e) {
        throw std::runtime_error(std::string("load config failed! ") + e.what());
    }
    be_.open(backendLib.empty() ? defaultBackendPath() : backendLib);
    cfx_config cc{};
    {
        // Lane::history (roadnet.cpp:900-915): an Archive carries it like the reference's (archive.cpp:286-294).  On the device it
        // rides in spare blocks of the action launch: +0.5 us per step at 30x30 (1.2 %), where that launch is bound by its slowest
        // chain — but +6.7 us (4.9 %) at 100x100 / 1 M vehicles, where the launch is bound by memory traffic and the lanes' threads
        // read every vehicle's speed a second time (profiles/r06_exp_lane_history_*); the dense layout (forced, or under lane
        // change, where the reference takes TWO records per step) pays a launch per record: 188 -> 197 us per lane-change step.
        // Not said in the config: kept where it is nearly free — the ring layout on networks up to kLaneHistoryAutoLanes lanes;
        // "cfx": {"laneHistory": true / false} decides otherwise.
        constexpr size_t kLaneHistoryAutoLanes = 20000;  // (the size from which the action phase takes its list form)
        EngineConfig ec = readEngineConfig(configFile);
        if (ec.laneHistory < 0)
            ec.laneHistory = (net_->lanes.size() <= kLaneHistoryAutoLanes && !laneChange_ && ec.layout != CFX_LAYOUT_DENSE) ? 1 : 0;
        ec.apply(cc);
    }
    laneHistory_ = cc.lane_history != 0;
    int32_t rc = be_.cfx_create(&net_->flat(), &cc, &dev_);
    if (rc != CFX_OK || !dev_) {
        const char *msg = be_.cfx_last_error(nullptr);
        throw std::runtime_error(std::string("cityflow_amd: cfx_create failed in ") + be_.path + ": " +
                                 (msg ? msg : "unknown error") + " (there is no CPU fallback)");
    }
    spawner_.setFinishedQuery([this](int vid) {
        uint8_t st = 0;
        check(be_.cfx_get_vehicle_status(dev_, vid, 1, &st), "cfx_get_vehicle_status");
        return st == 2;
    });
    uploadNewTablesIfAny();
}

EngineHost::~EngineHost() {
    if (dev_) be_.cfx_destroy(dev_);
}

void EngineHost::check(int32_t rc, const char *what) {
    if (rc == CFX_OK) return;
    const char *msg = be_.cfx_last_error(dev_);
    throw std::runtime_error(std::string("cityflow_amd: ") + what + " failed (" + std::to_string(rc) + "): " +
                             (msg ? msg : ""));
}

void EngineHost::uploadNewTablesIfAny() {
    int nt = (int) spawner_.templates.size();
    if (nt > templatesUploaded_) {
        check(be_.cfx_add_templates(dev_, nt - templatesUploaded_, spawner_.templates.data() + templatesUploaded_),
              "cfx_add_templates");
        templatesUploaded_ = nt;
    }
    const RouteTable &rt = spawner_.routes;
    int nr = rt.count();
    if (nr > routesUploaded_) {
        // rebase the CSR slices of the new routes to 0
        int r0 = routesUploaded_;
        int roadBase = rt.routeStart[r0];
        int nextBase = rt.nextStart[roadBase];
        std::vector<int32_t> routeStart, nextStart;
        for (int r = r0; r <= nr; ++r) routeStart.push_back(rt.routeStart[r] - roadBase);
        int nPos = rt.routeStart[nr] - roadBase;
        for (int p = 0; p <= nPos; ++p) nextStart.push_back(rt.nextStart[roadBase + p] - nextBase);
        check(be_.cfx_add_routes(dev_, nr - r0, routeStart.data(), rt.roads.data() + roadBase, nextStart.data(),
                                 rt.nextLL.data() + nextBase),
              "cfx_add_routes");
        routesUploaded_ = nr;
    }
}

// ---------------------------------------------------------------- stepping
bool EngineHost::onlyChangedPhases(std::vector<int32_t> &inters, std::vector<int32_t> &phases) {
    if (!rlTrafficLight_) return !inters.empty();  // (the device advances the lights itself: nothing is known here)
    if (knownPhase_.size() != net_->inters.size()) knownPhase_.assign(net_->inters.size(), -1);
    size_t keep = 0;
    for (size_t i = 0; i < inters.size(); ++i) {  // (in call order: a later call for the same intersection wins, as on the device)
        if (knownPhase_[(size_t) inters[i]] == phases[i]) continue;
        knownPhase_[(size_t) inters[i]] = phases[i];
        inters[keep] = inters[i];
        phases[keep] = phases[i];
        ++keep;
    }
    inters.resize(keep);
    phases.resize(keep);
    return keep > 0;
}

void EngineHost::flushPhases() {
    if (pendingPhaseInter_.empty()) return;
    if (onlyChangedPhases(pendingPhaseInter_, pendingPhaseValue_)) {
        const int32_t rc = be_.cfx_set_tl_phases(dev_, (int32_t) pendingPhaseInter_.size(), pendingPhaseInter_.data(),
                                                 pendingPhaseValue_.data());
        if (rc != CFX_OK) forgetPhases();  // (the cache was updated for phases the device never got: nothing is known any more)
        check(rc, "cfx_set_tl_phases");
    }
    pendingPhaseInter_.clear();
    pendingPhaseValue_.clear();
}

const std::vector<int32_t> &EngineHost::laneIdOrder() {
    if (laneIdOrder_.empty() && !net_->lanes.empty()) {
        std::vector<std::pair<std::string, int32_t>> ids;
        for (size_t l = 0; l < net_->lanes.size(); ++l) ids.emplace_back(net_->laneId((int) l), (int32_t) l);
        std::sort(ids.begin(), ids.end());
        for (auto &p : ids) laneIdOrder_.push_back(p.second);
    }
    return laneIdOrder_;
}

// Lane change: which vehicles got a shadow in the last step.  Asked as late as possible — the device engine only has to
// finish the schedule part of the step for it, and the rest of the step runs on while the host is back in the caller.
void EngineHost::settleLaneChange() {
    if (!lcPollPending_) return;
    lcPollPending_ = false;
    shadowParents_.resize((size_t) shadowPoolSize_);
    int32_t k = 0;
    check(be_.cfx_lane_change_poll(dev_, shadowPoolSize_, shadowParents_.data(), &k), "cfx_lane_change_poll");
    shadowParents_.resize((size_t) k);
    spawner_.commitShadows(shadowParents_);
    if (4 * k > shadowPoolSize_) shadowPoolSize_ = 8 * k;  // stay well clear of a step's demand
}

void EngineHost::dropAhead() {
    if (!aheadValid_) return;
    aheadValid_ = false;
    spawner_.rollbackAhead();
}

void EngineHost::nextStep() {
    // (where a call's host time went, for the slowest call since the last read: hostStats / Engine._host_stats)
    using clk = std::chrono::steady_clock;
    clk::time_point lap[7];
    lap[0] = clk::now();
    flushPhases();
    settleLaneChange();  // the generator must be past the last step's shadow draws before this step's spawns
    lap[1] = clk::now();
    if (aheadValid_) {  // this step's records were made while the device ran the last step
        aheadValid_ = false;
        spawner_.commitAhead();
        spawnBuf_.swap(aheadBuf_);
    } else {
        spawner_.step(step_, spawnBuf_);
    }
    lap[2] = clk::now();
    uploadNewTablesIfAny();
    if (laneChange_) {  // the priorities this step's shadows would draw, after the step's own spawn draws
        spawner_.peekShadowPriorities(shadowPoolSize_, shadowPool_);
        check(be_.cfx_lane_change_supply(dev_, (int32_t) shadowPool_.size(), shadowPool_.data()), "cfx_lane_change_supply");
    }
    lap[3] = clk::now();
    check(be_.cfx_step(dev_, spawnBuf_.data(), (int32_t) spawnBuf_.size()), "cfx_step");
    lap[4] = clk::now();
    lcPollPending_ = laneChange_;
    if (saveReplay_) updateLog();
    lap[5] = clk::now();
    struct Note {
        EngineHost *e;
        clk::time_point *lap;
        size_t at;
        ~Note() {
            lap[6] = clk::now();
            const double total = std::chrono::duration<double, std::micro>(lap[6] - lap[0]).count();
            if (total > e->slowest_.total) {
                e->slowest_.total = total;
                e->slowest_.step = (int64_t) at;
                for (int i = 0; i < 6; ++i) e->slowest_.part[i] = std::chrono::duration<double, std::micro>(lap[i + 1] - lap[i]).count();
            }
        }
    } note{this, lap, step_};
    step_ += 1;
    // The finished vehicles are forgotten once enough have been created since the last time (archive.cpp compactVehicles): the
    // reference frees a vehicle when it finishes; here host and device remember every vehicle number until then.
    // What it costs is proportional to the vehicles ALIVE (a snapshot and a load: ~0.15-0.3 us per vehicle), so the automatic
    // policy waits for 32 times as many vehicle numbers as there were vehicles alive last time (and 3.5 M at least — the
    // device's tables hold 4 M before they double): below 1 % of the run at 6x6 and 30x30, ~3 % at 100x100 / 1 M vehicles,
    // where host and device then hold up to ~3.5 GB of tables instead of growing without bound.  An explicit
    // "compactVehicles": N means every N vehicle numbers.
    if (compactAt_ > 0 && spawner_.vehicles.size() >= nextCompactAt_) {
        compactVehicles();
        const size_t alive = spawner_.vehicles.size();
        nextCompactAt_ = compactAuto_ ? std::max(compactAt_, 32 * alive) : alive + compactAt_;
    }
    if (spawnAhead_) {
        // The next step's spawner, now: its records depend on nothing the device is computing (a priority that collides with
        // a vehicle the host believes alive asks the device, as always — for the state after THIS step, which is the state the
        // next step would ask for), so the work overlaps the device's step instead of standing between a caller's
        // observation and its next step.  Any call other than nextStep / signals / counts takes it back first.
        // (step t WAS taken: whatever goes wrong in the step ahead — a device error in a priority-collision query — is left
        // to the next nextStep(), which takes that step plainly and raises it where it belongs)
        spawner_.beginAhead();
        try {
            spawner_.step(step_, aheadBuf_);
            aheadValid_ = true;
        } catch (...) {
            spawner_.rollbackAhead();
        }
    }
}

// Engine::updateLog (engine.cpp:518-554): the state after the step, lights after TrafficLight::passTime
void EngineHost::updateLog() {
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::vector<int32_t> phase;
    std::vector<double> remain;
    trafficLightState(phase, remain);
    replay_.writeStep(*net_, spawner_, s, phase);
}

void EngineHost::setReplayLogFile(const std::string &logFile) {
    if (!saveReplayInConfig_) {
        std::cerr << "saveReplay is not set to true in config file!" << std::endl;
        return;
    }
    replay_.open(dir_ + logFile);
}

void EngineHost::setSaveReplay(bool open) {
    if (!saveReplayInConfig_) {
        std::cerr << "saveReplay is not set to true in config file!" << std::endl;
        return;
    }
    saveReplay_ = open;
}

void EngineHost::sync() { check(be_.cfx_sync(dev_), "cfx_sync"); }

void EngineHost::profileEnable(bool on) { check(be_.cfx_profile_enable(dev_, on ? 1 : 0), "cfx_profile_enable"); }

std::map<std::string, std::pair<double, int64_t>> EngineHost::profileRead() {
    int n = be_.cfx_profile_kernel_count();
    std::vector<double> ms(n > 0 ? n : 1);
    std::vector<int64_t> cnt(n > 0 ? n : 1);
    check(be_.cfx_profile_read(dev_, ms.data(), cnt.data()), "cfx_profile_read");
    std::map<std::string, std::pair<double, int64_t>> out;
    for (int k = 0; k < n; ++k) out[be_.cfx_profile_kernel_name(k)] = std::make_pair(ms[k], cnt[k]);
    return out;
}

std::map<std::string, std::string> EngineHost::profileSymbols() {
    std::map<std::string, std::string> out;
    const int n = be_.cfx_profile_kernel_count();
    for (int k = 0; k < n; ++k) {
        const char *s = be_.cfx_profile_kernel_symbol(dev_, k);
        if (s && *s) out[be_.cfx_profile_kernel_name(k)] = s;
    }
    return out;
}

cfx_host_stats EngineHost::hostStats(bool reset) {
    cfx_host_stats s{};
    check(be_.cfx_get_host_stats(dev_, &s, reset ? 1 : 0), "cfx_get_host_stats");
    return s;
}

cfx_scalars EngineHost::scalars() {
    cfx_scalars s{};
    check(be_.cfx_get_scalars(dev_, &s), "cfx_get_scalars");
    return s;
}

void EngineHost::reset(bool resetRnd) {
    dropAhead();
    // (lane change: the generator must first get past the last step's shadow draws — the reference made them inside that
    // step, and without a reseed the stream goes on from there)
    settleLaneChange();
    pendingPhaseInter_.clear();  // TrafficLight::reset puts every light back to phase 0 anyway
    pendingPhaseValue_.clear();
    forgetPhases();
    check(be_.cfx_reset(dev_), "cfx_reset");
    spawner_.reset(resetRnd);
    waitingCustom_.clear();
    nextCompactAt_ = compactAt_;
    step_ = 0;
    vehicleEpoch_ += 1;
}

// ---------------------------------------------------------------- getters
size_t EngineHost::getVehicleCount() { return (size_t) scalars().active_vehicle_count; }

std::vector<int32_t> EngineHost::laneVehicleCountArray() {
    std::vector<int32_t> out(net_->lanes.size());
    check(be_.cfx_get_lane_counts(dev_, out.data()), "cfx_get_lane_counts");
    return out;
}

std::vector<int32_t> EngineHost::laneWaitingVehicleCountArray() {
    std::vector<int32_t> out(net_->lanes.size());
    check(be_.cfx_get_lane_waiting_counts(dev_, out.data()), "cfx_get_lane_waiting_counts");
    return out;
}

std::vector<std::string> EngineHost::laneIds() const {
    std::vector<std::string> ids(net_->lanes.size());
    for (size_t l = 0; l < ids.size(); ++l) ids[l] = net_->laneId((int) l);
    return ids;
}

std::vector<std::string> EngineHost::intersectionIds() const {
    std::vector<std::string> ids(net_->inters.size());
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = net_->inters[i].id;
    return ids;
}

std::map<std::string, int> EngineHost::getLaneVehicleCount() {
    std::vector<int32_t> cnt = laneVehicleCountArray();
    std::map<std::string, int> ret;
    for (size_t l = 0; l < cnt.size(); ++l) ret.emplace(net_->laneId((int) l), cnt[l]);
    return ret;
}

std::map<std::string, int> EngineHost::getLaneWaitingVehicleCount() {
    std::vector<int32_t> cnt = laneWaitingVehicleCountArray();
    std::map<std::string, int> ret;
    for (size_t l = 0; l < cnt.size(); ++l) ret.emplace(net_->laneId((int) l), cnt[l]);
    return ret;
}

void EngineHost::snapshotVehicles(VehicleSnapshot &s, unsigned fields) {
    dropAhead();
    settleLaneChange();
    int cap = (int) scalars().active_vehicle_count + 16;
    cfx_vehicle_view v{};
    v.capacity = cap;
    s.vid.resize(cap);
    v.vid = s.vid.data();
    auto want = [&](unsigned bit, auto &vec, auto *&ptr) {
        vec.resize(fields & bit ? cap : 0);
        ptr = fields & bit ? vec.data() : nullptr;
    };
    want(kSnapDrivable, s.drivable, v.drivable);
    want(kSnapPrev, s.prevDrivable, v.prev_drivable);
    want(kSnapLeader, s.leader, v.leader_vid);
    want(kSnapBlocker, s.blocker, v.blocker_vid);
    want(kSnapEnterLL, s.enterLLTime, v.enter_ll_time);
    want(kSnapRoutePos, s.routePos, v.route_pos);
    want(kSnapDis, s.dis, v.dis);
    want(kSnapSpeed, s.speed, v.speed);
    want(kSnapGap, s.gap, v.gap);
    if (laneChange_) fields |= kSnapLaneChange;  // ids depend on the shadow flag
    want(kSnapLaneChange, s.lcPartner, v.lc_partner_vid);
    want(kSnapLaneChange, s.lcFlags, v.lc_flags);
    want(kSnapLaneChange, s.lcOffset, v.lc_offset);
    want(kSnapLaneChange, s.lcLastDir, v.lc_last_dir);
    want(kSnapLaneChange, s.lcTarget, v.lc_target_lane);
    want(kSnapLaneChange, s.lcDirection, v.lc_direction);
    want(kSnapLaneChange, s.lcLastChangeTime, v.lc_last_change_time);
    want(kSnapLaneChange, s.lcWaitingTime, v.lc_waiting_time);
    check(be_.cfx_get_vehicles(dev_, &v), "cfx_get_vehicles");
    s.count = v.count;
    s.vid.resize(v.count);
    for (auto *vec : {&s.drivable, &s.prevDrivable, &s.leader, &s.blocker, &s.enterLLTime, &s.routePos})
        if (!vec->empty()) vec->resize(v.count);
    for (auto *vec : {&s.dis, &s.speed, &s.gap, &s.lcOffset, &s.lcLastChangeTime, &s.lcWaitingTime})
        if (!vec->empty()) vec->resize(v.count);
    for (auto *vec : {&s.lcPartner, &s.lcLastDir, &s.lcTarget, &s.lcDirection})
        if (!vec->empty()) vec->resize(v.count);
    if (!s.lcFlags.empty()) s.lcFlags.resize(v.count);
}

void EngineHost::waitingVehicles(std::vector<int32_t> &vid, std::vector<int32_t> &lane) {
    dropAhead();
    settleLaneChange();
    cfx_scalars sc = scalars();
    // spawned - finished - running, unless the state came from an archive (finished vehicles are not in its vehicle
    // table): then every vehicle of the table may be waiting
    int64_t cap = sc.spawned_vehicle_count - sc.finished_vehicle_count - sc.active_vehicle_count;
    if (cap < 0) cap = 0;
    int32_t n = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt) cap = sc.spawned_vehicle_count;
        vid.resize((size_t) cap + 16);
        lane.resize((size_t) cap + 16);
        const int32_t rc = be_.cfx_get_waiting(dev_, (int32_t) cap + 16, vid.data(), lane.data(), &n);
        if (rc == CFX_ERR_CAPACITY && attempt == 0) continue;
        check(rc, "cfx_get_waiting");
        break;
    }
    vid.resize(n);
    lane.resize(n);
}

// getVehicles engine.cpp:619-626 — vehiclePool (priority) order
std::vector<std::string> EngineHost::getVehicles(bool includeWaiting) {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::vector<std::pair<int32_t, int32_t>> byPriority;
    for (int i = 0; i < s.count; ++i)  // Engine::getRunningVehicles engine.cpp:780-790 skips shadows (isReal)
        if (!s.isShadow(i)) byPriority.emplace_back(spawner_.vehicles[s.vid[i]].priority, s.vid[i]);
    if (includeWaiting) {
        std::vector<int32_t> wv, wl;
        waitingVehicles(wv, wl);
        for (int32_t v : wv) byPriority.emplace_back(spawner_.vehicles[v].priority, v);
    }
    std::vector<std::pair<int32_t, std::string>> pushed;  // pushed since the last step: in vehiclePool already (engine.cpp:605-613)
    if (includeWaiting) spawner_.pendingPushed(pushed);
    for (size_t i = 0; i < pushed.size(); ++i) byPriority.emplace_back(pushed[i].first, -1 - (int32_t) i);
    std::sort(byPriority.begin(), byPriority.end());
    std::vector<std::string> ret;
    ret.reserve(byPriority.size());
    for (auto &p : byPriority) ret.emplace_back(p.second >= 0 ? spawner_.vehicleId(p.second) : pushed[(size_t) (-1 - p.second)].second);
    return ret;
}

std::map<std::string, std::vector<std::string>> EngineHost::getLaneVehicles() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::map<std::string, std::vector<std::string>> ret;
    const int L = (int) net_->lanes.size();
    std::vector<std::vector<std::string>> perLane(L);
    for (int i = 0; i < s.count; ++i)
        if (s.drivable[i] < L) perLane[s.drivable[i]].push_back(spawner_.vehicleId(s.vid[i], s.isShadow(i)));
    for (int l = 0; l < L; ++l) ret.emplace(net_->laneId(l), std::move(perLane[l]));
    return ret;
}

std::map<std::string, double> EngineHost::getVehicleSpeed() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::map<std::string, double> ret;
    for (int i = 0; i < s.count; ++i)
        if (!s.isShadow(i)) ret.emplace(spawner_.vehicleId(s.vid[i]), s.speed[i]);  // getRunningVehicles: real vehicles only
    return ret;
}

std::map<std::string, double> EngineHost::getVehicleDistance() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::map<std::string, double> ret;
    for (int i = 0; i < s.count; ++i)
        if (!s.isShadow(i)) ret.emplace(spawner_.vehicleId(s.vid[i]), s.dis[i]);
    return ret;
}

int EngineHost::vidOf(const std::string &id) {
    dropAhead();
    settleLaneChange();
    if (!laneChange_) return spawner_.vidOfId(id);
    // An id is carried by a chain of vehicles: the flow's vehicle, then each shadow that took it over when its lane change
    // completed (LaneChange::finishChanging lanechange.cpp:115-127).  Of the chain at most two are alive: the holder of
    // the id and, while it changes lane, its shadow "<id>_shadow".
    const std::string suffix = "_shadow";
    const bool wantShadow = id.size() > suffix.size() && id.compare(id.size() - suffix.size(), suffix.size(), suffix) == 0;
    const int root = spawner_.vidOfId(wantShadow ? id.substr(0, id.size() - suffix.size()) : id);
    if (root < 0) return -1;
    std::vector<int32_t> alive;
    for (int32_t v : spawner_.idChain(root)) {
        uint8_t st = 2;
        check(be_.cfx_get_vehicle_status(dev_, v, 1, &st), "cfx_get_vehicle_status");
        if (st != 2) alive.push_back(v);
    }
    if (wantShadow) return alive.size() >= 2 ? alive[1] : -1;
    return alive.empty() ? root : alive[0];
}

// A vehicle pushed (push_vehicle) since the last step: it gets its vehicle number with the next step's spawn records, but the
// reference's vehiclePool / vehicleMap hold it from the moment of the call (Engine::pushVehicle engine.cpp:605-613).
bool EngineHost::isPendingPushed(const std::string &id) {
    dropAhead();
    std::vector<std::pair<int32_t, std::string>> pushed;
    spawner_.pendingPushed(pushed);
    for (const auto &p : pushed)
        if (p.second == id) return true;
    return false;
}

// getLeader engine.cpp:836-850
std::string EngineHost::getLeader(const std::string &vehicleId) {
    dropAhead();
    int vid = vidOf(vehicleId);
    uint8_t st = 2;
    if (vid >= 0) check(be_.cfx_get_vehicle_status(dev_, vid, 1, &st), "cfx_get_vehicle_status");
    if (vid < 0 && isPendingPushed(vehicleId)) return "";  // known to the reference's vehicleMap, without a leader yet
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + vehicleId + "' not found");
    if (st == 0) return "";
    VehicleSnapshot s;
    snapshotVehicles(s);
    for (int i = 0; i < s.count; ++i)
        if (s.vid[i] == vid) {
            if (s.isShadow(i) && s.lcPartner[i] >= 0) {  // engine.cpp:842-845: a shadow answers with its partner's leader
                for (int j = 0; j < s.count; ++j)
                    if (s.vid[j] == s.lcPartner[i]) {
                        i = j;
                        break;
                    }
            }
            if (s.leader[i] < 0) return "";
            bool shadow = false;
            for (int j = 0; j < s.count && laneChange_; ++j)
                if (s.vid[j] == s.leader[i]) shadow = s.isShadow(j);
            return spawner_.vehicleId(s.leader[i], shadow);
        }
    return "";
}

// Vehicle::getInfo vehicle.cpp:435-457
std::map<std::string, std::string> EngineHost::getVehicleInfo(const std::string &vehicleId) {
    dropAhead();
    int vid = vidOf(vehicleId);
    uint8_t st = 2;
    if (vid >= 0) check(be_.cfx_get_vehicle_status(dev_, vid, 1, &st), "cfx_get_vehicle_status");
    // pushed since the last step: known to the reference's vehicleMap, not running (Vehicle::getInfo vehicle.cpp:437-438)
    if (vid < 0 && isPendingPushed(vehicleId)) return {{"running", "0"}};
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + vehicleId + "' not found");
    std::map<std::string, std::string> info;
    info["running"] = std::to_string(st == 1);
    if (st != 1) return info;
    VehicleSnapshot s;
    snapshotVehicles(s);
    for (int i = 0; i < s.count; ++i) {
        if (s.vid[i] != vid) continue;
        info["distance"] = std::to_string(s.dis[i]);
        info["speed"] = std::to_string(s.speed[i]);
        info["drivable"] = net_->drivableId(s.drivable[i]);
        if (s.drivable[i] < (int) net_->lanes.size()) {
            const HostRoad &road = net_->roads[net_->lanes[s.drivable[i]].road];
            info["road"] = road.id;
            info["intersection"] = net_->inters[road.endInter].id;
        }
        const RouteTable &rt = spawner_.routes;
        int r = spawner_.vehicles[vid].route;
        std::string route;
        for (int p = rt.routeStart[r] + s.routePos[i]; p < rt.routeStart[r + 1]; ++p)
            route += net_->roads[rt.roads[p]].id + " ";
        info["route"] = route;
    }
    return info;
}

// getAverageTravelTime engine.cpp:682-691: same summation order (vehiclePool = ascending priority),
// so the result is bit-identical to the single-threaded reference for any interval.
double EngineHost::getAverageTravelTime() {
    dropAhead();
    settleLaneChange();
    cfx_scalars sc = scalars();
    double tt = sc.cumulative_travel_time;
    int64_t n = sc.finished_vehicle_count;
    int total = (int) spawner_.vehicles.size();
    std::vector<uint8_t> st(total);
    if (total) check(be_.cfx_get_vehicle_status(dev_, 0, total, st.data()), "cfx_get_vehicle_status");
    std::vector<std::pair<int32_t, double>> live;
    for (int v = 0; v < total; ++v)
        if (st[v] != 2) live.emplace_back(spawner_.vehicles[v].priority, spawner_.vehicles[v].enterTime);
    std::sort(live.begin(), live.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) {
        return a.first < b.first;
    });
    double now = getCurrentTime();
    for (auto &p : live) {
        tt += now - p.second;
        n++;
    }
    {  // vehicles pushed since the last step are in the reference's vehiclePool already: they entered now (+ 0.0 each)
        std::vector<std::pair<int32_t, std::string>> pushed;
        spawner_.pendingPushed(pushed);
        n += (int64_t) pushed.size();
    }
    return n == 0 ? 0 : tt / n;
}

// ---------------------------------------------------------------- control
// Single sets are only queued here; one asynchronous cfx_set_tl_phases carries them to the device right before the
// next step (or before anything reads the light state), in call order.
void EngineHost::setTrafficLightPhaseIndexed(int inter, int phase) {
    if (inter < 0 || inter >= (int) net_->inters.size() || net_->inters[inter].isVirtual || phase < 0 ||
        phase >= (int) net_->inters[inter].phases.size())
        throw std::out_of_range("set_tl_phase: intersection or phase index out of range");
    pendingPhaseInter_.push_back(inter);
    pendingPhaseValue_.push_back(phase);
}

void EngineHost::setTrafficLightPhases(const std::vector<int32_t> &phases) { setTrafficLightPhases(phases.data(), phases.size()); }

// (an agent calls this every step and changes a signal every tenth: what the call costs when nothing changes is what it costs —
// one pass over the real intersections' entries against the phases the device is known to hold, nothing allocated)
void EngineHost::setTrafficLightPhases(const int32_t *phases, size_t n) {
    if (!rlTrafficLight_) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    if (n != net_->inters.size()) throw std::runtime_error("set_tl_phases: expected one phase per intersection");
    flushPhases();
    if (knownPhase_.size() != n) knownPhase_.assign(n, -1);
    if (realInter_.empty())
        for (size_t i = 0; i < n; ++i)
            if (!net_->inters[i].isVirtual) {
                realInter_.push_back((int32_t) i);
                realInterPhases_.push_back((int32_t) net_->inters[i].phases.size());
            }
    changedInter_.clear();
    changedPhase_.clear();
    for (size_t k = 0; k < realInter_.size(); ++k) {
        const int32_t i = realInter_[k], p = phases[i];
        if (p == knownPhase_[(size_t) i]) continue;  // (known = was valid)
        if (p < 0 || p >= realInterPhases_[k])
            throw std::out_of_range("set_tl_phases: phase out of range for intersection '" + net_->inters[(size_t) i].id + "'");
        changedInter_.push_back(i);
        changedPhase_.push_back(p);
    }
    if (changedInter_.empty()) return;
    for (size_t k = 0; k < changedInter_.size(); ++k) knownPhase_[(size_t) changedInter_[k]] = changedPhase_[k];
    const int32_t rc = be_.cfx_set_tl_phases(dev_, (int32_t) changedInter_.size(), changedInter_.data(), changedPhase_.data());
    if (rc != CFX_OK) forgetPhases();
    check(rc, "cfx_set_tl_phases");
}

// setTrafficLightPhase engine.cpp:719-725
void EngineHost::setTrafficLightPhase(const std::string &id, int phaseIndex) {
    if (!rlTrafficLight_) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    auto it = net_->interIndex.find(id);
    if (it == net_->interIndex.end()) throw std::runtime_error("Intersection '" + id + "' not found");
    setTrafficLightPhaseOf(it->second, phaseIndex);
}

// ... the same call for a caller that has resolved the id itself (the binding keeps a dict of the ids' Python strings)
void EngineHost::setTrafficLightPhaseOf(int inter, int phaseIndex) {
    if (!rlTrafficLight_) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    const HostInter &in = net_->inters.at((size_t) inter);
    if (in.isVirtual || phaseIndex < 0 || phaseIndex >= (int) in.phases.size())
        throw std::out_of_range("phase index " + std::to_string(phaseIndex) + " out of range for intersection '" + in.id + "'");
    pendingPhaseInter_.push_back(inter);
    pendingPhaseValue_.push_back(phaseIndex);
}

void EngineHost::trafficLightState(std::vector<int32_t> &phase, std::vector<double> &remain) {
    flushPhases();
    phase.resize(net_->inters.size());
    remain.resize(net_->inters.size());
    check(be_.cfx_get_tl_state(dev_, phase.data(), remain.data()), "cfx_get_tl_state");
}

void EngineHost::laneHistory(std::vector<int32_t> &len, std::vector<int32_t> &vehicleNum, std::vector<double> &averageSpeed,
                             std::vector<int32_t> &historyVehicleNum, std::vector<double> &historyAverageSpeed) {
    const size_t nL = net_->lanes.size();
    len.assign(nL, 0);
    vehicleNum.assign(nL * CFX_LANE_HISTORY_MAX, 0);
    averageSpeed.assign(nL * CFX_LANE_HISTORY_MAX, 0.0);
    historyVehicleNum.assign(nL, 0);
    historyAverageSpeed.assign(nL, 0.0);
    cfx_lane_history h{(int32_t) nL, len.data(), vehicleNum.data(), averageSpeed.data(), historyVehicleNum.data(), historyAverageSpeed.data()};
    check(be_.cfx_get_lane_history(dev_, &h), "cfx_get_lane_history");
}

// setVehicleSpeed engine.cpp:827-834
void EngineHost::setVehicleSpeed(const std::string &id, double speed) {
    dropAhead();
    int vid = vidOf(id);
    if (vid < 0) {
        // pushed since the last step: the reference finds it in vehicleMap and keeps the speed in the vehicle's buffer for
        // its first step (Vehicle::setCustomSpeed vehicle.h:128-131).  The device keeps it for the number the vehicle will get.
        const int future = spawner_.pendingPushedVid(id);
        if (future == -2) return;  // (its route is invalid: planRoute drops it at the next step; the speed dies with it)
        if (future >= 0) {
            check(be_.cfx_set_vehicle_speed(dev_, future, speed), "cfx_set_vehicle_speed");
            waitingCustom_[future] = speed;
            return;
        }
    }
    uint8_t st = 2;
    if (vid >= 0) check(be_.cfx_get_vehicle_status(dev_, vid, 1, &st), "cfx_get_vehicle_status");
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + id + "' not found");
    check(be_.cfx_set_vehicle_speed(dev_, vid, speed), "cfx_set_vehicle_speed");
    // (a vehicle still in its lane's waiting buffer keeps the speed for its first step; cfx_state carries custom speeds of
    // running vehicles only, so compactVehicles hands these over itself)
    if (st == 0) waitingCustom_[vid] = speed;
}

// Engine::setRoute engine.cpp:852-866 + Router::setRoute router.cpp:245-264
bool EngineHost::setRoute(const std::string &vehicleId, const std::vector<std::string> &anchorIds) {
    dropAhead();
    int vid = vidOf(vehicleId);
    if (vid < 0) return false;
    int32_t state = 2, drivable = -1, routePos = -1, route = -1;
    check(be_.cfx_get_vehicle(dev_, vid, &state, &drivable, &routePos, &route), "cfx_get_vehicle");
    if (state == 2) return false;
    std::vector<int> anchors;
    for (const auto &id : anchorIds) {
        auto it = net_->roadIndex.find(id);
        if (it == net_->roadIndex.end()) return false;
        anchors.push_back(it->second);
    }
    const int L = (int) net_->lanes.size();
    if (state == 0) {  // still in a waiting buffer: its drivable is its first lane, iCurRoad = begin
        drivable = spawner_.vehicles[vid].firstLane;
        routePos = 0;
    }
    if (drivable >= L) return false;  // on a laneLink (router.cpp:246)
    // Router::setRoute starts the new route at *iCurRoad.  That is the road of the vehicle's lane — except between a load and
    // the vehicle's next lane, when the restarted cursor (archive.cpp) still names the route's FIRST road: the reference then
    // plans from there, and the result is good where that path leads through the road the vehicle is on (its cursor catches
    // up at the next lane); where it does not, the reference walks off its own route (assertion, router.cpp:58) — only then
    // the new route starts at the lane's road here.
    const int laneRoad = net_->lanes[drivable].road;
    const RouteTable &rt = spawner_.routes;
    const int cursorRoad = rt.roads[rt.routeStart[route] + routePos];
    std::vector<int> seq;
    auto plan = [&](int from) {
        std::vector<int> newAnchors{from};
        newAnchors.insert(newAnchors.end(), anchors.begin(), anchors.end());
        return spawner_.expandRoute(newAnchors, seq);
    };
    auto positionOf = [&](int road) {
        for (size_t i = 0; i < seq.size(); ++i)
            if (seq[i] == road) return (int) i;
        return -1;
    };
    if (!plan(cursorRoad)) return false;
    int pos = positionOf(laneRoad);
    if (pos < 0) {
        if (!plan(laneRoad)) return false;
        pos = 0;
    }
    int newRoute = spawner_.internRoute(seq);
    // Router::onValidLane (router.h:66-68) under the new route: a next drivable exists or this is the last road
    const RouteTable &rt2 = spawner_.routes;
    int laneIdx = net_->lanes[drivable].index;
    bool hasNext = rt2.nextLL[rt2.nextStart[rt2.routeStart[newRoute] + pos] + laneIdx] >= 0;
    bool lastRoad = seq.back() == laneRoad;
    if (!hasNext && !lastRoad) return false;
    uploadNewTablesIfAny();
    check(be_.cfx_set_vehicle_route(dev_, vid, newRoute), "cfx_set_vehicle_route");
    spawner_.setVehicleRoute(vid, newRoute);
    return true;
}

// pushVehicle(map, vector) engine.cpp:693-717
void EngineHost::pushVehicle(const std::map<std::string, double> &info, const std::vector<std::string> &roads) {
    dropAhead();
    settleLaneChange();  // the new vehicle's priority is drawn now: after the last step's shadow draws, as in the reference
    auto get = [&info](const char *k, double d) {
        auto it = info.find(k);
        return it == info.end() ? d : it->second;
    };
    // VehicleInfo defaults vehicle.h:31-45
    cfx_vehicle_template t = spawner_.makeTemplate(get("length", 5), get("width", 2), get("maxPosAcc", 4.5),
                                                   get("maxNegAcc", 4.5), get("usualPosAcc", 2.5), get("usualNegAcc", 2.5),
                                                   get("minGap", 2), get("maxSpeed", 16.66667), get("headwayTime", 1), get("speed", 0));
    std::vector<int> anchors;
    for (auto &r : roads) {
        auto it = net_->roadIndex.find(r);
        if (it == net_->roadIndex.end()) throw std::runtime_error("Road '" + r + "' not found");
        anchors.push_back(it->second);
    }
    if (anchors.empty()) throw std::runtime_error("push_vehicle: empty route");
    spawner_.pushManual(spawner_.addTemplate(t), anchors, step_);
}

}  // namespace cfa
