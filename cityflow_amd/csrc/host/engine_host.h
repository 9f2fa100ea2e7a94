// C++ host of the drop-in `cityflow.Engine` (reference src/engine/engine.h:114-183, bound by
// src/cityflow.cpp:11-38).  It keeps everything that involves strings, JSON and the mt19937 stream on
// the CPU and drives the device engine exclusively through the C ABI of include/cityflow_amd.h, which
// it resolves from a shared library at run time:
//   * default: <package dir>/lib/libcfx_hip.so  — the HIP/gfx950 implementation (the product);
//   * tests may name another library exporting the same ABI (the CPU twin under oracle/).
// There is no CPU fallback: if the library cannot be loaded or cfx_create fails the constructor throws.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "archive.h"
#include "cityflow_amd.h"
#include "flow.h"
#include "replay.h"
#include "roadnet.h"

namespace cfa {

struct Backend {
    void *handle = nullptr;
    std::string path;
#define CFX_FN(name) decltype(&::name) name = nullptr;
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
    void open(const std::string &libPath);  // throws std::runtime_error
    ~Backend();
};

// Per-vehicle state downloaded from the device, in Drivable::vehicles order.
enum SnapshotField : unsigned {  // what snapshotVehicles should fetch besides the vehicle ids
    kSnapDrivable = 1, kSnapPrev = 2, kSnapLeader = 4, kSnapBlocker = 8, kSnapEnterLL = 16, kSnapRoutePos = 32,
    kSnapDis = 64, kSnapSpeed = 128, kSnapGap = 256, kSnapAll = 511,
    kSnapLaneChange = 512  // partner / flags / offset / lastDir (always fetched when the engine runs with laneChange)
};

struct VehicleSnapshot {
    std::vector<int32_t> vid, drivable, prevDrivable, leader, blocker, enterLLTime, routePos;
    std::vector<double> dis, speed, gap;
    std::vector<int32_t> lcPartner, lcLastDir, lcTarget, lcDirection;  // lane change (empty unless kSnapLaneChange)
    std::vector<uint8_t> lcFlags;               // CFX_LC_* bits
    std::vector<double> lcOffset, lcLastChangeTime, lcWaitingTime;
    int count = 0;
    bool isShadow(int i) const { return !lcFlags.empty() && (lcFlags[i] & CFX_LC_SHADOW); }
};

class EngineHost {
public:
    EngineHost(const std::string &configFile, int threadNum, const std::string &backendLib = "");
    ~EngineHost();
    EngineHost(const EngineHost &) = delete;
    EngineHost &operator=(const EngineHost &) = delete;

    // ---- reference API (engine.h:139-182) ----
    void nextStep();
    size_t getVehicleCount();
    std::vector<std::string> getVehicles(bool includeWaiting);
    std::map<std::string, int> getLaneVehicleCount();
    std::map<std::string, int> getLaneWaitingVehicleCount();
    std::map<std::string, std::vector<std::string>> getLaneVehicles();
    std::map<std::string, double> getVehicleSpeed();
    std::map<std::string, double> getVehicleDistance();
    std::string getLeader(const std::string &vehicleId);
    std::map<std::string, std::string> getVehicleInfo(const std::string &vehicleId);
    double getCurrentTime() const { return step_ * interval_; }
    double getAverageTravelTime();
    void setTrafficLightPhase(const std::string &id, int phaseIndex);
    bool keepsLaneHistory() const { return laneHistory_; }
    void setTrafficLightPhaseOf(int inter, int phaseIndex);  // the same, the id already resolved to its index in net().inters
    void setRandomSeed(int seed) {
        dropAhead();
        settleLaneChange();  // (the pending shadow draws belong to the old stream)
        spawner_.seed(seed);
    }
    void pushVehicle(const std::map<std::string, double> &info, const std::vector<std::string> &roads);
    void reset(bool resetRnd);
    void setVehicleSpeed(const std::string &id, double speed);                        // engine.cpp:827-834
    bool setRoute(const std::string &vehicleId, const std::vector<std::string> &anchors);  // engine.cpp:852-866
    Archive snapshot() { return snapshotImpl(true); }  // engine.h:177
    // Forget the finished vehicles (archive.cpp).  Automatic, between two steps, once `compactAt_` vehicles have been created
    // since the last time ("cfx": {"compactVehicles": N}; default 3.5 M = before the device's vehicle tables would double; 0 = never).
    void compactVehicles();
    size_t vehicleTableSize() const { return spawner_.vehicles.size(); }
    int64_t vehicleCompactions() const { return vehicleCompactions_; }
    void load(const Archive &archive);         // engine.h:176
    void loadFromFile(const std::string &path);  // engine.cpp:822-825
    void setReplayLogFile(const std::string &logFile);  // engine.cpp:727-734
    void setSaveReplay(bool open);                       // engine.cpp:736-742

    // ---- array getters (no string marshalling; for large grids / RL observation tensors) ----
    std::vector<int32_t> laneVehicleCountArray();
    std::vector<int32_t> laneWaitingVehicleCountArray();
    std::vector<std::string> laneIds() const;          // index -> id for the two arrays above
    const std::vector<int32_t> &laneIdOrder();         // lane indices in lexicographic id order (std::map order)
    std::vector<std::string> intersectionIds() const;
    void setTrafficLightPhaseIndexed(int inter, int phase);
    void setTrafficLightPhases(const std::vector<int32_t> &phases);  // [n_intersections]; virtual ones ignored
    void setTrafficLightPhases(const int32_t *phases, size_t n);
    void trafficLightState(std::vector<int32_t> &phase, std::vector<double> &remain);
    // Lane::history as the device keeps it ("cfx": {"laneHistory": true}; cfx_get_lane_history): lane-major, oldest record first
    void laneHistory(std::vector<int32_t> &len, std::vector<int32_t> &vehicleNum, std::vector<double> &averageSpeed,
                     std::vector<int32_t> &historyVehicleNum, std::vector<double> &historyAverageSpeed);
    void snapshotVehicles(VehicleSnapshot &out, unsigned fields = kSnapAll);
    // changes whenever vehicle numbers are reassigned (reset, load): per-vehicle caches of a language binding key on it
    uint64_t vehicleEpoch() const { return vehicleEpoch_; }
    void waitingVehicles(std::vector<int32_t> &vid, std::vector<int32_t> &lane);
    cfx_scalars scalars();
    void sync();
    void profileEnable(bool on);
    void deviceSpin(long long microseconds) { check(be_.cfx_device_spin(dev_, microseconds), "cfx_device_spin"); }
    std::pair<long long, long long> deviceMemory() {  // {free, total} bytes of the engine's device
        int64_t f = 0, t = 0;
        check(be_.cfx_device_memory(dev_, &f, &t), "cfx_device_memory");
        return {(long long) f, (long long) t};
    }
    std::map<std::string, std::pair<double, int64_t>> profileRead();  // kernel -> (total ms, launches)
    std::map<std::string, std::string> profileSymbols();  // timing slot -> symbol of the kernel launched last in it
    cfx_host_stats hostStats(bool reset);
    // the slowest nextStep() since the last clearing read: total us, the step, and its parts {phases + lane-change settle, this
    // step's spawn records (taken or made), table upload + lane-change supply, cfx_step, replay log, the spawner of the step ahead}
    struct SlowestStep {
        double total = 0, part[6] = {};
        int64_t step = -1;
    };
    SlowestStep slowestStep(bool reset) {
        SlowestStep s = slowest_;
        if (reset) slowest_ = SlowestStep{};
        return s;
    }

    std::shared_ptr<void> bindingCache;  // opaque per-engine cache owned by the language binding (lane id key objects)

    const HostRoadNet &net() const { return *net_; }
    const Spawner &spawner() {  // (the state as of the last step: a step taken ahead is taken back first)
        dropAhead();
        return spawner_;
    }
    std::string backendName() const { return be_.cfx_backend_name ? be_.cfx_backend_name() : "?"; }
    std::pair<int64_t, int> ringInfo() {
        int64_t slots = 0;
        int32_t scale = 1;
        be_.cfx_get_ring_info(dev_, &slots, &scale);
        return {slots, scale};
    }
    std::string layoutName() {
        const int l = be_.cfx_get_layout(dev_);
        return l == CFX_LAYOUT_RING ? "ring" : (l == CFX_LAYOUT_DENSE ? "dense" : "n/a");
    }
    std::string vehicleId(int vid, bool shadow = false) const { return spawner_.vehicleId(vid, shadow); }
    bool laneChange() const { return laneChange_; }
    bool isPendingPushed(const std::string &id);
    int vidOf(const std::string &id);  // -1 if unknown
    double interval() const { return interval_; }
    size_t step() const { return step_; }

private:
    void check(int32_t rc, const char *what);
    void uploadNewTablesIfAny();

    std::shared_ptr<HostRoadNet> net_ = std::make_shared<HostRoadNet>();
    Spawner spawner_;
    Backend be_;
    cfx_engine *dev_ = nullptr;
    double interval_ = 1.0;
    bool rlTrafficLight_ = false, laneChange_ = false, saveReplay_ = false, saveReplayInConfig_ = false;
    Archive snapshotImpl(bool hostState);
    std::map<int32_t, double> waitingCustom_;  // set_vehicle_speed on vehicles that were waiting (or not yet numbered) then
    size_t compactAt_ = 3500000, nextCompactAt_ = 3500000;
    bool compactAuto_ = true;
    int64_t vehicleCompactions_ = 0;
    bool laneHistory_ = false;  // "cfx": {"laneHistory": ...}; not said: kept on the ring layout up to 20 k lanes, not with lane change
    ReplayWriter replay_;
    void updateLog();  // Engine::updateLog engine.cpp:518-554
    int seed_ = 0, threadNum_ = 1;
    std::string dir_;
    size_t step_ = 0;
    int templatesUploaded_ = 0, routesUploaded_ = 0;
    std::vector<cfx_spawn> spawnBuf_;
    // The NEXT step's spawn records, computed right after this step was handed to the device (Spawner::beginAhead): consumed
    // by the next nextStep(), taken back (dropAhead) by every other call that could see or change the spawner's state.
    bool spawnAhead_ = true, aheadValid_ = false;
    SlowestStep slowest_;
    std::vector<cfx_spawn> aheadBuf_;
    void dropAhead();
    std::vector<int32_t> shadowPool_, shadowParents_;  // lane change: priorities offered to / parents reported by a step
    int shadowPoolSize_ = 1024;
    bool lcPollPending_ = false;
    void settleLaneChange();
    std::vector<int32_t> pendingPhaseInter_, pendingPhaseValue_;  // set_tl_phase calls since the last flush
    // rlTrafficLight: the phase the device shows at every intersection, as far as this host knows (-1: not known) — with agent
    // control nothing but these calls changes a phase (TrafficLight::passTime does not run, trafficlight.cpp:29-37), so an
    // agent that sets every signal every step uploads only the ones it actually changes; forgotten on reset / load
    std::vector<int32_t> knownPhase_;
    std::vector<int32_t> realInter_, realInterPhases_, changedInter_, changedPhase_;  // setTrafficLightPhases: the signals, scratch
    void forgetPhases() { knownPhase_.assign(knownPhase_.size(), -1); }
    // keeps of (inters, phases) what differs from knownPhase_ and records it; false: nothing left to send
    bool onlyChangedPhases(std::vector<int32_t> &inters, std::vector<int32_t> &phases);
    void flushPhases();
    std::vector<int32_t> laneIdOrder_;
    uint64_t vehicleEpoch_ = 0;
};

struct EngineConfig {  // Engine::loadConfig engine.cpp:37-84
    double interval = 1.0;
    bool rlTrafficLight = false, laneChange = false, saveReplay = false;
    int seed = 0;
    std::string dir, roadnetFile, flowFile;
    std::string roadnetLogFile, replayLogFile;  // required with saveReplay (engine.cpp:73-77)
    // optional "cfx" object (ignored by the reference): implementation choices that never change results
    int crossMode = CFX_CROSS_AUTO, layout = CFX_LAYOUT_AUTO, debugSync = 0, device = -1, ringLanesPerWave = 0, ringCapacityPercent = 0, denseForm = 0;
    bool exactShadowPeek = false;  // host: Spawner::exactPeekOnly
    int laneHistory = -1;          // keep Lane::history on the device (cfx_config::lane_history): Archive dumps then carry it, as the
                                   // reference's do.  -1 = not said: Engine keeps it where it is nearly free (ring layout, up to 20 k
                                   // lanes, no lane change: engine_host.cpp), VectorEngine and TiledEngine (no Archive of it) do not
    int hostThreads = -1;          // VectorEngine: worker threads for the per-environment host work (-1 auto, 0 serial)
    int64_t compactVehicles = -1;  // Engine: forget the finished vehicles once this many have been created since the last time (-1: 3.5 M; 0: never)
    bool spawnAhead = true;        // Engine: run the spawner of step t+1 right after step t is handed to the device (EngineHost::nextStep)
    void apply(cfx_config &cc) const;  // interval, flags, the choices above, device (config > CITYFLOW_AMD_DEVICE > LOCAL_RANK)
};
EngineConfig readEngineConfig(const std::string &configFile);  // throws std::runtime_error("load config failed! ...")

std::string defaultBackendPath();  // <dir of this shared object>/lib/libcfx_hip.so

}  // namespace cfa
