// Host-side flows, routes and the spawner.
//
// The spawner is the exact twin of the reference's per-step vehicle creation: it owns the engine's
// std::mt19937 and draws from it in the reference's order:
//   Flow::nextStep            flow.cpp:6-22      cadence, id flow_<i>_<cnt>
//   Vehicle::Vehicle          vehicle.cpp:38-47  priority = rnd() until unused
//   Engine::pushVehicle       engine.cpp:605-613 one more rnd() (% threadNum, value unused)
//   Engine::planRoute         engine.cpp:450-470 per road in JSON order, per vehicle in buffer order:
//   Router::getFirstDrivable  router.cpp:23-37   first lane = candidates[rnd() % n]
// Route expansion (Router::updateShortestPath router.cpp:228-243, dijkstra 160-226) is static for the
// LENGTH metric, so it is evaluated once per flow / per pushed vehicle instead of once per spawn.
#pragma once

#include <deque>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "cityflow_amd.h"
#include "flat_map.h"
#include "roadnet.h"

namespace cfa {

struct RouteTable {
    // CSR tables in the layout cfx_add_routes expects
    std::vector<int32_t> routeStart{0};
    std::vector<int32_t> roads;
    std::vector<int32_t> nextStart{0};
    std::vector<int32_t> nextLL;
    // host-only: candidate first lanes per route (Router::getFirstDrivable)
    std::vector<std::vector<int32_t>> firstLanes;

    int count() const { return (int) routeStart.size() - 1; }
    // Appends an (already expanded) road sequence and returns its route index.
    int add(const HostRoadNet &net, const std::vector<int> &roadSeq);
};

struct HostFlow {
    std::string id;
    int templ = -1;
    std::vector<int> anchors;
    int route = -1;  // -1: Dijkstra failed / route.size() <= 1  => invalid flow
    double interval = 0;
    int startTime = 0, endTime = -1;
    // dynamic state (Flow fields flow.h:23-31)
    double nowTime = 0, currentTime = 0;
    int cnt = 0;
    bool valid = true;
};

// What the host remembers about every vehicle it ever spawned (index = vid).
struct VehicleRecord {
    int32_t priority;
    int32_t flow;      // -1 for push_vehicle
    int32_t number;    // per-flow counter or manuallyPushCnt value
    int32_t templ, route;
    double enterTime;
    int32_t firstLane;  // lane whose waiting buffer it was pushed to (-1 while still in planRouteBuffer)
    int32_t root = -1;  // lane change: the flow / pushed vehicle whose id this shadow carries (-1: not a shadow)
};

struct FlowDyn {  // Flow fields that change while stepping (flow.h:23-31)
    double nowTime, currentTime;
    int cnt;
    bool valid;
};

class Spawner {
public:
    std::vector<cfx_vehicle_template> templates;
    RouteTable routes;
    std::vector<HostFlow> flows;
    // by vid, since the last reset.  A deque: it grows chunk by chunk — a vector of a million 40-byte records relocates
    // tens of megabytes when it doubles, a 5-10 ms next_step() in the middle of a long run (tests/test_steady_state.py)
    std::deque<VehicleRecord> vehicles;
    std::vector<std::vector<int32_t>> flowVids;  // [flow][per-flow number - flowVidBase[flow]] -> vid (-1: dropped, invalid route)
    std::vector<int32_t> manualVids;             // [manuallyPushCnt value - manualVidBase] -> vid or -1
    // numbers below the base belong to vehicles that have finished and been forgotten (compactedState)
    std::vector<int32_t> flowVidBase;
    int32_t manualVidBase = 0;
    std::mt19937 rnd;

    void init(const HostRoadNet *net, double interval, int threadNum, int seed);
    void loadFlows(const std::string &path);  // Engine::loadFlow engine.cpp:106-164

    // Expand anchors into a road route; returns false when the reference would mark it invalid.
    bool expandRoute(const std::vector<int> &anchors, std::vector<int> &out) const;
    int addTemplate(const cfx_vehicle_template &t);  // dedupes identical templates
    cfx_vehicle_template makeTemplate(double len, double width, double maxPosAcc, double maxNegAcc, double usualPosAcc,
                                      double usualNegAcc, double minGap, double maxSpeed, double headwayTime,
                                      double initialSpeed = 0.0) const;

    // push_vehicle (engine.cpp:693-717): queued into the first road's planRouteBuffer until the next step.
    void pushManual(int templ, const std::vector<int> &anchors, size_t step);

    // One step worth of spawn records (phases 0-1 of Engine::nextStep).
    void step(size_t stepIndex, std::vector<cfx_spawn> &out);

    // `isFinished(vid)` is consulted only when a freshly drawn priority collides with a vehicle the
    // host still believes alive (the host does not see vehicles finish; the device does).
    void setFinishedQuery(std::function<bool(int)> q) { isFinished_ = std::move(q); }

    void reset(bool reseed);  // Engine::reset engine.cpp:744-760 (flows, vehicles; RNG only if reseed)
    void seed(int s) { rnd.seed((std::mt19937::result_type) s); }

    // ---- lane change: shadows are `new Vehicle(*v, id + "_shadow")` drawing a priority from this generator
    //      (engine.cpp:812-820, vehicle.cpp:28-36)
    // The priorities the next n shadows WOULD get, from a copy of the generator (collisions with live vehicles redrawn).
    void peekShadowPriorities(int n, std::vector<int32_t> &out);
    bool exactPeekOnly = false;  // always take the exact redraw loop (test hook, config "cfx": {"exactShadowPeek": true})
    // The device created shadows of `parents` (in this order) with the first priorities of the last peek: advance the
    // generator past those draws and give the shadows their vehicle numbers.
    void commitShadows(const std::vector<int32_t> &parents);
    // vehicles that carry (or carried) the id of `root`: the root itself and every shadow made of its holders, ascending
    std::vector<int32_t> idChain(int root) const;

    std::string vehicleId(int vid, bool shadow = false) const;  // shadow: "<id>_shadow" (until its change completes)
    int vidOfId(const std::string &id) const;  // inverse of vehicleId (-1 if unknown)
    // vehicles pushed (push_vehicle) since the last step: they get their vehicle numbers with the next step's spawn records,
    // but the reference already lists them (Engine::pushVehicle puts them into vehiclePool at once, engine.cpp:605-613)
    void pendingPushed(std::vector<std::pair<int32_t, std::string>> &priorityAndId) const;
    // The vehicle number the next step will give a vehicle pushed since the last step (push_vehicle creates its vehicles
    // before the step's flows create theirs): -1 unknown id, -2 known but its route is invalid (it never enters the network)
    int pendingPushedVid(const std::string &id) const;
    // An integer that orders vehicles exactly like their id strings compare ("flow_<f>_<n>" / "manually_pushed_<n>", i.e.
    // the key order of the reference's std::map<std::string, ...> getters) without building or comparing strings.
    uint64_t idSortKey(int vid) const;
    int initialSeed() const { return seed_; }

    // Route index for an expanded road sequence, adding it if it is new (used by set_vehicle_route and
    // load_from_file, which would otherwise add one route per vehicle).
    int internRoute(const std::vector<int> &roadSeq);
    void setVehicleRoute(int vid, int route) { vehicles[vid].route = route; }

    // Everything that changes while stepping, for Archive-style snapshot / restore (archive.cpp:62-66,161-165).
    struct State {
        std::vector<FlowDyn> flows;
        std::deque<VehicleRecord> vehicles;
        std::vector<std::vector<int32_t>> flowVids;
        std::vector<int32_t> manualVids, lastWaitVid;
        std::vector<int32_t> flowVidBase;
        int32_t manualVidBase = 0;
        std::mt19937 rnd;
        int manualCnt = 0;
        FlatMapI32 livePriority;
        std::unordered_map<int32_t, std::vector<int32_t>> shadowChains;
    };
    State saveState() const;
    void loadState(const State &st);
    // The state with the finished vehicles forgotten and the others renumbered (newOfOld[vid] = new number, -1 = finished;
    // ascending, so that creation order — what exact-distance ties and the lane-change walk go by — is kept): what the host
    // remembers per vehicle is then bounded by the vehicles alive, not by the vehicles created (EngineHost::compactVehicles).
    State compactedState(const std::vector<int32_t> &newOfOld, int nLive) const;
    int manualCount() const { return manualCnt_; }

    // ---- a step taken AHEAD of time.  The host of a single engine runs the spawner of step t+1 right after it has handed
    //      step t to the device (the records depend on nothing the device computes, except through the rare priority
    //      collision, which asks the device as it always does): the work overlaps the device's step instead of standing
    //      between a caller's observation and its next step.  Until the next step consumes the records (commitAhead) any
    //      other call that could see or change the spawner's state takes the step back first (rollbackAhead): everything
    //      step() changes is journalled while `ahead` is on.
    void beginAhead();
    void commitAhead();
    void rollbackAhead();
    bool aheadActive() const { return journal_.active; }
    // vehicles created by the steps that were TAKEN (a journalled step ahead not counted)
    size_t committedVehicleCount() const { return journal_.active ? journal_.nVehicles : vehicles.size(); }

private:
    struct Pending {
        int index;  // into pendingRecords_
        int firstRoad;
    };
    void rebuildActiveFlows();
    int newVehicle(int flow, int number, int templ, const std::vector<int> &anchors, int route, size_t stepIndex,
                   const std::function<bool(int)> &isFinished);

    struct PriorityUndo {
        int32_t key, old;
        bool had;
    };
    struct Journal {
        bool active = false;
        std::mt19937 rnd;
        std::vector<int32_t> activeFlows;
        std::vector<FlowDyn> flowDyn;  // of activeFlows, in that order
        size_t nVehicles = 0;
        std::vector<PriorityUndo> priorities;
        std::vector<std::pair<int32_t, size_t>> vidTables;  // (flow, or -1 = manualVids; its size before the append)
        std::vector<std::pair<int32_t, int32_t>> lastWait;  // (lane, what it held)
    };
    Journal journal_;
    void prioritySet(int32_t key, int32_t value);
    void priorityErase(int32_t key);

    const HostRoadNet *net_ = nullptr;
    double interval_ = 1.0;
    int threadNum_ = 1;
    int seed_ = 0;
    int manualCnt_ = 0;
    std::function<bool(int)> isFinished_;
    std::vector<Pending> pending_;                 // planRouteBuffer contents, in push order
    std::vector<VehicleRecord> pendingRecords_;
    std::vector<int32_t> vidOfPending_;            // scratch of step(): vehicle number of every pending record
    std::vector<int32_t> lastWaitVid_;             // per lane: last vid pushed to its waitingBuffer
    FlatMapI32 livePriority_;  // priority -> vid (superset of live vehicles; -1 while in planRouteBuffer)
    std::vector<int32_t> activeFlows_;  // flows that may still spawn, ascending (Flow::nextStep early returns)
    std::vector<int32_t> peekPriorities_, peekDraws_;  // last peekShadowPriorities: values, cumulative raw draws
    std::vector<int64_t> peekTable_;  // scratch of peekShadowPriorities
    std::unordered_map<int32_t, std::vector<int32_t>> shadowChains_;  // root vid -> shadows carrying its id
    std::map<std::vector<int>, int> routeIndex_;          // expanded road sequence -> route index
};

}  // namespace cfa
