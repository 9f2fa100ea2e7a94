#include "flow.h"

#include <cstdio>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <queue>
#include <set>

#include "json.h"

namespace cfa {

// Per-position next-laneLink table: the value Router::getNextDrivable(const Drivable*) (router.cpp:49-76)
// returns for a lane of road[p], with the selection rule of selectLaneIndex (router.cpp:96-111):
// first candidate with the smallest |endLaneIndex - curLaneIndex|.
int RouteTable::add(const HostRoadNet &net, const std::vector<int> &seq) {
    const int n = (int) seq.size();
    for (int p = 0; p < n; ++p) {
        roads.push_back(seq[p]);
        const HostRoad &road = net.roads[seq[p]];
        for (int j = 0; j < road.nLanes; ++j) {
            int lane = road.laneStart + j;
            int chosen = -1;
            if (p < n - 1) {
                std::vector<int> cands = net.laneLinksToRoad(lane, seq[p + 1]);
                if (p < n - 2) {
                    std::vector<int> filtered;
                    for (int ll : cands)
                        if (!net.laneLinksToRoad(net.laneLinks[ll].endLane, seq[p + 2]).empty()) filtered.push_back(ll);
                    cands.swap(filtered);
                }
                int laneDiff = INT32_MAX;
                for (int ll : cands) {
                    int d = std::abs(net.lanes[net.laneLinks[ll].endLane].index - net.lanes[lane].index);
                    if (d < laneDiff) {
                        laneDiff = d;
                        chosen = ll;
                    }
                }
            }
            nextLL.push_back(chosen);
        }
        nextStart.push_back((int32_t) nextLL.size());
    }
    routeStart.push_back((int32_t) roads.size());
    // Router::getFirstDrivable router.cpp:23-37
    std::vector<int32_t> first;
    const HostRoad &r0 = net.roads[seq[0]];
    for (int j = 0; j < r0.nLanes; ++j) {
        int lane = r0.laneStart + j;
        if (n == 1 || !net.laneLinksToRoad(lane, seq[1]).empty()) first.push_back(lane);
    }
    firstLanes.push_back(std::move(first));
    return count() - 1;
}

void Spawner::init(const HostRoadNet *net, double interval, int threadNum, int seed) {
    net_ = net;
    interval_ = interval;
    threadNum_ = threadNum < 1 ? 1 : threadNum;
    seed_ = seed;
    rnd.seed((std::mt19937::result_type) seed);
    lastWaitVid_.assign(net->lanes.size(), -1);
}

cfx_vehicle_template Spawner::makeTemplate(double len, double width, double maxPosAcc, double maxNegAcc,
                                           double usualPosAcc, double usualNegAcc, double minGap, double maxSpeed,
                                           double headwayTime, double initialSpeed) const {
    cfx_vehicle_template t{};
    t.len = len;
    t.width = width;
    t.max_pos_acc = maxPosAcc;
    t.max_neg_acc = maxNegAcc;
    t.usual_pos_acc = usualPosAcc;
    t.usual_neg_acc = usualNegAcc;
    t.min_gap = minGap;
    t.max_speed = maxSpeed;
    t.headway_time = headwayTime;
    t.yield_distance = 5;    // VehicleInfo defaults vehicle.h:42-43 (not settable from flow JSON)
    t.turn_speed = 8.3333;
    // vehicle.cpp:42-44, same grouping
    t.approach_dist = maxSpeed * maxSpeed / usualNegAcc / 2 + maxSpeed * interval_ * 2;
    t.initial_speed = initialSpeed;
    return t;
}

int Spawner::addTemplate(const cfx_vehicle_template &t) {
    for (size_t i = 0; i < templates.size(); ++i)
        if (memcmp(&templates[i], &t, sizeof t) == 0) return (int) i;
    templates.push_back(t);
    return (int) templates.size() - 1;
}

// Shortest road path between two anchors of a route (reference: Router::dijkstra router.cpp:160-226, RouterType::LENGTH;
// DURATION is never selected by the reference's own API).  What has to match the reference is WHICH of several equally
// long paths comes out — grids are full of ties — and that is decided by the order in which the frontier is popped.  The
// reference keeps the frontier in a std::priority_queue ordered by cost alone, i.e. std::push_heap / std::pop_heap on a
// vector; the frontier below is that vector with those two calls and the same push sequence (roads of the end
// intersection in file order, every strict improvement pushed again, stale entries skipped when popped), so ties fall the
// same way.  Costs and parents live in dense per-road arrays that are reused from call to call (a stamp marks the
// entries of the current search), not in node-based maps.
namespace {
struct RoadSearch {
    std::vector<double> cost;
    std::vector<int> parent;
    std::vector<uint32_t> reached, settled;  // == stamp: valid for the search in progress
    std::vector<std::pair<int, double>> frontier;
    uint32_t stamp = 0;
    void begin(size_t nRoads) {
        if (cost.size() < nRoads) {
            cost.resize(nRoads);
            parent.resize(nRoads);
            reached.assign(nRoads, 0);
            settled.assign(nRoads, 0);
            stamp = 0;
        }
        if (++stamp == 0) {  // wrapped: forget everything
            std::fill(reached.begin(), reached.end(), 0u);
            std::fill(settled.begin(), settled.end(), 0u);
            stamp = 1;
        }
        frontier.clear();
    }
};
}  // namespace

static bool shortestRoadPath(const HostRoadNet &net, int start, int end, std::vector<int> &out) {
    static thread_local RoadSearch rs;
    rs.begin(net.roads.size());
    const auto costlier = [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.second > b.second; };
    rs.cost[start] = 0.0;
    rs.reached[start] = rs.stamp;
    rs.parent[start] = -1;
    rs.frontier.emplace_back(start, 0.0);
    bool arrived = false;
    while (!rs.frontier.empty()) {
        const int road = rs.frontier.front().first;
        if (road == end) {  // (checked before the pop, like the reference: the cheapest frontier entry is the target)
            arrived = true;
            break;
        }
        std::pop_heap(rs.frontier.begin(), rs.frontier.end(), costlier);
        rs.frontier.pop_back();
        if (rs.settled[road] == rs.stamp) continue;
        rs.settled[road] = rs.stamp;
        const double here = rs.cost[road];
        for (int next : net.inters[net.roads[road].endInter].roads) {
            if (!net.connectedToRoad(road, next)) continue;
            const double there = here + net.averageLength(next);
            if (rs.reached[next] == rs.stamp && !(there < rs.cost[next])) continue;
            rs.reached[next] = rs.stamp;
            rs.cost[next] = there;
            rs.parent[next] = road;
            rs.frontier.emplace_back(next, there);
            std::push_heap(rs.frontier.begin(), rs.frontier.end(), costlier);
        }
    }
    // the roads after `start` up to and including `end` (just `end` when it was never reached, like the reference)
    const size_t mark = out.size();
    out.push_back(end);
    for (int r = rs.reached[end] == rs.stamp ? rs.parent[end] : -1; r >= 0 && r != start; r = rs.parent[r]) out.push_back(r);
    std::reverse(out.begin() + (std::ptrdiff_t) mark, out.end());
    return arrived;
}

// Router::updateShortestPath router.cpp:228-243
bool Spawner::expandRoute(const std::vector<int> &anchors, std::vector<int> &out) const {
    out.clear();
    if (anchors.empty()) return false;
    out.push_back(anchors[0]);
    for (size_t i = 1; i < anchors.size(); ++i) {
        if (anchors[i - 1] == anchors[i]) continue;
        if (!shortestRoadPath(*net_, anchors[i - 1], anchors[i], out)) return false;
    }
    return out.size() > 1;
}

void Spawner::loadFlows(const std::string &path) {
    Json root = Json::parseFile(path);
    if (!root.isArray()) throw JsonError("flow file: expected type array");
    for (size_t i = 0; i < root.items.size(); ++i) {
        const Json &fv = root.items[i];
        HostFlow f;
        f.id = "flow_" + std::to_string(i);
        for (const Json &r : fv.arrayAt("route").items) {
            if (!r.isString()) throw JsonError("route: expected type string");
            auto it = net_->roadIndex.find(r.s);
            if (it == net_->roadIndex.end()) throw JsonError("No such road: " + r.s);
            f.anchors.push_back(it->second);
        }
        if (f.anchors.empty()) throw JsonError("flow[" + std::to_string(i) + "]: empty route");
        const Json &v = fv.objectAt("vehicle");
        f.templ = addTemplate(makeTemplate(v.numberAt("length"), v.numberAt("width"), v.numberAt("maxPosAcc"),
                                           v.numberAt("maxNegAcc"), v.numberAt("usualPosAcc"), v.numberAt("usualNegAcc"),
                                           v.numberAt("minGap"), v.numberAt("maxSpeed"), v.numberAt("headwayTime")));
        f.startTime = fv.intAt("startTime", 0);
        f.endTime = fv.intAt("endTime", -1);
        f.interval = fv.numberAt("interval");
        f.nowTime = f.interval;  // Flow ctor flow.h:30-36
        std::vector<int> seq;
        f.route = expandRoute(f.anchors, seq) ? internRoute(seq) : -1;
        flows.push_back(std::move(f));
    }
    flowVids.assign(flows.size(), {});
    flowVidBase.assign(flows.size(), 0);
    manualVidBase = 0;
    rebuildActiveFlows();
    // Size the priority table for the first simulated hour of this demand, so that it is not rebuilt (a stall of tens of
    // milliseconds at city scale) in the middle of a run; it still grows on demand after that.
    double expected = 0;
    for (const HostFlow &f : flows) {
        if (!f.valid || !(f.interval > 0)) continue;
        double until = f.endTime >= 0 ? std::min(3600.0, (double) f.endTime) : 3600.0;
        if (until > f.startTime) expected += (until - f.startTime) / f.interval + 1;
    }
    livePriority_.reserve((size_t) std::min(expected, 16.0e6));
}

void Spawner::rebuildActiveFlows() {
    activeFlows_.clear();
    for (size_t i = 0; i < flows.size(); ++i) activeFlows_.push_back((int32_t) i);
}

std::string Spawner::vehicleId(int vid, bool shadow) const {
    const VehicleRecord &r = vehicles[vid];
    std::string id = r.flow >= 0 ? flows[r.flow].id + "_" + std::to_string(r.number)
                                 : "manually_pushed_" + std::to_string(r.number);
    return shadow ? id + "_shadow" : id;
}

void Spawner::peekShadowPriorities(int n, std::vector<int32_t> &out) {
    // Fast path (practically always): n plain draws, none of which meets a live priority or repeats — then entry i is draw
    // i.  Anything else goes through the exact loop below (CFX_LC_PEEK_EXACT=1 forces it: the tests compare the two).
    static const bool exactOnly = getenv("CFX_LC_PEEK_EXACT") != nullptr;
    if (!exactOnly) {
        std::mt19937 peek = rnd;
        peekPriorities_.resize((size_t) n);
        for (int i = 0; i < n; ++i) {
            peekPriorities_[(size_t) i] = (int32_t) peek();
            livePriority_.prefetch(peekPriorities_[(size_t) i]);
        }
        bool clean = true;
        for (int i = 0; i < n && clean; ++i) clean = !livePriority_.contains(peekPriorities_[(size_t) i]);
        if (clean) {  // no value twice: a small open-addressing table (entries are value + 2^32, empty = -1)
            size_t cap = 64;
            while (cap < (size_t) n * 4) cap <<= 1;
            peekTable_.assign(cap, -1);
            for (int i = 0; i < n && clean; ++i) {
                const int64_t key = (int64_t) peekPriorities_[(size_t) i] + (1LL << 32);
                size_t h = ((uint64_t) key * 0x9E3779B97F4A7C15ULL) >> 20 & (cap - 1);
                while (peekTable_[h] != -1 && peekTable_[h] != key) h = (h + 1) & (cap - 1);
                clean = peekTable_[h] == -1;
                peekTable_[h] = key;
            }
        }
        if (clean) {
            peekDraws_.resize((size_t) n);
            for (int i = 0; i < n; ++i) peekDraws_[(size_t) i] = i + 1;
            out = peekPriorities_;
            return;
        }
    }
    std::mt19937 peek = rnd;
    peekPriorities_.clear();
    peekDraws_.clear();
    int draws = 0;
    for (int i = 0; i < n; ++i) {
        int32_t priority;
        for (;;) {  // `while (engine->checkPriority(priority = engine->rnd()));` vehicle.cpp:33
            priority = (int32_t) peek();
            ++draws;
            if (std::find(peekPriorities_.begin(), peekPriorities_.end(), priority) != peekPriorities_.end()) continue;
            int32_t owner;
            if (!livePriority_.lookup(priority, owner)) break;
            if (owner >= 0 && isFinished_ && isFinished_(owner)) {
                livePriority_.erase(priority);
                break;
            }
        }
        peekPriorities_.push_back(priority);
        peekDraws_.push_back(draws);
    }
    out = peekPriorities_;
}

void Spawner::commitShadows(const std::vector<int32_t> &parents) {
    if (parents.empty()) return;
    if (parents.size() > peekPriorities_.size()) throw std::runtime_error("lane change: more shadows than peeked priorities");
    rnd.discard((unsigned long long) peekDraws_[parents.size() - 1]);
    for (size_t i = 0; i < parents.size(); ++i) {
        VehicleRecord rec = vehicles[(size_t) parents[i]];  // Vehicle copy constructor: same info, route, enterTime
        rec.priority = peekPriorities_[i];
        rec.root = rec.root >= 0 ? rec.root : parents[i];
        const int vid = (int) vehicles.size();
        vehicles.push_back(rec);
        livePriority_.set(rec.priority, vid);
        shadowChains_[rec.root].push_back(vid);
    }
    peekPriorities_.clear();
    peekDraws_.clear();
}

std::vector<int32_t> Spawner::idChain(int root) const {
    std::vector<int32_t> c{root};
    auto it = shadowChains_.find(root);
    if (it != shadowChains_.end()) c.insert(c.end(), it->second.begin(), it->second.end());
    return c;
}

// Vehicle ctor priority loop (vehicle.cpp:45) + the extra draw of Engine::pushVehicle (engine.cpp:606).
int Spawner::newVehicle(int flow, int number, int templ, const std::vector<int> &anchors, int route, size_t stepIndex,
                        const std::function<bool(int)> &isFinished) {
    int32_t priority;
    for (;;) {
        priority = (int32_t) rnd();
        int32_t owner;
        if (!livePriority_.lookup(priority, owner)) break;
        // Host bookkeeping is a superset of the live set; settle it exactly before redrawing.
        if (owner >= 0 && isFinished && isFinished(owner)) {
            priorityErase(priority);
            break;
        }
    }
    (void) (rnd() % (std::mt19937::result_type) threadNum_);
    VehicleRecord rec{};
    rec.priority = priority;
    rec.flow = flow;
    rec.number = number;
    rec.templ = templ;
    rec.route = route;
    rec.enterTime = stepIndex * interval_;  // Engine::getCurrentTime engine.cpp:678-680
    rec.firstLane = -1;
    pendingRecords_.push_back(rec);
    prioritySet(priority, -1);
    Pending p;
    p.index = (int) pendingRecords_.size() - 1;
    p.firstRoad = anchors[0];
    pending_.push_back(p);
    return p.index;
}

void Spawner::pushManual(int templ, const std::vector<int> &anchors, size_t stepIndex) {
    std::vector<int> seq;
    int route = expandRoute(anchors, seq) ? internRoute(seq) : -1;
    newVehicle(-1, manualCnt_++, templ, anchors, route, stepIndex, isFinished_);
}

void Spawner::prioritySet(int32_t key, int32_t value) {
    if (journal_.active) {
        PriorityUndo u{key, -1, false};
        u.had = livePriority_.lookup(key, u.old);
        journal_.priorities.push_back(u);
    }
    livePriority_.set(key, value);
}

void Spawner::priorityErase(int32_t key) {
    if (journal_.active) {
        PriorityUndo u{key, -1, false};
        u.had = livePriority_.lookup(key, u.old);
        journal_.priorities.push_back(u);
    }
    livePriority_.erase(key);
}

void Spawner::beginAhead() {
    if (!pending_.empty() || !pendingRecords_.empty()) throw std::runtime_error("spawner: a step ahead with vehicles still pending");
    journal_.active = true;
    journal_.rnd = rnd;
    journal_.activeFlows = activeFlows_;
    journal_.flowDyn.resize(activeFlows_.size());
    for (size_t i = 0; i < activeFlows_.size(); ++i) {
        const HostFlow &f = flows[(size_t) activeFlows_[i]];
        journal_.flowDyn[i] = FlowDyn{f.nowTime, f.currentTime, f.cnt, f.valid};
    }
    journal_.nVehicles = vehicles.size();
    journal_.priorities.clear();
    journal_.vidTables.clear();
    journal_.lastWait.clear();
}

void Spawner::commitAhead() { journal_.active = false; }

void Spawner::rollbackAhead() {
    if (!journal_.active) return;
    journal_.active = false;
    rnd = journal_.rnd;
    activeFlows_ = journal_.activeFlows;
    for (size_t i = 0; i < activeFlows_.size(); ++i) {
        HostFlow &f = flows[(size_t) activeFlows_[i]];
        const FlowDyn &d = journal_.flowDyn[i];
        f.nowTime = d.nowTime;
        f.currentTime = d.currentTime;
        f.cnt = d.cnt;
        f.valid = d.valid;
    }
    for (size_t i = journal_.priorities.size(); i-- > 0;) {
        const PriorityUndo &u = journal_.priorities[i];
        if (u.had) livePriority_.set(u.key, u.old);
        else livePriority_.erase(u.key);
    }
    for (size_t i = journal_.vidTables.size(); i-- > 0;) {
        std::vector<int32_t> &tbl = journal_.vidTables[i].first >= 0 ? flowVids[(size_t) journal_.vidTables[i].first] : manualVids;
        tbl.resize(journal_.vidTables[i].second);
    }
    for (size_t i = journal_.lastWait.size(); i-- > 0;) lastWaitVid_[(size_t) journal_.lastWait[i].first] = journal_.lastWait[i].second;
    vehicles.resize(journal_.nVehicles);
    pending_.clear();
    pendingRecords_.clear();
}

void Spawner::step(size_t stepIndex, std::vector<cfx_spawn> &out) {
    out.clear();
    // The priority table is far larger than the caches at city scale and every spawn probes it at a random place.
    // Which flows spawn this step is pure timer arithmetic, and (collisions aside) each vehicle consumes exactly two
    // draws, so a COPY of the generator tells where the probes will land: prefetch them all, then run the real loop.
    {
        size_t n = 0;
        for (int32_t fi : activeFlows_) {
            const HostFlow &f = flows[(size_t) fi];
            if (!f.valid || (f.endTime != -1 && f.currentTime > f.endTime) || !(f.currentTime >= f.startTime)) continue;
            for (double t = f.nowTime; t >= f.interval; t -= f.interval) ++n;
        }
        if (n >= 8) {
            std::mt19937 peek = rnd;
            for (size_t i = 0; i < n; ++i) {
                livePriority_.prefetch((int32_t) peek());
                (void) peek();
            }
        }
    }
    // phase 0: Flow::nextStep for every flow in order (engine.cpp:567-568)
    size_t keep = 0;
    for (size_t ai = 0; ai < activeFlows_.size(); ++ai) {
        const size_t fi = (size_t) activeFlows_[ai];
        HostFlow &f = flows[fi];
        // both early returns of Flow::nextStep are permanent (currentTime only grows; valid is never restored)
        if (!f.valid) continue;
        if (f.endTime != -1 && f.currentTime > f.endTime) continue;
        activeFlows_[keep++] = (int32_t) fi;
        if (f.currentTime >= f.startTime) {
            while (f.nowTime >= f.interval) {
                newVehicle((int) fi, f.cnt++, f.templ, f.anchors, f.route, stepIndex, isFinished_);
                f.nowTime -= f.interval;
            }
            f.nowTime += interval_;
        }
        f.currentTime += interval_;
    }
    activeFlows_.resize(keep);
    // phase 1: Engine::planRoute — roads in JSON order, vehicles in buffer order
    std::stable_sort(pending_.begin(), pending_.end(),
                     [](const Pending &a, const Pending &b) { return a.firstRoad < b.firstRoad; });
    // Vehicle numbers follow the order the reference CREATES the vehicles in (Flow::nextStep, flow by flow), not the
    // order planRoute hands them to their lanes: lane change walks its candidates in creation order (include/
    // cityflow_amd.h "Lane change").  The records still go out in planRoute order (= waiting-buffer push order).
    const int firstVid = (int) vehicles.size();
    vidOfPending_.assign(pendingRecords_.size(), -1);
    {
        int next = firstVid;
        for (size_t i = 0; i < pendingRecords_.size(); ++i)
            if (pendingRecords_[i].route >= 0) vidOfPending_[i] = next++;
        vehicles.resize((size_t) next);
    }
    for (const Pending &p : pending_) {
        VehicleRecord &rec = pendingRecords_[p.index];
        if (rec.route >= 0) {
            const std::vector<int32_t> &cands = routes.firstLanes[rec.route];
            int lane = cands[rnd() % cands.size()];
            int vid = vidOfPending_[p.index];
            rec.firstLane = lane;
            vehicles[(size_t) vid] = rec;
            prioritySet(rec.priority, vid);
            {
                std::vector<int32_t> &tbl = rec.flow >= 0 ? flowVids[rec.flow] : manualVids;
                const int at = rec.number - (rec.flow >= 0 ? flowVidBase[rec.flow] : manualVidBase);  // (numbers only grow: never below the base)
                if (journal_.active) journal_.vidTables.emplace_back(rec.flow >= 0 ? rec.flow : -1, tbl.size());
                if ((int) tbl.size() <= at) tbl.resize(at + 1, -1);
                tbl[at] = vid;
            }
            cfx_spawn s{};
            s.vid = vid;
            s.priority = rec.priority;
            s.templ = rec.templ;
            s.route = rec.route;
            s.lane = lane;
            s.prev_wait = lastWaitVid_[lane];
            s.enter_time = rec.enterTime;
            if (journal_.active) journal_.lastWait.emplace_back(lane, lastWaitVid_[lane]);
            lastWaitVid_[lane] = vid;
            out.push_back(s);
        } else {
            if (rec.flow >= 0 && flows[rec.flow].valid) {
                std::cerr << "[warning] Invalid route '" << flows[rec.flow].id << "'. Omitted by default." << std::endl;
                flows[rec.flow].valid = false;
            }
            priorityErase(rec.priority);
        }
    }
    pending_.clear();
    pendingRecords_.clear();
}

// Lexicographic order of the decimal strings, folded into integers.  Inside the flow index a shorter number is followed by
// '_' (which sorts after every digit): position i holds digit (0..9), 10 for "the string ended here", then 0s.  The
// trailing counter is followed by the end of the string (which sorts before every character): digit + 1, then 0s.
uint64_t Spawner::idSortKey(int vid) const {
    auto fold = [](uint64_t x, int width, bool endIsGreater) {
        char buf[24];
        int len = snprintf(buf, sizeof buf, "%llu", (unsigned long long) x);
        uint64_t k = 0;
        for (int i = 0; i < width; ++i) {
            int c;
            if (i < len) c = (buf[i] - '0') + (endIsGreater ? 0 : 1);
            else c = (endIsGreater && i == len) ? 10 : 0;
            k = k * 11 + (uint64_t) c;
        }
        return k;
    };
    const VehicleRecord &r = vehicles[vid];
    const uint64_t pow11_10 = 25937424601ULL;  // 11^10
    if (r.flow < 0) return (1ULL << 63) | fold((uint64_t) r.number, 10, false);  // 'm' > 'f'
    return fold((uint64_t) r.flow, 8, true) * pow11_10 + fold((uint64_t) r.number, 10, false);
}

void Spawner::pendingPushed(std::vector<std::pair<int32_t, std::string>> &out) const {
    for (const VehicleRecord &r : pendingRecords_)
        if (r.flow < 0) out.emplace_back(r.priority, "manually_pushed_" + std::to_string(r.number));
}

int Spawner::pendingPushedVid(const std::string &id) const {
    int next = (int) vehicles.size();
    for (const VehicleRecord &r : pendingRecords_) {
        if (r.flow < 0 && id == "manually_pushed_" + std::to_string(r.number)) return r.route >= 0 ? next : -2;
        if (r.route >= 0) ++next;
    }
    return -1;
}

int Spawner::vidOfId(const std::string &id) const {
    auto number = [](const std::string &t, int &out) {
        if (t.empty() || t.size() > 9) return false;
        for (char c : t)
            if (c < '0' || c > '9') return false;
        out = atoi(t.c_str());
        return std::to_string(out) == t;
    };
    const std::string mp = "manually_pushed_";
    int n = 0;
    if (id.compare(0, mp.size(), mp) == 0) {
        if (!number(id.substr(mp.size()), n) || n < manualVidBase || n - manualVidBase >= (int) manualVids.size()) return -1;
        return manualVids[n - manualVidBase];
    }
    if (id.compare(0, 5, "flow_") != 0) return -1;
    size_t us = id.find('_', 5);
    if (us == std::string::npos) return -1;
    int f = 0;
    if (!number(id.substr(5, us - 5), f) || !number(id.substr(us + 1), n)) return -1;
    if (f >= (int) flowVids.size() || n < flowVidBase[f] || n - flowVidBase[f] >= (int) flowVids[f].size()) return -1;
    return flowVids[f][n - flowVidBase[f]];
}
int Spawner::internRoute(
const std::vector<int> &seq) {
    auto it = routeIndex_.find(seq);
    if (it != routeIndex_.end()) return it->second;
    int r = routes.add(*net_, seq);
    routeIndex_.emplace(seq, r);
    return r;
}

Spawner::State Spawner::saveState() const {
    State st;
    for (const HostFlow &f : flows) st.flows.push_back(FlowDyn{f.nowTime, f.currentTime, f.cnt, f.valid});
    st.vehicles = vehicles;
    st.flowVids = flowVids;
    st.manualVids = manualVids;
    st.flowVidBase = flowVidBase;
    st.manualVidBase = manualVidBase;
    st.lastWaitVid = lastWaitVid_;
    st.rnd = rnd;
    st.manualCnt = manualCnt_;
    st.livePriority = livePriority_;
    st.shadowChains = shadowChains_;
    return st;
}

Spawner::State Spawner::compactedState(const std::vector<int32_t> &newOfOld, int nLive) const {
    State st;
    for (const HostFlow &f : flows) st.flows.push_back(FlowDyn{f.nowTime, f.currentTime, f.cnt, f.valid});
    auto renumber = [&newOfOld](int32_t v) { return v >= 0 && (size_t) v < newOfOld.size() ? newOfOld[(size_t) v] : -1; };
    for (size_t v = 0; v < vehicles.size(); ++v)
        if (renumber((int32_t) v) >= 0) {
            VehicleRecord r = vehicles[v];
            r.root = r.root >= 0 ? renumber(r.root) : -1;  // (a shadow's root is the id it carries: alive as long as a shadow of it is — kept below)
            st.vehicles.push_back(r);
            st.livePriority.set(r.priority, renumber((int32_t) v));
        }
    if ((int) st.vehicles.size() != nLive) throw std::logic_error("Spawner::compactedState: inconsistent renumbering");
    // number -> vid tables: the finished front goes, the base moves up
    auto table = [&renumber](const std::vector<int32_t> &tbl, int32_t base, std::vector<int32_t> &out, int32_t &outBase) {
        size_t first = 0;
        while (first < tbl.size() && renumber(tbl[first]) < 0) ++first;
        outBase = base + (int32_t) first;
        out.clear();
        for (size_t i = first; i < tbl.size(); ++i) out.push_back(renumber(tbl[i]));
    };
    st.flowVids.resize(flowVids.size());
    st.flowVidBase.assign(flowVids.size(), 0);
    for (size_t f = 0; f < flowVids.size(); ++f) table(flowVids[f], flowVidBase[f], st.flowVids[f], st.flowVidBase[f]);
    table(manualVids, manualVidBase, st.manualVids, st.manualVidBase);
    // the last vehicle pushed to a lane's waiting buffer: one that has finished stands for "nobody" (a successor becomes the
    // head of the queue either way: kr_admit / k_spawn_link look at the predecessor's state)
    st.lastWaitVid = lastWaitVid_;
    for (int32_t &v : st.lastWaitVid) v = renumber(v);
    st.rnd = rnd;
    st.manualCnt = manualCnt_;
    for (const auto &kv : shadowChains_) {
        const int32_t root = renumber(kv.first);
        std::vector<int32_t> chain;
        for (int32_t v : kv.second)
            if (renumber(v) >= 0) chain.push_back(renumber(v));
        if (root >= 0 && !chain.empty()) st.shadowChains.emplace(root, std::move(chain));
    }
    return st;
}

void Spawner::loadState(const State &st) {
    for (size_t i = 0; i < flows.size() && i < st.flows.size(); ++i) {
        flows[i].nowTime = st.flows[i].nowTime;
        flows[i].currentTime = st.flows[i].currentTime;
        flows[i].cnt = st.flows[i].cnt;
        flows[i].valid = st.flows[i].valid;
    }
    vehicles = st.vehicles;
    flowVids = st.flowVids;
    manualVids = st.manualVids;
    flowVidBase = st.flowVidBase;
    flowVidBase.resize(flows.size(), 0);
    manualVidBase = st.manualVidBase;
    lastWaitVid_ = st.lastWaitVid;
    rnd = st.rnd;
    manualCnt_ = std::max(manualCnt_, st.manualCnt);  // manuallyPushCnt never goes back (engine.h:56)
    livePriority_ = st.livePriority;
    shadowChains_ = st.shadowChains;
    rebuildActiveFlows();
    pending_.clear();
    pendingRecords_.clear();
    journal_.active = false;  // (whatever a step ahead journalled is void)
}

void Spawner::reset(bool reseed) {
    for (HostFlow &f : flows) {  // Flow::reset flow.cpp:28-32 (valid is NOT restored)
        f.nowTime = f.interval;
        f.currentTime = 0;
        f.cnt = 0;
    }
    vehicles.clear();
    for (auto &v : flowVids) v.clear();
    std::fill(flowVidBase.begin(), flowVidBase.end(), 0);
    manualVids.clear();
    manualVidBase = 0;
    livePriority_.clear();
    shadowChains_.clear();
    peekPriorities_.clear();
    peekDraws_.clear();
    rebuildActiveFlows();
    pending_.clear();
    pendingRecords_.clear();
    journal_.active = false;  // (whatever a step ahead journalled is void)
    std::fill(lastWaitVid_.begin(), lastWaitVid_.end(), -1);
    if (reseed) rnd.seed((std::mt19937::result_type) seed_);
}

}  // namespace cfa
