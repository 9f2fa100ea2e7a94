#include "archive.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <stdexcept>

#include "engine_host.h"
#include "json.h"

namespace cfa {

namespace {

// A decimal literal for `v` that BOTH kinds of reader turn back into exactly `v`: a correctly rounding one (strtod, Python's
// json) and the reference's own (rapidjson's default number reader, json_number.h — which is an ulp or two off on many 16- and
// 17-digit literals, so that "%.17g" alone would not survive the reference's, or this engine's, load_from_file).  The shortest
// of %.15g / %.16g / %.17g that does it; failing those, one of the neighbouring 17-digit decimals inside v's rounding interval,
// written d.ddde±x; then integer significands with an exponent; then literals only the reference's reader returns exactly.
// Every candidate is checked AS IT IS EMITTED (an integer-valued literal gets its ".0" / "e0" before the check, not after:
// "65695431087769544.0" does not read back as 65695431087769544 through the reference's fraction path).  What is left — about
// one double in 10^5 at simulation magnitudes, mantissa close to 2: the reference's reader cannot produce it from ANY literal,
// its own dumps come back an ulp off there too — is written %.17g (exact for correctly rounding readers) and COUNTED:
// g_inexactNumbers, reported by Archive::dump on stderr and in Archive::lastDumpInexact.
thread_local long g_inexactNumbers = 0;
bool readsBackBothWays(const char *lit, double v) {
    const JsonNumber n = parseJsonNumber(lit, lit + strlen(lit));
    return n.ok && n.d == v && strtod(lit, nullptr) == v;
}
// `digits` from a %g conversion: the literal that goes into the file — with ".0" where it would otherwise be an integer event
// for the reader (the reference's Archive loader asks for doubles; an integer converts, but 2^53 and beyond would not survive
// a reader that keeps 64-bit integers exact and then casts) — or with "e0" where the ".0" form does not read back
bool emitChecked(std::string &o, const char *digits, double v) {
    char lit[80];
    if (strpbrk(digits, ".eEn")) {
        if (!readsBackBothWays(digits, v)) return false;
        o += digits;
        return true;
    }
    for (const char *suffix : {".0", "e0"}) {
        snprintf(lit, sizeof lit, "%s%s", digits, suffix);
        if (readsBackBothWays(lit, v)) {
            o += lit;
            return true;
        }
    }
    return false;
}
void num(std::string &o, double v) {  // NaN/Inf spelled the way python's json accepts
    char buf[64];
    if (v != v) {
        o += "NaN";
        return;
    }
    if (std::isinf(v)) {
        o += v > 0 ? "Infinity" : "-Infinity";
        return;
    }
    for (int prec = 15; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (emitChecked(o, buf, v)) return;
    }
    // v = m x 10^e with a 17-digit m: try m +- 1, 2, ...
    snprintf(buf, sizeof buf, "%.16e", std::fabs(v));  // d.dddddddddddddddde+xx
    long long m = 0;
    for (const char *q = buf; *q && *q != 'e'; ++q)
        if (*q != '.') m = m * 10 + (*q - '0');
    const int e10 = atoi(strchr(buf, 'e') + 1);
    for (int k = 1; k <= 40; ++k)
        for (int sign = -1; sign <= 1; sign += 2) {
            const long long mk = m + sign * k;
            if (mk < 10000000000000000LL || mk > 99999999999999999LL) continue;
            char digits[24];
            snprintf(digits, sizeof digits, "%lld", mk);
            snprintf(buf, sizeof buf, "%s%c.%se%d", v < 0 ? "-" : "", digits[0], digits + 1, e10);
            if (readsBackBothWays(buf, v)) {
                o += buf;
                return;
            }
        }
    // an integer A with 2^53 <= A < 2^64 that IS a double, and a power of ten: the reference's reader takes all of A's digits
    // into its 64-bit accumulator, converts without error and divides (or multiplies) once by an exact power of ten
    {
        const double a = std::fabs(v);
        for (int k = -22; k <= 22; ++k) {
            const double scaled = k >= 0 ? a * jsonnum::pow10Table(k) : a / jsonnum::pow10Table(-k);
            if (!(scaled >= 9007199254740992.0 && scaled < 18446744073709551615.0 * 0.999)) continue;
            for (int step = 0; step < 5; ++step) {
                double A = scaled;
                for (int i = 0; i < (step + 1) / 2; ++i) A = std::nextafter(A, step % 2 ? 1e300 : 0.0);
                if (!(A >= 9007199254740992.0 && A < 18446744073709549568.0)) continue;
                snprintf(buf, sizeof buf, "%s%llue%d", v < 0 ? "-" : "", (unsigned long long) A, -k);
                if (readsBackBothWays(buf, v)) {
                    o += buf;
                    return;
                }
            }
        }
    }
    // More digits than the reader's 64-bit accumulator holds: 19 significant digits and z zeros, times 10^-(k + z).  The
    // reader continues such an integer in a double (x 10 per further digit) and divides by the double nearest to a power of
    // ten beyond 10^22 — every z is one more, differently rounded, attempt at landing on v.
    {
        const long double a = std::fabs((long double) v);
        int k = 0;
        long double x = a;
        while (x < 1e18L && k < 60) x *= 10.0L, ++k;
        while (x >= 1e19L && k > -60) x /= 10.0L, --k;
        const unsigned long long m19 = (unsigned long long) (x + 0.5L);
        static const int deltas[] = {0, 1, -1, 2, -2, 4, -4, 7, -7, 11, -11, 16, -16, 24, -24, 36, -36};
        char zeros[24];
        for (int z = 1; z <= 16; ++z) {
            memset(zeros, '0', (size_t) z);
            zeros[z] = 0;
            for (int d : deltas) {
                snprintf(buf, sizeof buf, "%s%llu%se%d", v < 0 ? "-" : "", m19 + (long long) d, zeros, -(k + z));
                if (readsBackBothWays(buf, v)) {
                    o += buf;
                    return;
                }
            }
        }
    }
    // No literal that both kinds of reader return exactly (about 1 double in 7000 at simulation magnitudes: mantissa close to 2,
    // where v's rounding interval is narrower than the spacing of every significand the reference's reader can form for it).  The
    // file is for engines to load, so the reference's reader wins: an integer S with k > 22 digits behind it, S / 10^k inside v's
    // interval as that reader divides (by the double nearest to 10^k); a correctly rounding reader then sees a neighbour of v.
    {
        const long double a = std::fabs((long double) v);
        char big[400];
        for (int k = 23; k <= 290; ++k) {
            const double s0 = (double) (a * powl(10.0L, k));
            if (!(s0 < 1e300)) break;
            // (the reader builds S digit by digit in a double and may arrive a few units below the literal: the doubles around
            // the target are all tried, the reader itself decides)
            for (int step = 0; step < 15; ++step) {
                double S = s0;
                for (int i = 0; i < (step + 1) / 2; ++i) S = std::nextafter(S, step % 2 ? 1e308 : 0.0);
                snprintf(big, sizeof big, "%s%.0fe%d", v < 0 ? "-" : "", S, -k);
                const JsonNumber n = parseJsonNumber(big, big + strlen(big));
                if (n.ok && n.d == v) {
                    o += big;
                    return;
                }
            }
        }
    }
    // no literal at all for the reference's reader (see above): exact for a correctly rounding one, and counted
    ++g_inexactNumbers;
    snprintf(buf, sizeof buf, "%.17g", v);
    o += buf;
    if (!strpbrk(buf, ".eEn")) o += "e0";
}
void str(std::string &o, const std::string &s) {
    o += '"';
    for (char c : s) {
        if (c == '"' || c == '\\') o += '\\';
        o += c;
    }
    o += '"';
}
void key(std::string &o, const char *k) {
    o += '"';
    o += k;
    o += "\":";
}

}  // namespace

std::string formatJsonNumber(double v) {
    std::string o;
    num(o, v);
    return o;
}
long inexactJsonNumbers() { return g_inexactNumbers; }

std::string Archive::vehicleId(int vid) const {
    const VehicleRecord &r = host.vehicles[vid];
    std::string id = r.flow >= 0 ? flowIds[r.flow] + "_" + std::to_string(r.number) : "manually_pushed_" + std::to_string(r.number);
    // lane change: a shadow is "<id>_shadow" until its change completes (engine.cpp:814, lanechange.cpp:119-121)
    if (!dev.rLcFlags.empty())
        for (size_t i = 0; i < dev.rVid.size(); ++i)
            if (dev.rVid[i] == vid) return (dev.rLcFlags[i] & CFX_LC_SHADOW) ? id + "_shadow" : id;
    return id;
}

// Archive::dump archive.cpp:153-343 — same keys, same nesting; vehicles in vehiclePool (priority) order.
void Archive::dump(const std::string &path) const {
    const long inexactBefore = g_inexactNumbers;
    const bool lc = !dev.rLcFlags.empty();
    const int L = (int) net->lanes.size();
    const int nV = (int) host.vehicles.size();
    std::vector<int> runIndex(nV, -1);
    for (size_t i = 0; i < dev.rVid.size(); ++i) runIndex[dev.rVid[i]] = (int) i;
    std::vector<int> waitLane(nV, -1);
    for (size_t i = 0; i < dev.wVid.size(); ++i) waitLane[dev.wVid[i]] = dev.wLane[i];

    std::vector<std::string> idOf((size_t) nV);  // with lane change ids depend on the state: computed once
    for (int v = 0; v < nV; ++v) {
        const VehicleRecord &r = host.vehicles[v];
        idOf[v] = r.flow >= 0 ? flowIds[r.flow] + "_" + std::to_string(r.number) : "manually_pushed_" + std::to_string(r.number);
        if (lc && runIndex[v] >= 0 && (dev.rLcFlags[runIndex[v]] & CFX_LC_SHADOW)) idOf[v] += "_shadow";
    }
    auto vehicleId = [&idOf](int v) -> const std::string & { return idOf[(size_t) v]; };

    std::string o;
    o.reserve(1 << 20);
    o += '{';
    key(o, "step");
    o += std::to_string(dev.step);
    o += ',';
    key(o, "activeVehicleCount");
    o += std::to_string(dev.rVid.size());
    o += ',';
    key(o, "rnd");
    {
        std::ostringstream os;
        os << host.rnd;
        str(o, os.str());
    }
    o += ',';
    key(o, "vehicles");
    o += '[';
    std::vector<std::pair<int32_t, int>> byPriority;
    for (int v = 0; v < nV; ++v)
        if (dev.vState[v] != 2) byPriority.emplace_back(host.vehicles[v].priority, v);
    std::sort(byPriority.begin(), byPriority.end());
    bool first = true;
    for (auto &pv : byPriority) {
        const int v = pv.second;
        const VehicleRecord &r = host.vehicles[v];
        const cfx_vehicle_template &t = templates[r.templ];
        const int ri = runIndex[v];
        if (!first) o += ',';
        first = false;
        o += '{';
        key(o, "priority"); o += std::to_string(r.priority); o += ',';
        key(o, "id"); str(o, vehicleId(v)); o += ',';
        key(o, "enterTime"); num(o, r.enterTime); o += ',';
        key(o, "speed"); num(o, ri >= 0 ? dev.rSpeed[ri] : t.initial_speed); o += ',';
        key(o, "len"); num(o, t.len); o += ',';
        key(o, "width"); num(o, t.width); o += ',';
        key(o, "maxPosAcc"); num(o, t.max_pos_acc); o += ',';
        key(o, "maxNegAcc"); num(o, t.max_neg_acc); o += ',';
        key(o, "usualPosAcc"); num(o, t.usual_pos_acc); o += ',';
        key(o, "usualNegAcc"); num(o, t.usual_neg_acc); o += ',';
        key(o, "minGap"); num(o, t.min_gap); o += ',';
        key(o, "maxSpeed"); num(o, t.max_speed); o += ',';
        key(o, "headwayTime"); num(o, t.headway_time); o += ',';
        key(o, "yieldDistance"); num(o, t.yield_distance); o += ',';
        key(o, "turnSpeed"); num(o, t.turn_speed); o += ',';
        key(o, "route");
        o += '[';
        for (int p = routeStart[r.route]; p < routeStart[r.route + 1]; ++p) {
            if (p > routeStart[r.route]) o += ',';
            str(o, net->roads[routeRoads[p]].id);
        }
        o += "],";
        key(o, "dis"); num(o, ri >= 0 ? dev.rDis[ri] : 0.0); o += ',';
        int drivable = ri >= 0 ? dev.rDrivable[ri] : (waitLane[v] >= 0 ? waitLane[v] : r.firstLane);
        key(o, "drivable"); str(o, net->drivableId(drivable)); o += ',';
        if (ri >= 0 && dev.rPrevDrivable[ri] >= 0) {
            key(o, "prevDrivable"); str(o, net->drivableId(dev.rPrevDrivable[ri])); o += ',';
        }
        key(o, "approachingIntersectionDistance"); num(o, t.approach_dist); o += ',';
        key(o, "gap"); num(o, (ri >= 0 && (lc || dev.rLeader[ri] >= 0)) ? dev.rGap[ri] : 0.0); o += ',';
        key(o, "enterLaneLinkTime"); o += std::to_string((unsigned) (ri >= 0 ? dev.rEnterLLTime[ri] : INT_MAX)); o += ',';
        if (ri >= 0 && dev.rLeader[ri] >= 0) {
            key(o, "leader"); str(o, vehicleId(dev.rLeader[ri])); o += ',';
        }
        if (ri >= 0 && dev.rBlocker[ri] >= 0) {
            key(o, "blocker"); str(o, vehicleId(dev.rBlocker[ri])); o += ',';
        }
        key(o, "end"); o += "false,";
        key(o, "running"); o += ri >= 0 ? "true," : "false,";
        if (lc && ri >= 0) {  // archive.cpp:229-246
            const uint8_t f = dev.rLcFlags[ri];
            key(o, "partnerType"); o += (f & CFX_LC_SHADOW) ? "2," : ((f & CFX_LC_PARENT) ? "1," : "0,");
            if (dev.rLcPartner[ri] >= 0) {
                key(o, "partner"); str(o, vehicleId(dev.rLcPartner[ri])); o += ',';
            }
            key(o, "offset"); num(o, dev.rLcOffset[ri]); o += ',';
            if (f & CFX_LC_CHANGING) {  // the only signal that outlives a step
                key(o, "laneChangeUrgency"); o += "1,";
                key(o, "laneChangeDirection"); o += std::to_string(dev.rLcDirection[ri]); o += ',';
                if (dev.rLcTarget[ri] >= 0) {
                    key(o, "laneChangeTarget"); str(o, net->drivableId(dev.rLcTarget[ri])); o += ',';
                }
            }
            key(o, "laneChangeWaitingTime"); num(o, dev.rLcWaitingTime[ri]); o += ',';
            key(o, "laneChanging"); o += (f & CFX_LC_CHANGING) ? "true," : "false,";
            key(o, "laneChangeLastTime"); num(o, dev.rLcLastChangeTime[ri]);
        } else {
            key(o, "partnerType"); o += "0,";
            key(o, "offset"); o += "0.0,";
            key(o, "laneChangeWaitingTime"); o += "0.0,";
            key(o, "laneChanging"); o += "false,";
            key(o, "laneChangeLastTime"); o += "0.0";
        }
        o += '}';
    }
    o += "],";
    key(o, "drivables");
    o += '{';
    {
        const int D = L + (int) net->laneLinks.size();
        std::vector<std::vector<int>> perDrv(D), perLaneWait(L);
        for (size_t i = 0; i < dev.rVid.size(); ++i) perDrv[dev.rDrivable[i]].push_back(dev.rVid[i]);
        for (size_t i = 0; i < dev.wVid.size(); ++i) perLaneWait[dev.wLane[i]].push_back(dev.wVid[i]);
        for (int d = 0; d < D; ++d) {
            if (d) o += ',';
            str(o, net->drivableId(d));
            o += ":{";
            key(o, "vehicles");
            o += '[';
            for (size_t i = 0; i < perDrv[d].size(); ++i) {
                if (i) o += ',';
                str(o, vehicleId(perDrv[d][i]));
            }
            o += ']';
            if (d < L) {
                o += ',';
                key(o, "waitingBuffer");
                o += '[';
                for (size_t i = 0; i < perLaneWait[d].size(); ++i) {
                    if (i) o += ',';
                    str(o, vehicleId(perLaneWait[d][i]));
                }
                o += "],";
                // Lane::history (archive.cpp:286-294): {vehicle count, mean speed} pairs, flat.  It feeds only the never-selected
                // RouterType::DURATION and is kept only with "cfx": {"laneHistory": true}; empty otherwise
                key(o, "history");
                o += '[';
                const bool have = (int) dev.hLen.size() == L;
                for (int i = 0; have && i < dev.hLen[d]; ++i) {
                    if (i) o += ',';
                    o += std::to_string(dev.hVehicleNum[(size_t) d * CFX_LANE_HISTORY_MAX + i]);
                    o += ',';
                    num(o, dev.hAverageSpeed[(size_t) d * CFX_LANE_HISTORY_MAX + i]);
                }
                o += "],";
                key(o, "historyVehicleNum"); o += std::to_string(have ? dev.hHistoryVehicleNum[d] : 0); o += ',';
                key(o, "historyAverageSpeed"); num(o, have ? dev.hHistoryAverageSpeed[d] : 0.0);
            }
            o += '}';
        }
    }
    o += "},";
    key(o, "flows");
    o += '{';
    for (size_t f = 0; f < flowIds.size(); ++f) {
        if (f) o += ',';
        str(o, flowIds[f]);
        o += ":{";
        key(o, "nowTime"); num(o, host.flows[f].nowTime); o += ',';
        key(o, "currentTime"); num(o, host.flows[f].currentTime); o += ',';
        key(o, "cnt"); o += std::to_string(host.flows[f].cnt);
        o += '}';
    }
    o += "},";
    key(o, "trafficLights");
    o += '{';
    for (size_t i = 0; i < net->inters.size(); ++i) {
        if (i) o += ',';
        str(o, net->inters[i].id);
        o += ":{";
        key(o, "remainDuration"); num(o, dev.tlRemain[i]); o += ',';
        key(o, "curPhaseIndex"); o += std::to_string(dev.tlPhase[i]);
        o += '}';
    }
    o += "},";
    key(o, "finishedVehicleCnt"); o += std::to_string(dev.finished); o += ',';
    key(o, "cumulativeTravelTime"); num(o, dev.cumulativeTravelTime);
    o += '}';
    FILE *fp = fopen(path.c_str(), "w");
    if (!fp) throw std::runtime_error("Archive.dump: cannot open " + path);
    fwrite(o.data(), 1, o.size(), fp);
    fclose(fp);
    lastDumpInexact = g_inexactNumbers - inexactBefore;
    if (lastDumpInexact > 0)  // (never silent: such a value loads an ulp off in the reference — as the reference's own dump of it would)
        fprintf(stderr, "[warning] Archive.dump: %ld number(s) have no decimal literal that the reference's JSON reader returns "
                        "exactly; written with 17 digits (exact for correctly rounding readers): %s\n", lastDumpInexact, path.c_str());
}

// ------------------------------------------------------------------------------------------ EngineHost side
Archive EngineHost::snapshotImpl(bool hostState) {
    dropAhead();
    settleLaneChange();
    Archive a;
    if (hostState) a.host = spawner_.saveState();  // (compactVehicles builds its own)
    a.net = net_;
    a.templates = spawner_.templates;
    a.routeStart = spawner_.routes.routeStart;
    a.routeRoads = spawner_.routes.roads;
    for (const HostFlow &f : spawner_.flows) a.flowIds.push_back(f.id);
    DeviceState &d = a.dev;
    cfx_scalars sc = scalars();
    d.step = (int64_t) step_;
    d.finished = sc.finished_vehicle_count;
    d.vehicleSteps = sc.vehicle_steps;
    d.cumulativeTravelTime = sc.cumulative_travel_time;
    int nV = (int) spawner_.vehicles.size();
    d.vState.resize(nV);
    if (nV) check(be_.cfx_get_vehicle_status(dev_, 0, nV, d.vState.data()), "cfx_get_vehicle_status");
    if (laneHistory_) {
        const size_t nL = net_->lanes.size();
        d.hLen.resize(nL);
        d.hVehicleNum.assign(nL * CFX_LANE_HISTORY_MAX, 0);
        d.hAverageSpeed.assign(nL * CFX_LANE_HISTORY_MAX, 0.0);
        d.hHistoryVehicleNum.resize(nL);
        d.hHistoryAverageSpeed.resize(nL);
        cfx_lane_history h{(int32_t) nL, d.hLen.data(), d.hVehicleNum.data(), d.hAverageSpeed.data(), d.hHistoryVehicleNum.data(),
                           d.hHistoryAverageSpeed.data()};
        check(be_.cfx_get_lane_history(dev_, &h), "cfx_get_lane_history");
    }
    VehicleSnapshot s;
    snapshotVehicles(s);
    d.rVid = s.vid;
    d.rDrivable = s.drivable;
    d.rPrevDrivable = s.prevDrivable;
    d.rBlocker = s.blocker;
    d.rEnterLLTime = s.enterLLTime;
    // Router::iCurRoad of an archived vehicle: the Router copy constructor restarts it at route.begin() (router.cpp:11-14 — the
    // archive holds copies, Archive::copyVehiclePool) and Router::update brings it to the current road when the vehicle next
    // enters a lane; until then get_vehicle_info lists the whole route, as the reference's does after a load
    // (compactVehicles is not a load: the vehicles keep their place in their routes)
    if (hostState) d.rRoutePos.assign(s.routePos.size(), 0);
    else d.rRoutePos = s.routePos;
    d.rLeader = s.leader;
    d.rDis = s.dis;
    d.rSpeed = s.speed;
    d.rGap = s.gap;
    d.rLcPartner = s.lcPartner;  // all empty without lane change
    d.rLcFlags = s.lcFlags;
    d.rLcOffset = s.lcOffset;
    d.rLcLastDir = s.lcLastDir;
    d.rLcTarget = s.lcTarget;
    d.rLcDirection = s.lcDirection;
    d.rLcLastChangeTime = s.lcLastChangeTime;
    d.rLcWaitingTime = s.lcWaitingTime;
    d.rCustomSpeed.resize(s.count);
    if (s.count) check(be_.cfx_get_custom_speeds(dev_, s.count, d.rCustomSpeed.data()), "cfx_get_custom_speeds");
    waitingVehicles(d.wVid, d.wLane);
    trafficLightState(d.tlPhase, d.tlRemain);
    return a;
}

// Forget the finished vehicles (the reference frees a vehicle when it finishes, engine.cpp:296-310): the state as snapshot()
// reads it, the vehicles that are still waiting or running renumbered 0 .. n-1 in their old order — creation order, which is
// what breaks exact-distance ties — and loaded back through the very path Engine.load takes (cfx_load_state: the device's and
// the CPU twin's tables are rebuilt for the vehicles of the archive).  Between two steps; nothing a caller can see changes
// (ids, priorities, positions, counters, Lane::history), only the vehicle numbers behind the ABI — and what host and device
// remember per vehicle CREATED goes back to what they need per vehicle ALIVE.
void EngineHost::compactVehicles() {
    Archive a = snapshotImpl(/*hostState=*/false);
    const int nV = (int) spawner_.vehicles.size();
    std::vector<uint8_t> keep((size_t) nV, 0);
    for (int v = 0; v < nV; ++v)
        if (a.dev.vState[(size_t) v] != 2) {
            keep[(size_t) v] = 1;
            // lane change: an id travels along a chain of copies (the vehicle it was given to, its shadow, that one's shadow ...);
            // whoever carries it now is found through the whole chain, so the chain stays — as rows of finished vehicles
            const int root = spawner_.vehicles[(size_t) v].root >= 0 ? spawner_.vehicles[(size_t) v].root : v;
            if (laneChange_)
                for (int32_t w : spawner_.idChain(root))
                    if (w >= 0 && w < nV) keep[(size_t) w] = 1;
        }
    std::vector<int32_t> newOfOld((size_t) nV, -1);
    int nLive = 0;
    for (int v = 0; v < nV; ++v)
        if (keep[(size_t) v]) newOfOld[(size_t) v] = nLive++;
    auto renumber = [&](std::vector<int32_t> &vids) {
        for (int32_t &v : vids) v = v >= 0 && v < nV ? newOfOld[(size_t) v] : -1;
    };
    DeviceState &d = a.dev;
    renumber(d.rVid);
    renumber(d.rBlocker);
    renumber(d.rLeader);
    renumber(d.rLcPartner);
    renumber(d.wVid);
    for (int32_t v : d.rVid)
        if (v < 0) throw std::logic_error("compact_vehicles: a running vehicle counted as finished");
    for (int32_t v : d.wVid)
        if (v < 0) throw std::logic_error("compact_vehicles: a waiting vehicle counted as finished");
    std::vector<uint8_t> state((size_t) nLive);
    for (int v = 0; v < nV; ++v)
        if (newOfOld[(size_t) v] >= 0) state[(size_t) newOfOld[(size_t) v]] = d.vState[(size_t) v];
    d.vState.swap(state);
    a.host = spawner_.compactedState(newOfOld, nLive);
    // custom speeds of vehicles that are STILL waiting (cfx_state carries those of running vehicles only)
    std::map<int32_t, double> stillWaiting;
    for (const auto &kv : waitingCustom_)
        if (kv.first >= 0 && kv.first < nV && newOfOld[(size_t) kv.first] >= 0 && d.vState[(size_t) newOfOld[(size_t) kv.first]] == 0)
            stillWaiting[newOfOld[(size_t) kv.first]] = kv.second;
    load(a);
    for (const auto &kv : stillWaiting) check(be_.cfx_set_vehicle_speed(dev_, kv.first, kv.second), "cfx_set_vehicle_speed");
    waitingCustom_.swap(stillWaiting);
    vehicleCompactions_ += 1;
}

void EngineHost::load(const Archive &a) {
    dropAhead();
    settleLaneChange();
    if (laneChange_ != !a.dev.rLcFlags.empty() && !a.dev.rVid.empty())
        throw std::runtime_error("Engine.load: the archive was taken with a different laneChange setting");
    pendingPhaseInter_.clear();
    pendingPhaseValue_.clear();
    if (a.net.get() != net_.get() && a.net->lanes.size() != net_->lanes.size())
        throw std::runtime_error("Engine.load: archive belongs to a different road network");
    spawner_.loadState(a.host);
    uploadNewTablesIfAny();
    const DeviceState &d = a.dev;
    const int nV = (int) a.host.vehicles.size();
    std::vector<int32_t> prio(nV), templ(nV), route(nV);
    std::vector<double> enter(nV);
    for (int v = 0; v < nV; ++v) {
        prio[v] = a.host.vehicles[v].priority;
        templ[v] = a.host.vehicles[v].templ;
        route[v] = a.host.vehicles[v].route;
        enter[v] = a.host.vehicles[v].enterTime;
    }
    cfx_state st{};
    st.step = d.step;
    st.finished_vehicle_count = d.finished;
    st.vehicle_steps = d.vehicleSteps;
    st.cumulative_travel_time = d.cumulativeTravelTime;
    st.n_vehicles = nV;
    st.v_priority = prio.data();
    st.v_templ = templ.data();
    st.v_route = route.data();
    st.v_enter_time = enter.data();
    st.v_state = d.vState.data();
    st.n_running = (int) d.rVid.size();
    st.r_vid = d.rVid.data();
    st.r_drivable = d.rDrivable.data();
    st.r_prev_drivable = d.rPrevDrivable.data();
    st.r_blocker_vid = d.rBlocker.data();
    st.r_enter_ll_time = d.rEnterLLTime.data();
    st.r_route_pos = d.rRoutePos.data();
    st.r_dis = d.rDis.data();
    st.r_speed = d.rSpeed.data();
    st.r_custom_speed = d.rCustomSpeed.empty() ? nullptr : d.rCustomSpeed.data();
    if (d.rGap.size() == d.rVid.size() && !d.rGap.empty()) st.r_gap = d.rGap.data();  // stored state: the first step's gap
    if (laneChange_ && !d.rLcFlags.empty()) {
        st.r_lc_partner_vid = d.rLcPartner.data();
        st.r_lc_flags = d.rLcFlags.data();
        st.r_lc_offset = d.rLcOffset.data();
        st.r_lc_last_dir = d.rLcLastDir.data();
        st.r_lc_target_lane = d.rLcTarget.data();
        st.r_lc_direction = d.rLcDirection.data();
        st.r_lc_last_change_time = d.rLcLastChangeTime.data();
        st.r_lc_waiting_time = d.rLcWaitingTime.data();
    }
    st.n_waiting = (int) d.wVid.size();
    st.w_vid = d.wVid.data();
    st.w_lane = d.wLane.data();
    st.tl_phase = d.tlPhase.data();
    st.tl_remain = d.tlRemain.data();
    check(be_.cfx_load_state(dev_, &st), "cfx_load_state");
    forgetPhases();  // (the lights now show the archive's phases)
    if (laneHistory_ && d.hLen.size() == net_->lanes.size()) {  // Archive::resume archive.cpp:107-109 (an archive without it: as it is)
        cfx_lane_history h{(int32_t) d.hLen.size(), const_cast<int32_t *>(d.hLen.data()), const_cast<int32_t *>(d.hVehicleNum.data()),
                           const_cast<double *>(d.hAverageSpeed.data()), const_cast<int32_t *>(d.hHistoryVehicleNum.data()),
                           const_cast<double *>(d.hHistoryAverageSpeed.data())};
        check(be_.cfx_set_lane_history(dev_, &h), "cfx_set_lane_history");
    }
    step_ = (size_t) d.step;
    waitingCustom_.clear();  // (compactVehicles puts back what it carries over)
    vehicleEpoch_ += 1;  // vehicle numbers of the archive replace the current ones
}

// Archive(Engine&, filename) archive.cpp:345-550: rebuild an Archive from the reference's JSON format.
void EngineHost::loadFromFile(const std::string &path) {
    dropAhead();
    load(readArchiveFile(path, net_, spawner_, laneChange_));
}

Archive readArchiveFile(const std::string &path, const std::shared_ptr<HostRoadNet> &net_, Spawner &spawner_, bool laneChange_) {
    // The file is STREAMED (Json::parseFileStreamed): its `vehicles` array and `drivables` object — all but a few KB of an
    // Archive — are handed over child by child and never held as a DOM (1 M vehicles: 1.1 GB of text, 6 GB of nodes).
    Archive a;
    a.net = net_;
    const int L = (int) net_->lanes.size();
    std::map<std::string, int> drvIndex;
    for (int d = 0; d < L + (int) net_->laneLinks.size(); ++d) drvIndex[net_->drivableId(d)] = d;

    a.host = spawner_.saveState();  // flows' valid flags, manual counter; the rest is overwritten below
    a.host.vehicles.clear();
    for (auto &v : a.host.flowVids) v.clear();
    std::fill(a.host.manualVids.begin(), a.host.manualVids.end(), -1);
    std::fill(a.host.lastWaitVid.begin(), a.host.lastWaitVid.end(), -1);
    a.host.livePriority.clear();

    std::map<std::string, int> vidOf;
    struct Dyn {
        double dis, speed;
        int drivable, prev, ellt;
        std::string blocker;
        bool running;
        // lane change (archive.cpp:407-424,447-460)
        double gap = 0, offset = 0, lastTime = 0, waitingTime = 0;
        int partnerType = 0, direction = 0;
        bool changing = false, shadow = false;
        std::string partner, target;
    };
    a.host.shadowChains.clear();
    const std::string shadowSuffix = "_shadow";
    std::vector<std::pair<int, std::string>> shadowBase;  // (vid of a shadow, id it carries)
    std::vector<Dyn> dyn;
    bool vehiclesDone = false;
    auto oneVehicle = [&](const Json &jv) {
        cfx_vehicle_template t = spawner_.makeTemplate(jv.numberAt("len"), jv.numberAt("width"), jv.numberAt("maxPosAcc"),
                                                       jv.numberAt("maxNegAcc"), jv.numberAt("usualPosAcc"),
                                                       jv.numberAt("usualNegAcc"), jv.numberAt("minGap"),
                                                       jv.numberAt("maxSpeed"), jv.numberAt("headwayTime"));
        t.yield_distance = jv.numberAt("yieldDistance");
        t.turn_speed = jv.numberAt("turnSpeed");
        t.approach_dist = jv.numberAt("approachingIntersectionDistance");
        if (!jv.boolAt("running")) t.initial_speed = jv.numberAt("speed");  // a waiting vehicle enters with VehicleInfo::speed
        VehicleRecord r{};
        r.templ = spawner_.addTemplate(t);
        std::vector<int> seq;
        for (const Json &rd : jv.arrayAt("route").items) {
            auto it = net_->roadIndex.find(rd.s);
            if (it == net_->roadIndex.end()) throw std::runtime_error("load_from_file: unknown road " + rd.s);
            seq.push_back(it->second);
        }
        if (seq.empty()) throw std::runtime_error("load_from_file: vehicle without route");
        r.route = spawner_.internRoute(seq);
        r.priority = jv.intAt("priority");
        r.enterTime = jv.numberAt("enterTime");
        const std::string &fullId = jv.stringAt("id");
        const bool isShadow = laneChange_ && fullId.size() > shadowSuffix.size() &&
                              fullId.compare(fullId.size() - shadowSuffix.size(), shadowSuffix.size(), shadowSuffix) == 0;
        const std::string id = isShadow ? fullId.substr(0, fullId.size() - shadowSuffix.size()) : fullId;
        const std::string mp = "manually_pushed_";
        if (id.compare(0, mp.size(), mp) == 0) {
            r.flow = -1;
            r.number = atoi(id.c_str() + mp.size());
        } else {
            size_t us = id.rfind('_');
            if (id.compare(0, 5, "flow_") != 0 || us == std::string::npos || us <= 5)
                throw std::runtime_error("load_from_file: unsupported vehicle id " + id);
            r.flow = atoi(id.substr(5, us - 5).c_str());
            r.number = atoi(id.c_str() + us + 1);
            if (r.flow < 0 || r.flow >= (int) spawner_.flows.size())
                throw std::runtime_error("load_from_file: vehicle of unknown flow " + id);
        }
        Dyn y{};
        y.dis = jv.numberAt("dis");
        y.speed = jv.numberAt("speed");
        auto di = drvIndex.find(jv.stringAt("drivable"));
        if (di == drvIndex.end()) throw std::runtime_error("load_from_file: unknown drivable");
        y.drivable = di->second;
        const Json *pd = jv.find("prevDrivable");
        y.prev = pd ? drvIndex.at(pd->s) : -1;
        y.ellt = (int) jv.at("enterLaneLinkTime").i;
        const Json *bl = jv.find("blocker");
        if (bl) y.blocker = bl->s;
        y.running = jv.boolAt("running");
        // ControllerInfo::gap is read like every other field (archive.cpp:402); an archive written without it: recomputed
        y.gap = jv.find("gap") && jv.find("gap")->isNumber() ? jv.find("gap")->asDouble() : std::nan("");
        if (laneChange_) {
            y.shadow = isShadow;
            y.gap = jv.numberAt("gap");
            y.partnerType = jv.intAt("partnerType");
            y.offset = jv.numberAt("offset");
            if (const Json *pp = jv.find("partner")) y.partner = pp->s;
            y.changing = jv.boolAt("laneChanging");
            y.lastTime = jv.numberAt("laneChangeLastTime");
            y.waitingTime = jv.numberAt("laneChangeWaitingTime");
            if (jv.find("laneChangeUrgency")) {
                y.direction = jv.intAt("laneChangeDirection");
                if (const Json *tg = jv.find("laneChangeTarget")) y.target = tg->s;
            }
        }
        r.firstLane = y.running ? -1 : y.drivable;
        int vid = (int) a.host.vehicles.size();
        a.host.vehicles.push_back(r);
        a.host.livePriority.set(r.priority, vid);
        if (isShadow) {
            shadowBase.emplace_back(vid, id);  // its chain is attached once every vehicle is known
        } else {
            std::vector<int32_t> &tbl = r.flow >= 0 ? a.host.flowVids[r.flow] : a.host.manualVids;
            if ((int) tbl.size() <= r.number) tbl.resize(r.number + 1, -1);
            tbl[r.number] = vid;
        }
        vidOf[fullId] = vid;
        dyn.push_back(y);
    };
    // a drivable's lists (vehicles front to back, waiting buffer, history), in the order of the network's drivables
    const int D = L + (int) net_->laneLinks.size();
    int nextDv = 0;
    std::map<int, Json> heldDrivables;  // those that came before the vehicles, or out of the network's order
    DeviceState &d = a.dev;
    auto oneDrivable = [&](int dv, const Json &jd) {
        for (const Json &jid : jd.arrayAt("vehicles").items) {
            int vid = vidOf.at(jid.s);
            const Dyn &y = dyn[vid];
            d.vState[vid] = 1;
            d.rVid.push_back(vid);
            d.rDrivable.push_back(dv);
            d.rPrevDrivable.push_back(y.prev);
            d.rBlocker.push_back(y.blocker.empty() ? -1 : vidOf.at(y.blocker));
            d.rEnterLLTime.push_back(y.ellt);
            // Router::iCurRoad: the copy constructor restarts it at route.begin() (router.cpp:11-14); Router::update
            // advances it to the current road when the vehicle next enters a lane
            d.rRoutePos.push_back(0);
            d.rLeader.push_back(-1);
            d.rDis.push_back(y.dis);
            d.rSpeed.push_back(y.speed);
            d.rGap.push_back(y.gap);
            if (laneChange_) {
                d.rLcFlags.push_back((uint8_t) ((y.partnerType == 2 ? CFX_LC_SHADOW : 0) | (y.partnerType == 1 ? CFX_LC_PARENT : 0) |
                                                (y.changing ? CFX_LC_CHANGING : 0)));
                d.rLcPartner.push_back(y.partner.empty() ? -1 : vidOf.at(y.partner));
                d.rLcOffset.push_back(y.offset);
                d.rLcLastDir.push_back(0);  // not archived by the reference (the next clearSignal rewrites it)
                d.rLcTarget.push_back(y.target.empty() ? -1 : drvIndex.at(y.target));
                d.rLcDirection.push_back(y.direction);
                d.rLcLastChangeTime.push_back(y.lastTime);
                d.rLcWaitingTime.push_back(y.waitingTime);
            }
        }
        if (dv < L) {
            for (const Json &jid : jd.arrayAt("waitingBuffer").items) {
                int vid = vidOf.at(jid.s);
                d.wVid.push_back(vid);
                d.wLane.push_back(dv);
                a.host.lastWaitVid[dv] = vid;
            }
            // archive.cpp:508-521 (records beyond what the list can hold cannot have been written by either engine)
            if (d.hLen.empty()) {
                d.hLen.assign((size_t) L, 0);
                d.hVehicleNum.assign((size_t) L * CFX_LANE_HISTORY_MAX, 0);
                d.hAverageSpeed.assign((size_t) L * CFX_LANE_HISTORY_MAX, 0.0);
                d.hHistoryVehicleNum.assign((size_t) L, 0);
                d.hHistoryAverageSpeed.assign((size_t) L, 0.0);
            }
            const Json *jh = jd.find("history");
            if (jh && jh->isArray()) {
                const size_t pairs = std::min<size_t>(jh->items.size() / 2, CFX_LANE_HISTORY_MAX);
                for (size_t i = 0; i < pairs; ++i) {
                    d.hVehicleNum[(size_t) dv * CFX_LANE_HISTORY_MAX + i] = (int32_t) jh->items[2 * i].asDouble();
                    d.hAverageSpeed[(size_t) dv * CFX_LANE_HISTORY_MAX + i] = jh->items[2 * i + 1].asDouble();
                }
                d.hLen[(size_t) dv] = (int32_t) pairs;
            }
            if (const Json *x = jd.find("historyVehicleNum")) d.hHistoryVehicleNum[(size_t) dv] = (int32_t) x->asDouble();
            if (const Json *x = jd.find("historyAverageSpeed")) d.hHistoryAverageSpeed[(size_t) dv] = x->asDouble();
        }
    };
    auto afterVehicles = [&]() {
        const int nV = (int) a.host.vehicles.size();
        for (auto &sb : shadowBase) {  // the id a shadow carries belongs to the vehicle named so (its parent or an ancestor's heir)
            auto it = vidOf.find(sb.second);
            const int holder = it == vidOf.end() ? sb.first : it->second;
            a.host.vehicles[(size_t) sb.first].root = holder;
            a.host.shadowChains[holder].push_back(sb.first);
        }
        d.vState.assign(nV, 0);
        vehiclesDone = true;
    };
    auto drainHeld = [&]() {  // whatever can now be taken in the network's order
        for (auto it = heldDrivables.begin(); it != heldDrivables.end() && it->first == nextDv; it = heldDrivables.erase(it)) {
            oneDrivable(nextDv, it->second);
            ++nextDv;
        }
    };
    Json root = Json::parseFileStreamed(path, {"vehicles", "drivables"}, [&](const std::string &member, const std::string &key, Json &child) {
        if (member == "vehicles") {
            if (vehiclesDone) throw std::runtime_error("load_from_file: two `vehicles` members");
            oneVehicle(child);
            return;
        }
        auto di = drvIndex.find(key);
        if (di == drvIndex.end()) return;  // (a drivable this network does not have: ignored, as a DOM lookup by the network's ids would)
        if (!vehiclesDone && !a.host.vehicles.empty()) afterVehicles();  // (the `vehicles` array has ended)
        if (vehiclesDone && di->second == nextDv) {
            oneDrivable(nextDv, child);
            ++nextDv;
            drainHeld();
        } else {
            heldDrivables[di->second] = std::move(child);
        }
    });
    if (!vehiclesDone) afterVehicles();  // (no vehicles, or `drivables` came first in the file)
    drainHeld();
    if (nextDv != D) throw JsonError(net_->drivableId(nextDv) + " is required but missing in json file");
    {
        std::istringstream is(root.stringAt("rnd"));
        is >> a.host.rnd;
    }

    d.step = root.at("step").i;
    d.finished = root.intAt("finishedVehicleCnt");
    d.cumulativeTravelTime = root.numberAt("cumulativeTravelTime");
    d.vehicleSteps = 0;

    const Json &flows = root.objectAt("flows");
    for (size_t f = 0; f < spawner_.flows.size(); ++f) {
        const Json &jf = flows.objectAt(spawner_.flows[f].id.c_str());
        a.host.flows[f].nowTime = jf.numberAt("nowTime");
        a.host.flows[f].currentTime = jf.numberAt("currentTime");
        a.host.flows[f].cnt = jf.intAt("cnt");
    }
    const Json &lights = root.objectAt("trafficLights");
    d.tlPhase.resize(net_->inters.size());
    d.tlRemain.resize(net_->inters.size());
    for (size_t i = 0; i < net_->inters.size(); ++i) {
        const Json &jl = lights.objectAt(net_->inters[i].id.c_str());
        d.tlPhase[i] = jl.intAt("curPhaseIndex");
        d.tlRemain[i] = jl.numberAt("remainDuration");
    }
    a.templates = spawner_.templates;
    a.routeStart = spawner_.routes.routeStart;
    a.routeRoads = spawner_.routes.roads;
    for (const HostFlow &f : spawner_.flows) a.flowIds.push_back(f.id);
    return a;
}

}  // namespace cfa
