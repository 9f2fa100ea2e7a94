#include "replay.h"

#include <algorithm>
#include <charconv>
#include <cmath>

#include "engine_host.h"

namespace cfa {

namespace {

inline Pt sub(Pt a, Pt b) { return {a.x - b.x, a.y - b.y}; }
inline Pt add(Pt a, Pt b) { return {a.x + b.x, a.y + b.y}; }
inline Pt mul(Pt a, double k) { return {a.x * k, a.y * k}; }
inline double len(Pt a) { return sqrt(a.x * a.x + a.y * a.y); }  // Point::len utility.cpp:74-76
inline Pt unit(Pt a) {                                            // Point::unit utility.cpp:63-66
    double l = len(a);
    return {a.x / l, a.y / l};
}
inline double ang(Pt a) { return atan2(a.y, a.x); }
inline double cross(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }

void num(std::string &out, double x) {
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof buf, x);
    out.append(buf, r.ptr);
}

void quoted(std::string &out, const std::string &s) {
    out.push_back('"');
    for (char c : s) {
        if (c == '"' || c == '\\') out.push_back('\\');
        out.push_back(c);
    }
    out.push_back('"');
}

// getPointByDistance roadnet.cpp:17-29
Pt pointByDistance(const std::vector<Pt> &points, double dis) {
    double total = 0.0;
    for (size_t i = 0; i + 1 < points.size(); ++i) total += len(sub(points[i + 1], points[i]));
    double lo = dis > 0 ? dis : 0;  // max2double(dis, 0)
    dis = lo < total ? lo : total;  // min2double(., length)
    if (dis <= 0.0) return points[0];
    for (size_t i = 1; i < points.size(); ++i) {
        double l = len(sub(points[i - 1], points[i]));
        if (dis > l) dis -= l;
        else return add(points[i - 1], mul(sub(points[i], points[i - 1]), dis / l));
    }
    return points.back();
}

// Drivable::getDirectionByDistance roadnet.cpp:400-411
Pt directionByDistance(const std::vector<Pt> &points, double dis) {
    double remain = dis;
    for (int i = 0; i + 1 < (int) points.size(); ++i) {
        double l = len(sub(points[i + 1], points[i]));
        if (remain < l) return unit(sub(points[i + 1], points[i]));
        remain -= l;
    }
    return unit(sub(points[points.size() - 1], points[points.size() - 2]));
}

}  // namespace

// The "outline" polygon of an intersection in the roadnet log (what the frontend draws): the convex hull of the
// intersection's centre and, per adjoining road, the two corners of the road's mouth (plus the two corners one
// `deltaWidth` further up the road where the road is long enough).  Values must equal Intersection::getOutline
// (/root/reference/src/roadnet/roadnet.cpp:750-818) to the bit, which fixes three things the geometry leaves open:
//   * the corner arithmetic's operation order (FP64, no contraction);
//   * the pivot = FIRST candidate of minimal y (the reference's std::min_element), removed from the candidates by position;
//   * the order of candidates with EQUAL polar angle around the pivot: the reference hands an unstable std::sort a
//     comparator on the angle, so the permutation is libstdc++'s introsort's for that comparator's answers.  Sorting
//     (angle, point) pairs by the precomputed angle asks the same questions in the same order, so it moves the elements
//     the same way — and calls atan2 once per candidate instead of twice per comparison.
// The scan is Graham's: a candidate pops every hull vertex at which the path would not turn strictly one way.
namespace {
struct MouthCorners {
    Pt near[2], far[2];
    bool hasFar;
};
MouthCorners roadMouth(const HostRoadNet &net, const HostInter &in, int ii, int r) {
    const HostRoad &road = net.roads[r];
    Pt along = unit(sub(net.inters[road.endInter].point, net.inters[road.startInter].point));
    const Pt across{-along.y, along.x};  // Point::normal of the road's direction BEFORE it is turned towards the intersection
    if (road.startInter == ii) along = Pt{-along.x, -along.y};
    double roadWidth = 0;
    for (int j = 0; j < road.nLanes; ++j) roadWidth += net.lanes[road.laneStart + j].width;
    double step = 0.5 * (in.width < roadWidth ? in.width : roadWidth);  // min2double / max2double of utility.h
    step = step > 5 ? step : 5;
    MouthCorners m;
    m.near[0] = sub(in.point, mul(along, in.width));
    m.near[1] = sub(m.near[0], mul(across, roadWidth));
    m.hasFar = step < net.averageLength(r);
    if (m.hasFar) {
        m.far[0] = sub(m.near[0], mul(along, step));
        m.far[1] = sub(m.near[1], mul(along, step));
    }
    return m;
}
}  // namespace

std::vector<Pt> intersectionOutline(const HostRoadNet &net, int ii) {
    const HostInter &in = net.inters[ii];
    std::vector<Pt> cand{in.point};
    for (int r : in.roads) {
        const MouthCorners m = roadMouth(net, in, ii, r);
        cand.insert(cand.end(), m.near, m.near + 2);
        if (m.hasFar) cand.insert(cand.end(), m.far, m.far + 2);
    }
    size_t pivotAt = 0;
    for (size_t i = 1; i < cand.size(); ++i)
        if (cand[i].y < cand[pivotAt].y) pivotAt = i;
    const Pt pivot = cand[pivotAt];
    struct Keyed {
        double angle;
        Pt p;
    };
    std::vector<Keyed> rest;
    rest.reserve(cand.size());
    for (size_t i = 0; i < cand.size(); ++i)
        if (i != pivotAt) rest.push_back(Keyed{ang(sub(cand[i], pivot)), cand[i]});
    std::sort(rest.begin(), rest.end(), [](const Keyed &a, const Keyed &b) { return a.angle < b.angle; });
    std::vector<Pt> hull{pivot};
    for (const Keyed &k : rest) {
        const Pt &q = k.p;
        if (hull.size() == 1) {  // (only here does the reference drop a candidate that coincides with the hull's top)
            if (q.x != pivot.x || q.y != pivot.y) hull.push_back(q);
            continue;
        }
        while (hull.size() >= 2 && cross(sub(q, hull.back()), sub(hull.back(), hull[hull.size() - 2])) >= 0) hull.pop_back();
        hull.push_back(q);
    }
    return hull;
}

bool writeRoadnetLog(const HostRoadNet &net, const std::string &path) {
    std::string o = "{\"static\":{\"nodes\":[";
    for (size_t i = 0; i < net.inters.size(); ++i) {
        const HostInter &in = net.inters[i];
        if (i) o.push_back(',');
        o += "{\"id\":";
        quoted(o, in.id);
        o += ",\"point\":[";
        num(o, in.point.x);
        o.push_back(',');
        num(o, in.point.y);
        o += "],\"virtual\":";
        o += in.isVirtual ? "true" : "false";
        if (!in.isVirtual) {
            o += ",\"width\":";
            num(o, in.width);
        }
        o += ",\"outline\":[";
        bool first = true;
        for (const Pt &p : intersectionOutline(net, (int) i)) {
            if (!first) o.push_back(',');
            first = false;
            num(o, p.x);
            o.push_back(',');
            num(o, p.y);
        }
        o += "]}";
    }
    o += "],\"edges\":[";
    for (size_t r = 0; r < net.roads.size(); ++r) {
        const HostRoad &road = net.roads[r];
        if (r) o.push_back(',');
        o += "{\"id\":";
        quoted(o, road.id);
        o += ",\"from\":";
        quoted(o, road.startInter >= 0 ? net.inters[road.startInter].id : std::string("null"));
        o += ",\"to\":";
        quoted(o, road.endInter >= 0 ? net.inters[road.endInter].id : std::string("null"));
        o += ",\"points\":[";
        for (size_t j = 0; j < road.points.size(); ++j) {
            if (j) o.push_back(',');
            o.push_back('[');
            num(o, road.points[j].x);
            o.push_back(',');
            num(o, road.points[j].y);
            o.push_back(']');
        }
        o += "],\"nLane\":" + std::to_string(road.nLanes) + ",\"laneWidths\":[";
        for (int j = 0; j < road.nLanes; ++j) {
            if (j) o.push_back(',');
            num(o, net.lanes[road.laneStart + j].width);
        }
        o += "]}";
    }
    o += "]}}";
    std::ofstream f(path);
    if (!f) return false;
    f << o;
    return (bool) f;
}

bool ReplayWriter::open(const std::string &path) {
    close();
    out_.open(path);
    return out_.is_open();
}

void ReplayWriter::close() {
    if (out_.is_open()) out_.close();
}

void ReplayWriter::writeStep(const HostRoadNet &net, const Spawner &sp, const VehicleSnapshot &s, const std::vector<int32_t> &phase) {
    if (!out_.is_open()) return;
    const int L = (int) net.lanes.size();
    order_.clear();
    for (int i = 0; i < s.count; ++i)  // Engine::getRunningVehicles engine.cpp:780-790: real vehicles, vehiclePool order
        if (!s.isShadow(i)) order_.emplace_back(sp.vehicles[s.vid[i]].priority, i);
    std::sort(order_.begin(), order_.end());
    std::string &o = line_;
    o.clear();
    for (auto &pi : order_) {
        const int i = pi.second;
        const int d = s.drivable[i];
        const std::vector<Pt> &pts = d < L ? net.lanes[d].points : net.laneLinks[d - L].points;
        Pt pos = pointByDistance(pts, s.dis[i]);  // Vehicle::getPoint vehicle.cpp:81-105
        const double offset = s.lcOffset.empty() ? 0.0 : s.lcOffset[i];
        if (!(std::fabs(offset) < 1e-8) && d < L) {  // changing lane: blend towards the neighbour lane's point
            const HostLane &lane = net.lanes[d];
            const HostLane &other = net.lanes[offset > 0 ? d + 1 : d - 1];
            const Pt next = pointByDistance(other.points, s.dis[i]);
            const double percentage = (offset > 0 ? 2 : -2) * offset / (lane.width + other.width);
            Pt cur;
            cur.x = next.x * percentage + pos.x * (1 - percentage);
            cur.y = next.y * percentage + pos.y * (1 - percentage);
            pos = cur;
        }
        Pt dir = directionByDistance(pts, s.dis[i]);
        const cfx_vehicle_template &t = sp.templates[sp.vehicles[s.vid[i]].templ];
        num(o, pos.x);
        o.push_back(' ');
        num(o, pos.y);
        o.push_back(' ');
        num(o, atan2(dir.y, dir.x));
        o.push_back(' ');
        o += sp.vehicleId(s.vid[i]);
        o.push_back(' ');
        o += std::to_string(s.lcLastDir.empty() ? 0 : s.lcLastDir[i]);  // Vehicle::lastLaneChangeDirection
        o.push_back(' ');
        num(o, t.len);
        o.push_back(' ');
        num(o, t.width);
        o.push_back(',');
    }
    o.push_back(';');
    for (const HostRoad &road : net.roads) {
        const HostInter &end = net.inters[road.endInter];
        if (end.isVirtual) continue;
        o += road.id;
        for (int j = 0; j < road.nLanes; ++j) {
            if (end.phases.size() <= 1) {  // Intersection::isImplicitIntersection roadnet.cpp:820-822
                o += " i";
                continue;
            }
            bool canGo = true;
            for (int ll : net.lanes[road.laneStart + j].laneLinks)
                if (!end.phases[phase[road.endInter]].avail[net.laneLinks[ll].roadLink]) {
                    canGo = false;
                    break;
                }
            o += canGo ? " g" : " r";
        }
        o.push_back(',');
    }
    out_ << o << std::endl;
}

}  // namespace cfa
