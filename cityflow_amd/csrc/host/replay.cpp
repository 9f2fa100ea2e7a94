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

std::vector<Pt> intersectionOutline(const HostRoadNet &net, int ii) {
    const HostInter &in = net.inters[ii];
    std::vector<Pt> points;
    points.push_back(in.point);
    for (int r : in.roads) {
        const HostRoad &road = net.roads[r];
        Pt roadDirect = unit(sub(net.inters[road.endInter].point, net.inters[road.startInter].point));
        Pt pDirect{-roadDirect.y, roadDirect.x};  // Point::normal
        if (road.startInter == ii) roadDirect = Pt{-roadDirect.x, -roadDirect.y};
        double roadWidth = 0;
        for (int j = 0; j < road.nLanes; ++j) roadWidth += net.lanes[road.laneStart + j].width;
        double deltaWidth = 0.5 * (in.width < roadWidth ? in.width : roadWidth);
        deltaWidth = deltaWidth > 5 ? deltaWidth : 5;
        Pt pointA = sub(in.point, mul(roadDirect, in.width));
        Pt pointB = sub(pointA, mul(pDirect, roadWidth));
        points.push_back(pointA);
        points.push_back(pointB);
        if (deltaWidth < net.averageLength(r)) {
            points.push_back(sub(pointA, mul(roadDirect, deltaWidth)));
            points.push_back(sub(pointB, mul(roadDirect, deltaWidth)));
        }
    }
    auto minIter = std::min_element(points.begin(), points.end(), [](const Pt &a, const Pt &b) { return a.y < b.y; });
    const Pt p0 = *minIter;
    std::vector<Pt> stack{p0};
    points.erase(minIter);
    std::sort(points.begin(), points.end(), [&p0](const Pt &a, const Pt &b) { return ang(sub(a, p0)) < ang(sub(b, p0)); });
    for (const Pt &point : points) {
        Pt p2 = stack.back();
        if (stack.size() < 2) {
            if (point.x != p2.x || point.y != p2.y) stack.push_back(point);
            continue;
        }
        Pt p1 = stack[stack.size() - 2];
        while (stack.size() > 1 && cross(sub(point, p2), sub(p2, p1)) >= 0) {
            p2 = p1;
            stack.pop_back();
            if (stack.size() > 1) p1 = stack[stack.size() - 2];
        }
        stack.push_back(point);
    }
    return stack;
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
