#include "roadnet.h"

#include <algorithm>

#include "json.h"

namespace cfa {

// ---- 2-D helpers: same operations, same order as reference src/utility/utility.cpp:15-84 ----
namespace {
inline Pt sub(const Pt &a, const Pt &b) { return Pt{a.x - b.x, a.y - b.y}; }
inline Pt add(const Pt &a, const Pt &b) { return Pt{a.x + b.x, a.y + b.y}; }
inline Pt mul(const Pt &a, double k) { return Pt{a.x * k, a.y * k}; }
inline Pt neg(const Pt &a) { return Pt{-a.x, -a.y}; }
inline double len(const Pt &a) { return std::sqrt(a.x * a.x + a.y * a.y); }
inline Pt unit(const Pt &a) {
    double l = len(a);
    return Pt{a.x / l, a.y / l};
}
inline Pt normal(const Pt &a) { return Pt{-a.y, a.x}; }
inline double crossMul(const Pt &a, const Pt &b) { return a.x * b.y - a.y * b.x; }
inline double dotMul(const Pt &a, const Pt &b) { return a.x * b.x + a.y * b.y; }
constexpr double kEps = 1e-8;  // Point::eps utility.h:29
inline int sgn(double x) { return (x + kEps > 0) - (x < kEps); }  // Point::sign utility.cpp:82-84
inline double min2(double x, double y) { return x < y ? x : y; }   // utility.h:70-72
inline double max2(double x, double y) { return x > y ? x : y; }   // utility.h:66-68

// getLengthOfPoints roadnet.cpp:31-36
double polylineLength(const std::vector<Pt> &p) {
    double length = 0.0;
    for (size_t i = 0; i + 1 < p.size(); i++) length += len(sub(p[i + 1], p[i]));
    return length;
}

// getPointByDistance roadnet.cpp:17-29
Pt pointByDistance(const std::vector<Pt> &p, double dis) {
    dis = min2(max2(dis, 0), polylineLength(p));
    if (dis <= 0.0) return p[0];
    for (size_t i = 1; i < p.size(); i++) {
        double l = len(sub(p[i - 1], p[i]));
        if (dis > l)
            dis -= l;
        else
            return add(p[i - 1], mul(sub(p[i], p[i - 1]), dis / l));
    }
    return p.back();
}

// Drivable::getDirectionByDistance roadnet.cpp:400-410
Pt directionByDistance(const std::vector<Pt> &p, double dis) {
    double remain = dis;
    for (int i = 0; i + 1 < (int) p.size(); i++) {
        double l = len(sub(p[i + 1], p[i]));
        if (remain < l)
            return unit(sub(p[i + 1], p[i]));
        else
            remain -= l;
    }
    return unit(sub(p[p.size() - 1], p[p.size() - 2]));
}

// RoadNet::getPoint roadnet.cpp:38-40
inline Pt lerp(const Pt &p1, const Pt &p2, double a) { return Pt{(p2.x - p1.x) * a + p1.x, (p2.y - p1.y) * a + p1.y}; }

// calcIntersectPoint utility.cpp:30-36
inline Pt intersectPoint(const Pt &A, const Pt &B, const Pt &C, const Pt &D) {
    Pt u = sub(B, A);
    Pt v = sub(D, C);
    return add(A, mul(u, crossMul(sub(C, A), v) / crossMul(u, v)));
}

// onSegment utility.cpp:57-61
inline bool onSegment(const Pt &A, const Pt &B, const Pt &P) {
    double v1 = crossMul(sub(B, A), sub(P, A));
    double v2 = dotMul(sub(P, A), sub(P, B));
    return sgn(v1) == 0 && sgn(v2) <= 0;
}

Pt readPoint(const Json &v) { return Pt{v.numberAt("x"), v.numberAt("y")}; }
}  // namespace

// Lane centre lines of a road (reference: Road::initLanesPoints roadnet.cpp:456-505; called twice by the reference loader,
// roadnet.cpp:127-129 before intersections are read — widths 0, nothing virtual — and again at 307-308).  The road's
// polyline is shortened by the intersection widths at its ends; every lane is that polyline shifted sideways by the
// distance of the lane's middle from the road's left edge.  The shift direction of a vertex (the normal of the bisector
// of its two segments) does not depend on the lane, so it is computed once per vertex; every coordinate still goes
// through the same operations in the same order as in the reference, which the geometry hashes pin.
void HostRoadNet::initLanesPoints(int r) {
    HostRoad &road = roads[r];
    std::vector<Pt> axis = road.points;
    if (axis.size() < 2) throw JsonError("road " + road.id + ": needs at least 2 points");
    const size_t last = axis.size() - 1;
    if (!inters[road.startInter].isVirtual)
        axis[0] = add(axis[0], mul(unit(sub(axis[1], axis[0])), inters[road.startInter].width));
    if (!inters[road.endInter].isVirtual)
        axis[last] = sub(axis[last], mul(unit(sub(axis[last], axis[last - 1])), inters[road.endInter].width));
    std::vector<Pt> side(axis.size());  // unit vector pointing to the right of the direction of travel, per vertex
    for (size_t j = 0; j <= last; ++j) {
        Pt heading;
        if (j == 0) heading = unit(sub(axis[1], axis[0]));
        else if (j == last) heading = unit(sub(axis[j], axis[j - 1]));
        else heading = unit(add(unit(sub(axis[j + 1], axis[j])), unit(sub(axis[j], axis[j - 1]))));
        side[j] = neg(normal(heading));
    }
    double leftEdge = 0.0;
    for (int li = 0; li < road.nLanes; ++li) {
        HostLane &lane = lanes[road.laneStart + li];
        const double rightEdge = leftEdge + lane.width;
        const double middle = (leftEdge + rightEdge) / 2.0;
        lane.points.resize(axis.size());
        for (size_t j = 0; j <= last; ++j) lane.points[j] = add(axis[j], mul(side[j], middle));
        lane.length = polylineLength(lane.points);
        leftEdge += lane.width;
    }
}

namespace {
// Where polyline `a` first meets polyline `b`: a's segments in order and, for each of them, b's segments in order — two
// laneLinks can meet more than once and this scan order decides which meeting is "the" cross (reference:
// Intersection::initCrosses roadnet.cpp:515-576).  The distances are measured along each polyline up to the meeting
// point; like in the reference, segments of b that are parallel to the current segment of a are skipped WITHOUT
// counting their length (roadnet.cpp:535 vs 557).
struct Meeting {
    double alongA, alongB;
};
bool firstMeeting(const std::vector<Pt> &a, const std::vector<Pt> &b, Meeting &m) {
    double walkedA = 0.0;
    for (size_t sa = 0; sa + 1 < a.size(); ++sa) {
        const Pt a0 = a[sa], a1 = a[sa + 1];
        double walkedB = 0.0;
        for (size_t sb = 0; sb + 1 < b.size(); ++sb) {
            const Pt b0 = b[sb], b1 = b[sb + 1];
            if (sgn(crossMul(sub(a1, a0), sub(b1, b0))) == 0) continue;
            const Pt p = intersectPoint(a0, a1, b0, b1);
            if (onSegment(a0, a1, p) && onSegment(b0, b1, p)) {
                m.alongA = walkedA + len(sub(p, a0));
                m.alongB = walkedB + len(sub(p, b0));
                return true;
            }
            walkedB += len(sub(b1, b0));
        }
        walkedA += len(sub(a1, a0));
    }
    return false;
}
}  // namespace

// The conflict points of an intersection: one per pair of its laneLinks whose curves meet, then every laneLink's list of
// them ordered by distance along the laneLink.
void HostRoadNet::initCrosses(int ii) {
    HostInter &inter = inters[ii];
    std::vector<int> links;
    for (const auto &rl : inter.roadLinks)
        for (int k = 0; k < rl.nLaneLinks; ++k) links.push_back(rl.llStart + k);
    for (size_t i = 0; i < links.size(); ++i)
        for (size_t j = i + 1; j < links.size(); ++j) {
            Meeting m;
            if (!firstMeeting(laneLinks[links[i]].points, laneLinks[links[j]].points, m)) continue;
            HostCross c;
            c.ll[0] = links[i];
            c.ll[1] = links[j];
            c.dist[0] = m.alongA;
            c.dist[1] = m.alongB;
            inter.crosses.push_back(c);
        }
    const std::vector<HostCross> &xs = inter.crosses;
    for (int c = 0; c < (int) xs.size(); ++c)
        for (int side = 0; side < 2; ++side) laneLinks[xs[c].ll[side]].crosses.push_back(c);
    // std::sort on the same initial order with the same comparison gives the reference's permutation, also among equal
    // distances (the sibling laneLinks of one start lane all leave it at distance 0, and std::sort is not stable)
    for (int ll : links) {
        std::vector<int> &mine = laneLinks[ll].crosses;
        std::sort(mine.begin(), mine.end(), [ll, &xs](int ca, int cb) {
            return xs[ca].dist[xs[ca].ll[0] != ll] < xs[cb].dist[xs[cb].ll[0] != ll];
        });
    }
}

// RoadNet::loadFromJson roadnet.cpp:42-325 (same read order; errors become JsonError).
void HostRoadNet::load(const std::string &path) {
    Json doc = Json::parseFile(path);
    if (!doc.isObject()) throw JsonError("roadnet config file: expected type object");
    const Json &interValues = doc.arrayAt("intersections");
    const Json &roadValues = doc.arrayAt("roads");

    roads.resize(roadValues.items.size());
    inters.resize(interValues.items.size());
    for (size_t i = 0; i < roads.size(); ++i) {
        roads[i].id = roadValues.items[i].stringAt("id");
        roadIndex[roads[i].id] = (int) i;
    }
    for (size_t i = 0; i < inters.size(); ++i) {
        inters[i].id = interValues.items[i].stringAt("id");
        interIndex[inters[i].id] = (int) i;
    }

    // roads: endpoints, lanes, points (roadnet.cpp:77-125)
    for (size_t i = 0; i < roads.size(); ++i) {
        const Json &rv = roadValues.items[i];
        if (!rv.isObject()) throw JsonError("road[" + std::to_string(i) + "]: expected type object");
        auto s = interIndex.find(rv.stringAt("startIntersection"));
        auto e = interIndex.find(rv.stringAt("endIntersection"));
        if (s == interIndex.end()) throw JsonError("startIntersection does not exist.");
        if (e == interIndex.end()) throw JsonError("endIntersection does not exist.");
        roads[i].startInter = s->second;
        roads[i].endInter = e->second;
        roads[i].laneStart = (int) lanes.size();
        int laneIndex = 0;
        for (const Json &lv : rv.arrayAt("lanes").items) {
            if (!lv.isObject()) throw JsonError("lane: expected type object");
            HostLane lane;
            lane.road = (int) i;
            lane.index = laneIndex++;
            lane.width = lv.numberAt("width");
            lane.maxSpeed = lv.numberAt("maxSpeed");
            lanes.push_back(std::move(lane));
        }
        roads[i].nLanes = laneIndex;
        for (const Json &pv : rv.arrayAt("points").items) {
            if (!pv.isObject()) throw JsonError("point of road: expected type object");
            roads[i].points.push_back(readPoint(pv));
        }
    }

    // first geometry pass: intersections not read yet => width 0, nothing virtual (roadnet.cpp:127-129)
    for (size_t i = 0; i < roads.size(); ++i) initLanesPoints((int) i);

    // intersections (roadnet.cpp:131-291)
    for (size_t i = 0; i < inters.size(); ++i) {
        const Json &iv = interValues.items[i];
        if (!iv.isObject()) throw JsonError("intersection: expected type object");
        HostInter &inter = inters[i];
        const Json &pv = iv.objectAt("point");
        inter.isVirtual = iv.boolAt("virtual");
        inter.point = readPoint(pv);
        for (const Json &rn : iv.arrayAt("roads").items) {
            if (!rn.isString()) throw JsonError("roads: expected type string");
            auto it = roadIndex.find(rn.s);
            if (it == roadIndex.end()) throw JsonError("No such road: " + rn.s);
            inter.roads.push_back(it->second);
        }
        if (inter.isVirtual) continue;
        inter.width = iv.numberAt("width");

        const Json &rlValues = iv.arrayAt("roadLinks");
        inter.roadLinks.resize(rlValues.items.size());
        for (size_t r = 0; r < rlValues.items.size(); ++r) {
            const Json &rlv = rlValues.items[r];
            if (!rlv.isObject()) throw JsonError("roadLink: expected type object");
            HostRoadLink &rl = inter.roadLinks[r];
            const std::string &type = rlv.stringAt("type");
            if (type == "turn_left") rl.type = 2;
            else if (type == "turn_right") rl.type = 1;
            else if (type == "go_straight") rl.type = 3;
            else throw JsonError("unknown roadLink type: " + type);
            auto sr = roadIndex.find(rlv.stringAt("startRoad"));
            auto er = roadIndex.find(rlv.stringAt("endRoad"));
            if (sr == roadIndex.end() || er == roadIndex.end()) throw JsonError("roadLink: no such road");
            rl.startRoad = sr->second;
            rl.endRoad = er->second;
            rl.llStart = (int) laneLinks.size();
            for (const Json &llv : rlv.arrayAt("laneLinks").items) {
                if (!llv.isObject()) throw JsonError("laneLink: expected type object");
                int sIdx = llv.intAt("startLaneIndex");
                int eIdx = llv.intAt("endLaneIndex");
                if (sIdx >= roads[rl.startRoad].nLanes || sIdx < 0) throw JsonError("startLaneIndex out of range");
                if (eIdx >= roads[rl.endRoad].nLanes || eIdx < 0) throw JsonError("startLaneIndex out of range");
                HostLaneLink ll;
                ll.inter = (int) i;
                ll.roadLink = (int) r;
                ll.type = rl.type;
                ll.startLane = roads[rl.startRoad].laneStart + sIdx;
                ll.endLane = roads[rl.endRoad].laneStart + eIdx;
                const Json *pts = llv.find("points");
                if (pts && !pts->isArray()) throw JsonError("points in laneLink: expected type array");
                if (pts && !pts->items.empty()) {
                    for (const Json &p : pts->items) ll.points.push_back(readPoint(p));
                } else {
                    // generated curve, roadnet.cpp:212-247 (uses the FIRST-pass lane points)
                    const HostLane &sl = lanes[ll.startLane];
                    const HostLane &el = lanes[ll.endLane];
                    double sw = inters[roads[sl.road].endInter].width;
                    double ew = inters[roads[el.road].startInter].width;
                    Pt start = pointByDistance(sl.points, sl.length - sw);
                    Pt end = pointByDistance(el.points, 0.0 + ew);
                    double l = len(Pt{end.x - start.x, end.y - start.y});
                    Pt sd = directionByDistance(sl.points, sl.length - sw);
                    Pt ed = directionByDistance(el.points, 0.0 + ew);
                    double minGap = 5;
                    double gap1X = sd.x * l * 0.5;
                    double gap1Y = sd.y * l * 0.5;
                    double gap2X = -ed.x * l * 0.5;
                    double gap2Y = -ed.y * l * 0.5;
                    if (gap1X * gap1X + gap1Y * gap1Y < 25 && sw >= 5) {
                        gap1X = minGap * sd.x;
                        gap1Y = minGap * sd.y;
                    }
                    if (gap2X * gap2X + gap2Y * gap2Y < 25 && ew >= 5) {
                        gap2X = minGap * ed.x;
                        gap2Y = minGap * ed.y;
                    }
                    Pt mid1{start.x + gap1X, start.y + gap1Y};
                    Pt mid2{end.x + gap2X, end.y + gap2Y};
                    int numPoints = 10;
                    for (int q = 0; q <= numPoints; q++) {
                        double a = q / double(numPoints);
                        Pt p1 = lerp(start, mid1, a);
                        Pt p2 = lerp(mid1, mid2, a);
                        Pt p3 = lerp(mid2, end, a);
                        Pt p4 = lerp(p1, p2, a);
                        Pt p5 = lerp(p2, p3, a);
                        Pt p6 = lerp(p4, p5, a);
                        ll.points.push_back(p6);
                    }
                }
                ll.length = polylineLength(ll.points);
                lanes[ll.startLane].laneLinks.push_back((int) laneLinks.size());
                laneLinks.push_back(std::move(ll));
                rl.nLaneLinks++;
            }
        }

        const Json &tl = iv.objectAt("trafficLight");
        for (const Json &phv : tl.arrayAt("lightphases").items) {
            if (!phv.isObject()) throw JsonError("lightphase: expected type object");
            HostPhase ph;
            ph.time = phv.numberAt("time");
            ph.avail.assign(inter.roadLinks.size(), 0);
            for (const Json &a : phv.arrayAt("availableRoadLinks").items) {
                if (!a.isInt()) throw JsonError("availableRoadLink: expected type int");
                size_t idx = (size_t) (unsigned) a.i;
                if (idx >= ph.avail.size()) throw JsonError("index out of range");
                ph.avail[idx] = 1;
            }
            inter.phases.push_back(std::move(ph));
        }
        if (inter.phases.empty()) throw JsonError("intersection " + inter.id + ": no lightphases");
    }

    int xBase = 0;
    for (size_t i = 0; i < inters.size(); ++i) {
        initCrosses((int) i);
        inters[i].xBase = xBase;
        xBase += (int) inters[i].crosses.size();
    }
    // second geometry pass with the real intersection widths (roadnet.cpp:307-308)
    for (size_t i = 0; i < roads.size(); ++i) initLanesPoints((int) i);

    flatten();
}

std::vector<int> HostRoadNet::laneLinksToRoad(int lane, int road) const {
    std::vector<int> ret;
    for (int ll : lanes[lane].laneLinks)
        if (lanes[laneLinks[ll].endLane].road == road) ret.push_back(ll);
    return ret;
}

bool HostRoadNet::connectedToRoad(int from, int to) const {
    const HostRoad &r = roads[from];
    for (int li = 0; li < r.nLanes; ++li)
        for (int ll : lanes[r.laneStart + li].laneLinks)
            if (lanes[laneLinks[ll].endLane].road == to) return true;
    return false;
}

double HostRoadNet::averageLength(int road) const {
    const HostRoad &r = roads[road];
    double sum = 0;
    size_t laneNum = (size_t) r.nLanes;
    if (laneNum == 0) return 0;
    for (int li = 0; li < r.nLanes; ++li) sum += lanes[r.laneStart + li].length;
    return sum / laneNum;
}

void HostRoadNet::flatten() {
    const int L = (int) lanes.size(), K = (int) laneLinks.size(), R = (int) roads.size(), I = (int) inters.size();
    drvLength_.resize(L + K);
    drvMaxSpeed_.resize(L + K);
    laneRoad_.resize(L);
    laneIndex_.resize(L);
    laneLLStart_.assign(L + 1, 0);
    laneLL_.clear();
    for (int l = 0; l < L; ++l) {
        drvLength_[l] = lanes[l].length;
        drvMaxSpeed_[l] = lanes[l].maxSpeed;
        laneRoad_[l] = lanes[l].road;
        laneIndex_[l] = lanes[l].index;
        laneLLStart_[l] = (int) laneLL_.size();
        for (int ll : lanes[l].laneLinks) laneLL_.push_back(ll);
    }
    laneLLStart_[L] = (int) laneLL_.size();
    roadLaneStart_.resize(R + 1);
    for (int r = 0; r < R; ++r) roadLaneStart_[r] = roads[r].laneStart;
    roadLaneStart_[R] = L;

    llStartLane_.resize(K);
    llEndLane_.resize(K);
    llInter_.resize(K);
    llRoadLink_.resize(K);
    llType_.resize(K);
    llXStart_.assign(K + 1, 0);
    xDist_.clear();
    xPeer_.clear();
    xLL_.clear();
    // entry index of (global cross, side)
    int totalCross = 0;
    for (auto &in : inters) totalCross += (int) in.crosses.size();
    std::vector<int32_t> entryOf((size_t) totalCross * 2, -1);
    for (int k = 0; k < K; ++k) {
        const HostLaneLink &ll = laneLinks[k];
        drvLength_[L + k] = ll.length;
        drvMaxSpeed_[L + k] = 10000;  // LaneLink::maxSpeed roadnet.h:456
        llStartLane_[k] = ll.startLane;
        llEndLane_[k] = ll.endLane;
        llInter_[k] = ll.inter;
        llRoadLink_[k] = ll.roadLink;
        llType_[k] = ll.type;
        llXStart_[k] = (int) xDist_.size();
        const HostInter &in = inters[ll.inter];
        for (int c : ll.crosses) {
            int side = in.crosses[c].ll[0] != k;
            entryOf[(size_t) (in.xBase + c) * 2 + side] = (int32_t) xDist_.size();
            xDist_.push_back(in.crosses[c].dist[side]);
            xLL_.push_back(k);
            xPeer_.push_back(-1);
        }
    }
    llXStart_[K] = (int) xDist_.size();
    for (int k = 0; k < K; ++k) {
        const HostLaneLink &ll = laneLinks[k];
        const HostInter &in = inters[ll.inter];
        int e = llXStart_[k];
        for (int c : ll.crosses) {
            int side = in.crosses[c].ll[0] != k;
            xPeer_[e++] = entryOf[(size_t) (in.xBase + c) * 2 + (1 - side)];
        }
    }

    interVirtual_.resize(I);
    interNRoadLinks_.resize(I);
    interPhaseStart_.assign(I + 1, 0);
    interAvailStart_.resize(I);
    phaseTime_.clear();
    phaseAvail_.clear();
    for (int i = 0; i < I; ++i) {
        interVirtual_[i] = inters[i].isVirtual ? 1 : 0;
        interNRoadLinks_[i] = (int) inters[i].roadLinks.size();
        interPhaseStart_[i] = (int) phaseTime_.size();
        interAvailStart_[i] = (int) phaseAvail_.size();
        for (auto &ph : inters[i].phases) {
            phaseTime_.push_back(ph.time);
            phaseAvail_.insert(phaseAvail_.end(), ph.avail.begin(), ph.avail.end());
        }
    }
    interPhaseStart_[I] = (int) phaseTime_.size();

    // lane change: Lane::width and the segment count (Road::buildSegmentationByInterval roadnet.cpp:687-691 with the
    // interval of roadnet.cpp:310-312: (default len 5 + default minGap 2) * MAX_NUM_CARS_ON_SEGMENT 10, over the
    // road's own polyline)
    laneWidth_.resize(L);
    laneNumSegs_.resize(L);
    for (int l = 0; l < L; ++l) {
        laneWidth_[l] = lanes[l].width;
        const double interval = (5.0 + 2.0) * 10;
        laneNumSegs_[l] = (int32_t) std::max((size_t) std::ceil(polylineLength(roads[lanes[l].road].points) / interval), (size_t) 1);
    }

    flat_ = cfx_net{};
    flat_.n_roads = R;
    flat_.n_lanes = L;
    flat_.n_lanelinks = K;
    flat_.n_inters = I;
    flat_.n_xentries = (int) xDist_.size();
    flat_.n_phases = (int) phaseTime_.size();
    flat_.n_avail = (int) phaseAvail_.size();
    flat_.drv_length = drvLength_.data();
    flat_.drv_max_speed = drvMaxSpeed_.data();
    flat_.lane_road = laneRoad_.data();
    flat_.lane_index = laneIndex_.data();
    flat_.lane_ll_start = laneLLStart_.data();
    flat_.lane_ll = laneLL_.data();
    flat_.road_lane_start = roadLaneStart_.data();
    flat_.ll_start_lane = llStartLane_.data();
    flat_.ll_end_lane = llEndLane_.data();
    flat_.ll_inter = llInter_.data();
    flat_.ll_roadlink = llRoadLink_.data();
    flat_.ll_type = llType_.data();
    flat_.ll_x_start = llXStart_.data();
    flat_.x_dist = xDist_.data();
    flat_.x_peer = xPeer_.data();
    flat_.x_ll = xLL_.data();
    flat_.inter_virtual = interVirtual_.data();
    flat_.inter_n_roadlinks = interNRoadLinks_.data();
    flat_.inter_phase_start = interPhaseStart_.data();
    flat_.inter_avail_start = interAvailStart_.data();
    flat_.lane_width = laneWidth_.data();
    flat_.lane_n_segments = laneNumSegs_.data();
    flat_.phase_time = phaseTime_.data();
    flat_.phase_avail = phaseAvail_.data();
}

}  // namespace cfa
