// Host-side road network: loads the reference's roadnet JSON unchanged and flattens it into the
// index-based arrays of cfx_net (include/cityflow_amd.h).  Load-time geometry is on the parity path
// (lane / laneLink lengths and cross positions feed the step), so every formula below follows the
// reference's operation order exactly; each function cites the reference lines it restates.
#pragma once

#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "cityflow_amd.h"

namespace cfa {

struct Pt {
    double x = 0.0, y = 0.0;
};

struct HostRoad {
    std::string id;
    int startInter = -1, endInter = -1;
    std::vector<Pt> points;
    int laneStart = 0, nLanes = 0;
};

struct HostLane {
    int road = -1, index = 0;
    double width = 0, maxSpeed = 0, length = 0;
    std::vector<Pt> points;
    std::vector<int> laneLinks;  // laneLink ids, reference Lane::laneLinks order
};

struct HostCross {
    int ll[2];
    double dist[2];
};

struct HostLaneLink {
    int inter = -1, roadLink = -1 /*index inside the intersection*/, type = 0;
    int startLane = -1, endLane = -1;
    double length = 0;
    std::vector<Pt> points;
    std::vector<int> crosses;  // indices into HostInter::crosses, sorted like the reference
};

struct HostRoadLink {
    int type = 0, startRoad = -1, endRoad = -1;
    int llStart = 0, nLaneLinks = 0;  // global laneLink id range
};

struct HostPhase {
    double time = 0;
    std::vector<uint8_t> avail;
};

struct HostInter {
    std::string id;
    bool isVirtual = false;
    double width = 0.0;
    Pt point;
    std::vector<int> roads;
    std::vector<HostRoadLink> roadLinks;
    std::vector<HostPhase> phases;
    std::vector<HostCross> crosses;
    int xBase = 0;  // first global cross index of this intersection
};

class HostRoadNet {
public:
    std::vector<HostRoad> roads;
    std::vector<HostLane> lanes;          // reference RoadNet::lanes order
    std::vector<HostLaneLink> laneLinks;  // reference RoadNet::laneLinks order
    std::vector<HostInter> inters;
    std::map<std::string, int> roadIndex, interIndex;

    // Throws JsonError on malformed input (the Engine wrapper turns that into the reference's
    // "load config failed" behaviour).
    void load(const std::string &path);

    std::string laneId(int lane) const { return roads[lanes[lane].road].id + "_" + std::to_string(lanes[lane].index); }
    std::string laneLinkId(int ll) const {
        return laneId(laneLinks[ll].startLane) + "_TO_" + laneId(laneLinks[ll].endLane);
    }
    std::string drivableId(int drv) const {
        return drv < (int) lanes.size() ? laneId(drv) : laneLinkId(drv - (int) lanes.size());
    }

    // laneLinks of `lane` whose end lane belongs to `road` (reference Lane::getLaneLinksToRoad roadnet.cpp:447-454)
    std::vector<int> laneLinksToRoad(int lane, int road) const;
    bool connectedToRoad(int from, int to) const;  // Road::connectedToRoad roadnet.cpp:736-742
    double averageLength(int road) const;          // Road::averageLength roadnet.cpp:709-717

    // Flat views; valid while *this is alive and unchanged.
    const cfx_net &flat() const { return flat_; }

private:
    void initLanesPoints(int road);  // Road::initLanesPoints roadnet.cpp:456-505
    void initCrosses(int inter);     // Intersection::initCrosses roadnet.cpp:515-576
    void flatten();

    cfx_net flat_{};
    std::vector<double> drvLength_, drvMaxSpeed_, xDist_, phaseTime_, laneWidth_;
    std::vector<int32_t> laneRoad_, laneIndex_, laneLLStart_, laneLL_, roadLaneStart_, llStartLane_, llEndLane_,
        llInter_, llRoadLink_, llType_, llXStart_, xPeer_, xLL_, interVirtual_, interNRoadLinks_, interPhaseStart_,
        interAvailStart_, laneNumSegs_;
    std::vector<uint8_t> phaseAvail_;
};

}  // namespace cfa
