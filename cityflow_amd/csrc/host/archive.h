// Archive: snapshot / restore of the whole simulation state (reference src/engine/archive.{h,cpp}).
//   Engine.snapshot() -> Archive           in-memory deep copy (Archive(const Engine&) archive.cpp:9-37)
//   Engine.load(archive)                    Archive::resume archive.cpp:73-126
//   Archive.dump(path)                      JSON in the reference's own format (archive.cpp:153-343), so dumps
//   Engine.load_from_file(path)             travel in both directions between this engine and the reference
// The device part is read with the cfx getters and written back with cfx_load_state (include/cityflow_amd.h).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "flow.h"
#include "roadnet.h"

namespace cfa {

struct DeviceState {
    int64_t step = 0, finished = 0, vehicleSteps = 0;
    double cumulativeTravelTime = 0;
    std::vector<uint8_t> vState;  // per vid
    // running vehicles in Drivable::vehicles order
    std::vector<int32_t> rVid, rDrivable, rPrevDrivable, rBlocker, rEnterLLTime, rRoutePos, rLeader;
    std::vector<double> rDis, rSpeed, rGap, rCustomSpeed;
    // lane change (empty without it): the part of LaneChange / LaneChangeInfo that outlives a step
    std::vector<int32_t> rLcPartner, rLcLastDir, rLcTarget, rLcDirection;
    std::vector<uint8_t> rLcFlags;
    std::vector<double> rLcOffset, rLcLastChangeTime, rLcWaitingTime;
    std::vector<int32_t> wVid, wLane;  // waiting buffers, lane by lane
    std::vector<int32_t> tlPhase;
    std::vector<double> tlRemain;
    // Lane::history (cfx_lane_history, include/cityflow_amd.h); all empty where it is not kept ("cfx": {"laneHistory": true})
    std::vector<int32_t> hLen, hVehicleNum, hHistoryVehicleNum;  // [L], [L * CFX_LANE_HISTORY_MAX], [L]
    std::vector<double> hAverageSpeed, hHistoryAverageSpeed;
};

class Archive {
public:
    Spawner::State host;
    DeviceState dev;
    // immutable context needed to write ids / templates / routes
    std::shared_ptr<const HostRoadNet> net;
    std::vector<cfx_vehicle_template> templates;
    std::vector<int32_t> routeStart, routeRoads;
    std::vector<std::string> flowIds;

    void dump(const std::string &path) const;
    // numbers of the last dump() that NO decimal literal makes the reference's reader return exactly (about one double in 10^5:
    // archive.cpp num()); they are written with 17 digits — exact for a correctly rounding reader — and reported on stderr
    mutable long lastDumpInexact = 0;
    std::string vehicleId(int vid) const;
};

// Archive(Engine&, filename) archive.cpp:345-550: the reference's JSON format -> Archive.  Interns the vehicle templates
// and routes the file needs in `spawner` (their indices are what the archive's vehicle table refers to).
// The literal Archive::dump writes for a double: one that a correctly rounding reader AND the reference's (rapidjson's
// default number reader, json_number.h) both turn back into exactly `v`.
std::string formatJsonNumber(double v);
long inexactJsonNumbers();  // how many numbers formatted on this thread so far had no such literal (see Archive::lastDumpInexact)

Archive readArchiveFile(const std::string &path, const std::shared_ptr<HostRoadNet> &net, Spawner &spawner, bool laneChange);

}  // namespace cfa
