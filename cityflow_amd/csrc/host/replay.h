// Replay logging for the reference's web frontend: the static roadnet log written once
// (RoadNet::convertToJson roadnet.cpp:327-394 + Intersection::getOutline 750-818) and one text line per step
// (Engine::updateLog engine.cpp:518-554).  Numbers are printed in their shortest round-trip form (std::to_chars);
// the reference uses the Grisu2 printer of its bundled dtoa_milo.h, whose output also round-trips, so the two files
// parse to identical values (tests/test_replay.py) without being byte-identical.
#pragma once

#include <fstream>
#include <string>
#include <vector>

#include "flow.h"
#include "roadnet.h"

namespace cfa {

struct VehicleSnapshot;

std::vector<Pt> intersectionOutline(const HostRoadNet &net, int inter);  // Intersection::getOutline
bool writeRoadnetLog(const HostRoadNet &net, const std::string &path);   // false if the file cannot be written

class ReplayWriter {
public:
    bool open(const std::string &path);
    void close();
    bool isOpen() const { return out_.is_open(); }
    // `phase` = current phase index per intersection; vehicles of `s` in any order (sorted here by priority, the order of
    // Engine::getRunningVehicles = vehiclePool order)
    void writeStep(const HostRoadNet &net, const Spawner &sp, const VehicleSnapshot &s, const std::vector<int32_t> &phase);

private:
    std::ofstream out_;
    std::string line_;
    std::vector<std::pair<int32_t, int32_t>> order_;
};

}  // namespace cfa
