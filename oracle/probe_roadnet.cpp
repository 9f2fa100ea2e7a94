// TEST INFRASTRUCTURE ONLY (oracle/).
//
// Small driver around the reference's OWN RoadNet loader (reference: src/roadnet/roadnet.cpp:42-325,
// Road::initLanesPoints 456-505, Intersection::initCrosses 515-576).  It links against the unmodified
// reference objects (oracle/Makefile) and prints the load-time geometry the hot path depends on, so the
// product's from-scratch host loader (cityflow_amd/csrc/host) can be pinned against it:
//
//   L <laneId> <length> <maxSpeed> <width> <nLaneLinks>
//   K <laneLinkId> <length> <roadLinkType> <nCrosses>
//   X <distanceOnThisLaneLink> <peerLaneLinkId> <distanceOnPeer>        (nCrosses lines, ascending)
//   T <intersectionId> <virtual> <nPhases> [<time> <mask-as-01-string>]...
//
// Doubles are printed with %.17g (round-trip exact).
#include "roadnet/roadnet.h"

#include <cstdio>

using namespace CityFlow;

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s roadnet.json\n", argv[0]);
        return 2;
    }
    RoadNet net;
    if (!net.loadFromJson(argv[1])) return 1;
    for (const Lane *lane : net.getLanes()) {
        printf("L %s %.17g %.17g %.17g %zu\n", lane->getId().c_str(), lane->getLength(), lane->getMaxSpeed(),
               lane->getWidth(), lane->getLaneLinks().size());
    }
    for (LaneLink *ll : net.getLaneLinks()) {
        printf("K %s %.17g %d %zu\n", ll->getId().c_str(), ll->getLength(), (int) ll->getRoadLinkType(),
               ll->getCrosses().size());
        for (Cross *c : ll->getCrosses()) {
            LaneLink *peer = c->getLaneLink(0) == ll ? c->getLaneLink(1) : c->getLaneLink(0);
            printf("X %.17g %s %.17g\n", c->getDistanceByLane(ll), peer->getId().c_str(), c->getDistanceByLane(peer));
        }
    }
    for (Intersection &inter : net.getIntersections()) {
        auto &phases = inter.getTrafficLight().getPhases();
        printf("T %s %d %zu\n", inter.getId().c_str(), (int) inter.isVirtualIntersection(), phases.size());
    }
    return 0;
}
