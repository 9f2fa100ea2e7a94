// Device-side data model and per-vehicle arithmetic of the cfx HIP engine (gfx950).
//
// Layout: every running vehicle occupies one SLOT of a set of struct-of-arrays
// buffers; slots are ordered by (drivable, position in Drivable::vehicles), so
//   * a drivable's vehicles are the contiguous range [segStart[d], segStart[d] + cnt[d]),
//   * a vehicle's in-lane leader is simply slot-1 (adjacent => coalesced), and a vehicle is the head of
//     its drivable iff drv[slot-1] differs (no segStart gather needed),
//   * every lane segment carries ONE spare slot at its tail for this step's admission
//     (Engine::handleWaiting admits at most one vehicle per lane per step, engine.cpp:502-516).
// The order is rebuilt every step by a stable counting compaction (update-location phase).
//
// All arithmetic is FP64, compiled with -ffp-contract=off, and keeps the reference's operation order
// (reference file:line cited per function) so results are bit-identical to the x86 reference build.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cityflow_amd.h"

#define CFX_INT_MAX 2147483647

struct DevNet {
    int R, L, K, I, E;
    const double *drvLength, *drvMaxSpeed, *xDist, *phaseTime;
    const int32_t *laneRoad, *laneIndex, *laneLLStart, *laneLL, *llStartLane, *llEndLane, *llInter, *llRoadLink, *llType,
        *llXStart, *xPeer, *xLL, *interVirtual, *interNRL, *interPhaseStart, *interAvailStart;
    const uint8_t *phaseAvail;
    // derived at cfx_create (not part of the ABI)
    const int32_t *xPeerBit;        // [E] llLocal of the laneLink owning the peer entry
    const int32_t *llLocal;         // [K] index of the laneLink inside its intersection
    const int32_t *interMaskStart;  // [I+1] offsets (in 64-bit words) into the active-laneLink masks
    const double2 *drvLM;           // [D] {length, maxSpeed} packed: one 16-byte load per drivable
    const double2 *xDD;             // [E] {xDist of the entry, xDist of its peer entry}
    const int4 *xPack;              // [E] {peer laneLink, peer's llLocal bit, peer's RoadLinkType, peer's roadLink index}
    const int4 *llPack;             // [K] {first cross entry, end of cross entries, mask word base of its intersection, RoadLinkType}
    const int4 *laneLL4;            // [L] the lane's laneLinks (Lane::laneLinks order, -1 padded); x = -2: more than four, use the CSR
    const int4 *laneEnd4;           // [L] the end lanes of those laneLinks (same positions; -1 padded / unknown)
    // tiling (cfx_halo_config); both null for an engine that owns its whole network
    const uint8_t *laneGhost;       // [L] 1: lane owned by a neighbouring tile; its vehicles are frozen proxies
    const uint8_t *laneSpare;       // [L] spare slots behind the lane's vehicles (1 admission + halo migrants)
};

struct DevTables {
    const cfx_vehicle_template *templ;
    int nTempl;
    const int32_t *routeStart, *routeRoads, *nextStart, *nextLL;
};

// Committed per-slot state (double-buffered: rewritten in slot order by the compaction).
struct SlotArrays {
    int32_t *vid;       // -1: empty spare slot
    int32_t *drv;       // ControllerInfo::drivable (-1 in a spare slot)
    int32_t *prevDrv;   // ControllerInfo::prevDrivable (-1 none)
    int32_t *next;      // Router::getNextDrivable(0) for the current drivable, cached (-1 none)
    int32_t *blocker;   // ControllerInfo::blocker as a slot index of the PREVIOUS generation (-1 none;
                        // <= -2: -(vid + 2) of a proxy on a ghost lane, tiling only — ends blocker chains like -1);
                        // resolved through oldToNew[] (see blockerOf)
    int32_t *enterLLT;  // ControllerInfo::enterLaneLinkTime
    int32_t *routePos;  // Router::iCurRoad as index into the route
    int32_t *templ;     // vehicle template index
    int32_t *route;     // route index
    uint8_t *flags;     // bit 0: a custom speed is pending (Buffer::isCustomSpeedSet, vehicle.h:62); bit 2 (kFlagStateGap): the
                        // gap of this step's car following is the one the loaded state carried (vGapState), not the derived one
    double *dis;        // ControllerInfo::dis
    double *speed;      // VehicleInfo::speed
};

// Lane change (reference src/vehicle/lanechange.{h,cpp}, vehicle.h:74-79): per-vehicle state, sparse and rarely
// touched, so it lives in tables indexed by vid and the slot arrays / the compaction stay as they are.  `on == 0`
// unless the engine was created with cfx_config::lane_change.
#ifndef CFX_LC_ROAD_CAND
#define CFX_LC_ROAD_CAND 64
#endif
constexpr int kLcRoadCand = CFX_LC_ROAD_CAND;  // candidates per road kept in its list (more: the walk scans the road)
struct LcInsert {  // one shadow created in this step (Engine::insertShadow engine.cpp:812-820)
    int32_t parentVid, parentSlot, lane, recvFrom;  // recvFrom: signal the shadow received later in the same walk
    double dis;
    double gap;  // the parent's ControllerInfo::gap at the moment of the copy
    // where LaneChange::insertShadow puts it (lanechange.cpp:83-95): into the lane list right before its target follower
    // — `anchor` = index of the first EXISTING vehicle behind it (the lane's count if none), `seq` orders the shadows that
    // share an anchor — and into the target lane's segment with the PARENT's segment index
    int32_t anchor, seg;
    double seq;
};
struct LcDev {
    int on;
    const double *laneWidth;        // [L] Lane::width
    const int32_t *roadLaneStart;   // [R+1] lanes of a road are contiguous
    const int32_t *laneNumSegs;     // [L] Lane::segments.size()
    int32_t *segOfSlot;             // [slot capacity] Vehicle::segmentIndex as Lane::initSegments assigns it (lcInitSegments, k_admit)
    // LaneChangeInfo vehicle.h:74-79
    int8_t *ptype;                  // 0 none, 1 real vehicle of a changing pair, 2 shadow
    int32_t *partner;               // vid or -1
    double *offset;
    // LaneChange lanechange.h:27-44; signalSend = {present, target lane, urgency, direction}; of signalRecv only its
    // source is ever read
    int8_t *sigSend, *sendDir, *sendUrg, *lastDir, *changing, *lcFinished;
    int32_t *sendTarget, *recvFrom, *tLeader, *tFollower;
    double *leaderGap, *followerGap, *lastChangeTime;
    double *gap;                    // ControllerInfo::gap as stored state (makeSignal reads it without a leader)
    int32_t *slotOf;                // current slot of a running vehicle
    // vehicles whose step is finished by k_lc_resolve (changing pairs; vehicles signalled by an earlier changing vehicle):
    // their speed before the lane-change yield and their blocker, and the list of them (shadows go with their partner)
    double *bSpeed;
    int32_t *bBlocker;
    int32_t *parkList;              // [slot capacity]
    int32_t *parkIdx;               // [vid] index in parkList (valid for this step's parked real vehicles)
    int32_t *parkDep;               // [slot capacity] scratch of k_lc_resolve: the item each item has to wait for
    int32_t *parkCount;             // [4] parked items; of them, items that wait for another one; k_lc_resolve's ticket
    // neighbours that the schedule walk could only name provisionally (shadows of this very step): {vid, which, record}
    int32_t *fixList;               // [3 * fixCap]
    int32_t *fixCount;              // [1]
    int fixCap;
    // this step's scratch
    int32_t *roadCand;              // [R] candidates on the road (plan -> schedule)
    int2 *roadCandList;             // [R * kLcRoadCand] {vid, slot} of the first candidates of each road
    int32_t *candAll;               // [slot capacity] all candidates of the step
    int32_t *candAllEnv;            // [slot capacity] ... and the environment each belongs to (batched environments, below)
    int32_t *candAllCount;          // [1]
    int32_t *candPos;               // [vid] position of a candidate in the reference's walk (k_lc_order)
    int32_t *insHead, *insNext;     // [L] / [insCap] records of a target lane, linked
    LcInsert *ins;
    int32_t *insCount;              // [1]
    int insCap;
    int32_t *insKey;                // [insCap] walk position of the shadow's parent: shadows are numbered in that order
    int32_t *insLanes;              // [L] the lanes that get shadows in this step (k_lc_schedule -> k_lc_insert)
    int32_t *insLaneCount;          // [1]
    int32_t *newToOld;              // [slot capacity] inverse of oldToNew for the slots k_scatter filled
    const int32_t *pool;            // priorities the host's generator would hand out next (cfx_lane_change_supply)
    int firstShadowVid;
    // Batched environments (cfx_config::n_envs): environment of a road = road / roadsPerEnv (one environment: roadsPerEnv = R);
    // every environment has its own schedule walk (lcWalkPosition) and its own poolPerEnv priorities pool[env * poolPerEnv ...];
    // shadows are created environment by environment: LcDev::insKey = (environment << kLcEnvShift) + walk position
    int roadsPerEnv, poolPerEnv;
};
constexpr int kLcEnvShift = 20;

// Lane::history (roadnet.h:305-316) with cfx_config::lane_history: per lane a ring of kLaneHistoryMax {vehicle count, mean
// speed} records — stored record-major, [record][lane], so that the lanes' threads, which all write the same record number in a
// step, write adjacent words — and the two running aggregates.
constexpr int kLaneHistoryMax = CFX_LANE_HISTORY_MAX;
struct LaneHistDev {
    int32_t *num;    // [kLaneHistoryMax * L]
    double *avg;     // [kLaneHistoryMax * L]
    int32_t *head;   // [L] index of the oldest record
    int32_t *len;    // [L]
    int32_t *hNum;   // [L] Lane::historyVehicleNum
    double *hAvg;    // [L] Lane::historyAverageSpeed
    int L;
};
// Lane::updateHistory roadnet.cpp:900-915, expression for expression (the speeds summed in the lane's list order)
template <class SpeedAt>
__device__ inline void laneHistoryStep(const LaneHistDev &h, int lane, int n, SpeedAt speedAt) {
    int head = h.head[lane], len = h.len[lane], hn = h.hNum[lane];
    double speedSum = hn * h.hAvg[lane];
    while (len > 240) {
        const int fn = h.num[(size_t) head * h.L + lane];
        hn -= fn;
        speedSum -= fn * h.avg[(size_t) head * h.L + lane];
        head = head + 1 == kLaneHistoryMax ? 0 : head + 1;
        --len;
    }
    double curSpeedSum = 0;
    hn += n;
    // (the additions stay in list order — FP64 addition is not associative — but eight speeds are requested at a time: one
    // round trip to memory per eight vehicles instead of one per vehicle)
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = speedAt(i + k);
#pragma unroll
        for (int k = 0; k < 8; ++k) curSpeedSum += v[k];
    }
    {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k < n ? speedAt(i + k) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + k < n) curSpeedSum += v[k];
    }
    speedSum += curSpeedSum;
    int tail = head + len;
    if (tail >= kLaneHistoryMax) tail -= kLaneHistoryMax;
    h.num[(size_t) tail * h.L + lane] = n;
    h.avg[(size_t) tail * h.L + lane] = n ? curSpeedSum / n : 0;
    h.head[lane] = head;
    h.len[lane] = len + 1;
    h.hNum[lane] = hn;
    h.hAvg[lane] = hn ? speedSum / hn : 0;
}

// The last vehicle of a drivable, kept as ONE 32-byte record so that its readers — the leader search of every head of a
// drivable, Lane::canEnter, the admission check, the notify sources — do one load instead of a chain through
// {ring geometry, head, count} -> slot -> {dis, speed, template}.  `tag` is the step the record was written in: a record is
// the truth about the END of step `tag`, so "tag != step - 1" means nobody wrote one last step = the drivable was empty
// (every non-empty drivable's tail vehicle rewrites it every step; kr_commit does where the tail changed hands).
struct TailRec {
    double dis, speed;
    int32_t slot, templ, prevDrv, tag;
};
static_assert(sizeof(TailRec) == 32, "tail record layout");

// Per-laneLink notify sources beyond llDyn = {u, f, first vehicle on the laneLink, vehicles on it}: the state of u (the
// vehicle that just left onto the end lane) and f (the approaching vehicle on the start lane) and the two lengths the
// distances are measured with, so that a cross resolves "who was I notified of" from two records.
struct LLAux {
    double uDis, uSpeed, fDis, fSpeed, llLen, startLen;
    int32_t uTempl, fTempl;
};
static_assert(sizeof(LLAux) == 56, "laneLink aux record layout");

// What finishing a vehicle that leaves its drivable needs beyond its slot; valid = the caller requested it early
// (actionOneRing, round A), otherwise it is loaded here.
struct LeaverPrefetch {
    bool valid;
    double nextLen;  // length of the next drivable (nd0 >= 0)
    int vid, route, routePos;
    // ring layout, the LAST vehicle of its drivable (it rewrites the drivable's tail record): where it came from
    bool prevValid;
    int prevDrv;
    int templP1;  // the vehicle's template index + 1 where the caller holds it (0: read it from the slot)
    int blockerVidP2;  // the blocker's vehicle number + 2 where the caller holds it (0: read it from the blocker's slot)
};

// Everything the per-step kernels read.  Passed by value (kernel argument segment).
// Slot flag bits (SlotArrays::flags, the ring layout's meta.z): 1 = custom speed pending, 2 = on the last road of the route,
// 4 = the first step after cfx_load_state takes the vehicle's gap from the state (ControllerInfo::gap is stored state in the
// reference, include/cityflow_amd.h cfx_state::r_gap); like bit 0 it lives for one step.
constexpr int kFlagCustom = 1, kFlagLastRoad = 2, kFlagStateGap = 4;
#ifndef CFX_CROSS2_WAVES
#define CFX_CROSS2_WAVES 7
#endif
struct StepCtx {
    // wavefronts per SIMD k_cross2's register allocation leaves room for on this context (cfx_kernels.h; measured: 7 blocks
    // of the kernel per CU — what its LDS allows — instead of the 6 that 77 registers give: 58 -> 43-54 us at 1 M vehicles)
    static constexpr int kCross2Waves = CFX_CROSS2_WAVES;
    DevNet n;
    DevTables t;
    SlotArrays s;             // current generation
    const int32_t *segStart;  // [D+1]
    const int32_t *cnt;       // [D] live vehicles per drivable (spare excluded unless filled)
    const int32_t *admitStep; // [L] step index of the lane's latest admission
    int admissionsVisible;    // lane change, from k_action on: the leader search sees this step's admissions (lastSlotForLeader)
    const int32_t *curPhase;  // [I]
    const int32_t *oldToNew;  // slot of previous generation -> slot of current generation (-1 removed)
    const int32_t *vPriority; // [vid]
    const double *vCustomSpeed; // [vid] Buffer::customSpeed, valid where the slot flag / pending flag is set
    const double *vGapState;    // [vid] ControllerInfo::gap as cfx_load_state got it (cfx_state::r_gap), where flag bit 2 is set
    // per-laneLink notification sources of this step (phase 3, Engine::threadNotifyCross)
    // [K] {u: vehicle that just left onto the end lane (slot or -1), f: first vehicle of the start lane heading for
    //      this laneLink on green (slot or -1), segStart, cnt of the laneLink}: everything notifiedAt() needs
    int4 *llDyn;
    unsigned long long *interMask;  // active-laneLink bit masks, see DevNet::interMaskStart
    // per-step gate records written by k_admit (phase 2), read by k_action (phase 4)
    int2 *llGate;             // [K] {bit0 RoadLink::isAvailable, bits1-2 RoadLinkType, bit3 has crosses ; end lane}
    int32_t *laneTail;        // [L] Drivable::getLastVehicle() of the lane after this step's admission (slot or -1)
    int2 *admitRec;           // [L] {admitted vid, its successor in the lane's FIFO}: what k_scan needs to commit the pop
    // dense layout without lane change and tiling ("kd_" kernels, cfx_dense_kernels.h): the tail records of the ring layout
    // (two buffers by step parity + this step's view) and the wide gate records; null otherwise
    const TailRec *tailR;
    TailRec *tailW, *tailNow;
    int4 *llGate4;            // [K] {light | type | has crosses, end lane, first cross entry, end of cross entries}
    // cfx_config::dense_form bit 1 (kd_admit over the lanes only): a laneLink's gate record is rewritten only when its
    // intersection's phase has changed (kd_admit<true>), its tail as this step sees it is the committed record (linkTailNow)
    int laneAdmit;
    int32_t step;
    double interval;
    LcDev lc;
};

namespace cfxd {

// Developer build (-DCFX_TRACE): per-block wall-clock stamps (100 MHz) of ONE kernel's phases, chosen at build time with
// -DCFX_TRACE_KERNEL=<id> (0: the ring layout's kr_action / kr_cross as tools/trace_action.py reads them; 6 kd_action, 8 k_cross2, 9 k_lc_schedule, 10 kr_index, 11 kl_action;
// tools/trace_kernel.py); row = block index
#ifdef CFX_TRACE
__device__ long long *g_trace;  // [65536 * 8]
#ifndef CFX_TRACE_KERNEL
#define CFX_TRACE_KERNEL 0
#endif
// (-DCFX_TRACE_CYCLES: the shader clock instead, s_memtime — spans within a block only.  A kernel whose blocks all start
//  together queues up on the ONE wall clock: 1880 blocks stamping at once measured 10 us per stamp, round 5)
#ifdef CFX_TRACE_CYCLES
#define KCLOCK() clock64()
#else
#define KCLOCK() wall_clock64()
#endif
#define KSTAMP(id, k)                                                                                       \
    if (CFX_TRACE_KERNEL == (id) && threadIdx.x == 0 && blockIdx.x < 65536)                                  \
    g_trace[(size_t) blockIdx.x * 8 + (k)] = (long long) KCLOCK()
#define KNOTE(id, k, v)                                                                                     \
    if (CFX_TRACE_KERNEL == (id) && threadIdx.x == 0 && blockIdx.x < 65536) g_trace[(size_t) blockIdx.x * 8 + (k)] = (long long) (v)
#else
#define KSTAMP(id, k)
#define KNOTE(id, k, v)
#endif

__device__ __forceinline__ double min2(double x, double y) { return x < y ? x : y; }  // utility.h:70-72
__device__ __forceinline__ double max2(double x, double y) { return x > y ? x : y; }  // utility.h:66-68

// x86 cvttsd2si semantics for double -> int: out of range / NaN => INT_MIN.
__device__ __forceinline__ int d2i(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int) 0x80000000;
    return (int) x;
}

// One slot of a device-wide list for every lane of the wave that wants one: ONE atomic per wave on the list's counter (the
// lanes that are active here), not one per lane — thousands of same-address atomics in a kernel serialise in the L2.
// Returns the lane's index, -1 for a lane that did not ask.
__device__ __forceinline__ int waveListAppend(int32_t *counter, bool want) {
    const unsigned long long m = __ballot(want);
    if (m == 0ULL) return -1;
    const int lane = (int) (threadIdx.x & 63u), leader = __ffsll((long long) m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return want ? base + __popcll(m & ((1ULL << lane) - 1ULL)) : -1;
}

// New vehicle numbers start clean (LaneChange ctor lanechange.h:50, LaneChangeInfo vehicle.h:74-79)
__device__ __forceinline__ void lcInitVid(const LcDev &lc, int v) {
    lc.ptype[v] = 0;
    lc.partner[v] = -1;
    lc.offset[v] = 0.0;
    lc.sigSend[v] = 0;
    lc.sendDir[v] = 0;
    lc.sendUrg[v] = 0;
    lc.lastDir[v] = 0;
    lc.changing[v] = 0;
    lc.lcFinished[v] = 0;
    lc.sendTarget[v] = -1;
    lc.recvFrom[v] = -1;
    lc.tLeader[v] = -1;
    lc.tFollower[v] = -1;
    lc.leaderGap[v] = 0.0;
    lc.followerGap[v] = 0.0;
    lc.lastChangeTime[v] = 0.0;
    lc.gap[v] = 0.0;
    lc.slotOf[v] = -1;
}

// A step's few spawn records as kernel arguments of the admission kernel (kr_admit, cfx_ring_kernels.h, has the story)
constexpr int kAdmitRecs = 128;
// ... and kd_admit of a large network, whose stock flows alone produce a few hundred records per step (100x100: 400), takes
// up to kAdmitRecsBig of them the same way (24.6 KB of arguments with the firstNext column; the runtime takes 128 KB, tools/kernarg_probe.hip — but every block stages the
// lanes of ALL records, so more records per launch cost the kernel more than a k_spawn_link launch: round 6, profiles/r06_exp_*)
constexpr int kAdmitRecsBig = 1024;
template <int N> struct SpawnBatchT {
    int n, firstNewVid;
    double enterTime;
    int32_t lane[N], prevWait[N], route[N], priority[N];
    int32_t firstNext[N];  // VidTable::firstNext of the new vehicle (cfx_step looks it up in its host copy of the route tables)
    int16_t templ[N], vidOff[N];
};
using SpawnBatch = SpawnBatchT<kAdmitRecs>;
using SpawnBatchBig = SpawnBatchT<kAdmitRecsBig>;
// More records than the arguments hold (batched environments: 16 x 30x30 makes ~2 000 a step): the same columns, sorted by
// lane, in a pinned host buffer the admission kernel reads itself, and for every block of kBlock lanes where its lanes'
// records begin — a block stages its own few records, not everybody's (kr_admit; round 6: instead of k_spawn_link + a commit
// launch of its own)
struct SpawnBatchMem {
    int n, firstNewVid;
    double enterTime;
    const int32_t *lane, *prevWait, *route, *priority, *firstNext, *templ, *vidOff;
    const int32_t *blockOff;  // [nLaneBlocks + 1]; blockOff[0] = the records without a lane here (lane -1: rows only)
    int nLaneBlocks;
};

// Vehicles on drivable d as phases 3/4 see them.  cnt[] is the committed count; a lane's admission of THIS step
// (Engine::handleWaiting, phase 2) is not folded into it until the compaction (k_scan) — it is the flag
// admitStep[lane] == step plus the vehicle written into the lane's spare slot — so that nothing phase 2 writes is
// read by another lane's admission or by phases 3/4 of vehicles elsewhere.
__device__ __forceinline__ int cntNow(const StepCtx &c, int d) {
    return c.cnt[d] + ((d < c.n.L && c.admitStep[d] == c.step) ? 1 : 0);
}

// Drivable::getLastVehicle as every phase-3/4 reader sees it (this step's admission included).
__device__ __forceinline__ int lastSlot(const StepCtx &c, int d) {
    int n = cntNow(c, d);
    return n > 0 ? c.segStart[d] + n - 1 : -1;
}

// Drivable::getLastVehicle as the LEADER SEARCH saw it.  The reference evaluates leader/gap at the end
// of the previous step (engine.cpp:429-442), i.e. before this step's admissions, except for a vehicle
// admitted this step on lane B, whose search runs inside handleWaiting and therefore sees admissions on
// lanes A < B (engine.cpp:503,512).
// With lane change the reference runs the leader pass once more after the planning phase (engine.cpp:571-575): from then
// on — k_action, k_cross — every vehicle sees every admission of the step.
__device__ __forceinline__ int lastSlotForLeader(const StepCtx &c, int d, bool viewerNew, int viewerLane) {
    if (c.admissionsVisible) return lastSlot(c, d);
    int n = c.cnt[d];
    if (viewerNew && d < viewerLane && d < c.n.L && c.admitStep[d] == c.step) n += 1;
    return n > 0 ? c.segStart[d] + n - 1 : -1;
}

// RoadLink::isAvailable roadnet.h:429-431
template <class C> __device__ __forceinline__ bool llAvailable(const C &c, int k) {
    int in = c.n.llInter[k];
    return c.n.phaseAvail[c.n.interAvailStart[in] + c.curPhase[in] * c.n.interNRL[in] + c.n.llRoadLink[k]] != 0;
}
__device__ __forceinline__ bool typeIsTurn(int t) { return t == 1 || t == 2; }  // roadnet.h:433-435

// Router::getNextDrivable(const Drivable*) router.cpp:49-76 through the static per-route table.
__device__ __forceinline__ int nextOf(const DevNet &n, const DevTables &t, int d, int route, int routePos) {
    if (d >= n.L) return n.llEndLane[d - n.L];
    int road = n.laneRoad[d];
    int base = t.routeStart[route], len = t.routeStart[route + 1] - base;
    int p = routePos;
    while (p < len && t.routeRoads[base + p] != road) ++p;
    if (p >= len) return -1;
    int ll = t.nextLL[t.nextStart[base + p] + n.laneIndex[d]];
    return ll < 0 ? -1 : n.L + ll;
}

template <class C> __device__ __forceinline__ bool isLastRoad(const C &c, int d, int route) {  // router.cpp:131-134
    if (d >= c.n.L) return false;
    return c.n.laneRoad[d] == c.t.routeRoads[c.t.routeStart[route + 1] - 1];
}

// Bit 1 of a slot's flags: the vehicle is on the LAST road of its route (Router::isLastRoad) — kept where its next drivable
// is kept (set when it enters a lane with nothing behind it), so that Router::onValidLane (router.h:66-68) in every step's
// speed tail is a bit test instead of a walk route -> routeStart -> routeRoads.
template <class C> __device__ __forceinline__ int lastRoadBit(const C &c, int d, int route, int next) {
    return (next < 0 && isLastRoad(c, d, route)) ? 2 : 0;
}

// Layout accessors of the dense layout (the ring layout overloads them on its own context, cfx_ring_kernels.h)
__device__ __forceinline__ int committedCount(const StepCtx &c, int d) { return c.cnt[d]; }
__device__ __forceinline__ int firstSlot(const StepCtx &c, int d) { return c.segStart[d]; }  // Drivable::getFirstVehicle
__device__ __forceinline__ int slotAhead(const StepCtx &, int, int s) { return s - 1; }      // the in-drivable leader's slot
struct SegWalk {  // slots of one drivable from a starting slot towards the tail
    int first, base, mask;
    __device__ __forceinline__ int at(int i) const { return mask < 0 ? first + i : base + ((first - base + i) & mask); }
};
__device__ __forceinline__ SegWalk segWalk(const StepCtx &, int, int first) { return SegWalk{first, 0, -1}; }

// "The last vehicle of a drivable" as its readers need it: the head-of-drivable leader search, Lane::canEnter and the
// notify sources of a laneLink all look at a drivable's tail {slot, template, where it came from, dis, speed}.  The dense
// layout gathers these through the slot; the ring layout keeps a 32-byte record per drivable (cfx_ring_kernels.h).
struct Tail {
    int slot, templ, prevDrv;  // slot < 0: the drivable is empty
    double dis, speed;
};
__device__ __forceinline__ Tail tailAt(const StepCtx &c, int slot) {
    Tail t{slot, 0, -1, 0.0, 0.0};
    if (slot >= 0) {
        t.templ = c.s.templ[slot];
        t.prevDrv = c.s.prevDrv[slot];
        t.dis = c.s.dis[slot];
        t.speed = c.s.speed[slot];
    }
    return t;
}
// Drivable::getLastVehicle as phases 3 / 4 see it (this step's admission included)
__device__ __forceinline__ Tail tailNowOf(const StepCtx &c, int d) { return tailAt(c, d < c.n.L ? c.laneTail[d] : lastSlot(c, d)); }
// ... and as the leader search saw it (lastSlotForLeader)
__device__ __forceinline__ Tail tailForLeader(const StepCtx &c, int d, bool viewerNew, int viewerLane) {
    return tailAt(c, lastSlotForLeader(c, d, viewerNew, viewerLane));
}

__device__ __forceinline__ int4 gateRecord(const StepCtx &c, int k) { return c.llGate4[k]; }  // (cfx_dense_kernels.h)
// Drivable::getLastVehicle of laneLink d (an index >= L) as this step's phases 3 / 4 see it, as a tail record.  Nothing is admitted
// onto a laneLink, so it is the committed record with the "written last step" test — which is all kd_admit did for these
// records; with cfx_config::dense_form bit 1 that kernel no longer visits the laneLinks
__device__ __forceinline__ TailRec linkTailNow(const StepCtx &c, int d) {
    if (!c.laneAdmit) return c.tailNow[d];
    TailRec r = c.tailR[d];
    if (r.tag != c.step - 1) r.slot = -1;
    return r;
}
__device__ __forceinline__ TailRec firstHopRecord(const StepCtx &c, bool linkHead, int endLane, int firstLink) {  // (actionOneRounds)
    TailRec r = *(linkHead ? &c.tailR[endLane] : (c.laneAdmit ? &c.tailR[firstLink] : &c.tailNow[firstLink]));
    if (!linkHead && c.laneAdmit && r.tag != c.step - 1) r.slot = -1;
    return r;
}
__device__ __forceinline__ Tail tailOfRec(const TailRec &r) { return Tail{r.slot, r.templ, r.prevDrv, r.dis, r.speed}; }
__device__ __forceinline__ Tail tailIfCurrent(const TailRec &r, int wantTag) {
    Tail t = tailOfRec(r);
    if (r.tag != wantTag) t.slot = -1;
    return t;
}

// ControllerInfo::blocker of the vehicle in `slot`, as a current-generation slot (-1 none).
__device__ __forceinline__ int blockerOf(const StepCtx &c, int slot) {
    int b = c.s.blocker[slot];
    return b >= 0 ? c.oldToNew[b] : -1;
}

// Vehicle::getNoCollisionSpeed vehicle.cpp:200-209
__device__ __forceinline__ double noCollisionSpeed(double vL, double dL, double vF, double dF, double gap,
                                                   double interval, double targetGap) {
    double cc = vF * interval / 2 + targetGap - 0.5 * vL * vL / dL - gap;
    double a = 0.5 / dF;
    double b = 0.5 * interval;
    if (b * b < 4 * a * cc) return -100;
    double v1 = 0.5 / a * (sqrt(b * b - 4 * a * cc) - b);
    double v2 = 2 * vL - dL * interval + 2 * (gap - targetGap) / interval;
    return min2(v1, v2);
}

// The few per-vehicle values the intersection logic needs about "a vehicle" (self or foe).
struct VehRef {
    double speed;
    const cfx_vehicle_template *t;
};

__device__ __forceinline__ double minBrakeDistance(const VehRef &v) {  // vehicle.h:239
    return 0.5 * v.speed * v.speed / v.t->max_neg_acc;
}

// Vehicle::getBrakeDistanceAfterAccel vehicle.cpp:302-306
__device__ __forceinline__ double brakeDistanceAfterAccel(const VehRef &v, double acc, double dec, double interval) {
    double currentSpeed = v.speed;
    double nextSpeed = currentSpeed + acc * interval;
    return (currentSpeed + nextSpeed) * interval / 2 + (nextSpeed * nextSpeed / dec / 2);
}

// Vehicle::getStopBeforeSpeed vehicle.cpp:240-250
__device__ __forceinline__ double stopBeforeSpeed(const VehRef &v, double distance, double interval) {
    if (brakeDistanceAfterAccel(v, v.t->usual_pos_acc, v.t->usual_neg_acc, interval) < distance)
        return v.speed + v.t->usual_pos_acc * interval;
    double takeInterval = 2 * distance / (v.speed + 1e-8) / interval;
    if (takeInterval >= 1) {
        return v.speed - v.speed / d2i(takeInterval);
    } else {
        return v.speed - v.speed / takeInterval;
    }
}

// Vehicle::getDistanceUntilSpeed vehicle.cpp:275-282
__device__ __forceinline__ double distanceUntilSpeed(const VehRef &v, double speed, double acc, double interval) {
    if (speed <= v.speed) return 0;
    int stage1steps = d2i(floor((speed - v.speed) / acc / interval));
    double stage1speed = v.speed + stage1steps * acc / interval;
    double stage1dis = (v.speed + stage1speed) * (stage1steps * interval) / 2;
    return stage1dis + (stage1speed < speed ? ((stage1speed + speed) * interval / 2) : 0);
}

// Vehicle::getReachSteps vehicle.cpp:252-268
__device__ __forceinline__ int reachSteps(const VehRef &v, double distance, double targetSpeed, double acc,
                                          double interval) {
    if (distance <= 0) return 0;
    if (v.speed > targetSpeed) return d2i(ceil(distance / v.speed));
    double distanceUntilTargetSpeed = distanceUntilSpeed(v, targetSpeed, acc, interval);
    if (distanceUntilTargetSpeed > distance) {
        return d2i(ceil((sqrt(v.speed * v.speed + 2 * acc * distance) - v.speed) / acc / interval));
    } else {
        return d2i(ceil((targetSpeed - v.speed) / acc / interval) +
                   ceil((distance - distanceUntilTargetSpeed) / targetSpeed / interval));
    }
}

// Vehicle::getReachStepsOnLaneLink vehicle.cpp:270-273
__device__ __forceinline__ int reachStepsOnLaneLink(const VehRef &v, double distance, int llType, double interval) {
    return reachSteps(v, distance, typeIsTurn(llType) ? v.t->turn_speed : v.t->max_speed, v.t->usual_pos_acc, interval);
}

// Vehicle::canYield vehicle.cpp:284-287
__device__ __forceinline__ bool canYield(const VehRef &v, double dist) {
    return (dist > 0 && minBrakeDistance(v) < dist - v.t->yield_distance) || (dist < 0 && dist + v.t->len < 0);
}

}  // namespace cfxd
