// Per-step kernels of the cfx HIP engine.  One reference step (Engine::nextStep engine.cpp:566-594) is
//   k_spawn_link   phase 0/1 tail : append host-produced spawn records to lanes' waiting queues
//   k_admit        phase 2        : Engine::handleWaiting, one thread per lane
//   k_action       phase 4 + 5a   : leader/gap + Engine::vehicleControl, one thread per slot, up to the walk over
//                                   the crosses; classification stay / move / finish and per-drivable counts.
//                  phase 3        : its trailing blocks do the per-laneLink half of Engine::threadNotifyCross
//                                   (who may notify; sets the intersection's active-laneLink mask) for k_cross
//   k_cross        phase 4 cont.  : Cross::canPass for the queued vehicles, one 16-lane group per vehicle and
//                                   one cross per lane; the other half of threadNotifyCross (which vehicle a
//                                   given cross sees) is resolved on demand, only where the peer laneLink is active
//   k_scan         phase 5b       : new segment offsets (single-pass exclusive scan over drivables)
//   k_scatter      phase 5c/6/8   : stable compaction into the next generation = commit (Vehicle::update);
//                                   TrafficLight::passTime; the step's finish statistics (one block)
// Leader/gap (phase 7, engine.cpp:429-442) needs no kernel of its own: it is a pure function of the
// post-compaction order and is evaluated at the top of the next step's k_action (see lastSlotForLeader).
#pragma once

#include "cfx_device.h"

namespace cfxd {

constexpr int kBlock = 256;
#ifndef CFX_ACT_BLOCK
#define CFX_ACT_BLOCK 64
#endif
#ifndef CFX_CROSS_BLOCK
#define CFX_CROSS_BLOCK 256
#endif
constexpr int kActBlock = CFX_ACT_BLOCK;      // one wavefront per workgroup: ~100k vehicles spread over all 1024 SIMDs
constexpr int kCrossBlock = CFX_CROSS_BLOCK;  // k_cross workgroup (16-lane groups inside)
#ifndef CFX_LDS_TEMPL
#define CFX_LDS_TEMPL 32
#endif
constexpr int kLdsTempl = CFX_LDS_TEMPL;   // vehicle templates staged in LDS by k_action (104 B each)
// Wavefronts per SIMD the register allocation of a kernel is asked to leave room for (second argument of __launch_bounds__;
// 0 = whatever the compiler arrives at).  Build-time knobs for A/B runs (tools/exp_bench.py with differently built
// libraries); the defaults are what was measured best.
#ifndef CFX_SCATTER_WAVES
#define CFX_SCATTER_WAVES 0
#endif
#ifndef CFX_SCAN_WAVES
#define CFX_SCAN_WAVES 0
#endif
#if CFX_SCAN_WAVES > 0
#define CFX_SCAN_BOUNDS __launch_bounds__(kBlock, CFX_SCAN_WAVES)
#else
#define CFX_SCAN_BOUNDS __launch_bounds__(kBlock)
#endif
#if CFX_SCATTER_WAVES > 0
#define CFX_SCATTER_BOUNDS __launch_bounds__(kBlock, CFX_SCATTER_WAVES)
#else
#define CFX_SCATTER_BOUNDS
#endif

// ----------------------------------------------------------------------------------------------
// Vehicle table (indexed by vid, never permuted)
struct VidTable {
    int32_t *priority, *templ, *route, *nextWait;
    double *enterTime;
    double *customSpeed;     // Buffer::customSpeed (set_vehicle_speed)
    double *gapState;        // ControllerInfo::gap of the loaded state (StepCtx::vGapState)
    uint8_t *state;          // 0 waiting, 1 running, 2 finished
    uint8_t *pendingCustom;  // custom speed set while the vehicle was still waiting
    // Router::getNextDrivable(0) of a WAITING vehicle on the lane it waits on, known when it is created (the host holds the
    // route tables: cfx_step looks it up for every spawn record) — the admission then needs no walk route -> first road ->
    // row -> laneLink behind its decision: >= 0 the drivable, -1 none, -2 none and the lane is on the route's last road
    // (Router::isLastRoad: slot flag bit 1), kFirstNextUnknown: take the walk (after a load, a new route, a k_spawn_link batch)
    int32_t *firstNext;
};
constexpr int kFirstNextUnknown = -3;
// the next drivable and the last-road flag of a vehicle that is admitted onto `lane` now
template <class C>
__device__ __forceinline__ void admittedNext(const C &c, int fn, int lane, int road, int laneIdx, int route, int *next, int *onLast) {
    if (fn != kFirstNextUnknown) {
        *next = fn >= 0 ? fn : -1;
        *onLast = fn == -2 ? 2 : 0;
        return;
    }
    const int base = c.t.routeStart[route];
    if (c.t.routeRoads[base] == road) {  // the lane is on route position 0: the table row is known without walking the route
        const int ll = c.t.nextLL[c.t.nextStart[base] + laneIdx];
        *next = ll < 0 ? -1 : c.n.L + ll;
    } else {
        *next = nextOf(c.n, c.t, lane, route, 0);
    }
    *onLast = (*next < 0 && isLastRoad(c, lane, route)) ? 2 : 0;
}

struct DevScalars {
    long long active;          // Engine::activeVehicleCount
    long long finishedCnt;     // Engine::finishedVehicleCnt
    double cumulativeTravelTime;
    long long vehicleSteps;    // sum over steps of vehicles that ran phase 4
    int nFinishedStep;         // finished vehicles of the step in flight
    int overflow;              // set when an internal capacity was exceeded
    int nCrossJobs;            // vehicles queued for k_cross in the step in flight
    int nLeftUncounted;        // lane change: real vehicles of completed changes that left this step (not "finished")
    int ringNearFull;          // ring layout: some drivable's ring is within 8 vehicles of its capacity (sticky)
    int actionMaxT;            // ring layout: most vehicles one block of the action kernel had (blocks above 3/4 of a pass report)
    long long tieEvents;       // cfx_scalars::tie_events
    int tieDrv[8];             // cfx_scalars::tie_drivables (event i at index i % 8)
    // ring layout: vehicles admitted by the step of that parity, folded into `active` by that step's commit (the admission of
    // step t + 1 may run in the same launch as the commit of step t, which reads and rewrites `active`)
    long long admitPending[2];
};

struct HostMirror {  // pinned host copy of the end-of-step scalars (written by k_scatter's statistics block)
    DevScalars sc;
    int32_t slots;  // segStart[D] of the new generation
    int32_t pad;
    // (steps completed << 32) | running vehicles, one 8-byte store: the host may read it WITHOUT synchronising (a
    // possibly stale but never torn lower bound of progress) to refresh its slot-capacity bound
    unsigned long long progress;
};

// Per-drivable scratch of the compaction.
struct CompactScratch {
    int32_t *leaveCnt;     // [D] vehicles leaving (moved or finished)
    int32_t *maxLeaveIdx;  // [D] largest in-segment index among leavers (-1 none)
    int32_t *inCnt;        // [D] vehicles entering
    int32_t *inHead;       // [D] head of the linked list of entering slots (-1 none)
    int32_t *inNext;       // [slot] next entering slot of the same target
};

// Vehicles whose cross checks are done by k_cross: kJobShards independent queues (shard = block index & 15),
// counters one cache line apart.
constexpr int kJobShards = 16;
constexpr int kJobShardStride = 32;  // ints between two counters (128 B)
// The step's finished vehicles are listed the same way: kFinShards lists (shard = block index & 15), each with its own
// counter one cache line from the next.  With ONE list, the ~5 000 vehicles that finish in a step of a 1 M-vehicle network
// queue up behind a single word of the L2 (a returning atomic each, ~100-200 per microsecond and word) — tens of
// microseconds inside kernels that take 50.  The statistics blocks read the shards as one list through FinMap.
constexpr int kFinShards = 16;
struct FinMap {  // (in LDS) running totals of the shards' counts: list position i lives in the shard whose total first exceeds i
    int end[kFinShards];
};
// all threads of the block; returns the number of finishers
__device__ inline int finMapLoad(FinMap &m, const int32_t *finCount, int finCap) {
    if (threadIdx.x == 0) {
        const int capShard = finCap / kFinShards;
        int run = 0;
        for (int i = 0; i < kFinShards; ++i) {
            run += min(finCount[i * 32], capShard);
            m.end[i] = run;
        }
    }
    __syncthreads();
    return m.end[kFinShards - 1];
}
__device__ __forceinline__ int finAt(const FinMap &m, int finCap, int i) {  // where list position i is stored
    int sh = 0;
    while (i >= m.end[sh]) ++sh;
    return sh * (finCap / kFinShards) + (i - (sh ? m.end[sh - 1] : 0));
}
__device__ __forceinline__ void finCountsClear(int32_t *finCount) {
    for (int i = 0; i < kFinShards; ++i) finCount[i * 32] = 0;
}
// a place in the block's shard for a vehicle that finishes now (-1: the shard is full)
__device__ __forceinline__ int finPlace(int32_t *finCount, int finCap) {
    const int shard = blockIdx.x & (kFinShards - 1), capShard = finCap / kFinShards;
    const int idx = atomicAdd(&finCount[shard * 32], 1);
    return idx < capShard ? shard * capShard + idx : -1;
}

struct JobQueue {
    int32_t *count;    // [kJobShards * kJobShardStride]
    int32_t *jobs;     // [kJobShards * capacity]
    int capacity;      // per shard
    int32_t *overflow; // DevScalars::overflow (a shard ran out of room: code 9)
};

// Buffered (not yet committed) results of k_action: Vehicle::Buffer vehicle.h:54-72
struct ActionBuf {
    double *dis, *speed;
    int32_t *drv;      // -1 unchanged, -2 end of route, >= 0 new drivable
    int32_t *blocker;  // slot (current generation) or -1
};

// ----------------------------------------------------------------------------------------------
__global__ void k_spawn_link(const cfx_spawn *recs, int n, int firstNewVid, VidTable vt, int32_t *waitHead, LcDev lc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cfx_spawn r = recs[i];
    if (lc.on) lcInitVid(lc, r.vid);
    vt.priority[r.vid] = r.priority;
    vt.templ[r.vid] = r.templ;
    vt.route[r.vid] = r.route;
    vt.enterTime[r.vid] = r.enter_time;
    vt.state[r.vid] = 0;
    vt.pendingCustom[r.vid] = 0;
    vt.firstNext[r.vid] = kFirstNextUnknown;  // (this path is the rare one: the admission walks the route)
    if (r.lane < 0) return;  // tiling: the vehicle starts in another tile; only its static record is kept here
    // FIFO append (Lane::pushWaitingVehicle roadnet.h:365-367).  nextWait[] was pre-set to -1.
    if (r.prev_wait < 0) {
        waitHead[r.lane] = r.vid;
    } else if (r.prev_wait >= firstNewVid) {
        vt.nextWait[r.prev_wait] = r.vid;  // predecessor is in this very batch: certainly still queued
    } else if (vt.state[r.prev_wait] != 0) {
        waitHead[r.lane] = r.vid;  // predecessor already admitted => the FIFO is empty
    } else {
        vt.nextWait[r.prev_wait] = r.vid;
    }
}

// Lane::initSegments roadnet.cpp:863-875 (lane change only; the start of Engine::nextStep's planning phase, engine.cpp:571)
// walks a lane front to back and gives every vehicle the highest segment whose start (Lane::startPos roadnet.cpp:859) it has
// reached — as long as the vehicles in front of it did: segment(it) = min(segment(it - 1), highest i with start_i <= dis).
// The second term depends on the vehicle alone: whoever puts a vehicle into a lane slot leaves it in segOfSlot (k_scatter for
// the vehicles that stay in the network, 0 for an admission, k_lc_naive after a load), and the lane's thread in k_admit only
// takes the running minimum — no FP64 division and no dependent load in that walk (it was 12 us as a walk over distances).
__device__ __forceinline__ int lcNaiveSegment(const StepCtx &c, int lane, double dis) {
    const int nSeg = c.lc.laneNumSegs[lane];
    const double len = c.n.drvLength[lane];
    int i = (int) (dis * nSeg / len);  // an estimate; the comparisons below are the reference's own expression
    i = i < 0 ? 0 : (i > nSeg - 1 ? nSeg - 1 : i);
    while (i + 1 < nSeg && (i + 1) * len / nSeg <= dis) ++i;
    while (i > 0 && i * len / nSeg > dis) --i;
    return i;
}
__device__ inline void lcInitSegments(const StepCtx &c, int base, int n, bool admitted) {
    int32_t *seg = c.lc.segOfSlot + base;
    int run = CFX_INT_MAX;
    for (int it0 = 0; it0 < n; it0 += 8) {
        int v[8];
        for (int q = 0; q < 8; ++q) v[q] = it0 + q < n ? seg[it0 + q] : CFX_INT_MAX;
        for (int q = 0; q < 8; ++q)
            if (it0 + q < n) {
                run = run < v[q] ? run : v[q];
                seg[it0 + q] = run;
            }
    }
    if (admitted) seg[n] = 0;  // distance 0: segment 0, whatever is in front of it
}
// ... after cfx_load_state (and a reset): one thread per slot
__global__ void k_lc_naive(StepCtx c) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride) {
        const int d = c.s.drv[s];
        if (c.s.vid[s] >= 0 && d >= 0 && d < c.n.L) c.lc.segOfSlot[s] = lcNaiveSegment(c, d, c.s.dis[s]);
    }
}

// Engine::handleWaiting engine.cpp:502-516 + Lane::available roadnet.cpp:428-435 (engines with lane change: the others
// run kr_admit / kd_admit).  A step's few spawn records travel in the kernel arguments (SpawnBatch: kr_admit has the story):
// each lane's thread links its own records into its waiting queue, block 0 writes the vehicle table — no k_spawn_link launch.
__global__ __launch_bounds__(kBlock) void k_admit(StepCtx c, int32_t *admitStep, int32_t *waitHead, VidTable vt, CompactScratch cs,
                                                  const SpawnBatch batch) {
    __shared__ int sLane[kAdmitRecs];
    const int nRecs = batch.n, firstNewVid = batch.firstNewVid;
    if ((int) threadIdx.x < nRecs) sLane[threadIdx.x] = batch.lane[threadIdx.x];
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    const bool isLane = lane < c.n.L, inRange = lane < c.n.L + c.n.K;
    int w = -1, n = 0, base = 0;
    if (isLane) {
        w = waitHead[lane];
        n = c.cnt[lane];
        base = c.segStart[lane];
    }
    int wt = 0, route = 0, nextWait = -1;
    uint8_t pending = 0;
    if (w >= 0) {
        wt = vt.templ[w];
        route = vt.route[w];
        nextWait = vt.nextWait[w];
        pending = vt.pendingCustom[w];
    }
    __syncthreads();
    if (nRecs > 0) {
        // the vehicle table of the new vehicles (k_spawn_link): block 0.  Nobody reads these rows in this kernel — a vehicle
        // that is admitted in the step it appears in is taken from its record
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < nRecs; i += blockDim.x) {
                const int v = firstNewVid + batch.vidOff[i];
                vt.firstNext[v] = batch.firstNext[i];
                if (c.lc.on) lcInitVid(c.lc, v);
                vt.priority[v] = batch.priority[i];
                vt.templ[v] = batch.templ[i];
                vt.route[v] = batch.route[i];
                vt.enterTime[v] = batch.enterTime;
                vt.state[v] = 0;
                vt.pendingCustom[v] = 0;
            }
        if (isLane) {
            // FIFO append (Lane::pushWaitingVehicle roadnet.h:365-367; nextWait[] of a new vehicle was pre-set to -1): this
            // lane's records, in any order — each hangs behind its predecessor, or becomes the head where the predecessor
            // has left the queue
            int lo = 0, hi = nRecs;  // first record of this lane
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sLane[mid] < lane) lo = mid + 1;
                else hi = mid;
            }
            int headRec = -1;
            for (int j = lo; j < nRecs && sLane[j] == lane; ++j) {
                const int pv = batch.prevWait[j], v = firstNewVid + batch.vidOff[j];
                bool becomesHead = pv < 0;
                if (pv >= firstNewVid) vt.nextWait[pv] = v;      // predecessor in this very batch: certainly still queued
                else if (pv >= 0) {
                    if (vt.state[pv] != 0) becomesHead = true;  // predecessor already admitted => the FIFO is empty
                    else vt.nextWait[pv] = v;
                }
                if (becomesHead) headRec = j;
            }
            if (headRec >= 0) {
                w = firstNewVid + batch.vidOff[headRec];
                wt = batch.templ[headRec];
                route = batch.route[headRec];
                pending = 0;
                nextWait = -1;
                waitHead[lane] = w;
            }
            if (w >= 0)  // whoever was hung behind the head just now (this thread's own store: taken from the record)
                for (int j = lo; j < nRecs && sLane[j] == lane; ++j)
                    if (batch.prevWait[j] == w) nextWait = firstNewVid + batch.vidOff[j];
        }
    }
    if (!inRange) return;
    cs.leaveCnt[lane] = 0;  // compaction scratch of every drivable (lanes and laneLinks) for this step
    cs.maxLeaveIdx[lane] = -1;
    cs.inCnt[lane] = 0;
    cs.inHead[lane] = -1;
    if (!isLane) {
        // laneLink thread: the gate record k_action needs about "the next laneLink" in one load
        const int k = lane - c.n.L;
        int flags = (llAvailable(c, k) ? 1 : 0) | (c.n.llType[k] << 1) | (c.n.llXStart[k + 1] > c.n.llXStart[k] ? 8 : 0);
        c.llGate[k] = make_int2(flags, c.n.llEndLane[k]);
        return;
    }
    c.laneTail[lane] = n > 0 ? base + n - 1 : -1;  // overwritten below if a vehicle is admitted
    bool admit = w >= 0;
    if (admit && n > 0) {
        int tail = base + n - 1;
        if (!(c.s.dis[tail] > c.t.templ[c.s.templ[tail]].len + c.t.templ[wt].min_gap)) admit = false;
    }
    if (!admit) {
        if (c.lc.on) lcInitSegments(c, base, n, false);
        return;
    }
    int slot = base + n;  // the lane's spare slot
    c.s.vid[slot] = w;
    c.s.drv[slot] = lane;
    c.s.prevDrv[slot] = -1;
    c.s.next[slot] = nextOf(c.n, c.t, lane, route, 0);
    c.s.blocker[slot] = -1;
    c.s.enterLLT[slot] = CFX_INT_MAX;  // ControllerInfo ctor vehicle.cpp:10-13
    c.s.routePos[slot] = 0;
    c.s.templ[slot] = wt;
    c.s.route[slot] = route;
    c.s.flags[slot] = pending;
    c.s.dis[slot] = 0.0;
    c.s.speed[slot] = c.t.templ[wt].initial_speed;  // VehicleInfo::speed: 0 unless pushed with a speed
    c.laneTail[lane] = slot;
    c.admitRec[lane] = make_int2(w, nextWait);
    admitStep[lane] = c.step;  // cnt[], the FIFO pop and the running count follow in k_scan (see cntNow)
    if (c.lc.on) lcInitSegments(c, base, n, true);
}

// The per-slot columns the cross phase reads, through accessors: the ring layout keeps them in two 16-byte records per
// slot (cfx_ring_kernels.h: RingCtx::kin / meta) and overloads these.
template <class C> __device__ __forceinline__ double slotDis(const C &c, int s) { return c.s.dis[s]; }
template <class C> __device__ __forceinline__ double slotSpeed(const C &c, int s) { return c.s.speed[s]; }
template <class C> __device__ __forceinline__ int slotTempl(const C &c, int s) { return c.s.templ[s]; }
template <class C> __device__ __forceinline__ int slotNext(const C &c, int s) { return c.s.next[s]; }
template <class C> __device__ __forceinline__ int slotEnterLLT(const C &c, int s) { return c.s.enterLLT[s]; }

// Per-laneLink sources of Engine::threadNotifyCross (engine.cpp:317-372): the vehicle that just left
// onto the end lane (331-332), the vehicles on the laneLink (344), the first vehicle of the start lane if
// it heads here on green (362-363).  Which of them a particular cross sees is resolved by notifiedAt().
template <class C> __device__ inline void llstate(const C &c, int k) {
    if (k >= c.n.K) return;
    const int d = c.n.L + k;
    const int endLane = c.n.llEndLane[k], startLane = c.n.llStartLane[k];
    const Tail tu = tailNowOf(c, endLane);
    const int u = (tu.slot >= 0 && tu.prevDrv == d) ? tu.slot : -1;
    int f = cntNow(c, startLane) > 0 ? firstSlot(c, startLane) : -1;
    if (f >= 0 && !(c.s.next[f] == d && llAvailable(c, k))) f = -1;
    const int nOn = committedCount(c, d);
    c.llDyn[k] = make_int4(u, f, firstSlot(c, d), nOn);
    if (u >= 0 || f >= 0 || nOn > 0) {
        int in = c.n.llInter[k];
        int bit = c.n.llLocal[k];
        atomicOr(&c.interMask[c.n.interMaskStart[in] + (bit >> 6)], 1ULL << (bit & 63));
    }
}

// Cross::notifyVehicles / notifyDistances of cross entry `pe` (owned by laneLink k), i.e. what the sweep of
// engine.cpp:327-369 would have written there: entries are consumed far -> near by (1) the vehicle on the
// end lane while `crossDistance + vehDistance < 0`, (2) each vehicle on the laneLink, front to back, while
// it has not completely passed the cross, (3) the approaching vehicle for everything left.  All three
// conditions are monotone in the cross distance, so "first source that accepts this entry" is the same
// assignment as the sequential sweep.
template <class C>
__device__ inline int notifiedAt(const C &c, const cfx_vehicle_template *tv, int k, double x, double *distOut) {
    const int d = c.n.L + k;
    const int4 dyn = c.llDyn[k];  // {u, f, segStart, cnt} written by llstate(): one 16-byte load
    const int u = dyn.x;
    if (u >= 0) {
        double udis = slotDis(c, u);
        double vehDistance = udis - tv[slotTempl(c, u)].len;
        double crossDistance = c.n.drvLength[d] - x;
        if (crossDistance + vehDistance < 0.0) {
            *distOut = -(udis + crossDistance);
            return u;
        }
    }
    const int n = dyn.w;
    const SegWalk walk = segWalk(c, d, dyn.z);  // from the laneLink's first vehicle backwards
    for (int i = 0; i < n; ++i) {
        int w = walk.at(i);
        double vehDistance = slotDis(c, w);
        if (!(vehDistance > x) || (vehDistance - x - tv[slotTempl(c, w)].len <= 0.0)) {
            *distOut = x - vehDistance;
            return w;
        }
    }
    const int f = dyn.y;
    if (f >= 0) {
        int startLane = c.n.llStartLane[k];
        *distOut = (c.n.drvLength[startLane] - slotDis(c, f)) + x;
        return f;
    }
    return -1;
}

// The vehicle a cross has been notified of, with what Cross::canPass reads of it
struct Notified {
    int slot, templ;  // slot < 0: nobody
    double speed, dist;
    // requested together with the rest where the layout can (ring): the vehicle's enterLaneLinkTime and blocker record,
    // which the decision tree below may want after its arithmetic — `pre` says they are here
    bool pre;
    int enterLLT;
    int2 blk;
    int vid;  // (with `pre`) the vehicle's number: what a yielding vehicle records as its blocker
};
// the notified vehicle's blocker as a slot, from what notified() brought along if it did
template <class C> __device__ __forceinline__ int blockerOfNotified(const C &c, const Notified &nf) { return blockerOf(c, nf.slot); }
template <class C>
__device__ inline Notified notified(const C &c, const cfx_vehicle_template *tv, int k, double x) {
    Notified nf{-1, 0, 0.0, 0.0, false, 0, make_int2(-1, -1)};
    nf.slot = notifiedAt(c, tv, k, x, &nf.dist);
    if (nf.slot >= 0) {
        nf.templ = slotTempl(c, nf.slot);
        nf.speed = slotSpeed(c, nf.slot);
    }
    return nf;
}

// Cross::canPass roadnet.cpp:603-676 for a cross whose peer laneLink is active.  `e` = this laneLink's
// entry of the cross, `t1` its roadLink type.
template <class C>
__device__ inline bool canPassDecide(const C &c, const cfx_vehicle_template *tv, int selfSlot, const VehRef &self, double dOn,
                                     int t1, double distanceToLaneLinkStart, const Notified &nf, int t2, int *foeSlotOut);

template <class C>
__device__ inline bool canPassActive(const C &c, const cfx_vehicle_template *tv, int selfSlot, const VehRef &self,
                                     double dOn, int t1, double distanceToLaneLinkStart, int peLL, double peerDist,
                                     int t2, int *foeSlotOut) {
    return canPassDecide(c, tv, selfSlot, self, dOn, t1, distanceToLaneLinkStart, notified(c, tv, peLL, peerDist), t2, foeSlotOut);
}

// ... the decision once the cross's notified vehicle is known
template <class C>
__device__ inline bool canPassDecide(const C &c, const cfx_vehicle_template *tv, int selfSlot, const VehRef &self, double dOn,
                                     int t1, double distanceToLaneLinkStart, const Notified &nf, int t2, int *foeSlotOut) {
    const int foeSlot = nf.slot;
    const double d2 = nf.dist;
    *foeSlotOut = foeSlot;
    if (foeSlot < 0) return true;
    const double d1 = dOn - distanceToLaneLinkStart;
    if (!canYield(self, d1)) return true;
    VehRef foe{nf.speed, &tv[nf.templ]};
    int yield = 0;
    if (!canYield(foe, d2)) yield = 1;
    if (yield == 0) {
        if (t1 > t2) {
            yield = -1;
        } else if (t1 < t2) {
            if (d2 > 0) {
                int foeSteps = reachStepsOnLaneLink(foe, d2, t2, c.interval);
                int mySteps = reachStepsOnLaneLink(self, d1, t1, c.interval);
                if (foeSteps > mySteps) yield = -1;
            } else {
                if (d2 + foe.t->len < 0) yield = -1;
            }
            if (yield == 0) yield = 1;
        } else {
            if (d2 > 0) {
                int foeSteps = reachStepsOnLaneLink(foe, d2, t2, c.interval);
                int mySteps = reachStepsOnLaneLink(self, d1, t1, c.interval);
                if (foeSteps > mySteps) {
                    yield = -1;
                } else if (foeSteps < mySteps) {
                    yield = 1;
                } else {
                    int myT = slotEnterLLT(c, selfSlot), foeT = nf.pre ? nf.enterLLT : slotEnterLLT(c, foeSlot);
                    if (myT == foeT) {
                        if (d1 == d2) {
                            yield = c.vPriority[c.s.vid[selfSlot]] > c.vPriority[c.s.vid[foeSlot]] ? -1 : 1;
                        } else {
                            yield = d1 < d2 ? -1 : 1;
                        }
                    } else {
                        yield = myT < foeT ? -1 : 1;
                    }
                }
            } else {
                yield = d2 + foe.t->len < 0 ? -1 : 1;
            }
        }
    }
    if (yield == 1) {  // Floyd cycle walk over committed blockers (deadlock => pass), roadnet.cpp:662-674
        // (every link of the chain is looked up once: a lookup is two or three dependent loads)
        int fast = foeSlot, slow = foeSlot;
        int guard = 0;
        int fastBlocker = blockerOfNotified(c, nf);
        while (fastBlocker >= 0) {
            slow = blockerOf(c, slow);
            fast = blockerOf(c, fastBlocker);
            if (slow == fast) {
                yield = -1;
                break;
            }
            if (fast < 0 || ++guard > (1 << 22)) break;  // (the bound cannot be hit: Floyd terminates)
            fastBlocker = blockerOf(c, fast);
        }
    }
    return yield == -1;
}

// leader/gap (Vehicle::updateLeaderAndGap vehicle.cpp:157-196) for the HEAD of drivable d (slot s): the last vehicle of the
// drivables ahead on its route, within the look-ahead bound.  Returns the leader as a Tail (slot < 0: none).
template <class C>
__device__ inline Tail findHeadLeader(const C &c, const cfx_vehicle_template *tv, int s, int d, double myDis, double bound,
                                      int nd0, double dlen, double *gapOut, int4 firstHop = make_int4(-2, -2, -2, -2)) {
    // head of a lane whose only vehicle was admitted this step => it IS the admitted vehicle
    const bool viewerNew = d < c.n.L && c.admitStep[d] == c.step && committedCount(c, d) == 0;
    Tail best{-1, 0, -1, 0.0, 0.0};
    double gap = 0.0;
    double dist = dlen - myDis;
    int nd = nd0;
    int route = -1, routePos = 0;
    for (;;) {
        if (nd < 0) break;
        if (nd >= c.n.L) {
            // the last vehicles of ALL laneLinks that leave the lane this laneLink leaves (vehicle.cpp:170-181)
            auto consider = [&](int ll) {
                const Tail cand = tailNowOf(c, c.n.L + ll);
                if (cand.slot >= 0) {
                    double cg = dist + cand.dis - tv[cand.templ].len;
                    if (best.slot < 0 || cg < gap) {
                        best = cand;
                        gap = cg;
                    }
                }
            };
            if (firstHop.x != -2) {  // the caller already holds the list (the head's own lane, first hop)
                if (firstHop.x >= 0) consider(firstHop.x);
                if (firstHop.y >= 0) consider(firstHop.y);
                if (firstHop.z >= 0) consider(firstHop.z);
                if (firstHop.w >= 0) consider(firstHop.w);
                firstHop.x = -2;
            } else {
                int sl = c.n.llStartLane[nd - c.n.L];
                for (int q = c.n.laneLLStart[sl]; q < c.n.laneLLStart[sl + 1]; ++q) consider(c.n.laneLL[q]);
            }
            if (best.slot >= 0) break;
        } else {
            best = tailForLeader(c, nd, viewerNew, d);
            if (best.slot >= 0) {
                gap = dist + best.dis - tv[best.templ].len;
                break;
            }
        }
        dist += c.n.drvLength[nd];
        if (dist > bound) break;  // same expression as vehicle.cpp:190-191
        if (route < 0) {
            route = c.s.route[s];
            routePos = c.s.routePos[s];
        }
        nd = nextOf(c.n, c.t, nd, route, routePos);
    }
    *gapOut = gap;
    return best;
}

// ... for any vehicle, as a slot (getters, lane change)
template <class C>
__device__ inline int findLeader(const C &c, const cfx_vehicle_template *tv, int s, int d, bool head, double myDis,
                                 double bound, double *gapOut) {
    if (!head) {
        int ls = slotAhead(c, d, s);
        *gapOut = c.s.dis[ls] - tv[c.s.templ[ls]].len - myDis;
        return ls;
    }
    const int nd0 = c.s.next[s];
    // (the head's own lane: its laneLinks in one load, as the action kernels pass them)
    const int4 hop = (d < c.n.L && nd0 >= c.n.L) ? c.n.laneLL4[d] : make_int4(-2, -2, -2, -2);
    return findHeadLeader(c, tv, s, d, myDis, bound, nd0, c.n.drvLength[d], gapOut, hop).slot;
}

// Tail of Engine::vehicleControl for one vehicle once its intersection speed is known: the rest of
// Vehicle::getNextSpeed (vehicle.cpp:323-331), vehicleControl (engine.cpp:212-221), Vehicle::setDeltaDistance
// (vehicle.cpp:49-68), the buffered results, and the classification half of threadUpdateLocation
// (engine.cpp:290-310: per-drivable leave / enter counts for the compaction).
struct ActionOut {
    ActionBuf b;
    CompactScratch cs;
    VidTable vt;
    DevScalars *sc;
    int32_t *finList;
    int finCap;
    int32_t *finCount;  // [kFinShards * 32], see FinMap
    // a vehicle handed to the cross phase: its two partial speeds wait in the action buffer
    __device__ __forceinline__ void park(int s, double v, double iv) const {
        b.speed[s] = v;
        b.dis[s] = iv;
    }
    __device__ __forceinline__ double parkedSpeed(int s) const { return b.speed[s]; }
    __device__ __forceinline__ double parkedInterSpeed(int s) const { return b.dis[s]; }
    // tiling: a proxy on a ghost lane is not stepped here
    __device__ __forceinline__ void keep(int s, double dis, double speed) const {
        b.dis[s] = dis;
        b.speed[s] = speed;
        b.drv[s] = -1;
        b.blocker[s] = -1;
    }
};

// SimpleLaneChange::yieldSpeed lanechange.cpp:186-206: 100 unless another vehicle's lane-change signal reached this one
// (then: slow down so that the sender's gap behind it becomes safe; the sender's target leader never yields).
// `turn`: the position in the reference's walk over the vehicles at which this yield is evaluated (the vehicle's own vid;
// for a shadow its real vehicle's).  A sender that completed its change EARLIER in that walk has already cleared its
// signal's neighbours (LaneChange::finishChanging -> clearSignal); one that completes it later has not.
__device__ inline double lcYieldSpeed(const StepCtx &c, int vid, double speed, const cfx_vehicle_template &t, int turn) {
    const int src = c.lc.recvFrom[vid];
    if (src < 0) return 100;
    const bool cleared = src < turn && c.lc.lcFinished[src];
    if (!cleared && vid == c.lc.tLeader[src]) return 100;
    const cfx_vehicle_template *tv = c.t.templ;
    const int ss = c.lc.slotOf[src];
    double safeBefore = 0;  // safeGapBefore lanechange.cpp:213-215
    const int f = cleared ? -1 : c.lc.tFollower[src];
    if (f >= 0) {
        const int fs = c.lc.slotOf[f];
        const double fsp = c.s.speed[fs];
        safeBefore = 0.5 * fsp * fsp / tv[c.s.templ[fs]].max_neg_acc;
    }
    const double gap = c.lc.followerGap[src] - safeBefore;
    double v = noCollisionSpeed(c.s.speed[ss], tv[c.s.templ[ss]].max_neg_acc, speed, t.max_neg_acc, gap, c.interval, 0);
    if (v < 0) v = 100;  // "if the follower is too fast, let it go"
    return v;
}

// Engine::vehicleControl engine.cpp:212-221 + Vehicle::setDeltaDistance vehicle.cpp:49-68 for a known next speed
struct MoveOut {
    double v, ndis;
    int newDrv;  // -1 stays, -2 end of route, >= 0 new drivable
};
template <class C>
__device__ inline MoveOut computeMove(const C &c, const cfx_vehicle_template &t, int s, int d, double speed, double dis,
                                      double dlen, int nd0, double v) {
    const double interval = c.interval;
    double deltaDis;
    if (v < 0) {
        deltaDis = 0.5 * speed * speed / t.max_neg_acc;
        v = 0;
    } else {
        deltaDis = (speed + v) * interval / 2;
    }
    double ndis = deltaDis + dis;
    int newDrv = -1;
    if (ndis > dlen) {
        int drivable = d;
        int nxt = nd0;
        const int route = c.s.route[s];
        const int routePos = c.s.routePos[s];
        for (;;) {
            ndis -= c.n.drvLength[drivable];
            drivable = nxt;
            newDrv = drivable >= 0 ? drivable : -2;
            if (drivable < 0 || !(ndis > c.n.drvLength[drivable])) break;
            nxt = nextOf(c.n, c.t, drivable, route, routePos);
        }
    }
    return MoveOut{v, ndis, newDrv};
}

// The buffered results and the classification half of threadUpdateLocation (engine.cpp:290-310: per-drivable leave /
// enter counts for the compaction).  `counted` is false only for the real vehicle of a COMPLETED lane change: it leaves
// the simulation without being a finished vehicle (its shadow carries on, engine.cpp:299).
__device__ inline void commitMove(const StepCtx &c, const ActionOut &o, int s, int d, int vid, const MoveOut &m,
                                  int blockerSlot, bool counted) {
    o.b.dis[s] = m.ndis;
    o.b.speed[s] = m.v;
    o.b.drv[s] = m.newDrv;
    o.b.blocker[s] = blockerSlot;
    if (m.newDrv != -1) {
        const int k = s - c.segStart[d];
        atomicAdd(&o.cs.leaveCnt[d], 1);
        atomicMax(&o.cs.maxLeaveIdx[d], k);
        if (m.newDrv >= 0) {
            atomicAdd(&o.cs.inCnt[m.newDrv], 1);
            o.cs.inNext[s] = atomicExch(&o.cs.inHead[m.newDrv], s);
        } else {
            o.vt.state[vid] = 2;
            if (counted) {
                const int at = finPlace(o.finCount, o.finCap);
                if (at >= 0) o.finList[at] = s;
                else o.sc->overflow = 1;
            } else {
                atomicAdd(&o.sc->nLeftUncounted, 1);
            }
        }
    }
}

// The rest of Vehicle::getNextSpeed after the lane-change yield (vehicle.cpp:325-331): the brake on a lane that does not
// lead on (!Router::onValidLane router.h:66-68) and the deceleration limit.
template <class C>
__device__ inline double speedTail(const C &c, const cfx_vehicle_template &t, int s, int d, double speed, double dis,
                                   double dlen, int nd0, double v) {
    if (nd0 < 0 && !isLastRoad(c, d, c.s.route[s])) {
        double vn = noCollisionSpeed(0, 1, speed, t.max_neg_acc, dlen - dis, c.interval, t.min_gap);
        v = min2(v, vn);
    }
    return max2(v, speed - t.max_neg_acc * c.interval);
}

// LC: the engine runs with lane change (a separate instantiation of the step's kernels, so that the common configuration
// carries none of it)
template <bool LC>
__device__ inline void finishAction(const StepCtx &c, const ActionOut &o, const cfx_vehicle_template &t, int s, int d,
                                    int vid, double speed, double dis, double dlen, int nd0, double v, int blockerSlot,
                                    int /*idx*/ = -1, int /*nNow*/ = -1) {
    if constexpr (LC) {
        // Two kinds of vehicles cannot be finished here, because the reference's walk over the vehicles (creation order)
        // makes their speed depend on what happened to an EARLIER vehicle in the same walk (k_lc_resolve does them, in
        // that order): the two vehicles of a changing pair (common speed, engine.cpp:195-205), and a vehicle signalled by
        // an earlier changing vehicle — if that one completes its change in this step, its signal's neighbours are
        // already cleared when this vehicle evaluates yieldSpeed (LaneChange::finishChanging -> clearSignal).
        const int pt = c.lc.ptype[vid];
        const int src = c.lc.recvFrom[vid];
        if (pt != 0 || (src >= 0 && c.lc.changing[src] && src < vid)) {
            c.lc.bSpeed[vid] = v;  // before the yield
            c.lc.bBlocker[vid] = blockerSlot;
            if (pt != 2) {  // a shadow goes with its real vehicle
                const int idx = waveListAppend(c.lc.parkCount, true);  // (one atomic for the lanes of the wave that park here)
                c.lc.parkList[idx] = vid;
                c.lc.parkIdx[vid] = idx;
                // whom this item has to wait for in k_lc_resolve (lcResolveDep), as the tables stand in this phase — nothing
                // writes them between the schedule walk and the resolve: the changing vehicle whose signal it (or its
                // shadow) holds, if that one comes earlier in the walk.  Kept as that vehicle's number (-1: nobody).
                const int r = pt == 1 ? c.lc.partner[vid] : vid;
                const int from = c.lc.recvFrom[r];
                c.lc.parkDep[idx] = (from >= 0 && from < vid && c.lc.changing[from] && c.lc.ptype[from] == 1) ? from : -1;
            }
            return;
        }
        v = min2(v, lcYieldSpeed(c, vid, speed, t, vid));  // (nobody has completed a change yet at this point of the step)
    } else {
        v = min2(v, 100);  // SimpleLaneChange::yieldSpeed without signals
    }
    v = speedTail(c, t, s, d, speed, dis, dlen, nd0, v);
    commitMove(c, o, s, d, vid, computeMove(c, t, s, d, speed, dis, dlen, nd0, v), blockerSlot, true);
}

// Engine::threadGetAction / vehicleControl engine.cpp:188-251,402-413 with Vehicle::getNextSpeed
// vehicle.cpp:308-335: leader/gap, car following, and the first half of getIntersectionRelatedSpeed (red
// light / blocked exit lane / turn speed).  Vehicles that still have to look at the crosses of their laneLink
// are queued for k_cross (their speed so far parked in the action buffer); everybody else is finished here.
struct SlotIn {  // everything the action phase loads by slot index alone
    int vid, d, templIdx, templPrev, nd0, flags;
    double speed, dis, speedPrev, disPrev;
    bool head;       // first vehicle of its drivable
    int leaderSlot;  // slot of the vehicle ahead in the same drivable (valid unless head)
    int idx, nNow;   // position in the drivable's list and its length (ring layout only; idx -1 = not known)
    double2 lm;      // {length, max speed} of the drivable
    int4 hop;        // the laneLinks leaving the vehicle's lane if the loader holds them (x = -2: not), see findHeadLeader
    bool laneAdmitted;  // ring layout: the vehicle's lane admitted a vehicle this step
    int endLane;        // the lane behind the vehicle's next laneLink if the loader knows it (-1: read it from the gate record)
    int lastRoadFlags;  // ring layout: the slot's flags with bit 1 = "on the last road of its route" (-1: not known, walk the route)
};

// Every load that depends only on the slot index is issued up front, before the first branch, so the memory
// system sees them as ONE round (the kernels are bound by dependent-load rounds, not bytes).
__device__ __forceinline__ SlotIn loadSlot(const StepCtx &c, int s) {
    SlotIn in;
    const int sp = s > 0 ? s - 1 : 0;
    in.vid = c.s.vid[s];
    in.d = c.s.drv[s];
    const int dPrev = c.s.drv[sp];
    in.templIdx = c.s.templ[s];
    in.templPrev = c.s.templ[sp];
    in.speed = c.s.speed[s];
    in.dis = c.s.dis[s];
    in.speedPrev = c.s.speed[sp];
    in.disPrev = c.s.dis[sp];
    in.nd0 = c.s.next[s];
    in.flags = c.s.flags[s];
    in.head = s == 0 || dPrev != in.d;
    in.leaderSlot = sp;
    in.idx = -1;
    in.nNow = -1;
    // (an empty spare slot carries drivable -1; a slot beyond the ones in use, which kd_action reads before it knows the slot
    // count, anything)
    in.lm = c.n.drvLM[(in.d >= 0 && in.d < c.n.L + c.n.K) ? in.d : 0];
    in.hop = make_int4(-2, -2, -2, -2);
    in.laneAdmitted = false;
    in.endLane = -1;
    in.lastRoadFlags = -1;
    return in;
}

struct JobInfo {  // what the action phase knows about a vehicle it hands to the cross phase (the ring layout passes it on)
    int d, idx, nNow, templ, nd0, laneLink, gateFlags, xs, xe;
    double speed, dis, dlen, v, iv;
    int maskBase;  // second form of the ring step: first mask word of the laneLink's intersection
};
// the gate record of a laneLink: {light | type | has crosses, end lane} (dense) + {first, end cross entry} (ring)
__device__ __forceinline__ int gateXs(const int2 &) { return 0; }
__device__ __forceinline__ int gateXe(const int2 &) { return 0; }
__device__ __forceinline__ int gateXs(const int4 &g) { return g.z; }
__device__ __forceinline__ int gateXe(const int4 &g) { return g.w; }
// One vehicle's phase 4 up to the walk over the crosses; `push(s)` hands a vehicle that still has to look at the
// crosses of its laneLink to the cross phase (its two partial speeds are parked in the action buffer).
template <bool LC, class C, class Out, class Push>
__device__ __forceinline__ void actionOne(const C &c, const Out &o, const cfx_vehicle_template *tv, const int s,
                                          const SlotIn &in, Push push) {
    const int sp = in.leaderSlot;
    const int vid = in.vid, d = in.d, templIdx = in.templIdx, templPrev = in.templPrev;
    const double speed = in.speed, dis = in.dis, speedPrev = in.speedPrev, disPrev = in.disPrev;
    const int nd0 = in.nd0, flags = in.flags;
    if (vid < 0) return;
    if (c.n.laneGhost && d < c.n.L && c.n.laneGhost[d]) {  // tiling: proxy of a neighbour's vehicle, not stepped here
        o.keep(s, dis, speed);
        return;
    }
    const bool head = in.head;
    const cfx_vehicle_template &t = tv[templIdx];
    const double interval = c.interval;
    const double2 lm = in.lm;
    const double dlen = lm.x;

    // --- leader / gap
    double gap;
    int ls, leaderTempl = templPrev;
    double leaderSpeed = speedPrev;
    if (!head) {  // Vehicle::updateLeaderAndGap vehicle.cpp:158-160
        ls = sp;
        gap = disPrev - tv[templPrev].len - dis;
    } else {
        const Tail lead = findHeadLeader(c, tv, s, d, dis, t.approach_dist, nd0, dlen, &gap, in.hop);
        ls = lead.slot;
        leaderTempl = lead.templ;
        leaderSpeed = lead.speed;
    }
    // First step after a load: the state's gap (cfx_state::r_gap).  Not with lane change: there the reference refreshes every
    // leader and gap between planLaneChange and getAction (Engine::nextStep engine.cpp:571-575) — the stored gap is read by
    // makeSignal only (k_lc_plan).
    if constexpr (!LC) {
        if ((flags & kFlagStateGap) && ls >= 0) gap = c.vGapState[vid];
    }
    if constexpr (LC) {
        if (ls >= 0) c.lc.gap[vid] = gap;  // lane change reads ControllerInfo::gap as stored state
    }

    // --- Vehicle::getNextSpeed vehicle.cpp:308-335
    double v = t.max_speed;
    v = min2(v, speed + t.max_pos_acc * interval);
    v = min2(v, lm.y);

    // car following, Vehicle::getCarFollowSpeed vehicle.cpp:212-238
    double cf;
    const bool custom = (flags & 1) != 0;  // Vehicle::hasSetCustomSpeed
    if (ls < 0) {
        cf = custom ? c.vCustomSpeed[vid] : t.max_speed;
    } else if (custom) {
        const cfx_vehicle_template &tl = tv[leaderTempl];
        cf = min2(c.vCustomSpeed[vid], noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, t.max_neg_acc, gap, interval, 0));
    } else {
        const cfx_vehicle_template &tl = tv[leaderTempl];
        cf = noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, t.max_neg_acc, gap, interval, 0);
        double assumeDecel = 0;
        if (speed > leaderSpeed) assumeDecel = speed - leaderSpeed;
        cf = min2(cf, noCollisionSpeed(leaderSpeed, tl.usual_neg_acc, speed, t.usual_neg_acc, gap, interval, t.min_gap));
        cf = min2(cf, (gap + (leaderSpeed + assumeDecel / 2) * interval - speed * interval / 2) /
                          (t.headway_time + interval / 2));
    }
    v = min2(v, cf);

    // intersection logic, Vehicle::isIntersectionRelated vehicle.cpp:289-300
    const bool onLane = d < c.n.L;
    const bool related = !onLane || (nd0 >= c.n.L && dlen - dis <= t.approach_dist);
    if (related) {
        // Vehicle::getIntersectionRelatedSpeed vehicle.cpp:337-362
        VehRef self{speed, &t};
        double iv = t.max_speed;
        int laneLink = -1;
        bool done = false;
        int gateFlags, xs = 0, xe = 0;
        if (nd0 >= c.n.L) {
            laneLink = nd0 - c.n.L;
            const auto gate = c.llGate[laneLink];  // {available | type | has crosses, end lane, ...}, from k_admit
            gateFlags = gate.x;
            xs = gateXs(gate);
            xe = gateXe(gate);
            bool blocked = !(gate.x & 1);
            if (!blocked) {  // Lane::canEnter roadnet.cpp:437-445
                const Tail tail = tailNowOf(c, gate.y);
                if (tail.slot >= 0) blocked = !(tail.dis > tv[tail.templ].len + t.len || tail.speed >= 2);
            }
            if (blocked) {
                if (minBrakeDistance(self) > dlen - dis) {
                    // cannot stop before the line: run it
                } else {
                    iv = min2(iv, stopBeforeSpeed(self, dlen - dis, interval));
                    done = true;
                }
            }
        }
        if (!done) {
            if (laneLink < 0) {  // already on a laneLink
                laneLink = d - c.n.L;
                const auto gate = c.llGate[laneLink];
                gateFlags = gate.x;
                xs = gateXs(gate);
                xe = gateXe(gate);
            }
            if (nd0 >= c.n.L && typeIsTurn((gateFlags >> 1) & 3)) iv = min2(iv, t.turn_speed);
            if (gateFlags & 8) {
                // park the two partial speeds and hand the cross checks to k_cross
                o.park(s, v, iv);
                push(s, JobInfo{d, in.idx, in.nNow, templIdx, nd0, laneLink, gateFlags, xs, xe, speed, dis, dlen, v, iv});  // 16-lane groups do the cross checks
                return;
            }
        }
        v = min2(v, iv);
    }
    finishAction<LC>(c, o, t, s, d, vid, speed, dis, dlen, nd0, v, -1, in.idx, in.nNow);
}

// queue for the cross phase.  The counter is sharded: one word takes only ~88 returning atomics per us (MI355X guide,
// "dequeue"), and a step issues one per wave.
// A place in the block's shard of the job queue for every lane of the wavefront that pushes a job right now (= the lanes
// active at the call): ONE returning atomic per wavefront instead of one per vehicle.  A step of the 30x30 workload queues
// ~12 k vehicles; with one atomic each that is ~800 same-address atomics per shard, which the L2 serialises at ~90 per
// microsecond and word — several microseconds of queueing inside a 12 us kernel.
__device__ __forceinline__ int jobQueuePlace(const JobQueue &q) {
    const int shard = blockIdx.x & (kJobShards - 1);
    const unsigned long long m = __ballot(1);
    const int lane = (int) (threadIdx.x & 63u), leader = __ffsll((long long) m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&q.count[shard * kJobShardStride], __popcll(m));
    base = __shfl(base, leader, 64);
    return base + __popcll(m & ((1ULL << lane) - 1ULL));
}
struct PushJob {
    JobQueue q;
    __device__ __forceinline__ void operator()(int s, const JobInfo &) const {
        const int shard = blockIdx.x & (kJobShards - 1);
        const int idx = jobQueuePlace(q);
        if (idx < q.capacity) q.jobs[(size_t) shard * q.capacity + idx] = s;
        else *q.overflow = 9;
    }
};

template <bool LC>
__global__ __launch_bounds__(kActBlock) void k_action(StepCtx c, ActionOut o, JobQueue q, int nVehicleBlocks) {
    // The trailing blocks of the launch do the (independent) per-laneLink notify sources for k_cross.
    if ((int) blockIdx.x >= nVehicleBlocks) {
        llstate(c, ((int) blockIdx.x - nVehicleBlocks) * (int) blockDim.x + (int) threadIdx.x);
        return;
    }
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    const cfx_vehicle_template *tv = c.t.templ;
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        tv = sT;
    }
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = nVehicleBlocks * blockDim.x;
    const PushJob push{q};
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride) {
        SlotIn in = loadSlot(c, s);
        if (in.vid >= 0 && in.head && in.d < c.n.L && in.nd0 >= c.n.L) in.hop = c.n.laneLL4[in.d];  // (findHeadLeader's first hop)
        actionOne<LC>(c, o, tv, s, in, push);
    }
}

// Second half of Vehicle::getIntersectionRelatedSpeed (vehicle.cpp:357-375): the walk over the crosses of the
// vehicle's laneLink.  One kCrossGroup-lane group per queued vehicle, one cross per lane and round; the
// reference's "first cross (ascending distance) that cannot be passed" is the lowest failing lane of the
// first failing round.
constexpr int kCrossGroup = 16;

// tiling (dense layout): a blocker that sits on a ghost lane is a proxy whose slot is recycled by the halo exchange; keep
// it by vehicle id (-(vid + 2)).  Chain walks end there either way: proxies carry no blocker.
__device__ __forceinline__ int keepBlocker(const StepCtx &c, int blockerSlot) {
    if (blockerSlot >= 0 && c.n.laneGhost) {
        const int bd = c.s.drv[blockerSlot];
        if (bd < c.n.L && c.n.laneGhost[bd]) blockerSlot = -(c.s.vid[blockerSlot] + 2);
    }
    return blockerSlot;
}

template <bool LC, class C = StepCtx, class Out = ActionOut>
__global__ __launch_bounds__(kCrossBlock) void k_cross(C c, Out o, JobQueue q) {
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    const cfx_vehicle_template *tv = c.t.templ;
    // A block works on ONE shard of the queue (the one its index names), so it needs that shard's count only — no prefix over
    // the shards, no second barrier — and every group's first queue entry is requested together with the count and the
    // template table (its place follows from the block and group index; an entry beyond the count is read and dropped).
    const int g = threadIdx.x % kCrossGroup;                       // lane inside the group
    const int groupsPerBlock = blockDim.x / kCrossGroup;
    const int groupShift = (threadIdx.x & 63) & ~(kCrossGroup - 1);  // first wave-lane of this group
    const int shard = (int) blockIdx.x & (kJobShards - 1);
    const int blocksOfShard = ((int) gridDim.x - shard + kJobShards - 1) / kJobShards;
    const int jFirst = ((int) blockIdx.x / kJobShards) * groupsPerBlock + (int) threadIdx.x / kCrossGroup;
    const int32_t *const shardJobs = q.jobs + (size_t) shard * q.capacity;
    const int sFirstJob = shardJobs[jFirst < q.capacity ? jFirst : q.capacity - 1];
    const int nShard = min(q.count[shard * kJobShardStride], q.capacity);
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        tv = sT;
    }
    for (int j = jFirst; j < nShard; j += blocksOfShard * groupsPerBlock) {
        const int s = j == jFirst ? sFirstJob : shardJobs[j];
        const int d = c.s.drv[s];
        const cfx_vehicle_template &t = tv[c.s.templ[s]];
        const double speed = c.s.speed[s];
        const double dis = c.s.dis[s];
        const double dlen = c.n.drvLength[d];
        const int nd0 = c.s.next[s];
        const bool onLane = d < c.n.L;
        const int laneLink = onLane ? nd0 - c.n.L : d - c.n.L;
        const int4 lp = c.n.llPack[laneLink];  // {first entry, end of entries, mask word base, RoadLinkType}: one load
        const int t1 = lp.w;
        const double d0 = onLane ? -(dlen - dis) : dis;
        const int mb = lp.z;
        VehRef self{speed, &t};
        const int xs = lp.x, xe = lp.y;
        double iv = o.parkedInterSpeed(s);  // partial intersection speed parked by k_action
        int blockerSlot = -1;
        for (int e0 = xs; e0 < xe; e0 += kCrossGroup) {
            const int e = e0 + g;
            bool fail = false;
            int foe = -1;
            double dOn = 0.0;
            if (e < xe) {
                const double2 dd = c.n.xDD[e];   // {distance on this laneLink, distance on the peer laneLink}
                const int4 xp = c.n.xPack[e];    // {peer laneLink, peer bit, peer roadLink type, -}
                dOn = dd.x;
                // (Cross::canPass lets a vehicle that can no longer yield pass before it looks at the other laneLink's vehicle,
                // roadnet.cpp:617-618: tested here on what the lane already holds, in front of the notified vehicle's gathers)
                if (!(dOn < d0) && canYield(self, dOn - d0)) {
                    if ((c.interMask[mb + (xp.y >> 6)] >> (xp.y & 63)) & 1ULL)
                        fail = !canPassActive(c, tv, s, self, dOn, t1, d0, xp.x, dd.y, xp.z, &foe);
                }
            }
            const unsigned long long ball = __ballot(fail);
            const unsigned gm = (unsigned) ((ball >> groupShift) & ((1ULL << kCrossGroup) - 1ULL));
            if (gm != 0u) {
                const int first = __ffs(gm) - 1;  // lowest lane = smallest cross distance in this round
                const int src = groupShift + first;
                const double fdOn = __shfl(dOn, src, 64);
                blockerSlot = __shfl(foe, src, 64);
                iv = min2(iv, stopBeforeSpeed(self, fdOn - d0 - t.yield_distance, c.interval));
                break;
            }
        }
        if (g == 0) {
            double v = min2(o.parkedSpeed(s), iv);
            blockerSlot = keepBlocker(c, blockerSlot);
            finishAction<LC>(c, o, t, s, d, c.s.vid[s], speed, dis, dlen, nd0, v, blockerSlot);
        }
    }
}

// Same phase, organised for throughput (large networks): k_cross keeps a quarter wave busy for a vehicle's whole chain —
// finding the crosses that matter, the canPass arithmetic of the one or two that do, and the vehicle's finish by a single
// lane.  Here a block takes kCross2Jobs vehicles per batch and goes through three barrier-separated passes:
//   A  16-lane groups walk the crosses of each vehicle and only LIST the (vehicle, cross) pairs that need Cross::canPass
//      (cross still ahead, peer laneLink active);
//   B  one thread per listed pair evaluates it; the lowest failing cross of a vehicle is kept with an LDS atomicMin
//      (crosses are sorted by distance, so "lowest entry" is the reference's "first cross that cannot be passed");
//   C  one thread per vehicle turns that cross into the yield speed / blocker and finishes the vehicle.
// Same results as k_cross (which stops at the first failing round: the later rounds it skips cannot lower the minimum).
// (Round 4: starting pass A from 96-byte job records written by the action kernel — as kr_cross does — instead of queue ->
// slot columns -> laneLink record takes two dependent rounds out of pass A and one out of pass C; built, parity-green and
// measured at 1 M vehicles: k_cross2 58.3 -> 56.9 us on the dense layout against +0.9 us for writing the records, 66 -> 76 us
// on the ring layout (eight more registers: a wavefront less per SIMD), 80 -> 84 us for sixteen batched 30x30 networks.  Taken
// out again: this kernel's time is its three barrier-separated passes times the blocks that are not resident, not pass A's chain.)
constexpr int kCross2Block = 256;
constexpr int kCross2Jobs = 64;
// The vehicles of a batch follow from the queue's length and the grid — ceil(jobs / blocks), in sixteens, up to kCross2JobsMax
// — so that a grid of exactly the blocks the chip holds at once (the host asks for 7 per CU on the dense layout, 5 on the ring
// layout: LDS / registers) takes the whole queue in ONE residency round.  (Round 5, measured with per-block stamps at 1 M
// vehicles: 140 k queued vehicles were 2 200 fixed batches of 64 against 1 792 resident blocks — two rounds of a ~30 us batch,
// the second one a quarter full, behind 14 000 empty blocks: 61 -> 48 us.)
constexpr int kCross2JobsMax = 128;
#ifndef CFX_CROSS2_A2U
#define CFX_CROSS2_A2U 4
#endif
constexpr int kA2U = CFX_CROSS2_A2U;  // cross entries a thread of pass A2 requests per round
#ifndef CFX_CROSS2_WORK
#define CFX_CROSS2_WORK 1472  // (listed pairs per batch; what keeps the block's LDS at a seventh of a CU's)
#endif
constexpr int kCross2Work = CFX_CROSS2_WORK;

struct RingLights {  // TrafficLight::passTime of the step, done by the cross kernel when the step's commit is deferred (ring layout)
    int32_t *curPhase;
    double *remain;
    int on;
};
// TrafficLight::passTime trafficlight.cpp:29-37 for every intersection (threads gid, gid + stride, ...)
__device__ inline void passTimeAll(const DevNet &n, int32_t *curPhase, double *remain, double interval, int gid, int stride) {
    for (int i = gid; i < n.I; i += stride) {
        if (n.interVirtual[i]) continue;
        const int ps = n.interPhaseStart[i];
        const int np = n.interPhaseStart[i + 1] - ps;
        double rem = remain[i] - interval;
        int ph = curPhase[i];
        while (rem <= 0.0) {
            ph = (ph + 1) % np;
            rem += n.phaseTime[ps + ph];
        }
        remain[i] = rem;
        curPhase[i] = ph;
    }
}

template <bool LC, class C = StepCtx, class Out = ActionOut>
__global__ __launch_bounds__(kCross2Block, C::kCross2Waves) void k_cross2(C c, Out o, JobQueue q, RingLights lights = RingLights{nullptr, nullptr, 0}) {
    // (nothing in this kernel reads the lights: the approaching vehicles' light test is folded into llDyn by the action kernel)
    KSTAMP(8, 0);
    if (lights.on) passTimeAll(c.n, lights.curPhase, lights.remain, c.interval, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int shardEnd[kJobShards];
    __shared__ int sS[kCross2JobsMax], sT1[kCross2JobsMax], sTempl[kCross2JobsMax], sXs[kCross2JobsMax], sXe[kCross2JobsMax], sMb[kCross2JobsMax];
    __shared__ unsigned long long sMask0[kCross2JobsMax];
    __shared__ int sOff[kCross2JobsMax];
    // (lowest failing cross entry << 32) | the vehicle that cross yields to: entries are unique per vehicle, so the minimum of
    // these is the minimum over the entries, and pass C needs no second look at the cross (Cross::getFoeVehicle)
    __shared__ unsigned long long sFirst[kCross2JobsMax];
    __shared__ double sD0[kCross2JobsMax], sSpeed[kCross2JobsMax];
    __shared__ int sWorkJob[kCross2Work], sWorkE[kCross2Work];
    __shared__ int sNWork;
    const cfx_vehicle_template *tv = c.t.templ;
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < kJobShards; ++i) {
            run += min(q.count[i * kJobShardStride], q.capacity);
            shardEnd[i] = run;
        }
    }
    __syncthreads();
    const int nJ = shardEnd[kJobShards - 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) o.sc->nCrossJobs = nJ;  // (diagnostics; sizes the grids of the other forms)
    KSTAMP(8, 1);
    KNOTE(8, 5, nJ);
    const int tid = threadIdx.x;
    constexpr int kGroups = kCross2Block / kCrossGroup;
    // Cross::canPass for the vehicle in slot `s` at cross entry e; a failing cross competes for "first of the vehicle"
    auto evaluate = [&](int jl, int s, double speed, int templ, int t1, double d0, int e) {
        const double2 dd = c.n.xDD[e];
        const int4 xp = c.n.xPack[e];
        VehRef self{speed, &tv[templ]};
        int foe;
        if (!canPassActive(c, tv, s, self, dd.x, t1, d0, xp.x, dd.y, xp.z, &foe))
            atomicMin(&sFirst[jl], ((unsigned long long) (unsigned) e << 32) | (unsigned long long) (unsigned) foe);
    };
    const int jpb = min(kCross2JobsMax, max(kGroups, (((nJ + (int) gridDim.x - 1) / (int) gridDim.x) + kGroups - 1) / kGroups * kGroups));
    for (int j0 = blockIdx.x * jpb; j0 < nJ; j0 += gridDim.x * jpb) {
        if (tid < jpb) sFirst[tid] = ~0ULL;
        if (tid == 0) sNWork = 0;
        __syncthreads();
        // ---- pass A1: one thread per vehicle of the batch: its record (queue -> slot columns -> laneLink record) into LDS.
        // (Until round 5 every 16-lane group walked this chain for its own vehicle, rep after rep: with all of the chip's blocks
        // in pass A at the same time that was 5 us per rep and 26 of the batch's 39 us at 1 M vehicles.)
        if (tid < jpb) {
            int xs = 0, xe = 0;
            if (j0 + tid < nJ) {
                int shard = 0;
                while (j0 + tid >= shardEnd[shard]) ++shard;
                const int s = q.jobs[(size_t) shard * q.capacity + (j0 + tid - (shard ? shardEnd[shard - 1] : 0))];
                const int d = c.s.drv[s];
                const int templ = slotTempl(c, s);
                const double speed = slotSpeed(c, s);
                const double dis = slotDis(c, s);
                const int nd0 = slotNext(c, s);
                const bool onLane = d < c.n.L;
                const int laneLink = onLane ? nd0 - c.n.L : d - c.n.L;
                const int4 lp = c.n.llPack[laneLink];  // {first entry, end of entries, mask word base, RoadLinkType}
                sS[tid] = s;
                sT1[tid] = lp.w;
                sTempl[tid] = templ;
                sD0[tid] = onLane ? -(c.n.drvLength[d] - dis) : dis;
                sSpeed[tid] = speed;
                sMb[tid] = lp.z;
                sMask0[tid] = c.interMask[lp.z];  // (the intersection's first 64 laneLinks: all of them on most networks)
                xs = lp.x;
                xe = lp.y;
            }
            sXs[tid] = xs;
            sXe[tid] = xe;
        }
        __syncthreads();
        if (j0 == (int) blockIdx.x * jpb) { KSTAMP(8, 7); }
        // ---- pass A2: the batch's cross entries as ONE flat list (vehicle by vehicle, a prefix sum of their counts in LDS), four
        // per thread and round of loads: which of them need a look.  (16-lane groups walking vehicle after vehicle were one
        // dependent round per rep — 12 us of a 32 us batch with every block of the chip in this pass at once.)
        if (tid < kCross2JobsMax) {  // inclusive prefix sum of the vehicles' entry counts: two wavefronts, then the second adds the first's total
            int v = tid < jpb ? sXe[tid] - sXs[tid] : 0;
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(v, off, 64);
                if ((tid & 63) >= off) v += up;
            }
            sOff[tid] = v;
        }
        __syncthreads();
        if (tid >= 64 && tid < kCross2JobsMax) sOff[tid] += sOff[63];
        __syncthreads();
        const int nE = sOff[kCross2JobsMax - 1];
        for (int f0 = tid; f0 < nE; f0 += kA2U * kCross2Block) {
            int jl4[kA2U], e4[kA2U];
            double dOn4[kA2U];
            int bit4[kA2U];
#pragma unroll
            for (int u = 0; u < kA2U; ++u) {
                const int f = f0 + u * kCross2Block;
                jl4[u] = -1;
                if (f < nE) {
                    int lo = 0, hi = kCross2JobsMax - 1;  // first vehicle whose inclusive count exceeds f
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (sOff[mid] > f) hi = mid;
                        else lo = mid + 1;
                    }
                    jl4[u] = lo;
                    e4[u] = sXs[lo] + (f - (lo ? sOff[lo - 1] : 0));
                    dOn4[u] = c.n.xDD[e4[u]].x;
                    bit4[u] = c.n.xPack[e4[u]].y;
                }
            }
#pragma unroll
            for (int u = 0; u < kA2U; ++u) {
                const int jl = jl4[u];
                if (jl < 0) continue;
                const double d0 = sD0[jl], speed = sSpeed[jl];
                if (dOn4[u] < d0) continue;  // the cross is already behind the vehicle
                // a vehicle that can no longer yield at this cross passes it whoever comes (Cross::canPass roadnet.cpp:617-618,
                // tested before the other vehicle is looked at): such a pair needs no pass B
                const int templ = sTempl[jl];
                const VehRef self{speed, &tv[templ]};
                if (!canYield(self, dOn4[u] - d0)) continue;
                const int bit = bit4[u];
                const unsigned long long word = (bit >> 6) == 0 ? sMask0[jl] : c.interMask[sMb[jl] + (bit >> 6)];
                if (!((word >> (bit & 63)) & 1ULL)) continue;  // nobody to yield to there
                const int w = atomicAdd(&sNWork, 1);
                if (w < kCross2Work) {
                    sWorkJob[w] = jl;
                    sWorkE[w] = e4[u];
                } else {
                    evaluate(jl, sS[jl], speed, templ, sT1[jl], d0, e4[u]);  // list full (very busy junctions): look at it right away
                }
            }
        }
        __syncthreads();
        if (j0 == (int) blockIdx.x * jpb) { KSTAMP(8, 2); }
        // ---- pass B: one thread per listed pair
        const int nW = sNWork < kCross2Work ? sNWork : kCross2Work;
        for (int w = tid; w < nW; w += kCross2Block) {
            const int jl = sWorkJob[w];
            evaluate(jl, sS[jl], sSpeed[jl], sTempl[jl], sT1[jl], sD0[jl], sWorkE[w]);
        }
        __syncthreads();
        if (j0 == (int) blockIdx.x * jpb) { KSTAMP(8, 3); }
        // ---- pass C: one thread per vehicle
        if (tid < jpb && j0 + tid < nJ) {
            const int s = sS[tid];
            const cfx_vehicle_template &t = tv[sTempl[tid]];
            const double speed = sSpeed[tid], d0 = sD0[tid];
            const int d = c.s.drv[s];
            const double dis = slotDis(c, s);
            const double dlen = c.n.drvLength[d];
            const int nd0 = slotNext(c, s);
            double iv = o.parkedInterSpeed(s);  // partial intersection speed parked by k_action
            int blockerSlot = -1;
            const unsigned long long ff = sFirst[tid];
            if (ff != ~0ULL) {
                const int e = (int) (ff >> 32);
                blockerSlot = (int) (unsigned) ff;  // the vehicle that cross made us yield to (pass B kept it with the entry)
                VehRef self{speed, &t};
                iv = min2(iv, stopBeforeSpeed(self, c.n.xDD[e].x - d0 - t.yield_distance, c.interval));
                blockerSlot = keepBlocker(c, blockerSlot);
            }
            finishAction<LC>(c, o, t, s, d, c.s.vid[s], speed, dis, dlen, nd0, min2(o.parkedSpeed(s), iv), blockerSlot);
        }
        __syncthreads();
        if (j0 == (int) blockIdx.x * jpb) { KSTAMP(8, 4); }
    }
    KSTAMP(8, 6);
}

// Phase 5b in ONE launch: single-pass exclusive scan of the new segment sizes over drivables.
// A tile waits for its predecessors' totals, so every predecessor must be running or done: with at most
// kScanResidentTiles tiles the whole grid is co-resident (256 CUs x >= 2 such blocks) and tile = block index;
// larger grids hand tiles out by a ticket (every predecessor of a tile has then started: no dispatch-order assumption);
// each tile publishes its total as an 8-byte {epoch, value} granule written by one agent-scope store and
// sums its predecessors' granules, polling with agent-scope loads until their tag equals this step's epoch
// (MI355X guide, Guideline 16 form R2: the data is the flag).
constexpr int kScanItems = 8;                       // drivables per thread
constexpr int kScanTile = kBlock * kScanItems;      // drivables per tile
constexpr int kFinLds = 1024;                       // finished vehicles staged in LDS at a time
constexpr unsigned kSpinLimit = 1u << 26;
constexpr int kScanResidentTiles = 512;             // tiles (256-thread blocks, 28 VGPRs) that are certainly co-resident

__device__ __forceinline__ int newLiveCount(const int32_t *cnt, const CompactScratch &cs, int d, int admitted) {
    return cnt[d] + admitted - cs.leaveCnt[d] + cs.inCnt[d];
}

__device__ inline int blockReduceSum(int v, int *smem) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) smem[w] = v;
    __syncthreads();
    int tot = 0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int) (blockDim.x >> 6); ++i) tot += smem[i];
        smem[0] = tot;
    }
    __syncthreads();
    tot = smem[0];
    __syncthreads();
    return tot;
}

// cumulativeTravelTime += the step's travel times IN THE REFERENCE'S ORDER (term[] / finTerm[] hold them in that order).
// FP64 addition is not associative, so in general one thread has to add them one after the other.  But when the running
// sum and every term are multiples of 2^-10 and everything stays below 2^42 (interval 1.0, 0.5, ...: practically always),
// every partial sum of every order is exactly representable: no addition rounds, and the sum is the same in ANY order —
// then the whole block adds integers in parallel.  (Thousands of vehicles finish per step on large networks; the
// sequential tail was the longest single chain of the step.)  Executed by one whole block; returns the new sum.
__device__ inline double orderedSum(double cum, int F, double *term, const double *finTerm, bool inLds) {
    __shared__ long long sAcc[kBlock / 64];
    __shared__ int sBad[kBlock / 64];
    auto load = [&](int j) {
        return inLds ? term[j]
                     : __longlong_as_double((long long) __hip_atomic_load((const unsigned long long *) &finTerm[j],
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    const double lim = 4398046511104.0 * 1024.0;  // 2^42 in units of 2^-10
    long long acc = 0;
    int bad = 0;
    for (int j = threadIdx.x; j < F; j += blockDim.x) {
        const double sv = load(j) * 1024.0;
        if (sv == rint(sv) && fabs(sv) < lim) acc += (long long) fabs(sv);
        else bad = 1;
    }
    const double cs = cum * 1024.0;
    if (!(cs == rint(cs) && fabs(cs) < lim)) bad = 1;
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_down(acc, off, 64);
        bad |= __shfl_down(bad, off, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sAcc[threadIdx.x >> 6] = acc;
        sBad[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    long long absTotal = 0;
    bad = 0;
    for (int i = 0; i < (int) (blockDim.x >> 6); ++i) {
        absTotal += sAcc[i];
        bad |= sBad[i];
    }
    if (!bad && (double) absTotal + fabs(cs) < lim) {
        // travel times are never negative, so the sum of magnitudes IS the sum (a negative term would have set `bad`
        // through the comparison below); kept general: recompute signed when any term is negative
        long long total = 0;
        for (int j = threadIdx.x; j < F; j += blockDim.x) total += (long long) (load(j) * 1024.0);
        for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sAcc[threadIdx.x >> 6] = total;
        __syncthreads();
        total = 0;
        for (int i = 0; i < (int) (blockDim.x >> 6); ++i) total += sAcc[i];
        return (double) ((long long) cs + total) / 1024.0;
    }
    // the general case: one thread, in order (the loads run ahead of the dependent additions)
    __shared__ double sCum;
    if (threadIdx.x == 0) {
        int j = 0;
        for (; j + 4 <= F; j += 4) {
            const double a = load(j), b = load(j + 1), c2 = load(j + 2), d2 = load(j + 3);
            cum += a;
            cum += b;
            cum += c2;
            cum += d2;
        }
        for (; j < F; ++j) cum += load(j);
        sCum = cum;
    }
    __syncthreads();
    return sCum;
}

// Order-free finish statistics.  The host keeps track of whether the interval, every vehicle's enter time and the running
// sum are multiples of 2^-10 below 2^42 (cfx_engine::timesDyadic; interval 1.0, 0.5, ...: practically always).  Then every
// partial sum of the travel times is exactly representable in any order, no addition ever rounds, and the reference's
// sequential FP64 sum equals the integer sum: no rank sort, no ordered tail — the blocks add integers and the last one to
// arrive (ticket) folds the total into cumulativeTravelTime.  `finTicket[0]` = ticket, `finTicket[2..3]` = 64-bit total.
template <class VidAt>
__device__ inline bool exactFinishStatistics(double now, const VidTable &vt, DevScalars *sc, int F, VidAt vidAt, uint8_t *stateW,
                                             int32_t *finTicket, int part, int nParts, int nUncounted, int32_t *finCount,
                                             int32_t *slotOfW = nullptr) {
    __shared__ long long sAcc[kBlock / 64];
    __shared__ int lastShared;
    const int per = (F + nParts - 1) / nParts;
    const int lo = part * per, hi = min(F, lo + per);
    long long acc = 0;
    for (int i = lo + (int) threadIdx.x; i < hi; i += blockDim.x) {
        const int vid = vidAt(i);
        if (stateW) stateW[vid] = 2;
        if (slotOfW) slotOfW[vid] = -1;
        acc += (long long) ((now - vt.enterTime[vid]) * 1024.0);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sAcc[threadIdx.x >> 6] = acc;
    __syncthreads();
    long long total = 0;
    for (int i = 0; i < (int) (blockDim.x >> 6); ++i) total += sAcc[i];
    unsigned long long *accum = (unsigned long long *) (finTicket + 2);
    if (nParts > 1) {
        if (threadIdx.x == 0) {
            atomicAdd(accum, (unsigned long long) total);
            __threadfence();
            lastShared = atomicAdd(finTicket, 1) == nParts - 1;
        }
        __syncthreads();
        if (!lastShared) return false;
        if (threadIdx.x == 0) {
            __threadfence();
            total = (long long) __hip_atomic_load(accum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *accum = 0ULL;
            *finTicket = 0;
        }
    }
    if (threadIdx.x == 0) {
        sc->cumulativeTravelTime = (double) ((long long) (sc->cumulativeTravelTime * 1024.0) + total) / 1024.0;
        sc->vehicleSteps += sc->active;  // everybody counted as active took this step's phase 4
        sc->finishedCnt += F;
        sc->active -= F + nUncounted;
        finCountsClear(finCount);
        sc->nLeftUncounted = 0;
    }
    return true;
}

// The step's finish statistics in the reference's order (threadUpdateLocation with one thread walks
// drivables in RoadNet order, lists front to back, i.e. ascending slot; engine.cpp:296-310): rank sort of the
// finished slots in LDS, then one thread adds the travel times in that order (FP64 addition is not
// associative; the reference adds sequentially).  Executed by one whole block.
__device__ inline bool finishStatistics(const StepCtx &c, const VidTable &vt, DevScalars *sc, const int32_t *finList,
                                        double *finTerm, int finCap, int32_t *finTicket, int part, int nParts, int exactTimes,
                                        int32_t *finCount) {
    __shared__ int fin[kFinLds];
    __shared__ double term[kFinLds];
    __shared__ int lastShared;
    __shared__ FinMap fm;
    const int F = finMapLoad(fm, finCount, finCap);
    const double now = c.step * c.interval;  // Engine::getCurrentTime engine.cpp:678-680
    if (exactTimes)
        return exactFinishStatistics(now, vt, sc, F, [&](int i) { return c.s.vid[finList[finAt(fm, finCap, i)]]; }, nullptr, finTicket,
                                     part, nParts, sc->nLeftUncounted, finCount);
    // this block ranks finishers [lo, hi); every block walks the whole list, chunk by chunk through LDS
    const bool inLds = nParts == 1 && F <= kFinLds;  // the common case never leaves the block
    const int per = (F + nParts - 1) / nParts;
    const int lo = part * per, hi = min(F, lo + per);
    for (int base = lo; base < hi; base += blockDim.x) {
        const int i = base + (int) threadIdx.x;
        const int me = i < hi ? finList[finAt(fm, finCap, i)] : 0;
        int rank = 0;
        for (int cb = 0; cb < F; cb += kFinLds) {
            const int cn = min(kFinLds, F - cb);
            __syncthreads();
            for (int j = threadIdx.x; j < cn; j += blockDim.x) fin[j] = finList[finAt(fm, finCap, cb + j)];
            __syncthreads();
            if (i < hi)
                for (int j = 0; j < cn; ++j) rank += fin[j] < me;
        }
        if (i < hi) {
            const double t = now - vt.enterTime[c.s.vid[me]];
            if (inLds) term[rank] = t;
            else finTerm[rank] = t;
        }
    }
    bool last = true;
    if (nParts > 1) {  // the block that arrives last adds up
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) lastShared = atomicAdd(finTicket, 1) == nParts - 1;
        __syncthreads();
        last = lastShared != 0;
        if (!last) return false;
        if (threadIdx.x == 0) *finTicket = 0;
        __threadfence();
    }
    __syncthreads();
    const double cum = orderedSum(sc->cumulativeTravelTime, F, term, finTerm, inLds);
    if (threadIdx.x == 0) {
        sc->cumulativeTravelTime = cum;
        sc->vehicleSteps += sc->active;  // everybody counted as active took this step's phase 4
        sc->finishedCnt += F;
        sc->active -= F + sc->nLeftUncounted;
        finCountsClear(finCount);
        sc->nLeftUncounted = 0;
    }
    return last;
}

__global__ CFX_SCAN_BOUNDS void k_scan(int D, int L, const int32_t *cnt, CompactScratch cs,
                                                 unsigned long long *granules, int32_t *ticket, unsigned epoch,
                                                 int32_t *segStartNext, int32_t *cntNext, int32_t *vidNext,
                                                 int32_t *drvNext, DevScalars *sc, const uint8_t *laneSpare,
                                                 const int32_t *admitStep, int step, int32_t *waitHead, VidTable vt,
                                                 const uint8_t *laneGhost, const int2 *admitRec, int32_t *hostCnt, int hostAll) {
    __shared__ int smem[kBlock / 64];
    __shared__ int wsum[kBlock / 64];
    __shared__ int tileShared;
    int tile = (int) blockIdx.x;
    if (ticket) {  // grids too large to be co-resident: hand tiles out in start order
        if (threadIdx.x == 0) tileShared = atomicAdd(ticket, 1);
        __syncthreads();
        tile = tileShared;
        __syncthreads();
    }

    const int base = tile * kScanTile + threadIdx.x * kScanItems;
    int vals[kScanItems];
    int live[kScanItems];
    int sum = 0;
    // Loads first, 16 bytes at a time (a thread's 8 drivables are contiguous and the arrays are padded to whole tiles),
    // in one round; the admissions are committed (stores) after the loop.
    unsigned admittedMask = 0;
    int cntv[kScanItems], leavev[kScanItems], inv[kScanItems], admv[kScanItems];
    int2 rec[kScanItems];
    {
        const int4 *pc = (const int4 *) (cnt + base), *pl = (const int4 *) (cs.leaveCnt + base),
                   *pi = (const int4 *) (cs.inCnt + base), *pa = (const int4 *) (admitStep + base),
                   *pr = (const int4 *) (admitRec + base);
        const int4 c0 = pc[0], c1 = pc[1], l0 = pl[0], l1 = pl[1], i0 = pi[0], i1 = pi[1], a0 = pa[0], a1 = pa[1];
        const int4 r0 = pr[0], r1 = pr[1], r2 = pr[2], r3 = pr[3];
        cntv[0] = c0.x; cntv[1] = c0.y; cntv[2] = c0.z; cntv[3] = c0.w; cntv[4] = c1.x; cntv[5] = c1.y; cntv[6] = c1.z; cntv[7] = c1.w;
        leavev[0] = l0.x; leavev[1] = l0.y; leavev[2] = l0.z; leavev[3] = l0.w; leavev[4] = l1.x; leavev[5] = l1.y; leavev[6] = l1.z; leavev[7] = l1.w;
        inv[0] = i0.x; inv[1] = i0.y; inv[2] = i0.z; inv[3] = i0.w; inv[4] = i1.x; inv[5] = i1.y; inv[6] = i1.z; inv[7] = i1.w;
        admv[0] = a0.x; admv[1] = a0.y; admv[2] = a0.z; admv[3] = a0.w; admv[4] = a1.x; admv[5] = a1.y; admv[6] = a1.z; admv[7] = a1.w;
        rec[0] = make_int2(r0.x, r0.y); rec[1] = make_int2(r0.z, r0.w); rec[2] = make_int2(r1.x, r1.y); rec[3] = make_int2(r1.z, r1.w);
        rec[4] = make_int2(r2.x, r2.y); rec[5] = make_int2(r2.z, r2.w); rec[6] = make_int2(r3.x, r3.y); rec[7] = make_int2(r3.z, r3.w);
    }
    static_assert(kScanItems == 8, "k_scan loads its 8 items as two int4");
    for (int i = 0; i < kScanItems; ++i) {
        int d = base + i;
        const int admitted = (d < L && admv[i] == step) ? 1 : 0;
        admittedMask |= (unsigned) admitted << i;
        int nl = d < D ? cntv[i] + admitted - leavev[i] + inv[i] : 0;
        live[i] = nl;
        vals[i] = d < D ? nl + (d < L ? (laneSpare ? (int) laneSpare[d] : 1) : 0) : 0;
        sum += vals[i];
    }
    int nAdm = 0;
    if (admittedMask) {
        // commit this step's admissions (phase 2): the FIFO pop, the vehicle's state and the running count
        for (int i = 0; i < kScanItems; ++i)
            if ((admittedMask >> i) & 1u) {
                const int d = base + i;
                waitHead[d] = rec[i].y;
                vt.state[rec[i].x] = 1;
                // tiling: an admission onto a ghost lane only mirrors the owner's (same queue, same tail, same decision)
                nAdm += !(laneGhost && laneGhost[d]);
            }
    }
    {   // Engine::activeVehicleCount: one atomic per wavefront (a large network admits thousands of vehicles per step)
        int waveAdm = nAdm;
        for (int off = 32; off > 0; off >>= 1) waveAdm += __shfl_down(waveAdm, off, 64);
        if ((threadIdx.x & 63) == 0 && waveAdm) atomicAdd((unsigned long long *) &sc->active, (unsigned long long) waveAdm);
    }
    // in-wave inclusive scan of the per-thread sums, wave totals to LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < kBlock / 64; ++i) {
            smem[i] = run;
            run += wsum[i];
        }
        // publish this tile's total: ONE 8-byte agent-scope store {epoch, total}
        __hip_atomic_store(&granules[tile], ((unsigned long long) epoch << 32) | (unsigned) run, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int waveOff = smem[w];
    __syncthreads();
    // sum of the predecessors' totals
    int pre = 0;
    for (int p = threadIdx.x; p < tile; p += blockDim.x) {
        unsigned spins = 0;
        for (;;) {
            unsigned long long x = __hip_atomic_load(&granules[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned) (x >> 32) == epoch) {
                pre += (int) (unsigned) x;
                break;
            }
            if (++spins > kSpinLimit) {  // cannot happen unless a tile died; do not hang the device
                sc->overflow = 2;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const int tileOff = blockReduceSum(pre, smem);
    int off0 = tileOff + waveOff + incl - sum;
    int offs[kScanItems];
    for (int i = 0; i < kScanItems; ++i) {
        offs[i] = off0;
        for (int j = live[i]; j < vals[i]; ++j) {  // the lane's spare slot(s) of the next generation (vals is 0 past D)
            vidNext[off0 + j] = -1;
            drvNext[off0 + j] = -1;
        }
        off0 += vals[i];
    }
    // Two 16-byte stores per array.  Past D the entries are padding: live = 0 and the offset stays at the grand total,
    // which is exactly what segStartNext[D] has to hold.
    int4 *ps = (int4 *) (segStartNext + base), *pn = (int4 *) (cntNext + base);
    ps[0] = make_int4(offs[0], offs[1], offs[2], offs[3]);
    ps[1] = make_int4(offs[4], offs[5], offs[6], offs[7]);
    pn[0] = make_int4(live[0], live[1], live[2], live[3]);
    pn[1] = make_int4(live[4], live[5], live[6], live[7]);
    if (hostCnt)  // a caller observes the lane counts: the lanes whose count changed, straight into its pinned host array
        for (int i = 0; i < kScanItems; ++i)
            if (base + i < L && (hostAll || live[i] != cntv[i])) hostCnt[base + i] = live[i];
    if (base + kScanItems == D) segStartNext[D] = off0;  // D a multiple of 8 and this is the last thread with work
}

// Phase 5c + 6: stable compaction into the next generation and commit of the buffered action
// (Engine::threadUpdateLocation / updateLocation engine.cpp:282-315,477-494; Vehicle::update
// vehicle.cpp:107-143; Router::update router.cpp:78-94).  Low thread ids also advance the traffic
// lights (TrafficLight::passTime trafficlight.cpp:29-37) and clear the active-laneLink masks.
template <bool LC>  // (lane change as a separate instantiation: the common configuration carries none of its registers)
__global__ CFX_SCATTER_BOUNDS void k_scatter(StepCtx c, ActionBuf b, CompactScratch cs, SlotArrays nx, const int32_t *segStartNext,
                          int32_t *oldToNew, int32_t *curPhase, double *remain, int rlTrafficLight, int nMaskWords,
                          int32_t *scanTicket, VidTable vt, DevScalars *sc, const int32_t *finList, double *finTerm,
                          int finCap, int32_t *jobCount, HostMirror *hostMirror, int32_t *finTicket, int nStatBlocks,
                          int exactTimes, const int32_t *cntNext, int32_t *finCount) {
    // The launch carries extra blocks that only do the step's finish statistics (they read just the current
    // generation and the finish list, both complete before this kernel starts), in parallel with the compaction.
    const int nBody = (int) gridDim.x - nStatBlocks;
    if ((int) blockIdx.x >= nBody) {
        const int part = (int) blockIdx.x - nBody;
        if (part == 0 && threadIdx.x < kJobShards) jobCount[threadIdx.x * kJobShardStride] = 0;  // k_cross of this step is done
        const bool last = finishStatistics(c, vt, sc, finList, finTerm, finCap, finTicket, part, nStatBlocks, exactTimes, finCount);
        if (last && threadIdx.x == 0 && hostMirror) {
            // the step's scalars and slot count, also left in pinned host memory: a getter then needs the stream
            // synchronisation only, not a device-to-host copy on top of it
            hostMirror->sc = *sc;
            hostMirror->slots = segStartNext[c.n.L + c.n.K];
            __hip_atomic_store(&hostMirror->progress, ((unsigned long long) (c.step + 1) << 32) | (unsigned) sc->active,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = nBody * blockDim.x;
    if (gid == 0 && scanTicket) *scanTicket = 0;  // k_scan of this step is done; re-arm it for the next one
    if (LC && gid == 0 && c.lc.on) {  // the lane-change lists of this step are consumed
        *c.lc.insCount = 0;
        *c.lc.candAllCount = 0;
        *c.lc.insLaneCount = 0;
        *c.lc.fixCount = 0;
    }
    for (int i = gid; i < nMaskWords; i += stride) c.interMask[i] = 0ULL;
    if (!rlTrafficLight) {
        for (int i = gid; i < c.n.I; i += stride) {
            if (c.n.interVirtual[i]) continue;
            int ps = c.n.interPhaseStart[i];
            int np = c.n.interPhaseStart[i + 1] - ps;
            double rem = remain[i] - c.interval;
            int ph = curPhase[i];
            while (rem <= 0.0) {
                ph = (ph + 1) % np;
                rem += c.n.phaseTime[ps + ph];
            }
            remain[i] = rem;
            curPhase[i] = ph;
        }
    }
    const int S = c.segStart[c.n.L + c.n.K];
    for (int s = gid; s < S; s += stride) {
        // Round 1: everything indexed by the slot itself, issued before the first branch (the kernel is bound by
        // dependent-load rounds; spare slots hold garbage in most of these arrays, which is never used).
        const int vid = c.s.vid[s];
        const int nd = b.drv[s];
        const int d = c.s.drv[s];
        const int templ = c.s.templ[s];
        const double ndis = b.dis[s];
        const double nspeed = b.speed[s];
        const int nblocker = b.blocker[s];
        const int route = c.s.route[s];
        const int prevDrv = c.s.prevDrv[s];
        const int next = c.s.next[s];
        const int enterLLT = c.s.enterLLT[s];
        const int oldFlags = c.s.flags[s];
        int rp = c.s.routePos[s];
        if (vid < 0 || nd == -2) {  // spare slot, or finished: removed
            oldToNew[s] = -1;
            continue;
        }
        // Round 2: per-drivable values of the segment the vehicle ends up in (its own when it stays)
        const int t = nd >= 0 ? nd : d;
        const int tStartNext = segStartNext[t];
        const int tLeave = cs.leaveCnt[t];
        int ns;
        if (nd == -1) {
            // stays: rank among the stayers of its segment = k - (#leavers in front of it)
            const int k = s - c.segStart[d];
            int before;
            if (tLeave == 0) {
                before = 0;
            } else if (cs.maxLeaveIdx[d] + 1 == tLeave) {
                before = k < tLeave ? k : tLeave;  // leavers form a prefix (the normal case)
            } else {
                before = 0;
                for (int j = c.segStart[d]; j < s; ++j) before += (b.drv[j] != -1);
            }
            ns = tStartNext + (k - before);
        } else {
            // enters drivable nd: after its stayers, ordered by new distance descending
            // (std::sort with vehicleCmp engine.h:21-23; ties: lower vid first, as in the twin)
            int rank = 0;
            for (int j = cs.inHead[nd]; j >= 0; j = cs.inNext[j]) {
                if (j == s) continue;
                double od = b.dis[j];
                const bool tieBefore = od == ndis && c.s.vid[j] < vid;
                rank += (od > ndis) || tieBefore;
                if (tieBefore) sc->tieDrv[atomicAdd((unsigned long long *) &sc->tieEvents, 1ULL) & 7ULL] = nd;
            }
            ns = tStartNext + (cntNow(c, nd) - tLeave) + rank;
        }
        oldToNew[s] = ns;
        if (c.tailW && ns == tStartNext + cntNext[t] - 1) {
            // the vehicle that ends up last in its drivable leaves the tail record of this step's end (cfx_dense_kernels.h)
            TailRec r;
            r.dis = ndis;
            r.speed = nspeed;
            r.slot = ns;
            r.templ = templ;
            r.prevDrv = nd == -1 ? prevDrv : d;
            r.tag = c.step;
            c.tailW[t] = r;
        }
        nx.vid[ns] = vid;
        nx.templ[ns] = templ;
        nx.dis[ns] = ndis;
        nx.speed[ns] = nspeed;
        nx.blocker[ns] = nblocker;  // old-generation slot; resolved through oldToNew when read
        nx.route[ns] = route;
        if (nd == -1) {
            nx.flags[ns] = (uint8_t) (oldFlags & 2);  // Vehicle::update clears isCustomSpeedSet (vehicle.cpp:120-122); bit 1: lastRoadBit
            nx.drv[ns] = d;
            nx.prevDrv[ns] = prevDrv;
            nx.next[ns] = next;
            nx.enterLLT[ns] = enterLLT;
            nx.routePos[ns] = rp;
        } else {
            nx.drv[ns] = nd;
            nx.prevDrv[ns] = d;
            if (nd < c.n.L) {
                nx.enterLLT[ns] = CFX_INT_MAX;
                const int base = c.t.routeStart[route], n = c.t.routeStart[route + 1] - base;
                const int road = c.n.laneRoad[nd];
                while (rp < n && c.t.routeRoads[base + rp] != road) ++rp;
            } else {
                nx.enterLLT[ns] = c.step;
            }
            nx.routePos[ns] = rp;
            const int nextNew = nextOf(c.n, c.t, nd, route, rp);
            nx.next[ns] = nextNew;
            nx.flags[ns] = (uint8_t) lastRoadBit(c, nd, route, nextNew);
        }
        if (LC && c.lc.on) {
            // threadUpdateAction's clearSignal (engine.cpp:424, lanechange.cpp:129-138) for every vehicle that stays in the
            // network, and where it now is.  (Nothing in this kernel reads these tables.)
            const LcDev &lc = c.lc;
            lc.newToOld[ns] = s;  // (k_lc_insert: a vehicle it moves corrects its oldToNew entry)
            if (t < c.n.L) lc.segOfSlot[ns] = lcNaiveSegment(c, t, ndis);  // (lcInitSegments of the next step)
            lc.slotOf[vid] = ns;
            lc.tLeader[vid] = -1;
            lc.tFollower[vid] = -1;
            lc.lastDir[vid] = lc.sigSend[vid] ? lc.sendDir[vid] : 0;
            if (!lc.changing[vid]) {
                lc.sigSend[vid] = 0;
                lc.recvFrom[vid] = -1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- control
// Vehicle::setCustomSpeed (vehicle.h:128-131) for a running vehicle: find its slot and raise the flag.
__global__ void k_set_speed(StepCtx c, int vid) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride)
        if (c.s.vid[s] == vid) c.s.flags[s] |= 1;
}

// Router::setRoute (router.cpp:245-255) after the host's checks: new route, iCurRoad = begin
__global__ void k_set_route(StepCtx c, int vid, int route) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride)
        if (c.s.vid[s] == vid) {
            c.s.route[s] = route;
            c.s.routePos[s] = 0;
            const int next = nextOf(c.n, c.t, c.s.drv[s], route, 0);
            c.s.next[s] = next;
            c.s.flags[s] = (uint8_t) ((c.s.flags[s] & (kFlagCustom | kFlagStateGap)) | lastRoadBit(c, c.s.drv[s], route, next));
        }
}

// TrafficLight::setPhase (trafficlight.cpp:39-41) for n (intersection, phase) pairs read from pinned host memory
__global__ void k_set_phases(const int32_t *pairs, int n, int32_t *curPhase) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) curPhase[pairs[i]] = pairs[n + i];
}

__global__ void k_refresh_next(StepCtx c) {  // after cfx_load_state: Router::getNextDrivable(0) of every vehicle
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride)
        if (c.s.vid[s] >= 0) {
            const int d = c.s.drv[s], route = c.s.route[s], next = nextOf(c.n, c.t, d, route, c.s.routePos[s]);
            c.s.next[s] = next;
            c.s.flags[s] = (uint8_t) ((c.s.flags[s] & (kFlagCustom | kFlagStateGap)) | lastRoadBit(c, d, route, next));
        }
}

__global__ void k_find_vehicle(StepCtx c, int vid, int32_t *out /*[2]: drivable, routePos*/) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride)
        if (c.s.vid[s] == vid) {
            out[0] = c.s.drv[s];
            out[1] = c.s.routePos[s];
        }
}

// ---------------------------------------------------------------------------------------------- tiling halo
// One road network over several engines (include/cityflow_amd.h, "Tiling").  Runs on the generation produced by
// this step's k_scatter; cs.inCnt still holds the step's per-drivable entrant counts.
struct HaloDev {
    int nGhost, nImport;
    // block addresses are (peer, offset inside that peer's message); the staged path is one pseudo-peer 0 whose
    // "message" is the whole send / recv buffer
    const int32_t *ghostLane, *ghostSendOff, *ghostRecvOff, *importLane, *importRecvOff, *importSendOff;
    const int32_t *ghostPeer, *importPeer;
    const int32_t *llGlobal;         // [K] local laneLink -> global id
    const int32_t *llLocalOfGlobal;  // [global K] -> local laneLink or -1
    uint8_t *ghostHadEntrants;       // [nGhost] this step's export found entrants (the proxy is already current)
};

// Where this step's messages live: device staging buffers (cfx_halo_export / _import) or the peers' mailboxes in
// shared host memory (cfx_halo_post / _wait), and what to wait for before importing.
struct HaloIO {
    char *send[CFX_HALO_MAX_PEERS];
    const char *recv[CFX_HALO_MAX_PEERS];
    const unsigned long long *waitFlag[CFX_HALO_MAX_PEERS];
    int nWait;
    unsigned long long *signalFlag[CFX_HALO_MAX_PEERS];  // epoch word of every send mailbox (mailbox path only)
    int nSignal;
    int32_t *ticket;  // arrival counter of the export threads (re-armed by the last one)
    unsigned long long epoch;
};

struct HaloMigrant {
    int32_t vid, routePos, prevLL, pad;
    double dis, speed;
};
struct HaloTail {
    int32_t vid, prevLL;
    double dis, speed;
};
static_assert(sizeof(HaloMigrant) == 32 && sizeof(HaloTail) == CFX_HALO_TAIL_BYTES, "halo record layout");

// A slot whose vehicle now lives in a neighbouring tile: empty for every kernel (vid < 0), but it keeps the vehicle's
// identity as a tombstone -(vid + 2), so a committed blocker that still points here reports the right vehicle.
__device__ inline void haloClearSlot(const SlotArrays &s, int slot) {
    const int v = s.vid[slot];
    s.vid[slot] = v >= 0 ? -(v + 2) : v;
    s.drv[slot] = -1;
    s.blocker[slot] = -1;
}

// Tiles on the dense kernels with tail records (kd_admit / kd_action): the halo changes a cut lane's last vehicle after the
// step's scatter has written the lane's tail record — the proxy of a ghost lane, the migrants appended to an import lane —
// so the halo kernels rewrite that record.  They run after cfx_step has returned: `c` is already the NEXT step's context and
// the records of the step that has just finished are its tailR (tag = c.step - 1).
__device__ inline void haloWriteTail(const StepCtx &c, int lane, int slot) {
    if (!c.tailR) return;
    TailRec r{};
    r.slot = -1;
    if (slot >= 0) {
        r.dis = c.s.dis[slot];
        r.speed = c.s.speed[slot];
        r.slot = slot;
        r.templ = c.s.templ[slot];
        r.prevDrv = c.s.prevDrv[slot];
    }
    r.tag = c.step - 1;
    const_cast<TailRec *>(c.tailR)[lane] = r;
}

__device__ inline int haloGlobalPrev(const StepCtx &c, const HaloDev &h, int prevDrv) {
    if (prevDrv >= c.n.L) return h.llGlobal[prevDrv - c.n.L];
    if (prevDrv <= -2) return -prevDrv - 2;  // a migrant's laneLink of origin, kept as its global id
    return -1;
}

__device__ inline void haloExportLane(const StepCtx &c, int32_t *cnt, const HaloDev &h, const int32_t *inCnt, const HaloIO &io,
                                      DevScalars *sc, int i) {
    if (i < h.nGhost) {
        // upstream side: the vehicles that entered the ghost lane this step are the last `in` of its segment
        // (entrants are appended behind the stayers, already sorted like Lane::vehicles)
        const int g = h.ghostLane[i];
        const int base = c.segStart[g], n = cnt[g], in = inCnt[g];
        char *blk = io.send[h.ghostPeer[i]] + h.ghostSendOff[i];
        int m = in;
        if (m > CFX_HALO_MAX_MIGRANTS) {
            m = CFX_HALO_MAX_MIGRANTS;
            sc->overflow = 3;
        }
        ((int32_t *) blk)[0] = m;
        ((int32_t *) blk)[1] = 0;
        HaloMigrant *rec = (HaloMigrant *) (blk + 8);
        for (int j = 0; j < m; ++j) {
            const int s = base + n - in + j;
            HaloMigrant r;
            r.vid = c.s.vid[s];
            r.routePos = c.s.routePos[s];
            r.prevLL = haloGlobalPrev(c, h, c.s.prevDrv[s]);
            r.pad = 0;
            r.dis = c.s.dis[s];
            r.speed = c.s.speed[s];
            rec[j] = r;
        }
        h.ghostHadEntrants[i] = in > 0;
        if (in > 0) {
            // keep only the new tail as this lane's proxy, in the segment's first slot
            const int last = base + n - 1;
            if (last != base) {
                c.s.vid[base] = c.s.vid[last];
                c.s.prevDrv[base] = c.s.prevDrv[last];
                c.s.next[base] = c.s.next[last];
                c.s.enterLLT[base] = c.s.enterLLT[last];
                c.s.routePos[base] = c.s.routePos[last];
                c.s.templ[base] = c.s.templ[last];
                c.s.route[base] = c.s.route[last];
                c.s.dis[base] = c.s.dis[last];
                c.s.speed[base] = c.s.speed[last];
            }
            c.s.drv[base] = g;
            c.s.blocker[base] = -1;
            c.s.flags[base] = 0;
            for (int s = base + 1; s < base + n; ++s) haloClearSlot(c.s, s);
            cnt[g] = 1;
            atomicAdd((unsigned long long *) &sc->active, (unsigned long long) (-(long long) in));
            haloWriteTail(c, g, base);
        }
        return;
    }
    const int j = i - h.nGhost;
    if (j < h.nImport) {
        // downstream side: report the lane's tail (before this step's migrants are appended)
        const int l = h.importLane[j];
        const int n = cnt[l];
        HaloTail t;
        if (n > 0) {
            const int s = c.segStart[l] + n - 1;
            t.vid = c.s.vid[s];
            t.prevLL = haloGlobalPrev(c, h, c.s.prevDrv[s]);
            t.dis = c.s.dis[s];
            t.speed = c.s.speed[s];
        } else {
            t.vid = -1;
            t.prevLL = -1;
            t.dis = 0.0;
            t.speed = 0.0;
        }
        *(HaloTail *) (io.send[h.importPeer[j]] + h.importSendOff[j]) = t;
    }
}

// One thread per cut lane.  On the mailbox path the last thread to finish publishes the step's epoch in every peer's
// mailbox: each thread fences its stores at system scope before taking a ticket, the last one fences again after
// reading the final ticket and then does the release stores, so the peers' acquire loads see complete messages.
__global__ void k_halo_export(StepCtx c, int32_t *cnt, HaloDev h, const int32_t *inCnt, HaloIO io, DevScalars *sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = h.nGhost + h.nImport;
    if (i >= n) return;
    haloExportLane(c, cnt, h, inCnt, io, sc, i);
    if (io.nSignal == 0) return;
    __threadfence_system();
    if (atomicAdd(io.ticket, 1) != n - 1) return;
    __threadfence_system();
    *io.ticket = 0;
    for (int p = 0; p < io.nSignal; ++p)
        __hip_atomic_store(io.signalFlag[p], io.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_halo_import(StepCtx c, int32_t *cnt, HaloDev h, HaloIO io, VidTable vt, DevScalars *sc) {
    if (io.nWait > 0) {  // mailbox path: every block waits until all peers have published this epoch
        __shared__ int ok;
        if (threadIdx.x == 0) {
            ok = 1;
            for (int p = 0; p < io.nWait && ok; ++p) {
                unsigned spins = 0;
                while (__hip_atomic_load(io.waitFlag[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < io.epoch) {
                    if (++spins > (1u << 22)) {  // a peer died or fell far behind: flag it instead of hanging the device
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        __syncthreads();
        if (!ok) {
            sc->overflow = 4;
            return;
        }
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < h.nImport) {
        const int l = h.importLane[i];
        const char *blk = io.recv[h.importPeer[i]] + h.importRecvOff[i];
        const int m = ((const int32_t *) blk)[0];
        const HaloMigrant *rec = (const HaloMigrant *) (blk + 8);
        const int base = c.segStart[l], n = cnt[l];
        for (int j = 0; j < m; ++j) {
            const HaloMigrant r = rec[j];
            const int s = base + n + j;
            const int route = vt.route[r.vid];
            c.s.vid[s] = r.vid;
            c.s.drv[s] = l;
            c.s.prevDrv[s] = r.prevLL >= 0 ? -(r.prevLL + 2) : -1;
            const int nextM = nextOf(c.n, c.t, l, route, r.routePos);
            c.s.next[s] = nextM;
            c.s.blocker[s] = -1;  // a vehicle that left its laneLink this step was not yielding (no blocker set)
            c.s.enterLLT[s] = CFX_INT_MAX;
            c.s.routePos[s] = r.routePos;
            c.s.templ[s] = vt.templ[r.vid];
            c.s.route[s] = route;
            c.s.flags[s] = (uint8_t) lastRoadBit(c, l, route, nextM);
            c.s.dis[s] = r.dis;
            c.s.speed[s] = r.speed;
            vt.state[r.vid] = 1;
        }
        if (m > 0) {
            cnt[l] = n + m;
            atomicAdd((unsigned long long *) &sc->active, (unsigned long long) m);
            haloWriteTail(c, l, base + n + m - 1);
        }
        return;
    }
    const int j = i - h.nImport;
    if (j < h.nGhost && !h.ghostHadEntrants[j]) {
        // no entrant of our own this step: the owner's tail is the lane's tail
        const int g = h.ghostLane[j];
        const HaloTail t = *(const HaloTail *) (io.recv[h.ghostPeer[j]] + h.ghostRecvOff[j]);
        const int base = c.segStart[g], n = cnt[g];
        for (int s = base + (t.vid >= 0 ? 1 : 0); s < base + n; ++s) haloClearSlot(c.s, s);
        if (t.vid < 0) {
            cnt[g] = 0;
            haloWriteTail(c, g, -1);
            return;
        }
        int prev = -1;
        if (t.prevLL >= 0) {
            const int k = h.llLocalOfGlobal[t.prevLL];
            prev = k >= 0 ? c.n.L + k : -(t.prevLL + 2);
        }
        c.s.vid[base] = t.vid;
        c.s.drv[base] = g;
        c.s.prevDrv[base] = prev;
        c.s.next[base] = -1;
        c.s.blocker[base] = -1;
        c.s.enterLLT[base] = CFX_INT_MAX;
        c.s.routePos[base] = 0;
        c.s.templ[base] = vt.templ[t.vid];
        c.s.route[base] = vt.route[t.vid];
        c.s.flags[base] = 0;
        c.s.dis[base] = t.dis;
        c.s.speed[base] = t.speed;
        cnt[g] = 1;
        haloWriteTail(c, g, base);
    }
}

// ---------------------------------------------------------------------------------------------- getters
__global__ void k_leader_view(StepCtx c, int32_t *leaderSlot, double *gapOut) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride) {
        if (c.s.vid[s] < 0) {
            leaderSlot[s] = -1;
            continue;
        }
        int d = c.s.drv[s];
        double gap = 0;
        bool head = s == 0 || c.s.drv[s - 1] != d;
        leaderSlot[s] = findLeader(c, c.t.templ, s, d, head, c.s.dis[s], c.t.templ[c.s.templ[s]].approach_dist, &gap);
        if ((c.s.flags[s] & kFlagStateGap) && leaderSlot[s] >= 0) gap = c.vGapState[c.s.vid[s]];  // not stepped since the load
        gapOut[s] = gap;
    }
}

// Lane::updateHistory for every lane, as a part of Engine::threadUpdateLeaderAndGap (engine.cpp:429-442): at the end of a step,
// and with lane change also between planLaneChange and getAction (engine.cpp:571-575) — the lane lists then hold this step's
// admissions and shadows already (cntNow)
// cfx_get_lane_history / cfx_set_lane_history: the ABI's arrays — lane-major, oldest record first, zeros behind a lane's last
// record — from and to the device's record-major rings.  One thread per (record, lane), lanes fastest: the ring side coalesced.
__global__ void k_hist_export(LaneHistDev h, int32_t *num, double *avg) {
    const size_t q = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t) kLaneHistoryMax * h.L) return;
    const int l = (int) (q % h.L), i = (int) (q / h.L);
    const bool in = i < h.len[l];
    int r = h.head[l] + i;
    if (r >= kLaneHistoryMax) r -= kLaneHistoryMax;
    num[(size_t) l * kLaneHistoryMax + i] = in ? h.num[(size_t) r * h.L + l] : 0;
    avg[(size_t) l * kLaneHistoryMax + i] = in ? h.avg[(size_t) r * h.L + l] : 0.0;
}
__global__ void k_hist_import(LaneHistDev h, const int32_t *num, const double *avg) {  // (h.len is in place; the rings start at record 0)
    const size_t q = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t) kLaneHistoryMax * h.L) return;
    const int l = (int) (q % h.L), i = (int) (q / h.L);
    const bool in = i < h.len[l];
    h.num[(size_t) i * h.L + l] = in ? num[(size_t) l * kLaneHistoryMax + i] : 0;
    h.avg[(size_t) i * h.L + l] = in ? avg[(size_t) l * kLaneHistoryMax + i] : 0.0;
    if (i == 0) h.head[l] = 0;
}

__global__ void k_lane_history(StepCtx c, LaneHistDev h) {
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n.L) return;
    const int base = c.segStart[lane];
    laneHistoryStep(h, lane, cntNow(c, lane), [&](int i) { return c.s.speed[base + i]; });
}

__global__ void k_lane_waiting(StepCtx c, int32_t *out) {  // Engine::getLaneWaitingVehicleCount engine.cpp:636-648
    int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n.L) return;
    int base = c.segStart[lane], n = c.cnt[lane], k = 0;
    for (int i = 0; i < n; ++i) k += c.s.speed[base + i] < 0.1;
    out[lane] = k;
}

// Initial / reset layout: every lane owns just its spare slot, laneLinks are empty.
__global__ void k_init_layout(int D, int L, int32_t *segStart, int32_t *cnt, int32_t *vid, int32_t *drv) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > D) return;
    segStart[d] = d < L ? d : L;
    if (d < D) cnt[d] = 0;
    if (d < L) {
        vid[d] = -1;
        drv[d] = -1;
    }
}

__global__ void k_init_lights(DevNet n, int32_t *curPhase, double *remain) {  // TrafficLight::init trafficlight.cpp:6-11
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n.I) return;
    curPhase[i] = 0;
    remain[i] = n.interVirtual[i] ? 0.0 : n.phaseTime[n.interPhaseStart[i]];
}

}  // namespace cfxd
