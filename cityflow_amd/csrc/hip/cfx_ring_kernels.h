// RING layout of the vehicle order (cfx_config::layout = CFX_LAYOUT_RING; default whenever lane change is off).
//
// The reference keeps one std::list<Vehicle*> per drivable (roadnet.h:245); per step a vehicle either stays in its list,
// leaves it at the FRONT (it ran past the end of the drivable, engine.cpp:290-310) or is appended at the BACK of another
// one (engine.cpp:477-494; handleWaiting appends at most one, engine.cpp:502-516): a drivable is a deque.  So every
// drivable owns a fixed ring of slots in HBM
//     slots  base[d] .. base[d] + cap[d] - 1      cap a power of two, base aligned to it
//     live   base + ((head + i) & (cap - 1)),  i = 0 .. cnt - 1,  i = 0 the front (furthest ahead)
// and a step commits IN PLACE: a vehicle that stays keeps its slot and only its new {dis, speed} are written (into the
// other generation of those two arrays, 16 B); everything else a vehicle carries (vid, template, route cursor, ...) is
// written only when it enters a drivable (3-5 % of the vehicles per step).  There is no global scan and no scatter:
// `kr_commit` advances `head` past the leavers, appends the entrants behind the stayers in the reference's order and
// adjusts `cnt`, one thread per drivable.  The in-lane leader of a vehicle is the previous ring position — adjacent in
// memory, so a wave reads its leaders' {dis, speed, template} from the window it has staged in LDS.
//
// Blockers are kept by vehicle id with the step they were set in: Vehicle::update drops a blocker that was not set in
// that very step (vehicle.cpp:133-138), so a record is valid for exactly one step and nothing ever has to be cleared;
// two buffers alternate by step parity, so the records a step writes never overwrite the ones it reads.
//
// The arithmetic is the same code as the dense layout's (cfx_kernels.h: actionOne, canPassActive, findLeader, ...),
// instantiated on this file's context type.
#pragma once

#include <type_traits>

#include "cfx_kernels.h"


namespace cfxd {

struct RingCtx {
    // (k_cross2 on the ring layout needs 95 registers; asking for 7 wavefronts per SIMD spills and was measured slower,
    // 66 -> 72 us at 1 M vehicles: 5 = as the compiler has it)
#ifndef CFX_RING_CROSS2_WAVES
#define CFX_RING_CROSS2_WAVES 5  // (6: 80 registers, 21 spilled, 47.0 us at 1 M vehicles; 7: 72 / 44, 52.5 us; 5: 96 / 9, 43.7 us.  Round 6: 4 = 115
                                 //  registers, nothing spilled, no scratch segment — 37.9 against 38.2 us right behind a load (tools/exp_big.py),
                                 //  but 47.1-47.8 against 43.8 us in bench.py's 50-step leg: the spills are cheaper than the fifth wavefront)
#endif
    static constexpr int kCross2Waves = CFX_RING_CROSS2_WAVES;
    DevNet n;
    DevTables t;
    SlotArrays s;            // the COLD columns only: vid, drv, prevDrv, routePos, route (the rest is null here)
    // what every vehicle's action reads, as two 16-byte records per slot (a ring is sparsely filled: five separate
    // columns cost five partly used lines per lane, two records cost two):
    double2 *kin;            // [slot] {dis, speed}, the CURRENT generation
    double2 *kinN;           // the NEXT generation (all a vehicle that stays on its drivable writes)
    int4 *meta;              // [slot] {template, next drivable (Router::getNextDrivable(0), cached), flags (bit 0: a custom
                             // speed is pending), ControllerInfo::enterLaneLinkTime}; written when a vehicle enters a drivable
    // [slot] {blocker vid, step it was set in}, two buffers by step parity: a step WRITES the blockers it sets into
    // blkW (parity of this step) while every reader — the deadlock walk of Cross::canPass — looks at what the previous step
    // left in blkR, as the reference's buffered commit does (Vehicle::update vehicle.cpp:133-138)
    const int2 *blkR;
    int2 *blkW;
    int32_t *slotOf;         // [vid] slot of a running vehicle (-1: not on this engine)
    const uint8_t *vState;   // [vid] 0 waiting, 1 running, 2 finished
    const int2 *ringGeo;     // [D] {base, cap - 1}
    // tails: two buffers by step parity (a step reads what the previous one left in tailR and writes tailW), plus the view
    // of THIS step after its admissions (written by kr_admit for every drivable, valid while the step runs)
    const TailRec *tailR;
    TailRec *tailW;
    TailRec *tailNow;
    int betweenSteps;        // getters: no admission is pending, tailNow is stale — the committed records are the tails
    int32_t *head;           // [D] ring index of the front vehicle
    int32_t *cnt;            // [D] live vehicles (this step's admission excluded until kr_commit, see cntNow)
    const int32_t *admitStep;
    const int32_t *curPhase;
    const int32_t *vPriority;
    const double *vCustomSpeed;
    const double *vGapState;  // as StepCtx::vGapState
    int4 *llDyn;
    struct LLAux *llAux;     // [K] what a cross needs of a laneLink's notify sources beyond llDyn (written where active)
    unsigned long long *interMask;
    int4 *llGate;            // [K] {light | type | has crosses, end lane, first cross entry, end of cross entries}
    // [2 I] the phase each intersection's gate records stand for (-1: none), by step parity: kr_admit rewrites a laneLink's
    // record only when its intersection shows another phase than in the last step (kd_admit<true> has the story)
    int32_t *gatePhase;
    int32_t *laneTail;
    int2 *admitRec;
    int32_t step;
    double interval;
};

__device__ __forceinline__ int ringSlot(const int2 geo, int head, int i) { return geo.x + ((head + i) & geo.y); }

__device__ __forceinline__ int committedCount(const RingCtx &c, int d) { return c.cnt[d]; }
__device__ __forceinline__ int cntNow(const RingCtx &c, int d) {
    return c.cnt[d] + ((d < c.n.L && c.admitStep[d] == c.step) ? 1 : 0);
}
__device__ __forceinline__ int firstSlot(const RingCtx &c, int d) { return ringSlot(c.ringGeo[d], c.head[d], 0); }
__device__ __forceinline__ int lastSlot(const RingCtx &c, int d) {
    const int n = cntNow(c, d);
    return n > 0 ? ringSlot(c.ringGeo[d], c.head[d], n - 1) : -1;
}
__device__ __forceinline__ int lastSlotForLeader(const RingCtx &c, int d, bool viewerNew, int viewerLane) {
    int n = c.cnt[d];
    if (viewerNew && d < viewerLane && d < c.n.L && c.admitStep[d] == c.step) n += 1;
    return n > 0 ? ringSlot(c.ringGeo[d], c.head[d], n - 1) : -1;
}
__device__ __forceinline__ Tail tailCommitted(const RingCtx &c, int d) {  // Drivable::getLastVehicle after the last commit
    const TailRec r = c.tailR[d];
    Tail t = tailOfRec(r);
    if (r.tag != c.step - 1) t.slot = -1;
    return t;
}
__device__ __forceinline__ Tail tailNowOf(const RingCtx &c, int d) {
    if (c.betweenSteps) return tailCommitted(c, d);
    return tailOfRec(c.tailNow[d]);
}
__device__ __forceinline__ Tail tailForLeader(const RingCtx &c, int d, bool viewerNew, int viewerLane) {
    return (viewerNew && d < viewerLane && d < c.n.L) ? tailNowOf(c, d) : tailCommitted(c, d);
}
__device__ __forceinline__ int slotAhead(const RingCtx &c, int d, int s) {
    const int2 geo = c.ringGeo[d];
    return geo.x + ((s - geo.x - 1) & geo.y);
}
__device__ __forceinline__ SegWalk segWalk(const RingCtx &c, int d, int first) {
    const int2 geo = c.ringGeo[d];
    return SegWalk{first, geo.x, geo.y};
}
// ControllerInfo::blocker of the vehicle in `slot` as a slot (-1: none, expired, finished, or not on this engine)
__device__ __forceinline__ int blockerVid(const RingCtx &c, int slot) {
    const int2 b = c.blkR[slot];
    return (b.x >= 0 && b.y == c.step - 1 && c.vState[b.x] == 1) ? b.x : -1;
}
// (chain walks: a finished vehicle's slotOf entry is -1 — the finish statistics clear it — so two loads per link)
__device__ __forceinline__ int blockerOf(const RingCtx &c, int slot) {
    const int2 b = c.blkR[slot];
    return (b.x >= 0 && b.y == c.step - 1) ? c.slotOf[b.x] : -1;
}
__device__ __forceinline__ int keepBlocker(const RingCtx &, int blockerSlot) { return blockerSlot; }
// the shared cross-phase templates' view of a slot (cfx_kernels.h: slotDis ...)
__device__ __forceinline__ double slotDis(const RingCtx &c, int s) { return c.kin[s].x; }
__device__ __forceinline__ double slotSpeed(const RingCtx &c, int s) { return c.kin[s].y; }
__device__ __forceinline__ int slotTempl(const RingCtx &c, int s) { return c.meta[s].x; }
__device__ __forceinline__ int slotNext(const RingCtx &c, int s) { return c.meta[s].y; }
__device__ __forceinline__ int slotEnterLLT(const RingCtx &c, int s) { return c.meta[s].w; }

// What a vehicle that changes drivable leaves behind for kr_commit, at its OLD slot's index (written by 3-5 % of the
// vehicles; the ring arrays themselves are not touched, so the leavers' slots may be recycled while this is read).
struct MoverRec {
    int32_t vid, templ, route, routePos, oldDrv, newDrv, blockerVid, nextIn;  // nextIn: next entrant of the same drivable
    double dis, speed;
};
static_assert(sizeof(MoverRec) == 48, "mover record layout");

// Streaming accesses (experiment knob CFX_KL_NT, a bit mask: 1 the list entry read by kl_action, 2 the list entries written
// by kr_index, 4 the next-generation {dis, speed} record written by a stayer): data used once should not push the tail and
// gate records out of the L2.
#ifndef CFX_KL_NT
#define CFX_KL_NT 7  // (round 6, 1 M vehicles: kl_action 49.5 -> 47.9 us, kr_index 12.2 -> 10.9; profiles/r06_exp_kl_block_nt.txt)
#endif
typedef int cfx_v4i __attribute__((ext_vector_type(4)));
typedef double cfx_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int4 ntLoad4(const int4 *p) {
    const cfx_v4i v = __builtin_nontemporal_load((const cfx_v4i *) p);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 ntLoad2(const double2 *p) {
    const cfx_v2d v = __builtin_nontemporal_load((const cfx_v2d *) p);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void ntStore4(int4 *p, int4 x) {
    const cfx_v4i v = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(v, (cfx_v4i *) p);
}
__device__ __forceinline__ void ntStore2(double2 *p, double a, double b) {
    const cfx_v2d v = {a, b};
    __builtin_nontemporal_store(v, (cfx_v2d *) p);
}

struct RingOut {
    double2 *kinN;
    int2 *blk;
    int4 *scratch;       // [D] {vehicles leaving, largest list index among them, head of the entrant list, entrants}
    MoverRec *movers;    // [slot]
    DevScalars *sc;
    long long *finKey;   // finishers of the step: (drivable << 20 | list index) = the reference's removal order
    int32_t *finVid;
    int finCap;
    int32_t *finCount;   // [kFinShards * 32]: the finishers are listed per shard (FinMap, cfx_kernels.h)
    __device__ __forceinline__ void park(int s, double v, double iv) const { kinN[s] = make_double2(iv, v); }
    __device__ __forceinline__ double parkedSpeed(int s) const { return kinN[s].y; }
    __device__ __forceinline__ double parkedInterSpeed(int s) const { return kinN[s].x; }
    __device__ __forceinline__ void keep(int s, double dis, double speed) const { kinN[s] = make_double2(dis, speed); }
};
constexpr int kRingIdxBits = 20;  // list index inside a drivable (ring capacities stay far below 2^20)

// Tail of Engine::vehicleControl on the ring layout: stayers are committed right here (their slot does not move), leavers
// leave a MoverRec / a finish record and a mark in the next generation.
// Tail of Engine::vehicleControl on the ring layout (engine.cpp:212-221, Vehicle::setDeltaDistance vehicle.cpp:49-68):
// stayers are committed right here (their slot does not move), leavers leave a MoverRec / a finish record and a mark in
// the next generation.
template <bool LC>
__device__ inline void finishAction(const RingCtx &c, const RingOut &o, const cfx_vehicle_template &t, int s, int d, int /*vid*/,
                                    double speed, double dis, double dlen, int nd0, double v, int blockerSlot, int idx = -1,
                                    int nNow = -1, LeaverPrefetch lp = LeaverPrefetch{false, 0.0, 0, 0, 0}, int flags = -1) {
    static_assert(!LC, "the ring layout does not run lane change");
    v = min2(v, 100);  // SimpleLaneChange::yieldSpeed without signals
    // speedTail of cfx_kernels.h (vehicle.cpp:325-331).  "Is this the last road of the route" (Router::onValidLane, router.h:66-68)
    // is bit 1 of the slot's flags, set when the vehicle entered the lane: a vehicle on its last road otherwise walks
    // route -> routeStart -> routeRoads, three dependent loads in the middle of its wavefront's phase, every step
    if (nd0 < 0) {
        const bool lastRoad = flags >= 0 ? (flags & 2) != 0 : isLastRoad(c, d, c.s.route[s]);
        if (!lastRoad) v = min2(v, noCollisionSpeed(0, 1, speed, t.max_neg_acc, dlen - dis, c.interval, t.min_gap));
    }
    v = max2(v, speed - t.max_neg_acc * c.interval);
    // computeMove (cfx_kernels.h) with the first hop's two lengths in registers
    double deltaDis;
    if (v < 0) {
        deltaDis = 0.5 * speed * speed / t.max_neg_acc;
        v = 0;
    } else {
        deltaDis = (speed + v) * c.interval / 2;
    }
    double ndis = deltaDis + dis;
    int newDrv = -1;
    if (ndis > dlen) {
        if (!lp.valid) {
            lp.nextLen = nd0 >= 0 ? c.n.drvLength[nd0] : 0.0;
            lp.vid = c.s.vid[s];
            lp.route = c.s.route[s];
            lp.routePos = c.s.routePos[s];
        }
        ndis -= dlen;  // (== c.n.drvLength[d])
        int drivable = nd0;
        newDrv = drivable >= 0 ? drivable : -2;
        if (drivable >= 0 && ndis > lp.nextLen) {  // runs through a whole drivable in one step: the general walk goes on
            int nxt = nextOf(c.n, c.t, drivable, lp.route, lp.routePos);
            for (;;) {
                ndis -= c.n.drvLength[drivable];
                drivable = nxt;
                newDrv = drivable >= 0 ? drivable : -2;
                if (drivable < 0 || !(ndis > c.n.drvLength[drivable])) break;
                nxt = nextOf(c.n, c.t, drivable, lp.route, lp.routePos);
            }
        }
    }
    const int bv = blockerSlot >= 0 ? (lp.blockerVidP2 > 0 ? lp.blockerVidP2 - 2 : c.s.vid[blockerSlot]) : -1;
    if (idx < 0) {  // (cross phase of the generic kernels: the job carries the slot only)
        const int2 geo = c.ringGeo[d];
        idx = (s - geo.x - c.head[d]) & geo.y;
        nNow = cntNow(c, d);
    }
    if (newDrv == -1) {
        if (CFX_KL_NT & 4) ntStore2(&o.kinN[s], ndis, v);
        else o.kinN[s] = make_double2(ndis, v);
        if (bv >= 0) o.blk[s] = make_int2(bv, c.step);
        if (idx == nNow - 1) {  // the last vehicle of its drivable leaves the tail record of this step's end
            TailRec r;
            r.dis = ndis;
            r.speed = v;
            r.slot = s;
            r.templ = lp.templP1 > 0 ? lp.templP1 - 1 : c.meta[s].x;
            r.prevDrv = lp.prevValid ? lp.prevDrv : c.s.prevDrv[s];
            r.tag = c.step;
            c.tailW[d] = r;
        }
        return;
    }
    o.kinN[s].y = -1.0;  // "left its drivable": what kr_commit's general path looks at (a speed is never negative)
    atomicAdd(&o.scratch[d].x, 1);
    atomicMax(&o.scratch[d].y, idx);
    if (newDrv >= 0) {
        MoverRec r;
        r.vid = lp.vid;
        r.templ = lp.templP1 > 0 ? lp.templP1 - 1 : c.meta[s].x;
        r.route = lp.route;
        r.routePos = lp.routePos;
        r.oldDrv = d;
        r.newDrv = newDrv;
        r.blockerVid = bv;
        r.dis = ndis;
        r.speed = v;
        atomicAdd(&o.scratch[newDrv].w, 1);
        r.nextIn = atomicExch(&o.scratch[newDrv].z, s);
        o.movers[s] = r;
    } else {
        const int f = finPlace(o.finCount, o.finCap);
        if (f >= 0) {
            o.finKey[f] = ((long long) d << kRingIdxBits) | idx;
            o.finVid[f] = lp.vid;
        } else {
            o.sc->overflow = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------- phase 2
// Engine::handleWaiting engine.cpp:502-516 + Lane::available roadnet.cpp:428-435, one thread per drivable (laneLink
// threads write the gate records, as in the dense layout's k_admit).  The running count is raised here — the finish
// statistics of this step (extra blocks of kr_commit) read it while kr_commit's other blocks commit the rest.
// A step's few spawn records travel IN THE KERNEL ARGUMENTS of kr_admit (k_spawn_link's job, phase 0/1 tail, without its
// launch and without a read of host memory: every wave gets them where it gets its other arguments, through the scalar
// cache).  More records than fit: k_spawn_link runs first.
// The batch is kept by columns, SORTED BY LANE (a lane's thread finds its records by a binary search in the LDS copy of
// `lane[]`); record j is the vehicle firstNewVid + vidOff[j]; all vehicles of a step enter at the same time
// (Engine::getCurrentTime).  Kernel arguments stay at a few KB.
// (kAdmitRecs, SpawnBatch: cfx_device.h — the dense layouts' admission kernels take the batch too)
static_assert(sizeof(SpawnBatch) + sizeof(RingCtx) + sizeof(VidTable) + 256 <= 8192, "kr_admit's arguments: a few KB (the runtime takes 32 KB and more, tools/kernarg_probe.hip)");

struct RingCommit {
    int4 *scratch;
    const MoverRec *movers;
    int32_t *waitHead;
    int32_t *curPhase;
    double *remain;
    int rlTrafficLight, nMaskWords;
    DevScalars *sc;
    const long long *finKey;
    const int32_t *finVid;
    double *finTerm;
    int finCap;
    int32_t *jobCount;
    HostMirror *hostMirror;
    int32_t *finTicket;
    int nStatBlocks;
    uint8_t *vStateW;
    int32_t *slotOfW;  // a finished vehicle's entry becomes -1 (blocker chains end there)
    int exactTimes;  // every time involved is a multiple of 2^-10: the travel-time sum is order-free (exactFinishStatistics)
    int lightsDone;  // the step's cross kernel has already advanced the lights (kr_cross with lights.on)
    int32_t *hostCnt;  // pinned host copy of the lane counts, kept up by the commit while a caller observes them (or null)
    int32_t *finCount; // [kFinShards * 32] the finisher lists' counters
};
struct CommitOut {  // what a drivable's commit leaves, for the admission that follows it in the same thread
    int touched, head, n, tailWritten;
    int entrants;  // vehicles that entered the drivable in this step: the last `entrants` of its list (tiling: a ghost lane's migrants)
    TailRec tail;
};
__device__ inline void commitDrivable(const RingCtx &c, const RingCommit &k, const int d, CommitOut *out = nullptr);
__device__ inline void commitStatBlock(const RingCtx &c, const RingCommit &k, const VidTable &vt, int part);
__device__ inline int commitStatBlocks(const RingCommit &k);
__device__ inline void commitClearMasks(const RingCtx &c, const RingCommit &k, int gid, int stride);

// ---------------------------------------------------------------------------------------------- tiling on the rings
// One road network over several engines (include/cityflow_amd.h, "Tiling"; the dense layout's k_halo_export / k_halo_import
// are in cfx_kernels.h).  On the rings the EXPORT is part of the commit: the thread that commits a cut lane knows the lane's
// entrants of the step (a ghost lane's migrants: the last `entrants` of its list) and its new tail (an import lane's report),
// so the messages are written where those are known and no export kernel runs; the last cut lane to finish publishes the
// step's epoch in the peers' mailboxes.  The IMPORT stays a kernel (it has to wait for the neighbours' epochs): migrants are
// appended to the import lanes' rings, a ghost lane that had no entrant of its own takes its owner's tail as its proxy.
// A ghost lane holds at most its proxy (plus, for one step, a vehicle admitted here as on the owner's side): frozen, never
// stepped (actionOneRounds), leader and Lane::canEnter source for the vehicles upstream through its tail record.
struct RingHalo {
    int on;                   // 0: not a tile (everything below is unused)
    const int32_t *cutIndex;  // [L] -1: not cut; i < nGhost: ghost lane i; nGhost + j: import lane j
    HaloDev h;
    HaloIO io;
    long long *activeOut;     // vehicles that left this tile in the step (folded into DevScalars::active by the import kernel:
                              // the commit launch's own statistics block rewrites `active` while the lanes are committed)
};

__device__ inline int ringHaloGlobalPrev(const RingCtx &c, const HaloDev &h, int prevDrv) {
    if (prevDrv >= c.n.L) return h.llGlobal[prevDrv - c.n.L];
    if (prevDrv <= -2) return -prevDrv - 2;  // a migrant's laneLink of origin, kept as its global id
    return -1;
}

// The cut lane `d` after its commit (head / n as the commit left them, co.entrants = the step's entrants).
__device__ inline void ringHaloExport(const RingCtx &c, const RingCommit &k, const RingHalo &rh, int d, int ci, const CommitOut &co) {
    const HaloDev &h = rh.h;
    const int2 geo = c.ringGeo[d];
    int head = co.touched ? co.head : c.head[d];
    int n = co.touched ? co.n : c.cnt[d];
    if (ci < h.nGhost) {
        char *blk = rh.io.send[h.ghostPeer[ci]] + h.ghostSendOff[ci];
        const int in = co.entrants;
        int m = in;
        if (m > CFX_HALO_MAX_MIGRANTS) {
            m = CFX_HALO_MAX_MIGRANTS;
            k.sc->overflow = 3;
        }
        ((int32_t *) blk)[0] = m;
        ((int32_t *) blk)[1] = 0;
        HaloMigrant *rec = (HaloMigrant *) (blk + 8);
        for (int j = 0; j < m; ++j) {  // entrants are appended behind the stayers, already in Lane::vehicles order
            const int s = ringSlot(geo, head, n - in + j);
            const double2 kv = c.kinN[s];
            HaloMigrant r;
            r.vid = c.s.vid[s];
            r.routePos = c.s.routePos[s];
            r.prevLL = ringHaloGlobalPrev(c, h, c.s.prevDrv[s]);
            r.pad = 0;
            r.dis = kv.x;
            r.speed = kv.y;
            rec[j] = r;
        }
        h.ghostHadEntrants[ci] = in > 0;
        if (in > 0) {
            // keep only the new tail as this lane's proxy: the ring's head moves onto it (its slot, its records and the lane's
            // tail record, which the commit has just written, stay where they are); everybody else is no longer here
            for (int j = 0; j + 1 < n; ++j) c.slotOf[c.s.vid[ringSlot(geo, head, j)]] = -1;
            head = (head + n - 1) & geo.y;
            c.head[d] = head;
            c.cnt[d] = 1;
            atomicAdd((unsigned long long *) rh.activeOut, (unsigned long long) in);
        }
        return;
    }
    const int j = ci - h.nGhost;
    if (j < h.nImport) {  // downstream side: report the lane's tail (before this step's migrants are appended)
        HaloTail t;
        t.vid = -1;
        t.prevLL = -1;
        t.dis = 0.0;
        t.speed = 0.0;
        if (n > 0) {
            const int s = ringSlot(geo, head, n - 1);
            const double2 kv = c.kinN[s];
            t.vid = c.s.vid[s];
            t.prevLL = ringHaloGlobalPrev(c, h, c.s.prevDrv[s]);
            t.dis = kv.x;
            t.speed = kv.y;
        }
        *(HaloTail *) (rh.io.send[h.importPeer[j]] + h.importSendOff[j]) = t;
    }
}

// `c` is the NEXT step's context (cfx_step has returned): c.kin is the generation the step's commit wrote, c.tailR / c.blkR the
// records of the step that has just finished (tag c.step - 1), which the import rewrites where it changes a lane's last vehicle.
__device__ inline void ringHaloWriteTail(const RingCtx &c, int lane, int slot) {
    TailRec r{};
    r.slot = -1;
    if (slot >= 0) {
        const double2 kv = c.kin[slot];
        r.dis = kv.x;
        r.speed = kv.y;
        r.slot = slot;
        r.templ = c.meta[slot].x;
        r.prevDrv = c.s.prevDrv[slot];
    }
    r.tag = c.step - 1;
    const_cast<TailRec *>(c.tailR)[lane] = r;
}

// One cut lane's share of the import: `i` < nImport = import lane i (append the migrants), else ghost lane i - nImport (refresh
// the proxy unless the lane had an entrant of its own).  Called by the import kernel, one thread per cut lane, and — round 6,
// mailbox transports — by the lane's own thread at the top of the NEXT step's admission (kr_admit), which saves the launch.
__device__ inline void ringHaloImportCut(const RingCtx &c, const HaloDev &h, const HaloIO &io, const VidTable &vt, DevScalars *sc, int i) {
    if (i < h.nImport) {
        const int l = h.importLane[i];
        const char *blk = io.recv[h.importPeer[i]] + h.importRecvOff[i];
        const int m = ((const int32_t *) blk)[0];
        if (m <= 0) return;
        const HaloMigrant *rec = (const HaloMigrant *) (blk + 8);
        const int2 geo = c.ringGeo[l];
        const int head = c.head[l], n = c.cnt[l];
        if (n + m > geo.y) {
            sc->overflow = 8;
            return;
        }
        const int road = c.n.laneRoad[l];
        int last = -1;
        for (int j = 0; j < m; ++j) {
            const HaloMigrant r = rec[j];
            const int s = ringSlot(geo, head, n + j);
            const int route = vt.route[r.vid];
            const int next = nextOf(c.n, c.t, l, route, r.routePos);
            const int base = c.t.routeStart[route], len = c.t.routeStart[route + 1] - base;
            const int onLast = (next < 0 && c.t.routeRoads[base + len - 1] == road) ? 2 : 0;  // Router::isLastRoad: flags bit 1
            c.s.vid[s] = r.vid;
            c.s.drv[s] = l;
            c.s.prevDrv[s] = r.prevLL >= 0 ? -(r.prevLL + 2) : -1;
            c.s.routePos[s] = r.routePos;
            c.s.route[s] = route;
            c.meta[s] = make_int4(vt.templ[r.vid], next, onLast, CFX_INT_MAX);
            c.kin[s] = make_double2(r.dis, r.speed);
            const_cast<int2 *>(c.blkR)[s] = make_int2(-1, -1);  // a vehicle that left its laneLink this step was not yielding
            c.slotOf[r.vid] = s;
            vt.state[r.vid] = 1;
            last = s;
        }
        c.cnt[l] = n + m;
        atomicAdd((unsigned long long *) &sc->active, (unsigned long long) m);
        ringHaloWriteTail(c, l, last);
        return;
    }
    const int j = i - h.nImport;
    if (j < h.nGhost && !h.ghostHadEntrants[j]) {
        // no entrant of our own this step: the owner's tail is the lane's tail
        const int g = h.ghostLane[j];
        const HaloTail t = *(const HaloTail *) (io.recv[h.ghostPeer[j]] + h.ghostRecvOff[j]);
        const int2 geo = c.ringGeo[g];
        const int head = c.head[g], n = c.cnt[g];
        for (int q = 0; q < n; ++q) {  // whoever stood here (the old proxy, a vehicle admitted on both sides) is the owner's
            const int v = c.s.vid[ringSlot(geo, head, q)];
            if (v != t.vid) c.slotOf[v] = -1;
        }
        if (t.vid < 0) {
            c.cnt[g] = 0;
            ringHaloWriteTail(c, g, -1);
            return;
        }
        int prev = -1;
        if (t.prevLL >= 0) {
            const int k = h.llLocalOfGlobal[t.prevLL];
            prev = k >= 0 ? c.n.L + k : -(t.prevLL + 2);
        }
        const int s = ringSlot(geo, head, 0);
        c.s.vid[s] = t.vid;
        c.s.drv[s] = g;
        c.s.prevDrv[s] = prev;
        c.s.routePos[s] = 0;
        c.s.route[s] = vt.route[t.vid];
        c.meta[s] = make_int4(vt.templ[t.vid], -1, 0, CFX_INT_MAX);
        c.kin[s] = make_double2(t.dis, t.speed);
        const_cast<int2 *>(c.blkR)[s] = make_int2(-1, -1);
        c.slotOf[t.vid] = s;
        c.cnt[g] = 1;
        ringHaloWriteTail(c, g, s);
    }
}

// Waits (bounded) until peer `p` has published this epoch; false: it did not (flagged, never a hang).
__device__ inline bool ringHaloWaitPeer(const HaloIO &io, int p, DevScalars *sc) {
    unsigned spins = 0;
    while (__hip_atomic_load(io.waitFlag[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < io.epoch) {
        if (++spins > (1u << 22)) {
            sc->overflow = 4;
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    return true;
}

// What kr_admit needs to import the previous step's halo itself (on = 0: nothing pending).
struct RingHaloIn {
    int on;
    const int32_t *cutIndex;
    HaloDev h;
    HaloIO io;
    long long *activeOut;
};

__global__ void kr_halo_import(RingCtx c, HaloDev h, HaloIO io, VidTable vt, DevScalars *sc, long long *activeOut) {
    if (io.nWait > 0) {  // mailbox path: every block waits until all peers have published this epoch
        __shared__ int ok;
        if (threadIdx.x == 0) {
            ok = 1;
            for (int p = 0; p < io.nWait && ok; ++p) ok = ringHaloWaitPeer(io, p, sc) ? 1 : 0;
        }
        __syncthreads();
        if (!ok) return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {  // the vehicles the step's commit sent away (its own statistics block was rewriting `active` then)
        const long long out = *activeOut;
        if (out) {
            *activeOut = 0;
            atomicAdd((unsigned long long *) &sc->active, (unsigned long long) (-out));
        }
    }
    if (i < h.nImport + h.nGhost) ringHaloImportCut(c, h, io, vt, sc, i);
}

// COMMIT = true: the launch also commits the PREVIOUS step (kr_commit's work, drivable by drivable, in front of the drivable's
// admission; its statistics blocks at the end of the grid).  `cIn` is then the context of the step being committed and the
// admission runs on the next step's view of it.  What a lane's admission reads of the commit is what its own thread wrote —
// its tail, its count, its queue — except the lights, which the cross kernel of the committed step has already advanced.
// NB: spawn records the arguments hold (kAdmitRecsBig on large networks, as kd_admit) — or, with Batch = SpawnBatchMem (any
// number of records, in pinned host memory), the records of ONE block's lanes that are staged in LDS
template <bool COMMIT, int NB = kAdmitRecs, class Batch = SpawnBatchT<NB>>
__global__ __launch_bounds__(kBlock) void kr_admit(RingCtx cIn, int32_t *admitStep, int32_t *waitHead, VidTable vt, DevScalars *sc,
                                                   const Batch batch, const RingCommit k, const RingHaloIn hin) {
    constexpr bool kMem = std::is_same<Batch, SpawnBatchMem>::value;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const bool isLane = d < cIn.n.L, inRange = d < cIn.n.L + cIn.n.K;
    // (records in host memory: where this block's lanes' records lie — a trip over the link, requested before everything else)
    int b0 = 0, b1 = 0;
    if constexpr (kMem) {
        if ((int) blockIdx.x < batch.nLaneBlocks) {
            b0 = batch.blockOff[blockIdx.x];
            b1 = batch.blockOff[blockIdx.x + 1];
        }
    }
    // what the admission needs and the commit in front of it leaves alone — or changes through this very thread, which then
    // knows the new value: requested here, before the commit's own chain of loads, so that the two overlap
    TailRec committed{};
    int w = -1, n = 0, head = 0, road = 0, laneIdx = 0;
    int2 geo = make_int2(0, 0);
    int wt = 0, route = 0, nextWait = -1, fn = kFirstNextUnknown;
    uint8_t pending = 0;
    if constexpr (COMMIT) {
        const int nBody = (int) gridDim.x - commitStatBlocks(k);
        if ((int) blockIdx.x >= nBody) {
            commitStatBlock(cIn, k, vt, (int) blockIdx.x - nBody);
            return;
        }
        commitClearMasks(cIn, k, d, nBody * (int) blockDim.x);
        if (inRange) committed = cIn.tailW[d];  // (the record of the step being committed, unless its tail changes hands below)
        if (isLane) {
            w = waitHead[d];
            if (admitStep[d] == cIn.step) w = cIn.admitRec[d].y;  // (the commit below pops the vehicle admitted last step)
            n = cIn.cnt[d];
            geo = cIn.ringGeo[d];
            head = cIn.head[d];
            road = cIn.n.laneRoad[d];
            laneIdx = cIn.n.laneIndex[d];
            if (w >= 0) {
                wt = vt.templ[w];
                route = vt.route[w];
                nextWait = vt.nextWait[w];
                pending = vt.pendingCustom[w];
                fn = vt.firstNext[w];
            }
        }
        if (inRange) {
            CommitOut co;
            commitDrivable(cIn, k, d, &co);
            if (co.touched) {
                head = co.head;
                n = co.n;
                if (co.tailWritten) committed = co.tail;
            }
        }
    }
    RingCtx c = cIn;
    if constexpr (!COMMIT) {
        // tiling, mailbox transports: the previous step's halo import, by the cut lane's own thread, in front of the lane's
        // admission — what the import changes (the lane's count, its tail record, its slots) is what this thread reads next
        if (hin.on) {
            if (d == 0) {  // the vehicles the previous step's commit sent away (kr_halo_import's thread 0)
                const long long out = *hin.activeOut;
                if (out) {
                    *hin.activeOut = 0;
                    atomicAdd((unsigned long long *) &sc->active, (unsigned long long) (-out));
                }
            }
            const int ci = isLane ? hin.cutIndex[d] : -1;
            if (ci >= 0) {
                const int nG = hin.h.nGhost;
                const int peer = ci < nG ? hin.h.ghostPeer[ci] : hin.h.importPeer[ci - nG];
                if (hin.io.nWait == 0 || ringHaloWaitPeer(hin.io, peer, sc))
                    ringHaloImportCut(c, hin.h, hin.io, vt, sc, ci < nG ? hin.h.nImport + ci : ci - nG);
            }
        }
    }
    if constexpr (COMMIT) {  // the next step's view: what the commit wrote is what the admission reads
        c.step = cIn.step + 1;
        c.tailR = cIn.tailW;
        c.tailW = const_cast<TailRec *>(cIn.tailR);
        c.kin = cIn.kinN;
        c.kinN = cIn.kin;
        c.blkR = cIn.blkW;
        c.blkW = const_cast<int2 *>(cIn.blkR);
    }
    __shared__ int sAdmitted;
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sLane[NB];
    const cfx_vehicle_template *tv = c.t.templ;
    const int nRecs = batch.n, firstNewVid = batch.firstNewVid;
    if (threadIdx.x == 0) sAdmitted = 0;
    if constexpr (kMem) {
        for (int i = threadIdx.x; i < b1 - b0 && i < NB; i += blockDim.x) sLane[i] = batch.lane[b0 + i];
    } else {
        for (int i = threadIdx.x; i < nRecs; i += blockDim.x) sLane[i] = batch.lane[i];
    }
    // the lane of record j / the records this thread's lane can be found among
    auto laneOfRec = [&](int j) {
        if constexpr (kMem) return j - b0 < NB ? sLane[j - b0] : batch.lane[j];
        else return sLane[j];
    };
    const int recLo = kMem ? b0 : 0, recHi = kMem ? b1 : nRecs;
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    if constexpr (!COMMIT) {
        // ---- round 1: everything that hangs on the drivable alone (a lane's ring position is requested whether or not it
        //      will admit: the kernel is bound by its longest chain of dependent loads, not by bytes)
        if (inRange) committed = c.tailR[d];
        if (isLane) {
            w = waitHead[d];
            n = c.cnt[d];
            geo = c.ringGeo[d];
            head = c.head[d];
            road = c.n.laneRoad[d];
            laneIdx = c.n.laneIndex[d];
        }
        // ---- round 2: the head of the lane's waiting queue (as the last step left it)
        if (w >= 0) {
            wt = vt.templ[w];
            route = vt.route[w];
            nextWait = vt.nextWait[w];
            pending = vt.pendingCustom[w];
            fn = vt.firstNext[w];
        }
    }
    __syncthreads();  // (templates and the batch's lanes staged)
    if (nRecs > 0) {
        // the vehicle table of the new vehicles (k_spawn_link): block 0.  Nobody reads these rows in this kernel — a vehicle
        // that is admitted in the step it appears in is taken from its record
        auto writeRows = [&](int from, int to) {
            for (int i = from + (int) threadIdx.x; i < to; i += blockDim.x) {
                const int v = firstNewVid + batch.vidOff[i];
                vt.priority[v] = batch.priority[i];
                vt.templ[v] = batch.templ[i];
                vt.route[v] = batch.route[i];
                vt.enterTime[v] = batch.enterTime;
                vt.state[v] = 0;
                vt.pendingCustom[v] = 0;
                vt.firstNext[v] = batch.firstNext[i];
            }
        };
        if constexpr (kMem) {  // every block its own lanes' records; block 0 also the ones without a lane here
            writeRows(blockIdx.x == 0 ? 0 : b0, b1);
        } else {
            if (blockIdx.x == 0) writeRows(0, nRecs);
        }
        if (isLane) {
            // FIFO append (Lane::pushWaitingVehicle roadnet.h:365-367; nextWait[] of a new vehicle was pre-set to -1): this
            // lane's records, in any order — each hangs behind its predecessor, or becomes the head where the predecessor
            // has left the queue
            int lo = recLo, hi = recHi;  // first record of this lane
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (laneOfRec(mid) < d) lo = mid + 1;
                else hi = mid;
            }
            int headRec = -1;
            for (int j = lo; j < recHi && laneOfRec(j) == d; ++j) {
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
                fn = batch.firstNext[headRec];
                pending = 0;
                nextWait = -1;
                waitHead[d] = w;
            }
            if (w >= 0)  // whoever was hung behind the head just now (this thread's own store: taken from the record)
                for (int j = lo; j < recHi && laneOfRec(j) == d; ++j)
                    if (batch.prevWait[j] == w) nextWait = firstNewVid + batch.vidOff[j];
        }
    }
    if (inRange) {
        // this step's view of the drivable's tail: what the last step left, or the vehicle admitted below
        TailRec now = committed;
        if (committed.tag != c.step - 1) now.slot = -1;
        if (!isLane) {
            // RoadLink::isAvailable (roadnet.h:429-431) into the laneLink's gate record — only when the intersection's phase is
            // not the one the record was written for: the lights change every few seconds, the chain intersection -> phase ->
            // availability table behind the commit was walked for every laneLink in every step
            const int k = d - c.n.L, in = c.n.llInter[k];
            const int ph = c.curPhase[in];
            if (c.gatePhase[((c.step + 1) & 1) * c.n.I + in] != ph) {
                const int flags = (c.n.phaseAvail[c.n.interAvailStart[in] + ph * c.n.interNRL[in] + c.n.llRoadLink[k]] != 0 ? 1 : 0) |
                                  (c.n.llType[k] << 1) | (c.n.llXStart[k + 1] > c.n.llXStart[k] ? 8 : 0);
                c.llGate[k] = make_int4(flags, c.n.llEndLane[k], c.n.llXStart[k], c.n.llXStart[k + 1]);
            }
            if (c.n.llLocal[k] == 0) c.gatePhase[(c.step & 1) * c.n.I + in] = ph;
        } else {
            const int lane = d;
            bool admit = w >= 0;  // Lane::available roadnet.cpp:428-435
            if (admit && now.slot >= 0 && !(now.dis > tv[now.templ].len + tv[wt].min_gap)) admit = false;
            if (admit && n >= geo.y) {  // the ring is full (cannot happen with capacities from the shortest vehicle): refuse loudly
                sc->overflow = 8;
                admit = false;
            }
            if (admit) {
                // Router::getNextDrivable(0) (router.cpp:49-76) and Router::isLastRoad of the admitted vehicle: known since it was
                // created (VidTable::firstNext) — the walk route -> first road -> row -> laneLink used to sit behind the commit
                int next, onLast;
                admittedNext(c, fn, lane, road, laneIdx, route, &next, &onLast);
                const int slot = ringSlot(geo, head, n);
                const double v0 = tv[wt].initial_speed;  // VehicleInfo::speed: 0 unless pushed with a speed
                c.s.vid[slot] = w;
                c.s.drv[slot] = lane;
                c.s.prevDrv[slot] = -1;
                c.s.routePos[slot] = 0;
                c.s.route[slot] = route;
                c.meta[slot] = make_int4(wt, next, pending | onLast, CFX_INT_MAX);  // (enterLaneLinkTime: ControllerInfo ctor vehicle.cpp:10-13)
                c.kin[slot] = make_double2(0.0, v0);
                c.slotOf[w] = slot;
                c.admitRec[lane] = make_int2(w, nextWait);
                admitStep[lane] = c.step;  // cnt[] and the FIFO pop follow in kr_commit (see cntNow)
                now.dis = 0.0;
                now.speed = v0;
                now.slot = slot;
                now.templ = wt;
                now.prevDrv = -1;
                // tiling: an admission onto a ghost lane only mirrors the owner's (same queue, same tail, same decision)
                if (!(c.n.laneGhost && c.n.laneGhost[lane])) atomicAdd(&sAdmitted, 1);
            }
        }
        now.tag = c.step;
        c.tailNow[d] = now;
    }
    __syncthreads();
    // Engine::activeVehicleCount: one global atomic per block; the step's commit folds the sum into the running count
    if (threadIdx.x == 0 && sAdmitted) atomicAdd((unsigned long long *) &sc->admitPending[c.step & 1], (unsigned long long) sAdmitted);
}

// Per-laneLink sources of Engine::threadNotifyCross (llstate of cfx_kernels.h) on the ring layout: the tails come from
// the tail records, and the sources' state is left beside llDyn for the cross phase.
__device__ inline void llstateRing(const RingCtx &c, int k) {
    if (k >= c.n.K) return;
    const int d = c.n.L + k;
    const int endLane = c.n.llEndLane[k], startLane = c.n.llStartLane[k];
    const Tail tu = tailNowOf(c, endLane);
    const int u = (tu.slot >= 0 && tu.prevDrv == d) ? tu.slot : -1;
    int f = cntNow(c, startLane) > 0 ? firstSlot(c, startLane) : -1;
    // (RoadLink::isAvailable is bit 0 of the gate record kr_admit wrote for this step: one load instead of the chain
    // intersection -> phase -> availability table)
    const bool green = (c.llGate[k].x & 1) != 0;
    if (f >= 0 && !(green && c.meta[f].y == d)) f = -1;
    const int nOn = c.cnt[d];
    c.llDyn[k] = make_int4(u, f, firstSlot(c, d), nOn);
    if (u >= 0 || f >= 0 || nOn > 0) {
        LLAux a{};
        a.uDis = tu.dis;
        a.uSpeed = tu.speed;
        a.uTempl = tu.templ;
        if (f >= 0) {
            const double2 kf = c.kin[f];
            a.fDis = kf.x;
            a.fSpeed = kf.y;
            a.fTempl = c.meta[f].x;
        }
        a.llLen = c.n.drvLength[d];
        a.startLen = c.n.drvLength[startLane];
        c.llAux[k] = a;
        const int in = c.n.llInter[k];
        const int bit = c.n.llLocal[k];
        atomicOr(&c.interMask[c.n.interMaskStart[in] + (bit >> 6)], 1ULL << (bit & 63));  // (read by k_cross2 only)
    }
}

// notifiedAt + the notified vehicle's state (cfx_kernels.h) from the two laneLink records
__device__ __forceinline__ int blockerOfNotified(const RingCtx &c, const Notified &nf) {
    if (!nf.pre) return blockerOf(c, nf.slot);
    return (nf.blk.x >= 0 && nf.blk.y == c.step - 1) ? c.slotOf[nf.blk.x] : -1;
}
__device__ __forceinline__ void notifiedExtras(const RingCtx &c, Notified &nf) {  // requested now, used (maybe) much later
    nf.enterLLT = c.meta[nf.slot].w;
    nf.blk = c.blkR[nf.slot];
    nf.vid = c.s.vid[nf.slot];
    nf.pre = true;
}
__device__ inline Notified notifiedFrom(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x, const int4 dyn,
                                        const LLAux &a);
__device__ inline Notified notified(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x) {
    const int4 dyn = c.llDyn[k];
    const LLAux a = c.llAux[k];  // (issued together with llDyn: this phase is bound by rounds of dependent loads, not bytes)
    return notifiedFrom(c, tv, k, x, dyn, a);
}
__device__ inline Notified notifiedFrom(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x, const int4 dyn,
                                        const LLAux &a) {
    Notified nf{-1, 0, 0.0, 0.0, false, 0, make_int2(-1, -1)};
    if (dyn.x < 0 && dyn.y < 0 && dyn.w == 0) return nf;  // nobody to yield to on that laneLink
    if (dyn.x >= 0) {
        const double vehDistance = a.uDis - tv[a.uTempl].len;
        const double crossDistance = a.llLen - x;
        if (crossDistance + vehDistance < 0.0) {
            nf.slot = dyn.x;
            nf.templ = a.uTempl;
            nf.speed = a.uSpeed;
            nf.dist = -(a.uDis + crossDistance);
            notifiedExtras(c, nf);
            return nf;
        }
    }
    if (dyn.w > 0) {
        const SegWalk walk = segWalk(c, c.n.L + k, dyn.z);
        for (int i = 0; i < dyn.w; i += 2) {  // two vehicles per round of loads (the walk stops at the first that matches)
            const int w0 = walk.at(i), w1 = walk.at(i + 1 < dyn.w ? i + 1 : i);
            const double2 k0 = c.kin[w0], k1 = c.kin[w1];
            const int t0 = c.meta[w0].x, t1 = c.meta[w1].x;
            for (int q = 0; q < 2 && i + q < dyn.w; ++q) {
                const int w = q ? w1 : w0;
                const double2 kw = q ? k1 : k0;
                const double vehDistance = kw.x;
                const int wt = q ? t1 : t0;
                if (!(vehDistance > x) || (vehDistance - x - tv[wt].len <= 0.0)) {
                    nf.slot = w;
                    nf.templ = wt;
                    nf.speed = kw.y;
                    nf.dist = x - vehDistance;
                    notifiedExtras(c, nf);
                    return nf;
                }
            }
        }
    }
    if (dyn.y >= 0) {
        nf.slot = dyn.y;
        nf.templ = a.fTempl;
        nf.speed = a.fSpeed;
        nf.dist = (a.startLen - a.fDis) + x;
        notifiedExtras(c, nf);
    }
    return nf;
}

// The same for kr_cross, whose group holds a vehicle's whole chain and pays every dependent round in full (k_cross2 keeps the
// version above: its pass B is bound by resident wavefronts, and the columns held here cost one).
__device__ inline Notified notifiedEagerFrom(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x, const int4 dyn,
                                        const LLAux &a) {
    Notified nf{-1, 0, 0.0, 0.0, false, 0, make_int2(-1, -1)};
    if (dyn.x < 0 && dyn.y < 0 && dyn.w == 0) return nf;  // nobody to yield to on that laneLink
    // ONE round for what hangs on the laneLink's two records in the usual case: the first two vehicles on the laneLink, and of
    // the first one also the columns the decision may want of the notified vehicle (its enterLaneLinkTime comes with the
    // 16-byte meta record; its blocker record; its number).  Round 3 fetched those in a round of their own once the notified
    // vehicle was known; the other candidates (u, the second vehicle, f) still do — holding theirs too costs a wavefront per SIMD.
    const SegWalk walk = segWalk(c, c.n.L + k, dyn.z);
    const bool haveU = dyn.x >= 0, haveOn = dyn.w > 0;
    const int w0 = haveOn ? walk.at(0) : 0, w1 = haveOn ? walk.at(dyn.w > 1 ? 1 : 0) : 0;
    double2 k0 = make_double2(0.0, 0.0), k1 = k0;
    int4 m0 = make_int4(0, 0, 0, 0), m1 = m0;
    int2 b0 = make_int2(-1, -1);
    int v0 = 0;
    if (haveOn) {  // (the blocker record and the number only of the first vehicle — the usual answer; registers)
        k0 = c.kin[w0];
        k1 = c.kin[w1];
        m0 = c.meta[w0];
        m1 = c.meta[w1];
        b0 = c.blkR[w0];
        v0 = c.s.vid[w0];
    }
    if (haveU) {
        const double vehDistance = a.uDis - tv[a.uTempl].len;
        const double crossDistance = a.llLen - x;
        if (crossDistance + vehDistance < 0.0) {
            nf.slot = dyn.x;
            nf.templ = a.uTempl;
            nf.speed = a.uSpeed;
            nf.dist = -(a.uDis + crossDistance);
            notifiedExtras(c, nf);
            return nf;
        }
    }
    if (haveOn) {
        for (int q = 0; q < 2 && q < dyn.w; ++q) {  // the first two came with the round above
            const double2 kw = q ? k1 : k0;
            const int4 mw = q ? m1 : m0;
            const double vehDistance = kw.x;
            if (!(vehDistance > x) || (vehDistance - x - tv[mw.x].len <= 0.0)) {
                nf.slot = q ? w1 : w0;
                nf.templ = mw.x;
                nf.speed = kw.y;
                nf.dist = x - vehDistance;
                if (q == 0) {
                    nf.enterLLT = mw.w;
                    nf.blk = b0;
                    nf.vid = v0;
                    nf.pre = true;
                } else {
                    notifiedExtras(c, nf);
                }
                return nf;
            }
        }
        for (int i = 2; i < dyn.w; i += 2) {  // two vehicles per round of loads (the walk stops at the first that matches)
            const int x0 = walk.at(i), x1 = walk.at(i + 1 < dyn.w ? i + 1 : i);
            const double2 q0 = c.kin[x0], q1 = c.kin[x1];
            const int t0 = c.meta[x0].x, t1 = c.meta[x1].x;
            for (int q = 0; q < 2 && i + q < dyn.w; ++q) {
                const int w = q ? x1 : x0;
                const double2 kw = q ? q1 : q0;
                const double vehDistance = kw.x;
                const int wt = q ? t1 : t0;
                if (!(vehDistance > x) || (vehDistance - x - tv[wt].len <= 0.0)) {
                    nf.slot = w;
                    nf.templ = wt;
                    nf.speed = kw.y;
                    nf.dist = x - vehDistance;
                    notifiedExtras(c, nf);
                    return nf;
                }
            }
        }
    }
    if (dyn.y >= 0) {
        nf.slot = dyn.y;
        nf.templ = a.fTempl;
        nf.speed = a.fSpeed;
        nf.dist = (a.startLen - a.fDis) + x;
        notifiedExtras(c, nf);
    }
    return nf;
}

__device__ inline Notified notifiedEager(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x) {
    const int4 dyn = c.llDyn[k];
    const LLAux a = c.llAux[k];
    return notifiedEagerFrom(c, tv, k, x, dyn, a);
}

// A vehicle handed to the cross phase, with everything the action phase already knew about it: the cross phase starts
// from ONE record instead of the chain slot -> {drivable, template, speed, dis, next} -> {length, laneLink record}.
struct RingJob {
    int32_t slot, d, idx, nNow, xs, xe, t1, templ, nd0, pad0, pad1, pad2;
    double d0, speed, v, iv, dis, dlen;
};
static_assert(sizeof(RingJob) == 96, "cross job record layout");

struct RingPush {
    JobQueue q;
    RingJob *recs;
    int L;
    __device__ __forceinline__ void operator()(int s, const JobInfo &j) const {
        const int shard = blockIdx.x & (kJobShards - 1);
        const int idx = jobQueuePlace(q);  // (one atomic per wavefront)
        if (idx >= q.capacity) {
            *q.overflow = 9;
            return;
        }
        const size_t at = (size_t) shard * q.capacity + idx;
        q.jobs[at] = s;
        if (!recs) return;  // (the throughput form of the cross phase reads the slots)
        RingJob r{};
        r.slot = s;
        r.d = j.d;
        r.idx = j.idx;
        r.nNow = j.nNow;
        r.xs = j.xs;
        r.xe = j.xe;
        r.t1 = (j.gateFlags >> 1) & 3;
        r.templ = j.templ;
        r.nd0 = j.nd0;
        r.d0 = j.d < L ? -(j.dlen - j.dis) : j.dis;
        r.speed = j.speed;
        r.v = j.v;
        r.iv = j.iv;
        r.dis = j.dis;
        r.dlen = j.dlen;
        recs[at] = r;
    }
};

#ifdef CFX_TRACE
#define TRACE_STAMP(k) if (t == 0) g_trace[(size_t) w * 8 + (k)] = (long long) wall_clock64()
#else
#define TRACE_STAMP(k)
#endif

// Second half of Vehicle::getIntersectionRelatedSpeed (k_cross of cfx_kernels.h) from job records: one 16-lane group per
// queued vehicle, one cross per lane and round, first failing lane of the first failing round = the first cross that
// cannot be passed.  No active-laneLink mask: a lane reads the peer laneLink's two records directly.

#ifndef CFX_KR_CROSS_WAVES
#define CFX_KR_CROSS_WAVES 4  // (128 registers, 2 spilled, 12 B of scratch.  Round 6: 3 = 132 registers, none spilled: 13.9 against 14.0 us right
                              //  behind a load, 14.8-16.1 against 14.6-15.4 us in bench.py's runs: not kept)
#endif
#ifndef CFX_KR_ACTION_WAVES
#define CFX_KR_ACTION_WAVES 0
#endif
#if CFX_KR_ACTION_WAVES > 0
#define CFX_KR_ACTION_BOUNDS(B) __launch_bounds__(B, CFX_KR_ACTION_WAVES)
#else
#define CFX_KR_ACTION_BOUNDS(B) __launch_bounds__(B)
#endif
__global__ __launch_bounds__(kCrossBlock, CFX_KR_CROSS_WAVES) void kr_cross(RingCtx c, RingOut o, JobQueue q, const RingJob *recs, RingLights lights) {
    // (nothing in this kernel reads the lights: the approaching vehicles' light test is folded into llDyn by the action kernel.
    // They are advanced by the LAST blocks of the grid — the host sizes it with room to spare, so those have the fewest jobs)
    if (lights.on)
        passTimeAll(c.n, lights.curPhase, lights.remain, c.interval, ((int) gridDim.x - 1 - (int) blockIdx.x) * (int) blockDim.x + (int) threadIdx.x,
                    gridDim.x * blockDim.x);
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    const cfx_vehicle_template *tv = c.t.templ;
#ifdef CFX_TRACE
    const int traceRow = 4096 + (int) blockIdx.x;
#define XSTAMP(k) if (threadIdx.x == 0) g_trace[(size_t) traceRow * 8 + (k)] = (long long) wall_clock64()
#else
#define XSTAMP(k)
#endif
    XSTAMP(0);
    // A block works on ONE shard of the queue — the one its index names, as the action kernel's blocks fill them — so all it
    // needs to know is that shard's count (no prefix over the shards, no second barrier), and the first job record of every
    // group is requested together with that count and the template table instead of after them: the record's place is known
    // from the block and group index alone; a record beyond the shard's count is read (the shard's room is allocated) and dropped.
    // (All four 16-lane groups of a wavefront take a job although their chains differ and a wavefront walks divergent
    // branches one after the other: measured in round 4 with two / one working group per wavefront and a grid to match —
    // 15.4 -> 17.1 / 23.1 us at 30x30; what counts is how many wavefronts the launch needs.)
    const int g = threadIdx.x % kCrossGroup;
    const int groupsPerBlock = blockDim.x / kCrossGroup;
    const int groupShift = (threadIdx.x & 63) & ~(kCrossGroup - 1);
    const int shard = (int) blockIdx.x & (kJobShards - 1);
    const int blocksOfShard = ((int) gridDim.x - shard + kJobShards - 1) / kJobShards;
    const int jFirst = ((int) blockIdx.x / kJobShards) * groupsPerBlock + (int) threadIdx.x / kCrossGroup;
    const RingJob *const shardRecs = recs + (size_t) shard * q.capacity;
    RingJob jr = shardRecs[jFirst < q.capacity ? jFirst : q.capacity - 1];
    const int nShard = min(q.count[shard * kJobShardStride], q.capacity);
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (sizes the next steps' grids, through the host mirror)
        int nJ = 0;
        for (int i = 0; i < kJobShards; ++i) nJ += min(q.count[i * kJobShardStride], q.capacity);
        o.sc->nCrossJobs = nJ;
    }
    __syncthreads();
    XSTAMP(1);
    for (int j = jFirst; j < nShard; j += blocksOfShard * groupsPerBlock) {
        if (j != jFirst) jr = shardRecs[j];  // (the grid is sized for one job per group; a second one is loaded where it is needed)
        const int s = jr.slot;
        const cfx_vehicle_template &t = tv[jr.templ];
        const double d0 = jr.d0;
        VehRef self{jr.speed, &t};
        double iv = jr.iv;
        int blockerSlot = -1;
        // what the vehicle's finish may need, requested by the lane that will finish it while the group walks the crosses:
        // the identity columns of a vehicle that may leave its drivable, where the drivable's last vehicle came from
        LeaverPrefetch lp{false, 0.0, 0, 0, 0};
        if (g == 0) {
            if (jr.dlen - jr.dis <= (jr.speed + t.max_pos_acc * c.interval) * c.interval + 1.0) {
                lp.valid = true;
                lp.nextLen = jr.nd0 >= 0 ? c.n.drvLength[jr.nd0] : 0.0;
                lp.vid = c.s.vid[s];
                lp.route = c.s.route[s];
                lp.routePos = c.s.routePos[s];
            }
            if (jr.idx == jr.nNow - 1) {
                lp.prevValid = true;
                lp.prevDrv = c.s.prevDrv[s];
            }
            lp.templP1 = jr.templ + 1;
        }
        XSTAMP(2);
        for (int e0 = jr.xs; e0 < jr.xe; e0 += kCrossGroup) {
            const int e = e0 + g;
            bool fail = false;
            int foe = -1, foeVid = -2;
            double dOn = 0.0;
            if (e < jr.xe) {
                const double2 dd = c.n.xDD[e];  // {distance on this laneLink, distance on the peer laneLink}
                const int4 xp = c.n.xPack[e];   // {peer laneLink, peer bit, peer roadLink type, -}
                dOn = dd.x;
                // (a vehicle that can no longer yield passes whoever comes, Cross::canPass roadnet.cpp:617-618: tested in front
                // of the notified vehicle's gathers)
#ifdef CFX_NO_RING_PREFILTER
                if (!(dOn < d0)) {
#else
                if (!(dOn < d0) && canYield(self, dOn - d0)) {
#endif
                    const Notified nf = notifiedEager(c, tv, xp.x, dd.y);
                    fail = !canPassDecide(c, tv, s, self, dOn, jr.t1, d0, nf, xp.z, &foe);
                    if (nf.slot >= 0 && nf.pre) foeVid = nf.vid;  // (came with the notified vehicle's other columns)
                }
            }
            const unsigned long long ball = __ballot(fail);
            const unsigned gm = (unsigned) ((ball >> groupShift) & ((1ULL << kCrossGroup) - 1ULL));
            if (gm != 0u) {
                const int first = __ffs(gm) - 1;  // lowest lane = smallest cross distance in this round
                const int src = groupShift + first;
                const double fdOn = __shfl(dOn, src, 64);
                blockerSlot = __shfl(foe, src, 64);
                const int bvid = __shfl(foeVid, src, 64);
                if (blockerSlot >= 0 && bvid >= 0) lp.blockerVidP2 = bvid + 2;
                iv = min2(iv, stopBeforeSpeed(self, fdOn - d0 - t.yield_distance, c.interval));
                break;
            }
        }
        XSTAMP(3);
        if (g == 0)
            finishAction<false>(c, o, t, s, jr.d, 0, jr.speed, jr.dis, jr.dlen, jr.nd0, min2(jr.v, iv), blockerSlot, jr.idx, jr.nNow, lp);
    }
    XSTAMP(4);
#ifdef CFX_TRACE
    if (threadIdx.x == 0) g_trace[(size_t) traceRow * 8 + 5] = nShard;
#endif
}

// ---- one vehicle's phase 4 on the ring layout, organised by ROUNDS of memory accesses --------------------------------
// Same arithmetic as actionOne / finishAction (cfx_kernels.h), expression for expression.  What differs is when memory
// is touched.  A wave holds followers, heads of drivables, vehicles near an intersection and vehicles about to leave
// their drivable side by side; written naively each kind walks its own chain of dependent loads inside its own branch
// and the wave pays for the chains one after the other.  Here every lane first REQUESTS whatever its kind may need
// (round A: tails of the drivables ahead, the gate record of the next laneLink, the length of the next drivable and the
// identity columns of a vehicle that may leave), then everything that depends on those (round B: the tail of the lane
// behind the next laneLink), and only then is anything decided — the wave waits for memory twice instead of six times.
__device__ __forceinline__ int4 gateRecord(const RingCtx &c, int k) { return c.llGate[k]; }
__device__ __forceinline__ TailRec linkTailNow(const RingCtx &c, int d) { return c.tailNow[d]; }  // (kr_admit writes every drivable's)
// the first record a head of a drivable requests: a laneLink head's end lane as committed, or a lane head's first laneLink as of now
__device__ __forceinline__ TailRec firstHopRecord(const RingCtx &c, bool linkHead, int endLane, int firstLink) {
    return *(linkHead ? &c.tailR[endLane] : &c.tailNow[firstLink]);
}
__device__ __forceinline__ bool viewerIsNew(const RingCtx &, const SlotIn &in, int) { return in.laneAdmitted && in.nNow == 1; }

// GHOST: the engine is a tile (ghost lanes exist).  A template flag, not a run-time test: inlined into the single engine's kernels
// the frozen-proxy path cost kl_action 10 spilled registers under its five-wavefront bound (and as an out-of-line call, 74).
template <bool GHOST = false, class C, class Out, class Push>
__device__ __forceinline__ void actionOneRounds(const C &c, const Out &o, const cfx_vehicle_template *tv, const int s,
                                                const SlotIn &in, Push push) {
    const int d = in.d, templIdx = in.templIdx, nd0 = in.nd0, flags = in.flags, L = c.n.L;
    const double speed = in.speed, dis = in.dis;
    if constexpr (GHOST && std::is_same<C, RingCtx>::value) {
        // tiling: a vehicle on a ghost lane is the frozen proxy of a neighbour's vehicle (or a vehicle admitted on both sides):
        // not stepped here.  Its state goes into the next generation as it is, and the lane's last one rewrites the tail record.
        if (d < L && c.n.laneGhost[d]) {
            o.keep(s, dis, speed);
            if (in.idx == in.nNow - 1) {
                TailRec r;
                r.dis = dis;
                r.speed = speed;
                r.slot = s;
                r.templ = templIdx;
                r.prevDrv = c.s.prevDrv[s];
                r.tag = c.step;
                c.tailW[d] = r;
            }
            return;
        }
    }
    const cfx_vehicle_template &t = tv[templIdx];
    const double interval = c.interval;
    const double dlen = in.lm.x;
    const bool head = in.head, onLane = d < L, nextIsLink = nd0 >= L;
    const bool related = !onLane || (nextIsLink && dlen - dis <= t.approach_dist);  // Vehicle::isIntersectionRelated
    const bool custom = (flags & 1) != 0;

    // ================= round A: requests that depend on nothing but the slot
    const bool hopHead = head && onLane && nextIsLink && in.hop.x != -2 && in.hop.w < 0;  // head of a lane: tails of the (up to three) laneLinks leaving it
    const bool linkHead = head && !onLane;                                // head of a laneLink: tail of its end lane
    // (a head is either a lane's or a laneLink's: the first record is the lane head's first hop or the laneLink head's end lane)
    TailRec hopRec[3];
    double linkLen = 0.0;
    if ((hopHead && in.hop.x >= 0) || linkHead) hopRec[0] = firstHopRecord(c, linkHead, nd0, L + in.hop.x);
    if (hopHead) {
        if (in.hop.y >= 0) hopRec[1] = linkTailNow(c, L + in.hop.y);
        if (in.hop.z >= 0) hopRec[2] = linkTailNow(c, L + in.hop.z);
        linkLen = c.n.drvLength[nd0];
    }
    int4 gate = make_int4(0, 0, 0, 0);
    const int gateLink = onLane ? nd0 - L : d - L;
    if (related) gate = gateRecord(c, gateLink);
    // the lane behind the next laneLink (a lane's vehicle near the intersection): Lane::canEnter looks at it as of this step,
    // the leader search two drivables ahead as of the last commit.  Where the loader knows the lane (the static table of
    // the vehicle's own lane) the records are requested here, with everything else; otherwise they hang on the gate record
    const bool approaching = related && nextIsLink;  // (a vehicle ON a laneLink has nd0 = its end lane)
    const bool endKnown = approaching && in.endLane >= 0;
    TailRec laneNow{}, laneCommitted{};
    if (endKnown) {
        laneNow = c.tailNow[in.endLane];
        if (hopHead) laneCommitted = c.tailR[in.endLane];
    }
    // may it run past the end of its drivable this step?  (only a hint: decides what is requested early)
    LeaverPrefetch lp{false, 0.0, 0, 0, 0};
    if (dlen - dis <= (speed + t.max_pos_acc * interval) * interval + 1.0) {
        lp.valid = true;
        lp.nextLen = nd0 >= 0 ? c.n.drvLength[nd0] : 0.0;
        lp.vid = c.s.vid[s];
        lp.route = c.s.route[s];
        lp.routePos = c.s.routePos[s];
    }
    if (in.idx >= 0 && in.idx == in.nNow - 1) {  // (ring layout) the drivable's last vehicle: it will rewrite the tail record
        lp.prevValid = true;
        lp.prevDrv = c.s.prevDrv[s];
    }
    lp.templP1 = templIdx + 1;

    // ================= round B: what hangs on the gate record (only where the end lane was not known above)
    if (approaching && !endKnown) {
        laneNow = c.tailNow[gate.y];
        if (hopHead) laneCommitted = c.tailR[gate.y];
    }

    // ================= leader / gap (Vehicle::updateLeaderAndGap vehicle.cpp:157-196)
    double gap = 0.0;
    int ls, leaderTempl = in.templPrev;
    double leaderSpeed = in.speedPrev;
    if (!head) {
        ls = in.leaderSlot;
        gap = in.disPrev - tv[in.templPrev].len - dis;
    } else {
        Tail best{-1, 0, -1, 0.0, 0.0};
        bool resolved = false;
        double dist = dlen - dis;
        if (hopHead) {
            // first hop: the last vehicles of all laneLinks leaving this lane, closest first (findHeadLeader, `consider`)
            for (int q = 0; q < 3; ++q) {
                const int ll = q == 0 ? in.hop.x : (q == 1 ? in.hop.y : in.hop.z);
                if (ll < 0) continue;
                const Tail cand = tailOfRec(hopRec[q]);
                if (cand.slot >= 0) {
                    const double cg = dist + cand.dis - tv[cand.templ].len;
                    if (best.slot < 0 || cg < gap) {
                        best = cand;
                        gap = cg;
                    }
                }
            }
            resolved = best.slot >= 0;
            if (!resolved) {
                dist += linkLen;
                if (dist > t.approach_dist) {
                    resolved = true;  // nothing within the look-ahead bound (vehicle.cpp:190-191)
                } else {
                    // second hop: the lane behind the laneLink (dist <= bound implies `approaching`: its records are here).
                    // A vehicle admitted this step sees this step's admissions on lanes before its own (lastSlotForLeader).
                    const bool viewerNew = viewerIsNew(c, in, d);
                    const int endLane = gate.y;
                    best = (viewerNew && endLane < d) ? tailOfRec(laneNow) : tailIfCurrent(laneCommitted, c.step - 1);
                    if (best.slot >= 0) {
                        gap = dist + best.dis - tv[best.templ].len;
                        resolved = true;
                    } else {
                        dist += c.n.drvLength[endLane];
                        resolved = dist > t.approach_dist;
                    }
                }
            }
        } else if (linkHead) {
            best = tailIfCurrent(hopRec[0], c.step - 1);
            if (best.slot >= 0) {
                gap = dist + best.dis - tv[best.templ].len;
                resolved = true;
            } else {
                dist += c.n.drvLength[nd0];
                resolved = dist > t.approach_dist;
            }
        } else if (nd0 < 0) {
            resolved = true;  // end of the route: nothing ahead
        }
        if (!resolved) {  // anything else (more than four laneLinks, a search that goes on): the general walk from the start
            best = findHeadLeader(c, tv, s, d, dis, t.approach_dist, nd0, dlen, &gap, in.hop);
        }
        ls = best.slot;
        leaderTempl = best.templ;
        leaderSpeed = best.speed;
    }

    if ((flags & kFlagStateGap) && ls >= 0) gap = c.vGapState[in.vid];  // first step after a load: the state's gap (rare, like the custom speed)

    // ================= Vehicle::getNextSpeed vehicle.cpp:308-335
    double v = t.max_speed;
    v = min2(v, speed + t.max_pos_acc * interval);
    v = min2(v, in.lm.y);
    double cf;  // Vehicle::getCarFollowSpeed vehicle.cpp:212-238
    const double customSpeed = custom ? c.vCustomSpeed[in.vid] : 0.0;  // (rare: loaded where it is used)
    if (ls < 0) {
        cf = custom ? customSpeed : t.max_speed;
    } else if (custom) {
        const cfx_vehicle_template &tl = tv[leaderTempl];
        cf = min2(customSpeed, noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, t.max_neg_acc, gap, interval, 0));
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

    // ================= Vehicle::getIntersectionRelatedSpeed vehicle.cpp:337-362
    if (related) {
        VehRef self{speed, &t};
        double iv = t.max_speed;
        bool done = false;
        if (nextIsLink) {
            bool blocked = !(gate.x & 1);
            if (!blocked) {  // Lane::canEnter roadnet.cpp:437-445
                const Tail tail = tailOfRec(laneNow);
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
            if (nextIsLink && typeIsTurn((gate.x >> 1) & 3)) iv = min2(iv, t.turn_speed);
            if (gate.x & 8) {  // the crosses of the laneLink: the cross phase takes over, with everything known here
                o.park(s, v, iv);
                push(s, JobInfo{d, in.idx, in.nNow, templIdx, nd0, gateLink, gate.x, gate.z, gate.w, speed, dis, dlen, v, iv});
                return;
            }
        }
        v = min2(v, iv);
    }
    finishAction<false>(c, o, t, s, d, in.vid, speed, dis, dlen, nd0, v, -1, in.idx, in.nNow, lp, in.lastRoadFlags);
}

// ---------------------------------------------------------------------------------------------- phase 3 + 4
// A workgroup of B threads owns a run of consecutive drivables (G lanes, or B laneLinks): its first threads read the
// rings' {base, cap, head, cnt} and the drivables' {length, max speed}, a block-wide prefix sum turns the counts into the
// block's vehicle list, and the block walks that list B - 1 vehicles at a time (sized so that one pass is the rule:
// every block of the launch is resident at once, and the step is bound by the slowest block's chain of dependent
// loads).  Each pass stages the window's {dis, speed, template} in LDS; a vehicle's leader is the previous element of the
// window (thread 0 re-reads the last vehicle of the previous pass for that purpose only), so leader / follower state is
// read from HBM once, coalesced.
constexpr int kRingWave = 64;

// Lane::updateHistory (engine.cpp:429-442, roadnet.cpp:900-915) of the PREVIOUS step, taken by trailing blocks of this step's
// action launch: until this step's commit the lanes' lists (head, cnt — an admission of this step is not in them yet) and the
// generation the action phase reads ARE the end of the previous step, nobody in this launch writes either, and the lanes'
// threads have the whole launch to sum their speeds in: the history costs the step no launch and no link of anybody's chain.
// (Until round 6 the lane's thread of the commit took it: +4 us on the merged admission.)  The host launches kr_lane_history
// for a step whose successor has not been submitted when somebody asks (cfx_engine::settle).
struct RingHist {
    LaneHistDev h;   // (num == nullptr: not kept, or nothing pending)
    int firstBlock;  // the launch's blocks from here on are history blocks, one thread per lane
};
__device__ inline void ringLaneHistory(const RingCtx &c, const LaneHistDev &h, int lane) {
    if (lane >= c.n.L) return;
    const int2 geo = c.ringGeo[lane];
    const int head = c.head[lane];
    laneHistoryStep(h, lane, c.cnt[lane], [&](int i) { return c.kin[ringSlot(geo, head, i)].y; });
}

template <int B, bool GHOST = false>
__global__ CFX_KR_ACTION_BOUNDS(B) void kr_action(RingCtx c, RingOut o, JobQueue q, RingJob *jobRecs, int G, int nLaneBlocks, int nLLBlocks,
                                                             const RingHist rh) {
    const int w = (int) blockIdx.x, t = (int) threadIdx.x;
    TRACE_STAMP(0);
    if (w >= nLaneBlocks + nLLBlocks) {  // trailing blocks: the per-laneLink notify sources for the cross phase, then Lane::history
        if (w >= rh.firstBlock) ringLaneHistory(c, rh.h, (w - rh.firstBlock) * B + t);
        else llstateRing(c, (w - nLaneBlocks - nLLBlocks) * B + t);
        TRACE_STAMP(4);
        return;
    }
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sPre[B + 1];
    __shared__ int2 sGeo[B];
    __shared__ int sHead[B];
    __shared__ double2 sLM[B];
    __shared__ int4 sHop[B];
    __shared__ unsigned char sAdm[B];
    __shared__ double sDis[B], sSpeed[B];
    __shared__ int sTempl[B];
    __shared__ int sWave[B / 64];
    const cfx_vehicle_template *tv = c.t.templ;
    // ---- the block's drivables: G lanes, or B laneLinks
    const bool laneBlock = w < nLaneBlocks;
    const int g = laneBlock ? G : B;
    const int d0 = laneBlock ? w * G : c.n.L + (w - nLaneBlocks) * B;
    const int dEnd = laneBlock ? c.n.L : c.n.L + c.n.K;
    const int dMine = d0 + t;
    int n = 0;
    if (t < g && dMine < dEnd) {
        const int2 geo = c.ringGeo[dMine];
        const int head = c.head[dMine];
        n = c.cnt[dMine];
        const bool admitted = laneBlock && c.admitStep[dMine] == c.step;
        if (admitted) n += 1;
        sAdm[t] = admitted ? 1 : 0;
        sLM[t] = c.n.drvLM[dMine];
        sHop[t] = laneBlock ? c.n.laneLL4[dMine] : make_int4(-2, -2, -2, -2);  // a lane head's first hop, see findHeadLeader
        sGeo[t] = geo;
        sHead[t] = head;
    }
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = t; i < nd; i += B) dst[i] = src[i];
        tv = sT;
    }
    // block-wide inclusive prefix sum of the counts
    int incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += up;
    }
    if constexpr (B > 64) {
        if ((t & 63) == 63) sWave[t >> 6] = incl;
        __syncthreads();
        for (int i = 0; i < (t >> 6); ++i) incl += sWave[i];
    }
    sPre[t + 1] = incl;
    if (t == 0) sPre[0] = 0;
    __syncthreads();
    const int T = sPre[B];
    // feedback for the host's choice of lanes per block (a block that needs a second pass is the step's slowest)
    if (t == 0 && T > (B - 1) * 3 / 4) atomicMax(&o.sc->actionMaxT, T);
    // (letting the laneLink blocks compute their own laneLinks' notify sources instead of the trailing blocks was measured in
    // round 3: 12.4 -> 14.9 us at 30x30 — the laneLink blocks then become the longest ones)
    TRACE_STAMP(1);
    const RingPush push{q, jobRecs, c.n.L};
    for (int qb = 0; qb < T; qb += B - 1) {
        const int qv = qb + t - 1;  // thread 0 holds the vehicle ahead of the window (leader data only)
        const bool valid = qv >= 0 && qv < T;
        int i = 0, idx = 0, slot = 0;
        SlotIn in;
        in.vid = -1;
        if (valid) {
            // the drivable this list position belongs to: the last i with sPre[i] <= qv
            int lo = 0, hi = g;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (sPre[mid] <= qv) lo = mid;
                else hi = mid;
            }
            i = lo;
            idx = qv - sPre[i];
            slot = ringSlot(sGeo[i], sHead[i], idx);
            const double2 kv = c.kin[slot];  // the two records of the slot: 2 x 16 B, adjacent lanes adjacent in memory
            const int4 mv = c.meta[slot];
            in.dis = kv.x;
            in.speed = kv.y;
            in.templIdx = mv.x;
            in.nd0 = mv.y;
            in.flags = mv.z;
            in.lastRoadFlags = mv.z;
            sDis[t] = in.dis;
            sSpeed[t] = in.speed;
            sTempl[t] = in.templIdx;
        }
        __syncthreads();
        TRACE_STAMP(2);
        if (valid && t > 0) {
            in.vid = 0;  // (the vehicle number is loaded where it is needed: custom speed, leaving the drivable)
            in.d = d0 + i;
            in.head = idx == 0;
            in.idx = idx;
            in.nNow = sPre[i + 1] - sPre[i];
            in.leaderSlot = 0;
            if (idx > 0) {
                in.disPrev = sDis[t - 1];
                in.speedPrev = sSpeed[t - 1];
                in.templPrev = sTempl[t - 1];
                in.leaderSlot = ringSlot(sGeo[i], sHead[i], idx - 1);
            }
            if (in.flags & (kFlagCustom | kFlagStateGap)) {
                in.vid = c.s.vid[slot];
                c.meta[slot].z = in.flags & ~(kFlagCustom | kFlagStateGap);  // Vehicle::update clears isCustomSpeedSet (vehicle.cpp:120-122)
            }
            in.lm = sLM[i];
            in.hop = (in.nd0 >= c.n.L) ? sHop[i] : make_int4(-2, -2, -2, -2);
            in.laneAdmitted = sAdm[i] != 0;
            in.endLane = -1;  // (the end lanes from the lane's static tables, as kw_action has them: measured in round 4, 12.0 us either way)
            actionOneRounds<GHOST>(c, o, tv, slot, in, push);
        }
        __syncthreads();
        TRACE_STAMP(3);
    }
    TRACE_STAMP(4);
#ifdef CFX_TRACE
    if (t == 0) g_trace[(size_t) w * 8 + 5] = T;
#endif
}

// The same phase with WAVE-granular passes (the default).  The preamble is kr_action's — the block's first threads read the
// rings and lane tables of G lanes (or B laneLinks), one block-wide prefix sum, ONE barrier — but the block's vehicle list is
// then dealt out in chunks of 64 to the block's wavefronts, each of which runs on its own: a vehicle takes its leader's
// {dis, speed, template} from the lane below it in the wavefront (`__shfl_up`; lane 0 of a chunk loads the vehicle ahead of
// the chunk together with its own), so there is no LDS window and no barrier in the loop.  A block of 190 vehicles keeps
// three wavefronts busy and lets the fourth go at once (the block form keeps 256 threads through two barriers per pass);
// large networks take more lanes per block and every wavefront walks several chunks.  The end lanes of the laneLinks that
// leave a lane come with the lane's static tables, so the tail records behind the next laneLink are requested in round A.
template <int B, bool GHOST = false>
__global__ __launch_bounds__(B) void kw_action(RingCtx c, RingOut o, JobQueue q, RingJob *jobRecs, int G, int nLaneBlocks, int nLLBlocks,
                                                             const RingHist rh) {
    const int w = (int) blockIdx.x, t = (int) threadIdx.x;
    if (w >= nLaneBlocks + nLLBlocks) {  // trailing blocks: the per-laneLink notify sources for the cross phase, then Lane::history
        if (w >= rh.firstBlock) ringLaneHistory(c, rh.h, (w - rh.firstBlock) * B + t);
        else llstateRing(c, (w - nLaneBlocks - nLLBlocks) * B + t);
        return;
    }
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sPre[B + 1];
    __shared__ int2 sGeo[B];
    __shared__ int sHead[B];
    __shared__ double2 sLM[B];
    __shared__ int4 sHop[B];
    __shared__ int4 sEnd[B];
    __shared__ unsigned char sAdm[B];
    __shared__ int sWave[B / 64];
    const cfx_vehicle_template *tv = c.t.templ;
    const bool laneBlock = w < nLaneBlocks;
    const int g = laneBlock ? G : B;
    const int d0 = laneBlock ? w * G : c.n.L + (w - nLaneBlocks) * B;
    const int dEnd = laneBlock ? c.n.L : c.n.L + c.n.K;
    const int dMine = d0 + t;
    int n = 0;
    if (t < g && dMine < dEnd) {
        const int2 geo = c.ringGeo[dMine];
        const int head = c.head[dMine];
        n = c.cnt[dMine];
        const bool admitted = laneBlock && c.admitStep[dMine] == c.step;
        if (admitted) n += 1;
        sAdm[t] = admitted ? 1 : 0;
        sLM[t] = c.n.drvLM[dMine];
        sHop[t] = laneBlock ? c.n.laneLL4[dMine] : make_int4(-2, -2, -2, -2);
        sEnd[t] = laneBlock ? c.n.laneEnd4[dMine] : make_int4(-1, -1, -1, -1);
        sGeo[t] = geo;
        sHead[t] = head;
    }
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = t; i < nd; i += B) dst[i] = src[i];
        tv = sT;
    }
    int incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += up;
    }
    if constexpr (B > 64) {
        if ((t & 63) == 63) sWave[t >> 6] = incl;
        __syncthreads();
        for (int i = 0; i < (t >> 6); ++i) incl += sWave[i];
    }
    sPre[t + 1] = incl;
    if (t == 0) sPre[0] = 0;
    __syncthreads();
    const int T = sPre[B];
    // feedback for the host's choice of lanes per block (small networks: every wavefront should need one chunk only)
    if (t == 0 && T > B * 3 / 4) atomicMax(&o.sc->actionMaxT, T);
    const RingPush push{q, jobRecs, c.n.L};
    const int lane = t & 63;
    for (int q0 = (t >> 6) * 64; q0 < T; q0 += B) {  // chunk q0 / 64 belongs to wavefront (q0 / 64) mod (B / 64)
        const int qv = q0 + lane;
        const bool valid = qv < T;
        int i = 0, idx = 0, slot = 0;
        int2 geo = make_int2(0, 0);
        int head = 0;
        SlotIn in;
        in.vid = 0;  // (the vehicle number is loaded where it is needed: custom speed, leaving the drivable)
        in.dis = 0.0;
        in.speed = 0.0;
        in.templIdx = 0;
        in.nd0 = -1;
        in.flags = 0;
        double2 kp = make_double2(0.0, 0.0);
        int tp = 0;
        if (valid) {
            int lo = 0, hi = g;  // the drivable this list position belongs to: the last i with sPre[i] <= qv
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (sPre[mid] <= qv) lo = mid;
                else hi = mid;
            }
            i = lo;
            idx = qv - sPre[i];
            geo = sGeo[i];
            head = sHead[i];
            slot = ringSlot(geo, head, idx);
            const double2 kv = c.kin[slot];  // the two records of the slot: 2 x 16 B, adjacent lanes adjacent in memory
            const int4 mv = c.meta[slot];
            if (lane == 0 && idx > 0) {  // the vehicle ahead of the chunk: this lane's leader
                const int ls = ringSlot(geo, head, idx - 1);
                kp = c.kin[ls];
                tp = c.meta[ls].x;
            }
            in.dis = kv.x;
            in.speed = kv.y;
            in.templIdx = mv.x;
            in.nd0 = mv.y;
            in.flags = mv.z;
            in.lastRoadFlags = mv.z;
        }
        // the leader inside the drivable is the lane below (all lanes take part in the exchange)
        const double disUp = __shfl_up(in.dis, 1, 64), speedUp = __shfl_up(in.speed, 1, 64);
        const int templUp = __shfl_up(in.templIdx, 1, 64);
        if (!valid) continue;  // (a chunk's invalid lanes are its last ones: nobody reads them)
        in.d = d0 + i;
        in.head = idx == 0;
        in.idx = idx;
        in.nNow = sPre[i + 1] - sPre[i];
        in.leaderSlot = 0;
        in.disPrev = lane == 0 ? kp.x : disUp;
        in.speedPrev = lane == 0 ? kp.y : speedUp;
        in.templPrev = lane == 0 ? tp : templUp;
        if (idx > 0) in.leaderSlot = ringSlot(geo, head, idx - 1);
        if (in.flags & (kFlagCustom | kFlagStateGap)) {
            in.vid = c.s.vid[slot];
            c.meta[slot].z = in.flags & ~(kFlagCustom | kFlagStateGap);  // Vehicle::update clears isCustomSpeedSet (vehicle.cpp:120-122)
        }
        in.lm = sLM[i];
        in.hop = make_int4(-2, -2, -2, -2);
        in.endLane = -1;
        if (in.nd0 >= c.n.L) {
            in.hop = sHop[i];
            const int4 en = sEnd[i];
            const int ll = in.nd0 - c.n.L;
            in.endLane = in.hop.x == ll ? en.x : (in.hop.y == ll ? en.y : (in.hop.z == ll ? en.z : (in.hop.w == ll ? en.w : -1)));
        }
        in.laneAdmitted = sAdm[i] != 0;
        actionOneRounds<GHOST>(c, o, tv, slot, in, push);
    }
}

// The LIST form of the phase (large networks).  kw_action pays a block-wide preamble — ring geometry, prefix sum, a binary
// search per vehicle — before its first slot load, and a block's vehicles rarely fill its chunks (28 lanes x 4.3 vehicles =
// 120 of 256 threads at 1 M vehicles).  Here the step first writes the list of its vehicles, {slot, drivable, list index,
// vehicles on the drivable | admitted << 31} in drivable order (kr_index: an exclusive scan over the drivables' counts with
// the decoupled look-back k_scan uses, then the entries), and kl_action is
// one thread per list entry: round 1 the entry, round 2 the slot's two records, the leader's (wavefront exchange) and the
// drivable's static tables, then rounds A / B / C of actionOneRounds — the dense layout's chain of four rounds, on rings.
constexpr int kIndexBlock = 1024;                      // threads per tile (sixteen wavefronts share a tile's expansion)
constexpr int kIndexItems = 1;                         // drivables per thread
constexpr int kIndexTile = kIndexBlock * kIndexItems;  // drivables per tile: 100x100 is 470 tiles, all resident at once
constexpr int kIndexWindow = 8192;                // list entries a tile expands per pass (a tile holds about 4400 at 1 M vehicles)

// One tile = 1024 drivables, one per thread.  A tile publishes its TOTAL right after its block scan
// (nothing waits on a predecessor before publishing: no serial chain through the tiles), adds up its predecessors' totals,
// and expands its drivables into list entries through LDS: every thread marks the window positions its drivables own (two
// bytes each), then the block writes the window's entries side by side, 16 bytes per thread.  Tiles are block indices while
// the whole grid is certainly resident (1024 tiles = a million drivables); beyond that they are handed out in start order
// by a ticket — 11 ns each, the rate of returning atomics on one word: 1880 tiles of 256 drivables spent 21 of their 32 us
// queueing for it (profiles/r05_trace_kr_index.txt).
__global__ __launch_bounds__(kIndexBlock) void kr_index(RingCtx c, unsigned long long *granules, int32_t *ticket, unsigned epoch,
                                                   int4 *list, int listCap, int32_t *listCount, DevScalars *sc) {
    static_assert(kIndexTile <= 65536, "a window position names its drivable in two bytes");
    __shared__ int smem[kIndexBlock / 64];
    __shared__ int wsum[kIndexBlock / 64];
    __shared__ int tileShared;
    __shared__ int sOff[kIndexTile], sHead[kIndexTile], sN[kIndexTile];
    __shared__ int2 sGeo[kIndexTile];
    __shared__ unsigned short sOwner[kIndexWindow];
    const int t = (int) threadIdx.x;
    KSTAMP(10, 0);
    int tile = (int) blockIdx.x;
    if (ticket) {
        if (t == 0) tileShared = atomicAdd(ticket, 1);
        __syncthreads();
        tile = tileShared;
    }
    const int D = c.n.L + c.n.K;
    const int d0 = tile * kIndexTile + t * kIndexItems;
    int n[kIndexItems], head[kIndexItems], adm[kIndexItems];
    int2 geo[kIndexItems];
    int sum = 0;
    for (int i = 0; i < kIndexItems; ++i) {
        const int d = d0 + i;
        n[i] = 0;
        head[i] = 0;
        adm[i] = -1;
        geo[i] = make_int2(0, 0);
        if (d < D) {
            n[i] = c.cnt[d];
            head[i] = c.head[d];
            geo[i] = c.ringGeo[d];
            if (d < c.n.L) adm[i] = c.admitStep[d];
        }
    }
    for (int i = 0; i < kIndexItems; ++i) {
        adm[i] = (d0 + i < c.n.L && adm[i] == c.step) ? 1 : 0;
        n[i] += adm[i];
        sum += n[i];
    }
    const int lane = t & 63, w = t >> 6;
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < kIndexBlock / 64; ++i) {
            smem[i] = run;
            run += wsum[i];
        }
        tileShared = run;
        __hip_atomic_store(&granules[tile], ((unsigned long long) epoch << 32) | (unsigned) run, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    KSTAMP(10, 1);
    const int offT = smem[w] + incl - sum;  // this thread's first entry within the tile
    const int T = tileShared;
    __syncthreads();
    {
        int off = offT;
        for (int i = 0; i < kIndexItems; ++i) {
            const int o = t * kIndexItems + i;
            sOff[o] = off;
            sHead[o] = head[i];
            sN[o] = n[i] | (adm[i] << 31);
            sGeo[o] = geo[i];
            off += n[i];
        }
    }
    int pre = 0;
    for (int p = t; p < tile; p += kIndexBlock) {
        unsigned spins = 0;
        for (;;) {
            const unsigned long long x = __hip_atomic_load(&granules[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    KSTAMP(10, 2);
    for (int w0 = 0; w0 < T; w0 += kIndexWindow) {
        int off = offT;
        for (int i = 0; i < kIndexItems; ++i) {
            const int lo = max(0, w0 - off), hi = min(n[i], w0 + kIndexWindow - off);
            for (int j = lo; j < hi; ++j) sOwner[off + j - w0] = (unsigned short) (t * kIndexItems + i);
            off += n[i];
        }
        __syncthreads();
        const int wn = min(kIndexWindow, T - w0);
        for (int q = t; q < wn; q += kIndexBlock) {
            const int o = sOwner[q];
            const int idx = w0 + q - sOff[o];
            const int at = tileOff + w0 + q;
            if (at < listCap) {
                const int4 entry = make_int4(ringSlot(sGeo[o], sHead[o], idx), tile * kIndexTile + o, idx, sN[o]);
                if (CFX_KL_NT & 2) ntStore4(&list[at], entry);
                else list[at] = entry;
            }
        }
        __syncthreads();
    }
    KSTAMP(10, 3);
    if (tile == (int) gridDim.x - 1 && t == 0) {
        *listCount = min(tileOff + T, listCap);
        if (tileOff + T > listCap) sc->overflow = 13;  // the host sizes the list from its own bound of the running vehicles
    }
}

#ifndef CFX_KL_WAVES
#define CFX_KL_WAVES 5  // (97 registers unasked: one more than five wavefronts per SIMD allow; 48.5 us instead of 51.8.  6 spills: 73.7)
#endif
#ifndef CFX_KL_BLOCK
#define CFX_KL_BLOCK 256  // threads of a kl_action workgroup (a multiple of 64; the host sizes the list to whole workgroups)
#endif
constexpr int kListBlock = CFX_KL_BLOCK;
#define CFX_KL_BOUNDS __launch_bounds__(kListBlock, CFX_KL_WAVES)  // (second argument: wavefronts per SIMD)
template <bool GHOST = false>
__global__ CFX_KL_BOUNDS void kl_action(RingCtx c, RingOut o, JobQueue q, RingJob *jobRecs, const int4 *list,
                                                    const int32_t *listCount, int nVehBlocks, int32_t *ticket, const RingHist rh) {
    const int w = (int) blockIdx.x, t = (int) threadIdx.x;
    if (w >= nVehBlocks) {  // trailing blocks: the per-laneLink notify sources for the cross phase, then Lane::history
        if (w >= rh.firstBlock) ringLaneHistory(c, rh.h, (w - rh.firstBlock) * kListBlock + t);
        else llstateRing(c, (w - nVehBlocks) * kListBlock + t);
        return;
    }
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    const int qv = w * kListBlock + t;
    const int total = *listCount;
    if (qv == 0 && ticket) *ticket = 0;  // kr_index of this step is done; re-arm its tile counter for the next one
    KSTAMP(11, 0);
    const int4 e = (CFX_KL_NT & 1) ? ntLoad4(&list[qv]) : list[qv];  // (the list is allocated to whole blocks of the launch)
    int ls = 0;
    if ((t & 63) == 0 && qv > 0) ls = list[qv - 1].x;  // the vehicle ahead of the wavefront's first one, if it has a leader
    if (w * kListBlock >= total) return;
    const cfx_vehicle_template *tv = c.t.templ;
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = t; i < nd; i += kListBlock) dst[i] = src[i];
        tv = sT;
        __syncthreads();
    }
    const bool valid = qv < total;
    const int lane = t & 63;
    const int slot = valid ? e.x : 0, d = e.y, idx = valid ? e.z : 0;
    KSTAMP(11, 1);
    SlotIn in;
    in.vid = 0;  // (the vehicle number is loaded where it is needed: custom speed, leaving the drivable)
    in.dis = 0.0;
    in.speed = 0.0;
    in.templIdx = 0;
    in.nd0 = -1;
    in.flags = 0;
    double2 kp = make_double2(0.0, 0.0);
    int tp = 0;
    double2 lm = make_double2(0.0, 0.0);
    int4 hop = make_int4(-2, -2, -2, -2), en = make_int4(-1, -1, -1, -1);
    if (valid) {
        const double2 kv = (CFX_KL_NT & 8) ? ntLoad2(&c.kin[slot]) : c.kin[slot];
        const int4 mv = (CFX_KL_NT & 16) ? ntLoad4(&c.meta[slot]) : c.meta[slot];
        lm = c.n.drvLM[d];
        if (d < c.n.L) {
            hop = c.n.laneLL4[d];
            en = c.n.laneEnd4[d];
        }
        if (idx > 0 && lane == 0) {
            kp = c.kin[ls];
            tp = c.meta[ls].x;
        }
        in.dis = kv.x;
        in.speed = kv.y;
        in.templIdx = mv.x;
        in.nd0 = mv.y;
        in.flags = mv.z;
        in.lastRoadFlags = mv.z;
    }
    const int slotUp = __shfl_up(slot, 1, 64);
    const double disUp = __shfl_up(in.dis, 1, 64), speedUp = __shfl_up(in.speed, 1, 64);
    const int templUp = __shfl_up(in.templIdx, 1, 64);
    if (!valid) return;
    if (lane != 0) ls = slotUp;
    in.d = d;
    in.head = idx == 0;
    in.idx = idx;
    in.nNow = e.w & 0x7fffffff;
    in.leaderSlot = idx > 0 ? ls : 0;
    in.disPrev = lane == 0 ? kp.x : disUp;
    in.speedPrev = lane == 0 ? kp.y : speedUp;
    in.templPrev = lane == 0 ? tp : templUp;
    if (in.flags & (kFlagCustom | kFlagStateGap)) {
        in.vid = c.s.vid[slot];
        c.meta[slot].z = in.flags & ~(kFlagCustom | kFlagStateGap);  // Vehicle::update clears isCustomSpeedSet (vehicle.cpp:120-122)
    }
    in.lm = lm;
    in.hop = make_int4(-2, -2, -2, -2);
    in.endLane = -1;
    if (in.nd0 >= c.n.L) {
        in.hop = hop;
        const int ll = in.nd0 - c.n.L;
        in.endLane = hop.x == ll ? en.x : (hop.y == ll ? en.y : (hop.z == ll ? en.z : (hop.w == ll ? en.w : -1)));
    }
    in.laneAdmitted = e.w < 0;
    const RingPush push{q, jobRecs, c.n.L};
    KSTAMP(11, 2);
    actionOneRounds<GHOST>(c, o, tv, slot, in, push);
    KSTAMP(11, 3);
}

// ---------------------------------------------------------------------------------------------- phase 5 + 6 + 8
// Commit: one thread per drivable.  Leavers are (almost always) a prefix of the list: the head moves past them.  Entrants
// are appended behind the stayers by descending new distance (std::sort with vehicleCmp engine.h:21-23; ties: lower vid
// first, as in the twin).  Also commits the step's admission (FIFO pop, running count), Router::update of the entrants,
// TrafficLight::passTime, and re-arms the step's scratch.  Extra blocks do the finish statistics in the reference's order.

__device__ inline void ringCopySlot(const RingCtx &c, int from, int to) {  // general path only: one list element moves
    c.s.vid[to] = c.s.vid[from];
    c.s.drv[to] = c.s.drv[from];
    c.s.prevDrv[to] = c.s.prevDrv[from];
    c.s.routePos[to] = c.s.routePos[from];
    c.s.route[to] = c.s.route[from];
    c.meta[to] = c.meta[from];
    c.blkW[to] = c.blkW[from];  // (the other buffer's records expire with this step)
    c.kinN[to] = c.kinN[from];
    c.slotOf[c.s.vid[from]] = to;
}

// The step's finish statistics (finishStatistics of the dense layout with 64-bit order keys): rank sort of the
// finishers by (drivable, list index) = the order a one-thread reference removes them in (engine.cpp:296-310), travel
// times added up in that order by one thread.
__device__ inline bool ringFinishStatistics(const RingCtx &c, const VidTable &vt, const RingCommit &k, int part, int nParts) {
    __shared__ long long fin[kFinLds];
    __shared__ double term[kFinLds];
    __shared__ int lastShared;
    __shared__ FinMap fm;
    DevScalars *sc = k.sc;
    const int F = finMapLoad(fm, k.finCount, k.finCap);
    const double now = c.step * c.interval;
    // this step's admissions (kr_admit counted them by step parity) belong to the vehicles that took the step
    auto foldAdmissions = [&]() {
        const long long a = sc->admitPending[c.step & 1];
        sc->admitPending[c.step & 1] = 0;
        sc->vehicleSteps += a;
        sc->active += a;
    };
    if (k.exactTimes) {
        const bool lastBlock = exactFinishStatistics(now, vt, sc, F, [&](int i) { return k.finVid[finAt(fm, k.finCap, i)]; }, k.vStateW,
                                                     k.finTicket, part, nParts, 0, k.finCount, k.slotOfW);
        if (lastBlock && threadIdx.x == 0) foldAdmissions();
        return lastBlock;
    }
    const bool inLds = nParts == 1 && F <= kFinLds;
    const int per = (F + nParts - 1) / nParts;
    const int lo = part * per, hi = min(F, lo + per);
    for (int base = lo; base < hi; base += blockDim.x) {
        const int i = base + (int) threadIdx.x;
        const int at = i < hi ? finAt(fm, k.finCap, i) : 0;
        const long long me = i < hi ? k.finKey[at] : 0;
        int rank = 0;
        for (int cb = 0; cb < F; cb += kFinLds) {
            const int cn = min(kFinLds, F - cb);
            __syncthreads();
            for (int j = threadIdx.x; j < cn; j += blockDim.x) fin[j] = k.finKey[finAt(fm, k.finCap, cb + j)];
            __syncthreads();
            if (i < hi)
                for (int j = 0; j < cn; ++j) rank += fin[j] < me;
        }
        if (i < hi) {
            const int vid = k.finVid[at];
            const double tt = now - vt.enterTime[vid];
            k.vStateW[vid] = 2;
            k.slotOfW[vid] = -1;
            if (inLds) term[rank] = tt;
            else k.finTerm[rank] = tt;
        }
    }
    bool last = true;
    if (nParts > 1) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) lastShared = atomicAdd(k.finTicket, 1) == nParts - 1;
        __syncthreads();
        last = lastShared != 0;
        if (!last) return false;
        if (threadIdx.x == 0) *k.finTicket = 0;
        __threadfence();
    }
    __syncthreads();
    const double cum = orderedSum(sc->cumulativeTravelTime, F, term, k.finTerm, inLds);
    if (threadIdx.x == 0) {
        sc->cumulativeTravelTime = cum;
        sc->vehicleSteps += sc->active;
        sc->finishedCnt += F;
        sc->active -= F;
        finCountsClear(k.finCount);
        foldAdmissions();
    }
    return last;
}

// One drivable's commit (the body of kr_commit's loop; kr_admit<true> runs it in front of the NEXT step's admission).
__device__ inline void commitDrivable(const RingCtx &c, const RingCommit &k, const int d, CommitOut *out) {
    const int4 sc = k.scratch[d];  // {leavers, largest list index among them, entrant list, entrants}
    const bool admitted = d < c.n.L && c.admitStep[d] == c.step;
    if (out) {
        out->touched = 0;
        out->entrants = 0;
    }
    if (!admitted && sc.x == 0 && sc.z < 0) return;  // nothing happened on this drivable: nothing is written
    const int2 geo = c.ringGeo[d];
    int head = c.head[d];
    int n = c.cnt[d];
    if (admitted) {  // commit this step's admission (phase 2): the FIFO pop and the vehicle's state
        const int2 rec = c.admitRec[d];
        k.waitHead[d] = rec.y;
        k.vStateW[rec.x] = 1;
        n += 1;
    }
    if (sc.x > 0) {
        if (sc.y + 1 != sc.x) {
            // leavers are not a prefix of the list (a vehicle ran past the end of its drivable before the one ahead of
            // it did): close the gaps, stayers keep their order and move towards the tail
            int wr = n - 1;
            for (int r = n - 1; r >= 0; --r) {
                const int from = ringSlot(geo, head, r);
                if (c.kinN[from].y < 0.0) continue;
                if (wr != r) ringCopySlot(c, from, ringSlot(geo, head, wr));
                --wr;
            }
        }
        head = (head + sc.x) & geo.y;
        n -= sc.x;
    }
    int m = 0, tailRank = -1;
    TailRec tail{};
    for (int e = sc.z; e >= 0; e = k.movers[e].nextIn) {
        const MoverRec r = k.movers[e];
        int rank = 0;
        for (int f = sc.z; f >= 0; f = k.movers[f].nextIn) {
            if (f == e) continue;
            const double od = k.movers[f].dis;
            const bool tieBefore = od == r.dis && k.movers[f].vid < r.vid;
            rank += (od > r.dis) || tieBefore;
            if (tieBefore) k.sc->tieDrv[atomicAdd((unsigned long long *) &k.sc->tieEvents, 1ULL) & 7ULL] = d;
        }
        ++m;
        if (n + rank > geo.y) {  // the ring is full: refuse (reported as an error by the next cfx_step / getter)
            k.sc->overflow = 8;
            continue;
        }
        const int slot = ringSlot(geo, head, n + rank);
        int rp = r.routePos;
        c.s.vid[slot] = r.vid;
        c.s.drv[slot] = d;
        c.s.prevDrv[slot] = r.oldDrv;
        c.s.route[slot] = r.route;
        c.blkW[slot] = make_int2(r.blockerVid, c.step);
        int next, enterLLT, onLast = 0;
        if (d < c.n.L) {  // Router::update router.cpp:78-94, then Router::getNextDrivable from the road it stopped at
            enterLLT = CFX_INT_MAX;
            const int base = c.t.routeStart[r.route], len = c.t.routeStart[r.route + 1] - base;
            const int road = c.n.laneRoad[d], laneIdx = c.n.laneIndex[d];
            // The road is almost always at the vehicle's route position or the one behind it: both are looked at in one
            // round, together with their rows of the next-drivable table and the route's last road (the loop form walked
            // routeRoads -> routeRoads -> nextStart -> nextLL one load after the other, per entrant).
            const int p0 = rp < len ? rp : len - 1, p1 = rp + 1 < len ? rp + 1 : len - 1;
            const int road0 = c.t.routeRoads[base + p0], road1 = c.t.routeRoads[base + p1];
            int row = c.t.nextStart[base + p0];
            const int row1 = c.t.nextStart[base + p1], roadLast = c.t.routeRoads[base + len - 1];
            if (rp < len && road0 != road) {
                ++rp;
                row = row1;
                if (rp < len && road1 != road) {
                    ++rp;
                    while (rp < len && c.t.routeRoads[base + rp] != road) ++rp;
                    if (rp < len) row = c.t.nextStart[base + rp];
                }
            }
            next = -1;
            if (rp < len) {
                const int ll = c.t.nextLL[row + laneIdx];
                next = ll < 0 ? -1 : c.n.L + ll;
            }
            if (next < 0 && road == roadLast) onLast = 2;  // Router::isLastRoad: flags bit 1 (finishAction)
        } else {
            enterLLT = c.step;
            next = c.n.llEndLane[d - c.n.L];
        }
        c.s.routePos[slot] = rp;
        c.meta[slot] = make_int4(r.templ, next, onLast, enterLLT);
        c.kinN[slot] = make_double2(r.dis, r.speed);
        c.slotOf[r.vid] = slot;
        if (rank > tailRank) {  // the last of the entrants becomes the drivable's tail
            tailRank = rank;
            tail.dis = r.dis;
            tail.speed = r.speed;
            tail.slot = slot;
            tail.templ = r.templ;
            tail.prevDrv = r.oldDrv;
        }
    }
    n += m;
    // the tail record of this step's end where the tail changed hands (a tail that stayed wrote its own, finishAction)
    if (tailRank >= 0) {
        tail.tag = c.step;
        c.tailW[d] = tail;
    } else if (sc.x > 0) {
        if (n == 0) {
            tail.slot = -1;
        } else {  // (the general path above may have moved the tail vehicle; with a prefix of leavers this rewrites the same)
            const int ts = ringSlot(geo, head, n - 1);
            const double2 kt = c.kinN[ts];
            tail.dis = kt.x;
            tail.speed = kt.y;
            tail.slot = ts;
            tail.templ = c.meta[ts].x;
            tail.prevDrv = c.s.prevDrv[ts];
        }
        tail.tag = c.step;
        c.tailW[d] = tail;
    }
    if (n > geo.y) k.sc->overflow = 8;
    else if (n + min(8, (geo.y + 1) / 2) > geo.y) k.sc->ringNearFull = 1;  // (the host doubles every capacity)
    c.head[d] = head;
    c.cnt[d] = n;
    if (k.hostCnt && d < c.n.L) k.hostCnt[d] = n;  // (the host's copy: a lane that was not touched keeps its value there too)
    k.scratch[d] = make_int4(0, -1, -1, 0);
    if (out) {
        out->touched = 1;
        out->head = head;
        out->n = n;
        out->tailWritten = (tailRank >= 0 || sc.x > 0) ? 1 : 0;
        out->entrants = m;
        out->tail = tail;
    }
}

__device__ inline int commitStatBlocks(const RingCommit &k) { return k.nStatBlocks; }
// one of the statistics blocks at the end of a commit's grid
__device__ inline void commitStatBlock(const RingCtx &c, const RingCommit &k, const VidTable &vt, int part) {
    if (part == 0 && threadIdx.x < kJobShards) k.jobCount[threadIdx.x * kJobShardStride] = 0;
    const bool last = ringFinishStatistics(c, vt, k, part, k.nStatBlocks);
    if (last && threadIdx.x == 0 && k.hostMirror) {
        k.hostMirror->sc = *k.sc;
        k.sc->actionMaxT = 0;
        k.hostMirror->slots = (int32_t) k.sc->active;
        __hip_atomic_store(&k.hostMirror->progress, ((unsigned long long) (c.step + 1) << 32) | (unsigned) k.sc->active,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ inline void commitClearMasks(const RingCtx &c, const RingCommit &k, int gid, int stride) {
    for (int i = gid; i < k.nMaskWords; i += stride) c.interMask[i] = 0ULL;
}

__global__ __launch_bounds__(kBlock) void kr_commit(RingCtx c, RingCommit k, VidTable vt, RingHalo rh) {
    const int nBody = (int) gridDim.x - k.nStatBlocks;
    if ((int) blockIdx.x >= nBody) {
        commitStatBlock(c, k, vt, (int) blockIdx.x - nBody);
        return;
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = nBody * blockDim.x;
    commitClearMasks(c, k, gid, stride);
    if (!k.rlTrafficLight && !k.lightsDone) passTimeAll(c.n, k.curPhase, k.remain, c.interval, gid, stride);
    const int D = c.n.L + c.n.K;
    for (int d = gid; d < D; d += stride) {
        CommitOut co;
        commitDrivable(c, k, d, &co);
        if (rh.on && d < c.n.L) {  // tiling: a cut lane's message, written by the thread that has just committed the lane
            const int ci = rh.cutIndex[d];
            if (ci >= 0) {
                ringHaloExport(c, k, rh, d, ci, co);
                if (rh.io.nSignal > 0) {  // mailbox path: the last cut lane to finish publishes the step's epoch (k_halo_export)
                    const int nCut = rh.h.nGhost + rh.h.nImport;
                    __threadfence_system();
                    if (atomicAdd(rh.io.ticket, 1) == nCut - 1) {
                        __threadfence_system();
                        *rh.io.ticket = 0;
                        for (int p = 0; p < rh.io.nSignal; ++p)
                            __hip_atomic_store(rh.io.signalFlag[p], rh.io.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- slow paths
// Getters / archive / growth: the ring order as dense arrays (Drivable::vehicles order: by drivable, front to back).
// `off` = exclusive prefix sum of cnt over drivables.
struct RingDense {
    int32_t *vid, *drv, *prevDrv, *blockerVid, *enterLLT, *routePos, *leaderVid;
    uint8_t *flags;
    double *dis, *speed, *gap;
};

__global__ void kr_gather(RingCtx c, const int32_t *off, RingDense out, int wantLeader) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= c.n.L + c.n.K) return;
    const int n = c.cnt[d];
    if (n == 0) return;
    const int2 geo = c.ringGeo[d];
    const int head = c.head[d], o = off[d];
    for (int i = 0; i < n; ++i) {
        const int s = ringSlot(geo, head, i);
        out.vid[o + i] = c.s.vid[s];
        out.drv[o + i] = d;
        out.prevDrv[o + i] = c.s.prevDrv[s];
        out.blockerVid[o + i] = blockerVid(c, s);
        const double2 kv = c.kin[s];
        const int4 mv = c.meta[s];
        out.enterLLT[o + i] = mv.w;
        out.routePos[o + i] = c.s.routePos[s];
        out.flags[o + i] = (uint8_t) (mv.z & (kFlagCustom | kFlagStateGap));
        out.dis[o + i] = kv.x;
        out.speed[o + i] = kv.y;
        if (wantLeader) {  // Vehicle::updateLeaderAndGap as of now (findLeader of cfx_kernels.h)
            double gap = 0;
            int ls;
            if (i > 0) {
                ls = ringSlot(geo, head, i - 1);
                gap = c.kin[ls].x - c.t.templ[c.meta[ls].x].len - kv.x;
            } else {
                ls = findHeadLeader(c, c.t.templ, s, d, kv.x, c.t.templ[mv.x].approach_dist, mv.y, c.n.drvLength[d], &gap).slot;
            }
            if ((mv.z & kFlagStateGap) && ls >= 0) gap = c.vGapState[c.s.vid[s]];  // not stepped since the load: the state's gap
            out.leaderVid[o + i] = ls >= 0 ? c.s.vid[ls] : -1;
            out.gap[o + i] = gap;
        }
    }
}

// The inverse (cfx_load_state, growth): dense arrays -> rings.  templ / route come from the vehicle table.
__global__ void kr_scatter_in(RingCtx c, const int32_t *off, RingDense in, VidTable vt) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= c.n.L + c.n.K) return;
    const int n = off[d + 1] - off[d];
    c.head[d] = 0;
    c.cnt[d] = n;
    const int2 geo = c.ringGeo[d];
    for (int i = 0; i < n; ++i) {
        const int s = geo.x + i, j = off[d] + i;
        const int v = in.vid[j];
        const int route = vt.route[v];
        c.s.vid[s] = v;
        c.s.drv[s] = d;
        c.s.prevDrv[s] = in.prevDrv[j];
        c.s.routePos[s] = in.routePos[j];
        c.s.route[s] = route;
        const_cast<int2 *>(c.blkR)[s] = make_int2(in.blockerVid[j], c.step - 1);
        c.kin[s] = make_double2(in.dis[j], in.speed[j]);
        const int nextD = nextOf(c.n, c.t, d, route, in.routePos[j]);
        c.meta[s] = make_int4(vt.templ[v], nextD, (in.flags[j] & (kFlagCustom | kFlagStateGap)) | ((nextD < 0 && isLastRoad(c, d, route)) ? 2 : 0), in.enterLLT[j]);
        c.slotOf[v] = s;
        if (i == n - 1) {
            TailRec r;
            r.dis = in.dis[j];
            r.speed = in.speed[j];
            r.slot = s;
            r.templ = vt.templ[v];
            r.prevDrv = in.prevDrv[j];
            r.tag = c.step - 1;
            const_cast<TailRec *>(c.tailR)[d] = r;
        }
    }
}

// Exclusive prefix sum of the per-drivable counts (out[0] = 0 ... out[n - 1]): where each drivable's vehicles start in the
// dense staging view of the getters / snapshot / ring growth — off the step's path, so ONE 1024-thread block walks the array
// 4096 elements at a time (four per thread, a wavefront scan, the wavefront totals through LDS) and carries the running
// total; 44 k drivables (30x30) are 11 rounds, 481 k (100x100) 118.
constexpr int kOffBlock = 1024;
__global__ __launch_bounds__(kOffBlock) void kr_offsets(const int32_t *in, int32_t *out, int n) {
    __shared__ int sWaveSum[kOffBlock / 64];
    __shared__ int sCarry;
    const int t = (int) threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) sCarry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kOffBlock * 4) {
        const int i0 = base + t * 4;
        int v[4];
        for (int j = 0; j < 4; ++j) v[j] = i0 + j < n ? in[i0 + j] : 0;
        const int mine = v[0] + v[1] + v[2] + v[3];
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) sWaveSum[wv] = incl;
        __syncthreads();
        int before = sCarry + incl - mine;
        for (int i = 0; i < wv; ++i) before += sWaveSum[i];
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < n) out[i0 + j] = before;
            before += v[j];
        }
        __syncthreads();
        if (t == kOffBlock - 1) sCarry = before;  // (the last thread's running total = everything up to the end of this round)
        __syncthreads();
    }
}

__global__ void kr_reset(int D, int32_t *head, int32_t *cnt, int4 *scratch) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    head[d] = 0;
    cnt[d] = 0;
    scratch[d] = make_int4(0, -1, -1, 0);
}

// Lane::history of the last committed step as a launch of its own (the action launch of the NEXT step takes it otherwise: RingHist)
__global__ void kr_lane_history(RingCtx c, LaneHistDev h) { ringLaneHistory(c, h, blockIdx.x * blockDim.x + threadIdx.x); }

__global__ void kr_lane_waiting(RingCtx c, int32_t *out) {  // Engine::getLaneWaitingVehicleCount engine.cpp:636-648
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n.L) return;
    const int2 geo = c.ringGeo[lane];
    const int head = c.head[lane], n = c.cnt[lane];
    int k = 0;
    for (int i = 0; i < n; ++i) k += c.kin[ringSlot(geo, head, i)].y < 0.1;
    out[lane] = k;
}

// Vehicle::setCustomSpeed / Router::setRoute / lookup of one running vehicle: its slot is known
__global__ void kr_set_speed(RingCtx c, int vid) {
    const int s = c.slotOf[vid];
    if (s >= 0) c.meta[s].z |= 1;
}
__global__ void kr_set_route(RingCtx c, int vid, int route) {
    const int s = c.slotOf[vid];
    if (s < 0) return;
    c.s.route[s] = route;
    c.s.routePos[s] = 0;
    const int dNow = c.s.drv[s], nextD = nextOf(c.n, c.t, dNow, route, 0);
    c.meta[s].y = nextD;
    c.meta[s].z = (c.meta[s].z & (kFlagCustom | kFlagStateGap)) | ((nextD < 0 && isLastRoad(c, dNow, route)) ? 2 : 0);
}
__global__ void kr_find_vehicle(RingCtx c, int vid, int32_t *out /*[2]: drivable, routePos*/) {
    const int s = c.slotOf[vid];
    if (s < 0) return;
    out[0] = c.s.drv[s];
    out[1] = c.s.routePos[s];
}

// Developer aid (cfx_config::debug_sync): invariants of the ring state between the action and the cross phase.
// report[0] = first violation code (0 none), report[1..5] = context.
__global__ void kr_validate(RingCtx c, JobQueue q, int nVid, int nRoutes, int ringSlots, int32_t *report) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = c.n.L + c.n.K;
    auto bad = [&](int code, int a, int b, int cc, int dd) {
        if (atomicCAS(&report[0], 0, code) == 0) {
            report[1] = a;
            report[2] = b;
            report[3] = cc;
            report[4] = dd;
        }
    };
    if (gid < D) {
        const int d = gid, n = cntNow(c, d);
        const int2 geo = c.ringGeo[d];
        if (n < 0 || n > geo.y + 1) bad(1, d, n, geo.y, 0);
        for (int i = 0; i < n && i <= geo.y; ++i) {
            const int s = ringSlot(geo, c.head[d], i);
            if (c.s.drv[s] != d) bad(2, d, i, s, c.s.drv[s]);
            if ((unsigned) c.meta[s].x >= (unsigned) c.t.nTempl) bad(3, d, i, s, c.meta[s].x);
            if ((unsigned) c.s.vid[s] >= (unsigned) nVid) bad(4, d, i, s, c.s.vid[s]);
            else if (c.slotOf[c.s.vid[s]] != s) bad(5, d, i, s, c.slotOf[c.s.vid[s]]);
            if ((unsigned) c.s.route[s] >= (unsigned) nRoutes) bad(6, d, i, s, c.s.route[s]);
            if (c.meta[s].y < -1 || c.meta[s].y >= D) bad(7, d, i, s, c.meta[s].y);
            // (a tile: a migrant's previous laneLink lies in the neighbour's tile and is kept as -(its global number + 2), ringHaloImportCut)
            if ((c.s.prevDrv[s] < -1 && !c.n.laneGhost) || c.s.prevDrv[s] >= D) bad(8, d, i, s, c.s.prevDrv[s]);
            const int2 b = c.blkR[s];
            if (b.x < -1 || b.x >= nVid) bad(9, d, i, s, b.x);
        }
    }
    if (gid < c.n.K) {
        const int4 dyn = c.llDyn[gid];
        if (dyn.x < -1 || dyn.x >= ringSlots) bad(10, gid, dyn.x, 0, 0);
        if (dyn.y < -1 || dyn.y >= ringSlots) bad(11, gid, dyn.y, 0, 0);
        if (dyn.z < 0 || dyn.z >= ringSlots) bad(12, gid, dyn.z, dyn.w, 0);
        if (dyn.w < 0 || dyn.w > c.ringGeo[c.n.L + gid].y + 1) bad(13, gid, dyn.w, 0, 0);
    }
    if (gid < kJobShards) {
        const int n = min(q.count[gid * kJobShardStride], q.capacity);
        for (int j = 0; j < n; ++j) {
            const int s = q.jobs[(size_t) gid * q.capacity + j];
            if ((unsigned) s >= (unsigned) ringSlots) bad(14, gid, j, s, n);
            else if ((unsigned) c.s.drv[s] >= (unsigned) D) bad(15, gid, j, s, c.s.drv[s]);
        }
    }
}

}  // namespace cfxd
