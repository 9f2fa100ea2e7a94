// The dense layout's admission and action phases with what the ring layout taught (cfx_ring_kernels.h): per-drivable
// TAIL RECORDS instead of chains through {segment offset, count} -> slot -> {dis, speed, template}, and an action phase
// organised by rounds of memory accesses (actionOneRounds).  Used by dense engines without lane change and tiling — the
// large networks, where the step is bound by (dependent rounds per wave) x (waves / resident waves); lane change and tiles
// keep k_admit / k_action of cfx_kernels.h.  Same arithmetic, same results (tests/test_parity_pins.py forces this path
// on every pinned workload).
#pragma once

#include "cfx_ring_kernels.h"

namespace cfxd {

__device__ __forceinline__ bool viewerIsNew(const StepCtx &c, const SlotIn &in, int d) {
    return d < c.n.L && c.admitStep[d] == c.step && c.cnt[d] == 0;  // head of a lane whose only vehicle was admitted this step
}

// finishAction of cfx_kernels.h for the dense layout with the first hop's lengths and the identity columns in registers
template <bool LC>
__device__ inline void finishAction(const StepCtx &c, const ActionOut &o, const cfx_vehicle_template &t, int s, int d, int vid,
                                    double speed, double dis, double dlen, int nd0, double v, int blockerSlot, int /*idx*/,
                                    int /*nNow*/, LeaverPrefetch lp, int flags = -1) {
    static_assert(!LC, "lane change runs on k_action");
    v = min2(v, 100);  // SimpleLaneChange::yieldSpeed without signals
    // speedTail of cfx_kernels.h (vehicle.cpp:325-331) with Router::onValidLane's "last road" from the slot's flags (lastRoadBit)
    if (nd0 < 0) {
        const bool lastRoad = flags >= 0 ? (flags & 2) != 0 : isLastRoad(c, d, c.s.route[s]);
        if (!lastRoad) v = min2(v, noCollisionSpeed(0, 1, speed, t.max_neg_acc, dlen - dis, c.interval, t.min_gap));
    }
    v = max2(v, speed - t.max_neg_acc * c.interval);
    MoveOut m;
    double deltaDis;
    if (v < 0) {
        deltaDis = 0.5 * speed * speed / t.max_neg_acc;
        v = 0;
    } else {
        deltaDis = (speed + v) * c.interval / 2;
    }
    m.v = v;
    m.ndis = deltaDis + dis;
    m.newDrv = -1;
    if (m.ndis > dlen) {
        if (!lp.valid) {
            lp.nextLen = nd0 >= 0 ? c.n.drvLength[nd0] : 0.0;
            lp.route = c.s.route[s];
            lp.routePos = c.s.routePos[s];
        }
        m.ndis -= dlen;  // (== c.n.drvLength[d])
        int drivable = nd0;
        m.newDrv = drivable >= 0 ? drivable : -2;
        if (drivable >= 0 && m.ndis > lp.nextLen) {  // runs through a whole drivable in one step: the general walk goes on
            int nxt = nextOf(c.n, c.t, drivable, lp.route, lp.routePos);
            for (;;) {
                m.ndis -= c.n.drvLength[drivable];
                drivable = nxt;
                m.newDrv = drivable >= 0 ? drivable : -2;
                if (drivable < 0 || !(m.ndis > c.n.drvLength[drivable])) break;
                nxt = nextOf(c.n, c.t, drivable, lp.route, lp.routePos);
            }
        }
    }
    commitMove(c, o, s, d, vid, m, blockerSlot, true);
}

// Engine::handleWaiting + Lane::available from the lane's tail record; this step's view of every drivable's tail, the wide
// gate records, and the compaction scratch of the step (k_admit of cfx_kernels.h does the same through the slots).
// A step's few spawn records travel in the kernel arguments (SpawnBatch, cfx_ring_kernels.h: kr_admit does the same): each
// lane's thread links its own records into its waiting queue, block 0 writes the vehicle table — no k_spawn_link launch.
// (Tiles: a record whose lane belongs to another tile has lane -1; only its vehicle-table row is written.)
// LANES (cfx_config::dense_form bit 1): the grid covers the lanes only.  What the kernel did for each of the ~3 L laneLinks in
// every step regardless of traffic is gone or has moved: a laneLink's gate record is rewritten only when its intersection
// shows another phase than the one the records were written for (`gatePhase`, a comparison per laneLink and step — whoever
// changed the light: the commit's TrafficLight::passTime, cfx_set_tl_phase(s), a load, a reset, which sets gatePhase to
// -1), their tails "as of this step" are the committed records (linkTailNow), and the compaction scratch of all drivables
// is cleared by the lanes' threads in strides (16 B per drivable, written, nothing read).
template <bool LANES, int NB>
__global__ __launch_bounds__(kBlock) void kd_admit(StepCtx c, int32_t *admitStep, int32_t *waitHead, VidTable vt, CompactScratch cs,
                                                   const SpawnBatchT<NB> batch, int32_t *gatePhase) {
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sLane[NB];
    const int nRecs = batch.n, firstNewVid = batch.firstNewVid;
    for (int i = threadIdx.x; i < nRecs; i += blockDim.x) sLane[i] = batch.lane[i];
    const cfx_vehicle_template *tv = c.t.templ;
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const bool isLane = d < c.n.L, inRange = LANES ? isLane : d < c.n.L + c.n.K;
    TailRec committed{};
    int w = -1, n = 0, base = 0, road = 0, laneIdx = 0;
    if (inRange) committed = c.tailR[d];
    if (isLane) {
        w = waitHead[d];
        n = c.cnt[d];
        base = c.segStart[d];
        road = c.n.laneRoad[d];
        laneIdx = c.n.laneIndex[d];
    }
    int wt = 0, route = 0, nextWait = -1, fn = kFirstNextUnknown;
    uint8_t pending = 0;
    if (w >= 0) {
        wt = vt.templ[w];
        route = vt.route[w];
        nextWait = vt.nextWait[w];
        pending = vt.pendingCustom[w];
        fn = vt.firstNext[w];
    }
    __syncthreads();
    if (nRecs > 0) {
        // the vehicle table of the new vehicles (k_spawn_link): block 0.  Nobody reads these rows in this kernel — a vehicle
        // that is admitted in the step it appears in is taken from its record
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < nRecs; i += blockDim.x) {
                const int v = firstNewVid + batch.vidOff[i];
                vt.priority[v] = batch.priority[i];
                vt.templ[v] = batch.templ[i];
                vt.route[v] = batch.route[i];
                vt.enterTime[v] = batch.enterTime;
                vt.state[v] = 0;
                vt.pendingCustom[v] = 0;
                vt.firstNext[v] = batch.firstNext[i];
            }
        if (isLane) {
            // FIFO append (Lane::pushWaitingVehicle roadnet.h:365-367; nextWait[] of a new vehicle was pre-set to -1): this
            // lane's records, in any order — each hangs behind its predecessor, or becomes the head where the predecessor
            // has left the queue
            int lo = 0, hi = nRecs;  // first record of this lane
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sLane[mid] < d) lo = mid + 1;
                else hi = mid;
            }
            int headRec = -1;
            for (int j = lo; j < nRecs && sLane[j] == d; ++j) {
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
                for (int j = lo; j < nRecs && sLane[j] == d; ++j)
                    if (batch.prevWait[j] == w) nextWait = firstNewVid + batch.vidOff[j];
        }
    }
    if constexpr (LANES) {
        const int D = c.n.L + c.n.K, stride = (int) (gridDim.x * blockDim.x);
        // the gate record of a laneLink whose intersection shows another phase than in the last step (one thread per laneLink,
        // not per intersection: a thread that walked its intersection's 36 laneLinks was the kernel's longest chain).  The
        // phases the records stand for are kept twice, by step parity: this step reads last step's and writes its own
        const int32_t *const seenPrev = gatePhase + ((c.step + 1) & 1) * c.n.I;
        int32_t *const seenNow = gatePhase + (c.step & 1) * c.n.I;
        for (int k = d; k < c.n.K; k += stride) {
            const int in = c.n.llInter[k];
            const int ph = c.curPhase[in];
            if (seenPrev[in] != ph) {
                const int flags = (c.n.phaseAvail[c.n.interAvailStart[in] + ph * c.n.interNRL[in] + c.n.llRoadLink[k]] != 0 ? 1 : 0) |
                                  (c.n.llType[k] << 1) | (c.n.llXStart[k + 1] > c.n.llXStart[k] ? 8 : 0);
                c.llGate[k] = make_int2(flags, c.n.llEndLane[k]);
                c.llGate4[k] = make_int4(flags, c.n.llEndLane[k], c.n.llXStart[k], c.n.llXStart[k + 1]);
            }
            if (c.n.llLocal[k] == 0) seenNow[in] = ph;
        }
        for (int j = d; j < D; j += stride) {
            cs.leaveCnt[j] = 0;
            cs.maxLeaveIdx[j] = -1;
            cs.inCnt[j] = 0;
            cs.inHead[j] = -1;
        }
    }
    if (!inRange) return;
    if constexpr (!LANES) {
        cs.leaveCnt[d] = 0;
        cs.maxLeaveIdx[d] = -1;
        cs.inCnt[d] = 0;
        cs.inHead[d] = -1;
    }
    TailRec now = committed;
    if (committed.tag != c.step - 1) now.slot = -1;
    if (!isLane) {
        const int k = d - c.n.L;
        int flags = (llAvailable(c, k) ? 1 : 0) | (c.n.llType[k] << 1) | (c.n.llXStart[k + 1] > c.n.llXStart[k] ? 8 : 0);
        c.llGate[k] = make_int2(flags, c.n.llEndLane[k]);  // (k_cross / llstate of the generic kernels)
        c.llGate4[k] = make_int4(flags, c.n.llEndLane[k], c.n.llXStart[k], c.n.llXStart[k + 1]);
    } else {
        const int lane = d;
        c.laneTail[lane] = n > 0 ? base + n - 1 : -1;  // overwritten below if a vehicle is admitted
        bool admit = w >= 0;  // Lane::available roadnet.cpp:428-435
        if (admit && now.slot >= 0 && !(now.dis > tv[now.templ].len + tv[wt].min_gap)) admit = false;
        if (admit) {
            int next, onLast;  // Router::getNextDrivable(0) and Router::isLastRoad: known since the vehicle was created (VidTable::firstNext)
            admittedNext(c, fn, lane, road, laneIdx, route, &next, &onLast);
            const int slot = base + n;  // the lane's spare slot
            const double v0 = tv[wt].initial_speed;
            c.s.vid[slot] = w;
            c.s.drv[slot] = lane;
            c.s.prevDrv[slot] = -1;
            c.s.next[slot] = next;
            c.s.blocker[slot] = -1;
            c.s.enterLLT[slot] = CFX_INT_MAX;  // ControllerInfo ctor vehicle.cpp:10-13
            c.s.routePos[slot] = 0;
            c.s.templ[slot] = wt;
            c.s.route[slot] = route;
            c.s.flags[slot] = (uint8_t) (pending | onLast);
            c.s.dis[slot] = 0.0;
            c.s.speed[slot] = v0;
            c.laneTail[lane] = slot;
            c.admitRec[lane] = make_int2(w, nextWait);
            admitStep[lane] = c.step;  // cnt[], the FIFO pop and the running count follow in k_scan (see cntNow)
            now.dis = 0.0;
            now.speed = v0;
            now.slot = slot;
            now.templ = wt;
            now.prevDrv = -1;
        }
    }
    now.tag = c.step;
    c.tailNow[d] = now;
}

// llstate of cfx_kernels.h (the per-laneLink sources of Engine::threadNotifyCross) from the records this path keeps: the end
// lane's tail is ONE 32-byte record (tailNow) instead of laneTail -> slot -> {template, previous drivable, dis, speed}, and
// the light is the bit kd_admit has just put into the laneLink's gate record instead of the chain intersection -> phase ->
// availability table.  Same values, two dependent rounds instead of four — these blocks are a quarter of kd_action's grid
// on a large network and start last.
__device__ inline void llstateTails(const StepCtx &c, int k) {
    if (k >= c.n.K) return;
    const int d = c.n.L + k;
    const int endLane = c.n.llEndLane[k], startLane = c.n.llStartLane[k];
    const int gateFlags = c.llGate4[k].x;
    const int nOn = c.cnt[d], firstOn = c.segStart[d];
    const TailRec tu = c.tailNow[endLane];
    const int nStart = cntNow(c, startLane);
    int f = nStart > 0 ? c.segStart[startLane] : -1;
    const int u = (tu.slot >= 0 && tu.prevDrv == d) ? tu.slot : -1;
    if (f >= 0 && !((gateFlags & 1) && c.s.next[f] == d)) f = -1;
    c.llDyn[k] = make_int4(u, f, firstOn, nOn);
    if (u >= 0 || f >= 0 || nOn > 0) {
        const int in = c.n.llInter[k];
        const int bit = c.n.llLocal[k];
        atomicOr(&c.interMask[c.n.interMaskStart[in] + (bit >> 6)], 1ULL << (bit & 63));
    }
}

// k_action of cfx_kernels.h with the rounds-organised per-vehicle phase.  One slot per thread, no loop: nothing is kept
// alive across iterations, which is the difference between four and five waves per SIMD (86 registers against 109); the
// host sizes the grid to its bound on the slots, and a bound that turns out too small is an error, not a skipped vehicle.
// (workgroup size measured at 1 M vehicles, round 4: 64 -> 43.7 us, 128 -> 44.1, 256 -> 45.6, 512 -> 49.2; a wavefront that
// has finished its slots makes room for the next one without waiting for three others)
#ifndef CFX_KD_ACT_BLOCK
#define CFX_KD_ACT_BLOCK 64
#endif
constexpr int kDenseActBlock = CFX_KD_ACT_BLOCK;
#ifndef CFX_KD_ACTION_WAVES
#define CFX_KD_ACTION_WAVES 5
#endif
__global__ __launch_bounds__(kDenseActBlock, CFX_KD_ACTION_WAVES) void kd_action(StepCtx c, ActionOut o, JobQueue q, int nVehicleBlocks) {
    KSTAMP(6, 0);
    if ((int) blockIdx.x >= nVehicleBlocks) {  // trailing blocks: the per-laneLink notify sources for the cross phase
        llstateTails(c, ((int) blockIdx.x - nVehicleBlocks) * (int) blockDim.x + (int) threadIdx.x);
        KSTAMP(6, 4);
        KNOTE(6, 5, 1);
        return;
    }
    // (the host's grid covers its BOUND on the slots in use; a block wholly beyond the slots that are in use — a sixth of the
    // grid at 1 M vehicles — leaves before it stages anything)
    const int S = c.segStart[c.n.L + c.n.K];
    if ((int) (blockIdx.x * blockDim.x) >= S) return;
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
    KSTAMP(6, 1);
    // (requesting the slot's columns in front of this barrier — so that they travel with the template table and the slot
    // count — was measured in round 4: 9.5 -> 9.3 us at 30x30, 46.8 -> 48.5 us at 1 M vehicles; not kept)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 && S > nVehicleBlocks * (int) blockDim.x) o.sc->overflow = 10;
    if (s < S) {
        SlotIn in = loadSlot(c, s);
        if (in.vid >= 0) {
            if (c.n.laneGhost && in.d < c.n.L && c.n.laneGhost[in.d]) {  // tiling: proxy of a neighbour's vehicle, not stepped here
                o.keep(s, in.dis, in.speed);
            } else {
                in.lastRoadFlags = in.flags;  // (k_scatter / kd_admit / the halo import keep bit 1 up on this path)
                if (in.d < c.n.L && in.nd0 >= c.n.L) in.hop = c.n.laneLL4[in.d];  // (requested with the slot's columns)
                actionOneRounds(c, o, tv, s, in, PushJob{q});
            }
        }
    }
    KSTAMP(6, 4);
}

// The tail records after cfx_load_state / cfx_reset (k_scatter keeps them up afterwards): one thread per drivable
__global__ void kd_init_tails(StepCtx c, TailRec *tailBoth0, TailRec *tailBoth1) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= c.n.L + c.n.K) return;
    TailRec r{};
    r.slot = -1;
    r.tag = -5;
    tailBoth0[d] = r;
    tailBoth1[d] = r;
    const int n = c.cnt[d];
    if (n > 0) {
        const int s = c.segStart[d] + n - 1;
        r.dis = c.s.dis[s];
        r.speed = c.s.speed[s];
        r.slot = s;
        r.templ = c.s.templ[s];
        r.prevDrv = c.s.prevDrv[s];
        r.tag = c.step - 1;
        const_cast<TailRec *>(c.tailR)[d] = r;
    }
}

}  // namespace cfxd
