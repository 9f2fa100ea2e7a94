// Second form of the step on the RING layout (the default; cfx_ring_kernels.h holds the layout itself, the commit kernel,
// the slow paths and the first form, which stays selectable for comparison).
//
// What changed, and why (round 2's profile: three quarters of the action kernel's bytes were not vehicle state):
//   * no per-step GATE records.  What a vehicle near an intersection needs to know about the laneLink ahead — its end lane,
//     type, roadLink, whether it has crosses — is static and comes with the static tables of the vehicle's own LANE, which the
//     block reads once, coalesced.  The only dynamic part, the light, is one bit of a 64-bit word per intersection
//     (`interGreen`, rebuilt by the admission kernel from the current phases: a few KB that live in the caches);
//   * the notify sources of Engine::threadNotifyCross (engine.cpp:317-372) are written per LANE, by the lane's admission
//     thread: a lane's last vehicle is the source "just left laneLink k" of at most one laneLink, its first vehicle the
//     source "approaching laneLink k" of at most one — two 32-byte halves of `LLSrc[k]`, valid for the step in their tag.
//     Round 2 computed both per laneLink (K threads of random reads, K records written, every step);
//   * `tailNow` (the lanes as of this step, after handleWaiting) is written only where a lane admitted a vehicle; a reader
//     takes that record if its tag is this step and the committed tail otherwise;
//   * a vehicle is handed to the cross phase only if a laneLink its own laneLink CROSSES has a source this step or vehicles
//     on it (static peer mask & two dynamic words of the intersection): everybody else's walk over the crosses could only
//     say "pass";
//   * the admission kernel has one thread per lane (it had one per drivable: laneLinks only copied records).
// Same arithmetic, expression for expression, as the first form (each block cites the reference lines it follows).
#pragma once

#include "cfx_ring_kernels.h"

namespace cfxd {

__device__ __forceinline__ void markActive(const RingCtx &c, int k) {
    const int bit = c.n.llLocal[k];
    atomicOr(&c.interMask[c.n.llPack[k].z + (bit >> 6)], 1ULL << (bit & 63));
}

// ---------------------------------------------------------------------------------------------- phase 2 (+ sources of phase 3)
// Engine::handleWaiting engine.cpp:502-516 + Lane::available roadnet.cpp:428-435, one thread per LANE (kr_admit's lane
// branch), followed by what this lane's vehicles are to the crosses of the laneLinks around it (first half of
// Engine::threadNotifyCross engine.cpp:331-342, 362-363).  The first I threads also rebuild the intersections' green words
// (RoadLink::isAvailable roadnet.h:429-431 for every roadLink of the current phase).
__global__ __launch_bounds__(kBlock) void kr2_admit(RingCtx c, int32_t *admitStep, int32_t *waitHead, VidTable vt, DevScalars *sc,
                                                    const SpawnBatch batch) {
    __shared__ int sAdmitted;
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sLane[kAdmitRecs];
    const cfx_vehicle_template *tv = c.t.templ;
    const int nRecs = batch.n, firstNewVid = batch.firstNewVid;
    if (threadIdx.x == 0) sAdmitted = 0;
    if ((int) threadIdx.x < nRecs) sLane[threadIdx.x] = batch.lane[threadIdx.x];
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const bool isLane = d < c.n.L;
    if (d < c.n.I) {
        const int nrl = c.n.interNRL[d], as = c.n.interAvailStart[d], ph = c.curPhase[d];
        unsigned long long green = 0;
        for (int r = 0; r < nrl && r < 64; ++r)
            if (c.n.phaseAvail[as + ph * nrl + r] != 0) green |= 1ULL << r;
        c.interGreen[d] = green;
    }
    // ---- round 1: everything that hangs on the lane alone
    TailRec committed{};
    committed.slot = -1;
    committed.tag = -5;
    int w = -1, n = 0, head = 0, road = 0, laneIdx = 0;
    int2 geo = make_int2(0, 0);
    double laneLen = 0.0;
    if (isLane) {
        committed = c.tailR[d];
        w = waitHead[d];
        n = c.cnt[d];
        geo = c.ringGeo[d];
        head = c.head[d];
        road = c.n.laneRoad[d];
        laneIdx = c.n.laneIndex[d];
        laneLen = c.n.drvLength[d];
    }
    // ---- round 2: the head of the lane's waiting queue (as the last step left it); the lane's first vehicle and the
    //      blocker records of its first and last one (what a cross may want to know about a notify source)
    int wt = 0, route = 0, nextWait = -1;
    uint8_t pending = 0;
    if (w >= 0) {
        wt = vt.templ[w];
        route = vt.route[w];
        nextWait = vt.nextWait[w];
        pending = vt.pendingCustom[w];
    }
    const bool tailLive = isLane && committed.tag == c.step - 1 && committed.slot >= 0;
    int fs = -1;
    double2 fk = make_double2(0.0, 0.0);
    int4 fm = make_int4(0, -1, 0, 0);
    int2 fb = make_int2(-1, -1), ub = make_int2(-1, -1);
    if (isLane && n > 0) {
        fs = ringSlot(geo, head, 0);
        fk = c.kin[fs];
        fm = c.meta[fs];
        fb = c.blkR[fs];
    }
    if (tailLive && committed.prevDrv >= c.n.L) ub = c.blkR[committed.slot];
    __syncthreads();  // (templates and the batch's lanes staged)
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
                pending = 0;
                nextWait = -1;
                waitHead[d] = w;
            }
            if (w >= 0)  // whoever was hung behind the head just now (this thread's own store: taken from the record)
                for (int j = lo; j < nRecs && sLane[j] == d; ++j)
                    if (batch.prevWait[j] == w) nextWait = firstNewVid + batch.vidOff[j];
        }
    }
    if (isLane) {
        const int lane = d;
        // this step's view of the lane's tail: what the last step left, or the vehicle admitted below
        TailRec now = committed;
        if (!tailLive) now.slot = -1;
        bool admit = w >= 0;  // Lane::available roadnet.cpp:428-435
        if (admit && now.slot >= 0 && !(now.dis > tv[now.templ].len + tv[wt].min_gap)) admit = false;
        if (admit && n >= geo.y) {  // the ring is full (cannot happen with capacities from the shortest vehicle): refuse loudly
            sc->overflow = 8;
            admit = false;
        }
        int aSlot = -1, aNext = -1;
        double aSpeed = 0.0;
        if (admit) {
            // Router::getNextDrivable for a vehicle on the first road of its route (router.cpp:49-76): its lane is on
            // route position 0, so the table row is known without walking the route; anything else takes the walk
            const int base = c.t.routeStart[route];
            int next;
            if (c.t.routeRoads[base] == road) {
                const int ll = c.t.nextLL[c.t.nextStart[base] + laneIdx];
                next = ll < 0 ? -1 : c.n.L + ll;
            } else {
                next = nextOf(c.n, c.t, lane, route, 0);
            }
            const int slot = ringSlot(geo, head, n);
            const double v0 = tv[wt].initial_speed;  // VehicleInfo::speed: 0 unless pushed with a speed
            c.s.vid[slot] = w;
            c.s.drv[slot] = lane;
            c.s.prevDrv[slot] = -1;
            c.s.routePos[slot] = 0;
            c.s.route[slot] = route;
            c.meta[slot] = make_int4(wt, next, pending | ((next < 0 && isLastRoad(c, lane, route)) ? 2 : 0), CFX_INT_MAX);  // (ControllerInfo ctor vehicle.cpp:10-13; flags bit 1: finishAction)
            c.kin[slot] = make_double2(0.0, v0);
            c.slotOf[w] = slot;
            c.admitRec[lane] = make_int2(w, nextWait);
            admitStep[lane] = c.step;  // cnt[] and the FIFO pop follow in kr_commit (see cntNow)
            now.dis = 0.0;
            now.speed = v0;
            now.slot = slot;
            now.templ = wt;
            now.prevDrv = -1;
            now.tag = c.step;
            c.tailNow[lane] = now;  // (only lanes that admitted have a record of this step, see tailNowOf)
            atomicAdd(&sAdmitted, 1);
            aSlot = slot;
            aNext = next;
            aSpeed = v0;
        }
        // ---- what this lane's vehicles are to the crosses around it (Engine::threadNotifyCross, first half)
        // its LAST vehicle, if it has just come off a laneLink, is that laneLink's source "on the end lane" (engine.cpp:331-342)
        if (now.slot >= 0 && now.prevDrv >= c.n.L) {
            const int k = now.prevDrv - c.n.L;
            int4 *r = (int4 *) &c.llSrc[k];
            r[0] = make_int4(now.slot, now.templ, c.step, (ub.x >= 0 && ub.y == c.step - 1) ? ub.x : -1);
            ((double2 *) r)[1] = make_double2(now.dis, now.speed);
            markActive(c, k);
        }
        // its FIRST vehicle, if it heads for a laneLink, is that laneLink's source "approaching" (engine.cpp:362-363; whether
        // the light lets it — `laneLink->isAvailable()` there — is tested by the reader, against the same green word)
        int fSlot = -1, fNext = -1, fTempl = 0, fBlk = -1;
        double fDis = 0.0, fSpeed = 0.0;
        if (n > 0) {
            fSlot = fs;
            fNext = fm.y;
            fTempl = fm.x;
            fDis = fk.x;
            fSpeed = fk.y;
            fBlk = (fb.x >= 0 && fb.y == c.step - 1) ? fb.x : -1;
        } else if (aSlot >= 0) {
            fSlot = aSlot;
            fNext = aNext;
            fTempl = wt;
            fSpeed = aSpeed;
        }
        if (fSlot >= 0 && fNext >= c.n.L) {
            const int k = fNext - c.n.L;
            int4 *r = (int4 *) &c.llSrc[k];
            r[2] = make_int4(fSlot, fTempl, c.step, fBlk);
            ((double2 *) r)[3] = make_double2(laneLen - fDis, fSpeed);
            markActive(c, k);
        }
    }
    __syncthreads();
    // Engine::activeVehicleCount: one global atomic per block (phase 4 of this very step counts the admitted vehicles)
    if (threadIdx.x == 0 && sAdmitted) atomicAdd((unsigned long long *) &sc->admitPending[c.step & 1], (unsigned long long) sAdmitted);
}

// Cross::notifyVehicles / notifyDistances of a cross on laneLink k (what the sweep of engine.cpp:327-369 would have written
// there), resolved on demand from the laneLink's source record, its ring and the light: notifiedFrom of cfx_ring_kernels.h
// with the records of this form.  x = the cross's distance on laneLink k, rest = length(k) - x.
__device__ inline Notified notified2(const RingCtx &c, const cfx_vehicle_template *tv, int k, double x, double rest, bool green) {
    Notified nf{-1, 0, 0.0, 0.0, false, 0, make_int2(-1, -1)};
    const int4 *sp = (const int4 *) &c.llSrc[k];
    const int4 uh = sp[0];
    const double2 uk = ((const double2 *) sp)[1];
    const int4 fh = sp[2];
    const double2 fk = ((const double2 *) sp)[3];
    const int dk = c.n.L + k;
    const int2 geo = c.ringGeo[dk];
    const int head = c.head[dk], nOn = c.cnt[dk];
    if (uh.z == c.step) {
        const double vehDistance = uk.x - tv[uh.y].len;
        if (rest + vehDistance < 0.0) {
            nf.slot = uh.x;
            nf.templ = uh.y;
            nf.speed = uk.y;
            nf.dist = -(uk.x + rest);
            nf.pre = true;
            nf.enterLLT = CFX_INT_MAX;  // (a vehicle on a lane)
            nf.blk = make_int2(uh.w, uh.w >= 0 ? c.step - 1 : -1);
            return nf;
        }
    }
    if (nOn > 0) {
        const SegWalk walk{ringSlot(geo, head, 0), geo.x, geo.y};
        for (int i = 0; i < nOn; ++i) {
            const int w = walk.at(i);
            const double2 kw = c.kin[w];
            const double vehDistance = kw.x;
            const int wt = c.meta[w].x;
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
    if (fh.z == c.step && green) {
        nf.slot = fh.x;
        nf.templ = fh.y;
        nf.speed = fk.y;
        nf.dist = fk.x + x;
        nf.pre = true;
        nf.enterLLT = CFX_INT_MAX;
        nf.blk = make_int2(fh.w, fh.w >= 0 ? c.step - 1 : -1);
    }
    return nf;
}

// ---------------------------------------------------------------------------------------------- phase 4
// kw_action of cfx_ring_kernels.h (wave-granular chunks of the block's vehicle list, leaders by __shfl_up) with the
// per-vehicle phase of this form: everything a vehicle may need is requested in ONE round after its slot's two records —
// tails of the drivables ahead (heads), both views of the lane behind the next laneLink (vehicles near an intersection),
// the intersection's words, the peer mask of the laneLink (vehicles that may have to walk its crosses), the identity columns
// (vehicles that may leave) — and nothing hangs on a per-step gate record.
struct RingPush2 {
    JobQueue q;
    RingJob *recs;
    int L;
    __device__ __forceinline__ void operator()(int s, const JobInfo &j) const {
        const int shard = blockIdx.x & (kJobShards - 1);
        const int idx = atomicAdd(&q.count[shard * kJobShardStride], 1);
        if (idx >= q.capacity) {
            *q.overflow = 9;
            return;
        }
        const size_t at = (size_t) shard * q.capacity + idx;
        q.jobs[at] = s;
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
        r.pad0 = j.laneLink;
        r.pad1 = j.maskBase;
        r.d0 = j.d < L ? -(j.dlen - j.dis) : j.dis;
        r.speed = j.speed;
        r.v = j.v;
        r.iv = j.iv;
        r.dis = j.dis;
        r.dlen = j.dlen;
        recs[at] = r;
    }
};

__device__ __forceinline__ int pick4(const int4 &v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }

template <int B>
__global__ __launch_bounds__(B, 4) void kw2_action(RingCtx c, RingOut o, JobQueue q, RingJob *jobRecs, int G, int nLaneBlocks) {
    const int w = (int) blockIdx.x, t = (int) threadIdx.x;
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    __shared__ int sPre[B + 1];
    __shared__ int2 sGeo[B];
    __shared__ int sHead[B];
    __shared__ double2 sLM[B];
    __shared__ int4 sA[B];  // lanes: the laneLinks leaving the lane (laneLL4)      laneLinks: {first, end cross entry, mask word, type}
    __shared__ int4 sB[B];  // lanes: their end lanes (laneEnd4)                    laneLinks: {peer mask low, high, -, -}
    __shared__ int4 sC[B];  // lanes: their roadLink | type | has crosses (laneInfo4)
    __shared__ int4 sD[B];  // lanes: {intersection, its mask word, mask words, roadLinks}
    __shared__ unsigned char sAdm[B];
    __shared__ int sWave[B / 64];
    const cfx_vehicle_template *tv = c.t.templ;
    const int L = c.n.L;
    const bool laneBlock = w < nLaneBlocks;
    const int g = laneBlock ? G : B;
    const int d0 = laneBlock ? w * G : L + (w - nLaneBlocks) * B;
    const int dEnd = laneBlock ? L : L + c.n.K;
    const int dMine = d0 + t;
    int n = 0;
    if (t < g && dMine < dEnd) {
        const int2 geo = c.ringGeo[dMine];
        const int head = c.head[dMine];
        n = c.cnt[dMine];
        bool admitted = false;
        sLM[t] = c.n.drvLM[dMine];
        if (laneBlock) {
            admitted = c.admitStep[dMine] == c.step;
            sA[t] = c.n.laneLL4[dMine];
            sB[t] = c.n.laneEnd4[dMine];
            sC[t] = c.n.laneInfo4[dMine];
            sD[t] = c.n.laneInter4[dMine];
        } else {
            sA[t] = c.n.llPack[dMine - L];
            sB[t] = c.n.llPeer[dMine - L];
        }
        if (admitted) n += 1;
        sAdm[t] = admitted ? 1 : 0;
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
    const RingPush2 push{q, jobRecs, L};
    const int lane = t & 63;
    const double interval = c.interval;
    for (int q0 = (t >> 6) * 64; q0 < T; q0 += B) {  // chunk q0 / 64 belongs to wavefront (q0 / 64) mod (B / 64)
        const int qv = q0 + lane;
        const bool valid = qv < T;
        int i = 0, idx = 0, s = 0;
        int2 geo = make_int2(0, 0);
        int head = 0;
        double dis = 0.0, speed = 0.0;
        int templIdx = 0, nd0 = -1, flags = 0;
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
            s = ringSlot(geo, head, idx);
            const double2 kv = c.kin[s];  // the two records of the slot: 2 x 16 B, adjacent lanes adjacent in memory
            const int4 mv = c.meta[s];
            if (lane == 0 && idx > 0) {  // the vehicle ahead of the chunk: this lane's leader
                const int ls = ringSlot(geo, head, idx - 1);
                kp = c.kin[ls];
                tp = c.meta[ls].x;
            }
            dis = kv.x;
            speed = kv.y;
            templIdx = mv.x;
            nd0 = mv.y;
            flags = mv.z;
        }
        // the leader inside the drivable is the lane below (all lanes take part in the exchange)
        const double disUp = __shfl_up(dis, 1, 64), speedUp = __shfl_up(speed, 1, 64);
        const int templUp = __shfl_up(templIdx, 1, 64);
        if (!valid) continue;  // (a chunk's invalid lanes are its last ones: nobody reads them)
        const int d = d0 + i;
        const bool isHead = idx == 0, onLane = laneBlock, nextIsLink = nd0 >= L;
        const int nNow = sPre[i + 1] - sPre[i];
        const double disPrev = lane == 0 ? kp.x : disUp, speedPrev = lane == 0 ? kp.y : speedUp;
        const int templPrev = lane == 0 ? tp : templUp;
        const cfx_vehicle_template &tt = tv[templIdx];
        const double2 lm = sLM[i];
        const double dlen = lm.x;
        const bool related = !onLane || (nextIsLink && dlen - dis <= tt.approach_dist);  // Vehicle::isIntersectionRelated
        const bool custom = (flags & 1) != 0;
        int vid = 0;
        if (custom) {
            vid = c.s.vid[s];
            c.meta[s].z = flags & ~1;  // Vehicle::update clears isCustomSpeedSet (vehicle.cpp:120-122)
        }
        // which of its lane's laneLinks the vehicle takes next
        int4 hop = make_int4(-2, -2, -2, -2);
        int hopq = -1;
        if (onLane && nextIsLink) {
            hop = sA[i];
            const int ll = nd0 - L;
            hopq = hop.x == ll ? 0 : (hop.y == ll ? 1 : (hop.z == ll ? 2 : (hop.w == ll ? 3 : -1)));
        }

        // ================= the one round of requests
        const bool hopHead = isHead && onLane && nextIsLink && hop.x != -2 && hop.w < 0;  // head of a lane: tails of the (up to three) laneLinks leaving it
        const bool linkHead = isHead && !onLane;                                          // head of a laneLink: tail of its end lane
        TailRec hopRec[3];
        double linkLen = 0.0;
        if ((hopHead && hop.x >= 0) || linkHead) hopRec[0] = c.tailR[linkHead ? nd0 : L + hop.x];
        if (hopHead) {
            if (hop.y >= 0) hopRec[1] = c.tailR[L + hop.y];
            if (hop.z >= 0) hopRec[2] = c.tailR[L + hop.z];
            linkLen = c.n.drvLength[nd0];
        }
        // the laneLink ahead (a lane's vehicle near the intersection) or underneath (a vehicle on a laneLink)
        const bool approaching = related && nextIsLink;
        const int kx = onLane ? nd0 - L : d - L;
        int endLane = -1, info = 0;
        if (approaching) {
            if (hopq >= 0) {
                endLane = pick4(sB[i], hopq);
                info = pick4(sC[i], hopq);
            } else {  // a lane with more than four laneLinks: the per-laneLink tables
                endLane = c.n.llEndLane[kx];
                info = c.n.llRoadLink[kx] | (c.n.llType[kx] << 16) | ((c.n.llXStart[kx + 1] > c.n.llXStart[kx] ? 1 : 0) << 18);
            }
        }
        TailRec laneR{}, laneN{};
        if (approaching) {
            laneR = c.tailR[endLane];    // the lane behind the laneLink as of the last commit ...
            laneN = c.tailNow[endLane];  // ... and the vehicle it admitted this step, if it did (tag)
        }
        const bool hasX = related && (onLane ? ((info >> 18) & 1) != 0 : sA[i].y > sA[i].x);
        unsigned long long green = 0, active = 0, peerMask = ~0ULL;
        int mb = 0, xs = 0, xe = 0, nRL = 0;
        if (related) {
            if (onLane) {
                const int4 i4 = sD[i];
                mb = i4.y;
                nRL = i4.w;
                green = c.interGreen[i4.x];
            } else {
                mb = sA[i].z;
            }
            if (hasX) {
                active = c.interMask[mb] | c.llOcc[mb];
                const int4 pm = onLane ? c.n.llPeer[kx] : sB[i];
                if (!onLane) {
                    xs = sA[i].x;
                    xe = sA[i].y;
                } else {
                    xs = pm.z;
                    xe = pm.w;
                }
                peerMask = ((unsigned long long) (unsigned) pm.x) | ((unsigned long long) (unsigned) pm.y << 32);
            }
        }
        // may it run past the end of its drivable this step?  (only a hint: decides what is requested early)
        LeaverPrefetch lp{false, 0.0, 0, 0, 0};
        if (dlen - dis <= (speed + tt.max_pos_acc * interval) * interval + 1.0) {
            lp.valid = true;
            lp.nextLen = nd0 >= 0 ? c.n.drvLength[nd0] : 0.0;
            lp.vid = c.s.vid[s];
            lp.route = c.s.route[s];
            lp.routePos = c.s.routePos[s];
        }

        // ================= leader / gap (Vehicle::updateLeaderAndGap vehicle.cpp:157-196)
        const bool laneAdmitted = sAdm[i] != 0;
        const Tail laneCommittedTail = tailIfCurrent(laneR, c.step - 1);
        const Tail laneNowTail = laneN.tag == c.step ? tailOfRec(laneN) : laneCommittedTail;
        double gap = 0.0;
        int ls, leaderTempl = templPrev;
        double leaderSpeed = speedPrev;
        if (!isHead) {
            ls = ringSlot(geo, head, idx - 1);
            gap = disPrev - tv[templPrev].len - dis;
        } else {
            Tail best{-1, 0, -1, 0.0, 0.0};
            bool resolved = false;
            double dist = dlen - dis;
            if (hopHead) {
                // first hop: the last vehicles of all laneLinks leaving this lane, closest first (findHeadLeader, `consider`)
                for (int qq = 0; qq < 3; ++qq) {
                    const int ll = qq == 0 ? hop.x : (qq == 1 ? hop.y : hop.z);
                    if (ll < 0) continue;
                    const Tail cand = tailIfCurrent(hopRec[qq], c.step - 1);
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
                    if (dist > tt.approach_dist) {
                        resolved = true;  // nothing within the look-ahead bound (vehicle.cpp:190-191)
                    } else {
                        // second hop: the lane behind the laneLink (dist <= bound implies `approaching`: its records are here).
                        // A vehicle admitted this step sees this step's admissions on lanes before its own (lastSlotForLeader).
                        const bool viewerNew = laneAdmitted && nNow == 1;
                        best = (viewerNew && endLane < d) ? laneNowTail : laneCommittedTail;
                        if (best.slot >= 0) {
                            gap = dist + best.dis - tv[best.templ].len;
                            resolved = true;
                        } else {
                            dist += c.n.drvLength[endLane];
                            resolved = dist > tt.approach_dist;
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
                    resolved = dist > tt.approach_dist;
                }
            } else if (nd0 < 0) {
                resolved = true;  // end of the route: nothing ahead
            }
            if (!resolved) {  // anything else (more than four laneLinks, a search that goes on): the general walk from the start
                best = findHeadLeader(c, tv, s, d, dis, tt.approach_dist, nd0, dlen, &gap, hop);
            }
            ls = best.slot;
            leaderTempl = best.templ;
            leaderSpeed = best.speed;
        }

        // ================= Vehicle::getNextSpeed vehicle.cpp:308-335
        double v = tt.max_speed;
        v = min2(v, speed + tt.max_pos_acc * interval);
        v = min2(v, lm.y);
        double cf;  // Vehicle::getCarFollowSpeed vehicle.cpp:212-238
        const double customSpeed = custom ? c.vCustomSpeed[vid] : 0.0;  // (rare: loaded where it is used)
        if (ls < 0) {
            cf = custom ? customSpeed : tt.max_speed;
        } else if (custom) {
            const cfx_vehicle_template &tl = tv[leaderTempl];
            cf = min2(customSpeed, noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, tt.max_neg_acc, gap, interval, 0));
        } else {
            const cfx_vehicle_template &tl = tv[leaderTempl];
            cf = noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, tt.max_neg_acc, gap, interval, 0);
            double assumeDecel = 0;
            if (speed > leaderSpeed) assumeDecel = speed - leaderSpeed;
            cf = min2(cf, noCollisionSpeed(leaderSpeed, tl.usual_neg_acc, speed, tt.usual_neg_acc, gap, interval, tt.min_gap));
            cf = min2(cf, (gap + (leaderSpeed + assumeDecel / 2) * interval - speed * interval / 2) /
                              (tt.headway_time + interval / 2));
        }
        v = min2(v, cf);

        // ================= Vehicle::getIntersectionRelatedSpeed vehicle.cpp:337-362
        if (related) {
            VehRef self{speed, &tt};
            double iv = tt.max_speed;
            bool done = false;
            const int type = (info >> 16) & 3;
            if (nextIsLink) {
                // RoadLink::isAvailable roadnet.h:429-431: a bit of the intersection's green word
                const bool avail = nRL <= 64 ? ((green >> (info & 0xFFFF)) & 1ULL) != 0 : llAvailable(c, kx);
                bool blocked = !avail;
                if (!blocked) {  // Lane::canEnter roadnet.cpp:437-445
                    const Tail tail = laneNowTail;
                    if (tail.slot >= 0) blocked = !(tail.dis > tv[tail.templ].len + tt.len || tail.speed >= 2);
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
                if (nextIsLink && typeIsTurn(type)) iv = min2(iv, tt.turn_speed);
                // the crosses of the laneLink: the cross phase takes over — if any laneLink this one crosses has a notify
                // source this step or vehicles on it; otherwise every Cross::canPass on the way would say "pass"
                if (hasX && (peerMask == ~0ULL || (peerMask & active) != 0)) {
                    o.park(s, v, iv);
                    const int t1 = onLane ? type : sA[i].w;
                    JobInfo job{d, idx, nNow, templIdx, nd0, kx, t1 << 1, xs, xe, speed, dis, dlen, v, iv};
                    job.maskBase = mb;
                    push(s, job);
                    continue;
                }
            }
            v = min2(v, iv);
        }
        finishAction<false>(c, o, tt, s, d, vid, speed, dis, dlen, nd0, v, -1, idx, nNow, lp, flags);
    }
}

// ---------------------------------------------------------------------------------------------- phase 4, the crosses
// kr_cross of cfx_ring_kernels.h on this form's records: a lane tests the peer laneLink's bit in the intersection's two words
// and reads its source record and ring directly; a notify source that sits on a lane brings what Cross::canPass may ask of it.
__global__ __launch_bounds__(kCrossBlock) void kr2_cross(RingCtx c, RingOut o, JobQueue q, const RingJob *recs) {
    __shared__ cfx_vehicle_template sT[kLdsTempl];
    const cfx_vehicle_template *tv = c.t.templ;
    __shared__ int shardEnd[kJobShards];
    if (threadIdx.x < kJobShards) shardEnd[threadIdx.x] = min(q.count[threadIdx.x * kJobShardStride], q.capacity);
    if (c.t.nTempl <= kLdsTempl) {
        const int nd = c.t.nTempl * (int) (sizeof(cfx_vehicle_template) / sizeof(double));
        const double *src = (const double *) c.t.templ;
        double *dst = (double *) sT;
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        tv = sT;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < kJobShards; ++i) {
            run += shardEnd[i];
            shardEnd[i] = run;
        }
    }
    __syncthreads();
    const int nJ = shardEnd[kJobShards - 1];
    const int g = threadIdx.x % kCrossGroup;
    const int groupsPerBlock = blockDim.x / kCrossGroup;
    const int groupShift = (threadIdx.x & 63) & ~(kCrossGroup - 1);
    for (int j = blockIdx.x * groupsPerBlock + threadIdx.x / kCrossGroup; j < nJ; j += gridDim.x * groupsPerBlock) {
        int shard = 0;
        while (j >= shardEnd[shard]) ++shard;
        const RingJob jr = recs[(size_t) shard * q.capacity + (j - (shard ? shardEnd[shard - 1] : 0))];
        const int s = jr.slot;
        const cfx_vehicle_template &t = tv[jr.templ];
        const double d0 = jr.d0;
        VehRef self{jr.speed, &t};
        double iv = jr.iv;
        int blockerSlot = -1;
        // this intersection's words (requested with the crosses' static records)
        const int mb = jr.pad1;
        const int inter = c.n.llInter[jr.pad0];
        const unsigned long long act0 = c.interMask[mb] | c.llOcc[mb];
        const unsigned long long green = c.interGreen[inter];
        const int nRL = c.n.interNRL[inter];
        for (int e0 = jr.xs; e0 < jr.xe; e0 += kCrossGroup) {
            const int e = e0 + g;
            bool fail = false;
            int foe = -1;
            double dOn = 0.0;
            if (e < jr.xe) {
                const double2 dd = c.n.xDD[e];  // {distance on this laneLink, distance on the peer laneLink}
                const int4 xp = c.n.xPack[e];   // {peer laneLink, peer bit, peer roadLink type, peer roadLink}
                const double rest = c.n.xPeerRest[e];
                dOn = dd.x;
                if (!(dOn < d0)) {
                    unsigned long long word = act0;
                    if (xp.y >= 64) word = c.interMask[mb + (xp.y >> 6)] | c.llOcc[mb + (xp.y >> 6)];
                    if ((word >> (xp.y & 63)) & 1ULL) {  // otherwise nobody to yield to on that laneLink
                        const bool peerGreen = nRL <= 64 ? ((green >> xp.w) & 1ULL) != 0 : llAvailable(c, xp.x);
                        const Notified nf = notified2(c, tv, xp.x, dd.y, rest, peerGreen);
                        fail = !canPassDecide(c, tv, s, self, dOn, jr.t1, d0, nf, xp.z, &foe);
                    }
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
        if (g == 0)
            finishAction<false>(c, o, t, s, jr.d, 0, jr.speed, jr.dis, jr.dlen, jr.nd0, min2(jr.v, iv), blockerSlot, jr.idx, jr.nNow);
    }
}

}  // namespace cfxd
