// Per-step kernels of the cfx HIP engine.  One reference step (Engine::nextStep engine.cpp:566-594) is
//   k_spawn_link   phase 0/1 tail : append host-produced spawn records to lanes' waiting queues
//   k_admit        phase 2        : Engine::handleWaiting, one thread per lane
//   k_notify       phase 3        : Engine::threadNotifyCross, one thread per laneLink
//   k_action       phase 4        : leader/gap + Engine::vehicleControl, one thread per slot
//   k_count        phase 5a (+8)  : classify stay / move / finish, per-drivable counts; traffic lights
//   k_scan_*       phase 5b       : new segment offsets (exclusive scan over drivables); finish stats
//   k_scatter      phase 5c/6     : stable compaction into the next generation = commit (Vehicle::update)
// Leader/gap (phase 7, engine.cpp:429-442) needs no kernel of its own: it is a pure function of the
// post-compaction order and is evaluated at the top of the next step's k_action (see lastSlotForLeader).
#pragma once

#include "cfx_device.h"

namespace cfxd {

constexpr int kBlock = 256;

// ----------------------------------------------------------------------------------------------
// Vehicle table (indexed by vid, never permuted)
struct VidTable {
    int32_t *priority, *templ, *route, *nextWait;
    double *enterTime;
    uint8_t *state;  // 0 waiting, 1 running, 2 finished
};

struct DevScalars {
    long long active;          // Engine::activeVehicleCount
    long long finishedCnt;     // Engine::finishedVehicleCnt
    double cumulativeTravelTime;
    long long vehicleSteps;    // sum over steps of vehicles that ran phase 4
    int nFinishedStep;         // finished vehicles of the step in flight
    int overflow;              // set when an internal capacity was exceeded
};

// Per-drivable scratch of the compaction.
struct CompactScratch {
    int32_t *leaveCnt;     // [D] vehicles leaving (moved or finished)
    int32_t *maxLeaveIdx;  // [D] largest in-segment index among leavers (-1 none)
    int32_t *inCnt;        // [D] vehicles entering
    int32_t *inHead;       // [D] head of the linked list of entering slots (-1 none)
    int32_t *inNext;       // [slot] next entering slot of the same target
};

// Buffered (not yet committed) results of k_action: Vehicle::Buffer vehicle.h:54-72
struct ActionBuf {
    double *dis, *speed;
    int32_t *drv;      // -1 unchanged, -2 end of route, >= 0 new drivable
    int32_t *blocker;  // slot (current generation) or -1
};

// ----------------------------------------------------------------------------------------------
__global__ void k_spawn_link(const cfx_spawn *recs, int n, int firstNewVid, VidTable vt, int32_t *waitHead) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cfx_spawn r = recs[i];
    vt.priority[r.vid] = r.priority;
    vt.templ[r.vid] = r.templ;
    vt.route[r.vid] = r.route;
    vt.enterTime[r.vid] = r.enter_time;
    vt.state[r.vid] = 0;
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

// Engine::handleWaiting engine.cpp:502-516 + Lane::available roadnet.cpp:428-435
__global__ void k_admit(StepCtx c, int32_t *cnt, int32_t *admitStep, int32_t *waitHead, VidTable vt, CompactScratch cs,
                        DevScalars *sc) {
    int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n.L) return;
    cs.leaveCnt[lane] = 0;
    cs.maxLeaveIdx[lane] = -1;
    cs.inCnt[lane] = 0;
    cs.inHead[lane] = -1;
    int w = waitHead[lane];
    if (w < 0) return;
    int n = cnt[lane];
    int base = c.segStart[lane];
    int wt = vt.templ[w];
    if (n > 0) {
        int tail = base + n - 1;
        if (!(c.s.dis[tail] > T(c, tail).len + c.t.templ[wt].min_gap)) return;
    }
    int slot = base + n;  // the lane's spare slot
    c.s.vid[slot] = w;
    c.s.drv[slot] = lane;
    c.s.prevDrv[slot] = -1;
    c.s.blocker[slot] = -1;
    c.s.enterLLT[slot] = CFX_INT_MAX;  // ControllerInfo ctor vehicle.cpp:10-13
    c.s.routePos[slot] = 0;
    c.s.templ[slot] = wt;
    c.s.route[slot] = vt.route[w];
    c.s.dis[slot] = 0.0;
    c.s.speed[slot] = 0.0;
    cnt[lane] = n + 1;
    admitStep[lane] = c.step;
    waitHead[lane] = vt.nextWait[w];
    vt.state[w] = 1;
    atomicAdd((unsigned long long *) &sc->active, 1ULL);
}

// Engine::threadNotifyCross engine.cpp:317-372 + Cross::notify roadnet.cpp:595-601
__global__ void k_notify(StepCtx c, CompactScratch cs) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.n.K) return;
    const int d = c.n.L + k;
    cs.leaveCnt[d] = 0;
    cs.maxLeaveIdx[d] = -1;
    cs.inCnt[d] = 0;
    cs.inHead[d] = -1;

    const int xb = c.n.llXStart[k], xe = c.n.llXStart[k + 1];
    if (xb == xe) return;
    const int endLane = c.n.llEndLane[k], startLane = c.n.llStartLane[k];
    // the three vehicle sources
    int u = lastSlot(c, endLane);
    if (u >= 0 && c.s.prevDrv[u] != d) u = -1;
    const int nOn = c.cnt[d];
    int f = c.cnt[startLane] > 0 ? c.segStart[startLane] : -1;
    if (f >= 0 && !(nextOf(c, startLane, c.s.route[f], c.s.routePos[f]) == d && llAvailable(c, k))) f = -1;
    if (u < 0 && nOn == 0 && f < 0) return;  // nothing to notify: entries stay stale (llStamp != step+1)

    int r = xe - 1;
    const double llLen = c.n.drvLength[d];
    if (u >= 0) {
        double udis = c.s.dis[u];
        double vehDistance = udis - T(c, u).len;
        while (r >= xb) {
            double crossDistance = llLen - c.n.xDist[r];
            if (crossDistance + vehDistance < 0.0) {
                c.nSlot[r] = u;
                c.nDist[r] = -(udis + crossDistance);
                --r;
            } else
                break;
        }
    }
    const int base = c.segStart[d];
    for (int i = 0; i < nOn && r >= xb; ++i) {
        int w = base + i;
        double vehDistance = c.s.dis[w];
        double wlen = T(c, w).len;
        while (r >= xb) {
            double crossDistance = c.n.xDist[r];
            if (vehDistance > crossDistance) {
                if (vehDistance - crossDistance - wlen <= 0.0) {
                    c.nSlot[r] = w;
                    c.nDist[r] = crossDistance - vehDistance;
                } else
                    break;
            } else {
                c.nSlot[r] = w;
                c.nDist[r] = crossDistance - vehDistance;
            }
            --r;
        }
    }
    if (f >= 0) {
        double vehDistance = c.n.drvLength[startLane] - c.s.dis[f];
        while (r >= xb) {
            c.nSlot[r] = f;
            c.nDist[r] = vehDistance + c.n.xDist[r];
            --r;
        }
    }
    while (r >= xb) {  // Cross::clearNotify for the entries nobody claimed
        c.nSlot[r] = -1;
        --r;
    }
    c.llStamp[k] = c.step + 1;
}

// leader/gap (Vehicle::updateLeaderAndGap vehicle.cpp:157-196) for the vehicle in slot s
__device__ inline int findLeader(const StepCtx &c, int s, int d, int k, double *gapOut) {
    if (k > 0) {
        int ls = s - 1;
        *gapOut = c.s.dis[ls] - T(c, ls).len - c.s.dis[s];
        return ls;
    }
    // k == 0 and the lane's only vehicle was admitted this step => it IS the admitted vehicle
    const bool viewerNew = d < c.n.L && c.admitStep[d] == c.step && c.cnt[d] == 1;
    const int route = c.s.route[s], routePos = c.s.routePos[s];
    const double bound = T(c, s).approach_dist;  // same expression as vehicle.cpp:190-191
    int ls = -1;
    double gap = 0.0;
    double dist = c.n.drvLength[d] - c.s.dis[s];
    int cur = d;
    for (;;) {
        int nd = nextOf(c, cur, route, routePos);
        if (nd < 0) break;
        if (nd >= c.n.L) {
            int sl = c.n.llStartLane[nd - c.n.L];
            for (int q = c.n.laneLLStart[sl]; q < c.n.laneLLStart[sl + 1]; ++q) {
                int cand = lastSlot(c, c.n.L + c.n.laneLL[q]);
                if (cand >= 0) {
                    double cg = dist + c.s.dis[cand] - T(c, cand).len;
                    if (ls < 0 || cg < gap) {
                        ls = cand;
                        gap = cg;
                    }
                }
            }
            if (ls >= 0) break;
        } else {
            ls = lastSlotForLeader(c, nd, viewerNew, d);
            if (ls >= 0) {
                gap = dist + c.s.dis[ls] - T(c, ls).len;
                break;
            }
        }
        dist += c.n.drvLength[nd];
        if (dist > bound) break;
        cur = nd;
    }
    *gapOut = gap;
    return ls;
}

// Engine::threadGetAction / vehicleControl engine.cpp:188-251,402-413 with Vehicle::getNextSpeed
// vehicle.cpp:308-335 and everything below it.
__global__ void k_action(StepCtx c, ActionBuf b) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride) {
        if (c.s.vid[s] < 0) continue;
        const int d = c.s.drv[s];
        const int k = s - c.segStart[d];
        const cfx_vehicle_template &t = T(c, s);
        const double interval = c.interval;
        const double speed = c.s.speed[s];
        const double dis = c.s.dis[s];
        const double dlen = c.n.drvLength[d];
        const int route = c.s.route[s], routePos = c.s.routePos[s];

        // --- leader / gap
        double gap;
        const int ls = findLeader(c, s, d, k, &gap);

        // --- Vehicle::getNextSpeed vehicle.cpp:308-335
        double v = t.max_speed;
        v = min2(v, speed + t.max_pos_acc * interval);
        v = min2(v, c.n.drvMaxSpeed[d]);

        // car following, Vehicle::getCarFollowSpeed vehicle.cpp:212-238
        double cf;
        if (ls < 0) {
            cf = t.max_speed;
        } else {
            const cfx_vehicle_template &tl = T(c, ls);
            const double leaderSpeed = c.s.speed[ls];
            cf = noCollisionSpeed(leaderSpeed, tl.max_neg_acc, speed, t.max_neg_acc, gap, interval, 0);
            double assumeDecel = 0;
            if (speed > leaderSpeed) assumeDecel = speed - leaderSpeed;
            cf = min2(cf, noCollisionSpeed(leaderSpeed, tl.usual_neg_acc, speed, t.usual_neg_acc, gap, interval, t.min_gap));
            cf = min2(cf, (gap + (leaderSpeed + assumeDecel / 2) * interval - speed * interval / 2) /
                              (t.headway_time + interval / 2));
        }
        v = min2(v, cf);

        // intersection logic, Vehicle::isIntersectionRelated vehicle.cpp:289-300
        const int nd0 = nextOf(c, d, route, routePos);
        int blockerSlot = -1;
        const bool onLane = d < c.n.L;
        bool related = !onLane || (nd0 >= c.n.L && dlen - dis <= t.approach_dist);
        if (related) {
            // Vehicle::getIntersectionRelatedSpeed vehicle.cpp:337-376
            VehRef self{speed, &t};
            double iv = t.max_speed;
            int laneLink = -1;
            bool done = false;
            if (nd0 >= c.n.L) {
                laneLink = nd0 - c.n.L;
                bool blocked = !llAvailable(c, laneLink);
                if (!blocked) {  // Lane::canEnter roadnet.cpp:437-445
                    int tail = lastSlot(c, c.n.llEndLane[laneLink]);
                    if (tail >= 0) blocked = !(c.s.dis[tail] > T(c, tail).len + t.len || c.s.speed[tail] >= 2);
                }
                if (blocked) {
                    if (minBrakeDistance(self) > dlen - dis) {
                        // cannot stop before the line: run it
                    } else {
                        iv = min2(iv, stopBeforeSpeed(self, dlen - dis, interval));
                        done = true;
                    }
                }
                if (!done && llIsTurn(c, laneLink)) iv = min2(iv, t.turn_speed);
            }
            if (!done) {
                if (laneLink < 0 && !onLane) laneLink = d - c.n.L;
                double d0 = onLane ? -(dlen - dis) : dis;
                for (int e = c.n.llXStart[laneLink]; e < c.n.llXStart[laneLink + 1]; ++e) {
                    double dOn = c.n.xDist[e];
                    if (dOn < d0) continue;
                    int foe;
                    if (!canPass(c, s, self, e, d0, &foe)) {
                        iv = min2(iv, stopBeforeSpeed(self, dOn - d0 - t.yield_distance, interval));
                        blockerSlot = foe;
                        break;
                    }
                }
            }
            v = min2(v, iv);
        }
        v = min2(v, 100);  // SimpleLaneChange::yieldSpeed without signals
        if (nd0 < 0 && !isLastRoad(c, d, route)) {  // !Router::onValidLane router.h:66-68
            double vn = noCollisionSpeed(0, 1, speed, t.max_neg_acc, dlen - dis, interval, t.min_gap);
            v = min2(v, vn);
        }
        v = max2(v, speed - t.max_neg_acc * interval);

        // --- Engine::vehicleControl engine.cpp:212-221
        double deltaDis;
        if (v < 0) {
            deltaDis = 0.5 * speed * speed / t.max_neg_acc;
            v = 0;
        } else {
            deltaDis = (speed + v) * interval / 2;
        }
        // --- Vehicle::setDeltaDistance vehicle.cpp:49-68
        double nd = deltaDis + dis;
        int drivable = d;
        int newDrv = -1;
        while (drivable >= 0 && nd > c.n.drvLength[drivable]) {
            nd -= c.n.drvLength[drivable];
            drivable = nextOf(c, drivable, route, routePos);
            newDrv = drivable >= 0 ? drivable : -2;
        }
        b.dis[s] = nd;
        b.speed[s] = v;
        b.drv[s] = newDrv;
        b.blocker[s] = blockerSlot;
    }
}

// Phase 5a: classification + per-drivable counts (Engine::threadUpdateLocation engine.cpp:282-315, first
// half) and, on the low thread ids, TrafficLight::passTime trafficlight.cpp:29-37.
__global__ void k_count(StepCtx c, ActionBuf b, CompactScratch cs, VidTable vt, DevScalars *sc, int32_t *finList,
                        int finCap, int32_t *curPhase, double *remain, int rlTrafficLight) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
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
        int vid = c.s.vid[s];
        if (vid < 0) continue;
        int nd = b.drv[s];
        if (nd == -1) continue;  // stays
        int d = c.s.drv[s];
        int k = s - c.segStart[d];
        atomicAdd(&cs.leaveCnt[d], 1);
        atomicMax(&cs.maxLeaveIdx[d], k);
        if (nd >= 0) {
            atomicAdd(&cs.inCnt[nd], 1);
            cs.inNext[s] = atomicExch(&cs.inHead[nd], s);
        } else {
            vt.state[vid] = 2;
            int idx = atomicAdd(&sc->nFinishedStep, 1);
            if (idx < finCap) finList[idx] = s;
            else sc->overflow = 1;
        }
    }
}

// Phase 5b: exclusive scan of the new segment sizes over drivables, 3 launches.
constexpr int kScanItems = 8;                       // drivables per thread
constexpr int kScanTile = kBlock * kScanItems;      // drivables per block
constexpr int kFinLds = 2048;                       // finished vehicles per step staged in LDS

__device__ __forceinline__ int newLiveCount(const int32_t *cnt, const CompactScratch &cs, int d) {
    return cnt[d] - cs.leaveCnt[d] + cs.inCnt[d];
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

__global__ void k_scan_reduce(int D, int L, const int32_t *cnt, CompactScratch cs, int32_t *blockSums) {
    __shared__ int smem[kBlock / 64];
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int sum = 0;
    for (int i = 0; i < kScanItems; ++i) {
        int d = base + i;
        if (d < D) sum += newLiveCount(cnt, cs, d) + (d < L ? 1 : 0);
    }
    int tot = blockReduceSum(sum, smem);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}

// Single block: scan of the block sums + the step's finish statistics in the reference's order
// (threadUpdateLocation with one thread walks drivables in RoadNet order, lists front to back, i.e.
// ascending slot; engine.cpp:296-310).
__global__ void k_scan_top(int nBlocks, int32_t *blockSums, StepCtx c, VidTable vt, DevScalars *sc, int32_t *finList,
                           int32_t *finSorted, int finCap) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    // sequential-by-chunks exclusive scan (nBlocks is small: D / 2048)
    for (int b0 = 0; b0 < nBlocks; b0 += blockDim.x) {
        int i = b0 + threadIdx.x;
        int v = i < nBlocks ? blockSums[i] : 0;
        // inclusive scan inside the chunk via shared memory (Hillis-Steele)
        __shared__ int buf[kBlock];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < (int) blockDim.x; off <<= 1) {
            int add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        int incl = buf[threadIdx.x];
        int total = buf[blockDim.x - 1];
        if (i < nBlocks) blockSums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    // finish statistics: order the step's finished slots (rank sort in LDS), then one thread adds the
    // travel times in that order (FP64 addition is not associative; the reference adds sequentially)
    __shared__ int fin[kFinLds];
    __shared__ double term[kFinLds];
    int F = sc->nFinishedStep;
    if (F > finCap) F = finCap;
    const double now = c.step * c.interval;  // Engine::getCurrentTime engine.cpp:678-680
    if (F <= kFinLds) {
        for (int i = threadIdx.x; i < F; i += blockDim.x) fin[i] = finList[i];
        __syncthreads();
        for (int i = threadIdx.x; i < F; i += blockDim.x) {
            int me = fin[i];
            int rank = 0;
            for (int j = 0; j < F; ++j) rank += fin[j] < me;
            term[rank] = now - vt.enterTime[c.s.vid[me]];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double cum = sc->cumulativeTravelTime;
            for (int i = 0; i < F; ++i) cum += term[i];
            sc->cumulativeTravelTime = cum;
        }
    } else {  // more finishers in one step than the LDS staging holds: same algorithm through global memory
        for (int i = threadIdx.x; i < F; i += blockDim.x) {
            int me = finList[i];
            int rank = 0;
            for (int j = 0; j < F; ++j) rank += finList[j] < me;
            finSorted[rank] = me;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double cum = sc->cumulativeTravelTime;
            for (int i = 0; i < F; ++i) cum += now - vt.enterTime[c.s.vid[finSorted[i]]];
            sc->cumulativeTravelTime = cum;
        }
    }
    if (threadIdx.x == 0) {
        sc->vehicleSteps += sc->active;  // everybody counted as active took this step's phase 4
        sc->finishedCnt += F;
        sc->active -= F;
        sc->nFinishedStep = 0;
    }
}

__global__ void k_scan_apply(int D, int L, const int32_t *cnt, CompactScratch cs, const int32_t *blockSums,
                             int32_t *segStartNext, int32_t *cntNext, int32_t *vidNext) {
    __shared__ int smem[kBlock / 64];
    __shared__ int wsum[kBlock / 64];
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int vals[kScanItems];
    int live[kScanItems];
    int sum = 0;
    for (int i = 0; i < kScanItems; ++i) {
        int d = base + i;
        int nl = d < D ? newLiveCount(cnt, cs, d) : 0;
        live[i] = nl;
        vals[i] = d < D ? nl + (d < L ? 1 : 0) : 0;
        sum += vals[i];
    }
    // exclusive scan of per-thread sums across the block
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
    }
    __syncthreads();
    int off0 = blockSums[blockIdx.x] + smem[w] + incl - sum;
    for (int i = 0; i < kScanItems; ++i) {
        int d = base + i;
        if (d < D) {
            segStartNext[d] = off0;
            cntNext[d] = live[i];
            if (d < L) vidNext[off0 + live[i]] = -1;  // the lane's spare slot of the next generation
            off0 += vals[i];
            if (d == D - 1) segStartNext[D] = off0;
        }
    }
}

// Phase 5c + 6: stable compaction into the next generation and commit of the buffered action
// (Engine::threadUpdateLocation / updateLocation engine.cpp:282-315,477-494; Vehicle::update
// vehicle.cpp:107-143; Router::update router.cpp:78-94).
__global__ void k_scatter(StepCtx c, ActionBuf b, CompactScratch cs, SlotArrays nx, const int32_t *segStartNext,
                          int32_t *oldToNew) {
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += stride) {
        const int vid = c.s.vid[s];
        if (vid < 0) {
            oldToNew[s] = -1;
            continue;
        }
        const int nd = b.drv[s];
        if (nd == -2) {  // finished: removed
            oldToNew[s] = -1;
            continue;
        }
        const int d = c.s.drv[s];
        int ns;
        if (nd == -1) {
            // stays: rank among the stayers of its segment = k - (#leavers in front of it)
            const int k = s - c.segStart[d];
            const int lc = cs.leaveCnt[d];
            int before;
            if (cs.maxLeaveIdx[d] + 1 == lc) {
                before = k < lc ? k : lc;  // leavers form a prefix (the normal case)
            } else {
                before = 0;
                for (int j = c.segStart[d]; j < s; ++j) before += (b.drv[j] != -1);
            }
            ns = segStartNext[d] + (k - before);
        } else {
            // enters drivable nd: after its stayers, ordered by new distance descending
            // (std::sort with vehicleCmp engine.h:21-23; ties: lower vid first, as in the twin)
            const double myDis = b.dis[s];
            int rank = 0;
            for (int j = cs.inHead[nd]; j >= 0; j = cs.inNext[j]) {
                if (j == s) continue;
                double od = b.dis[j];
                rank += (od > myDis) || (od == myDis && c.s.vid[j] < vid);
            }
            ns = segStartNext[nd] + (c.cnt[nd] - cs.leaveCnt[nd]) + rank;
        }
        oldToNew[s] = ns;
        nx.vid[ns] = vid;
        nx.templ[ns] = c.s.templ[s];
        nx.route[ns] = c.s.route[s];
        nx.dis[ns] = b.dis[s];
        nx.speed[ns] = b.speed[s];
        nx.blocker[ns] = b.blocker[s];  // old-generation slot; resolved through oldToNew when read
        if (nd == -1) {
            nx.drv[ns] = d;
            nx.prevDrv[ns] = c.s.prevDrv[s];
            nx.enterLLT[ns] = c.s.enterLLT[s];
            nx.routePos[ns] = c.s.routePos[s];
        } else {
            nx.drv[ns] = nd;
            nx.prevDrv[ns] = d;
            int rp = c.s.routePos[s];
            if (nd < c.n.L) {
                nx.enterLLT[ns] = CFX_INT_MAX;
                const int route = c.s.route[s];
                const int base = c.t.routeStart[route], n = c.t.routeStart[route + 1] - base;
                const int road = c.n.laneRoad[nd];
                while (rp < n && c.t.routeRoads[base + rp] != road) ++rp;
            } else {
                nx.enterLLT[ns] = c.step;
            }
            nx.routePos[ns] = rp;
        }
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
        leaderSlot[s] = findLeader(c, s, d, s - c.segStart[d], &gap);
        gapOut[s] = gap;
    }
}

__global__ void k_lane_waiting(StepCtx c, int32_t *out) {  // Engine::getLaneWaitingVehicleCount engine.cpp:636-648
    int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= c.n.L) return;
    int base = c.segStart[lane], n = c.cnt[lane], k = 0;
    for (int i = 0; i < n; ++i) k += c.s.speed[base + i] < 0.1;
    out[lane] = k;
}

__global__ void k_fill_i32(int32_t *p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Initial / reset layout: every lane owns just its spare slot, laneLinks are empty.
__global__ void k_init_layout(int D, int L, int32_t *segStart, int32_t *cnt, int32_t *vid) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > D) return;
    segStart[d] = d < L ? d : L;
    if (d < D) cnt[d] = 0;
    if (d < L) vid[d] = -1;
}

__global__ void k_init_lights(DevNet n, int32_t *curPhase, double *remain) {  // TrafficLight::init trafficlight.cpp:6-11
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n.I) return;
    curPhase[i] = 0;
    remain[i] = n.interVirtual[i] ? 0.0 : n.phaseTime[n.interPhaseStart[i]];
}

}  // namespace cfxd
