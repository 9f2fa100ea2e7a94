// Lane change on the device (reference src/vehicle/lanechange.cpp, engine.cpp:374-400,571-575,792-820; the semantics are
// the CPU twin's, oracle/twin/twin.cpp, which is pinned against the reference).  Only launched when the engine was created
// with cfx_config::lane_change; the per-step order is
//   k_spawn_link, k_admit                     as always (the admission sits in the lane's spare slot) + Lane::initSegments
//   k_lc_plan       threadPlanLaneChange: every real vehicle makes its signal
//   k_lc_schedule   scheduleLaneChange: one wave per road takes the position of its candidates in the reference's walk
//                   (creation order put through std::sort) and walks them in that order
//   k_lc_insert     Engine::insertShadow: vehicle numbers and priorities of the step's shadows, in walk order, and
//                   LaneChange::insertShadow: the lanes that get shadows move behind the layout's end, shadows in place
//   k_action, k_cross                         as always; a changing pair parks its two next speeds
//   k_lc_resolve    the vehicles whose step depends on an earlier vehicle of the reference's walk: changing pairs (common
//                   speed, lateral offset, finish / abort, engine.cpp:195-205,223-244) and vehicles they signalled
//   k_scan, k_scatter                         as always + LaneChange::clearSignal for every vehicle (engine.cpp:424)
// The reference finds a vehicle's neighbours in another lane through per-lane segment lists (Lane::initSegments,
// getVehicleAfterDistance / BeforeDistance, roadnet.cpp:863-898) — indexed with the segment number the vehicle has on its
// OWN lane, and inserts a shadow before the follower found that way.  On roads whose lanes differ in length that is not
// "the neighbours by distance", and the lane list can even lose its distance order.  lcInitSegments (k_admit) assigns the segment
// numbers exactly as initSegments does and every search below walks the lists the way the reference does.
#pragma once

#include "cfx_kernels.h"

namespace cfxd {

constexpr int kLcRoadInserts = 32;  // shadows one road can get in one step (more: CFX_ERR_CAPACITY)

// LaneChange::planChange lanechange.cpp:23-25
__device__ __forceinline__ bool lcPlanChange(const LcDev &lc, int vid, int drv) {
    return (lc.sigSend[vid] && lc.sendTarget[vid] >= 0 && lc.sendTarget[vid] != drv) || lc.changing[vid];
}

// SimpleLaneChange::estimateGap lanechange.cpp:221-226 for the two neighbour lanes of a vehicle AT ONCE (either may be
// switched off): Lane::getVehicleAfterDistance(dis, seg) roadnet.cpp:889-898 over a lane's vehicles — segments seg, seg+1,
// ... each back to front = the list from the last vehicle of a segment >= seg towards the front (segment numbers never
// increase along the list, lcInitSegments: that vehicle is found by bisection).  The two searches are chains of dependent
// loads of the same length; they advance in lockstep, every round's two loads in flight together.
__device__ inline void lcEstimateGap2(const StepCtx &c, const cfx_vehicle_template *tv, bool doA, int laneA, bool doB, int laneB,
                                      double dis, int seg, double *estA, double *estB) {
    int baseA = 0, nA = 0, baseB = 0, nB = 0;
    double lenA = 0.0, lenB = 0.0;
    if (doA) {
        baseA = c.segStart[laneA];
        nA = cntNow(c, laneA);
        lenA = c.n.drvLength[laneA];
    }
    if (doB) {
        baseB = c.segStart[laneB];
        nB = cntNow(c, laneB);
        lenB = c.n.drvLength[laneB];
    }
    int loA = 0, hiA = nA, loB = 0, hiB = nB;
    while (loA < hiA || loB < hiB) {  // first index whose segment number is < seg
        const bool a = loA < hiA, b = loB < hiB;
        const int midA = (loA + hiA) >> 1, midB = (loB + hiB) >> 1;
        int sa = 0, sb = 0;
        if (a) sa = c.lc.segOfSlot[baseA + midA];
        if (b) sb = c.lc.segOfSlot[baseB + midB];
        if (a) {
            if (sa < seg) hiA = midA;
            else loA = midA + 1;
        }
        if (b) {
            if (sb < seg) hiB = midB;
            else loB = midB + 1;
        }
    }
    int kA = loA - 1, kB = loB - 1;  // ... and from there towards the front: the first vehicle at or beyond `dis`
    bool fa = kA < 0, fb = kB < 0;
    double dA = 0.0, dB = 0.0;
    while (!fa || !fb) {
        double xa = 0.0, xb = 0.0;
        if (!fa) xa = c.s.dis[baseA + kA];
        if (!fb) xb = c.s.dis[baseB + kB];
        if (!fa) {
            if (xa >= dis) {
                fa = true;
                dA = xa;
            } else if (--kA < 0) {
                fa = true;
            }
        }
        if (!fb) {
            if (xb >= dis) {
                fb = true;
                dB = xb;
            } else if (--kB < 0) {
                fb = true;
            }
        }
    }
    int tA = 0, tB = 0;
    if (kA >= 0) tA = c.s.templ[baseA + kA];
    if (kB >= 0) tB = c.s.templ[baseB + kB];
    if (doA) *estA = kA >= 0 ? dA - dis - tv[tA].len : lenA - dis;
    if (doB) *estB = kB >= 0 ? dB - dis - tv[tB].len : lenB - dis;
}

// threadPlanLaneChange engine.cpp:374-390 + SimpleLaneChange::makeSignal lanechange.cpp:151-184, one thread per slot.
// Also: the stored gap as the leader pass at the end of the previous step (and handleWaiting for this step's admissions)
// left it, and the slot of every vehicle.
__global__ void k_lc_plan(StepCtx c) {
    const cfx_vehicle_template *tv = c.t.templ;
    const LcDev &lc = c.lc;
    const int S = c.segStart[c.n.L + c.n.K];
    const int stride = gridDim.x * blockDim.x;
    // (whole waves go round together: the step's candidate list is appended to once per wave, waveListAppend)
    for (int s0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); s0 < S; s0 += stride) {
        const int s = s0 + (int) (threadIdx.x & 63u);
        const int vid = s < S ? c.s.vid[s] : -1;
        bool isCand = false;
        int d = -1;
        if (vid >= 0) {
            d = c.s.drv[s];
            lc.slotOf[vid] = s;
            const cfx_vehicle_template &t = tv[c.s.templ[s]];
            const double dis = c.s.dis[s];
            {
                const bool head = s == 0 || c.s.drv[s - 1] != d;
                double gap;
                const int ls = findLeader(c, tv, s, d, head, dis, t.approach_dist, &gap);
                if (ls >= 0) lc.gap[vid] = (c.s.flags[s] & kFlagStateGap) ? c.vGapState[vid] : gap;  // (first step after a load: the state's)
            }
            if (lc.ptype[vid] == 2) {
                // shadows make no signals (isReal)
            } else if (lc.changing[vid]) {  // keeps the signal it started with; still a candidate (it signals its neighbours)
                isCand = d < c.n.L;
            } else if (!(c.step * c.interval - lc.lastChangeTime[vid] < 3 /*coolingTime*/)) {
                int target = -1, dir = 0, urgency = 0;
                if (d < c.n.L) {
                    const double dlen = c.n.drvLength[d];
                    bool go = !(dlen - dis < 30);
                    const double gap = lc.gap[vid];
                    const double expectedGap = 2 * t.len + 4 * c.interval * t.max_speed;
                    if (go && (gap > expectedGap || gap < 1.5 * t.len)) go = false;
                    if (go) {
                        const int road = c.n.laneRoad[d];
                        const int nLanes = lc.roadLaneStart[road + 1] - lc.roadLaneStart[road];
                        const int route = c.s.route[s], routePos = c.s.routePos[s];
                        const bool lastRoad = isLastRoad(c, d, route);
                        const int li = c.n.laneIndex[d];
                        // the outer lane first, then the inner one if it is better still (makeSignal lanechange.cpp:160-175)
                        const bool tryOut = li < nLanes - 1, tryIn = li > 0;
                        int nOut = -1, nIn = -1;
                        if (!lastRoad) {
                            if (tryOut) nOut = nextOf(c.n, c.t, d + 1, route, routePos);
                            if (tryIn) nIn = nextOf(c.n, c.t, d - 1, route, routePos);
                        }
                        const bool evalOut = tryOut && (lastRoad || nOut >= 0), evalIn = tryIn && (lastRoad || nIn >= 0);
                        double outerEst = 0, innerEst = 0;
                        if (evalOut || evalIn)
                            lcEstimateGap2(c, tv, evalOut, d + 1, evalIn, d - 1, dis, c.lc.segOfSlot[s], &outerEst, &innerEst);
                        if (evalOut && outerEst > gap + t.len) target = d + 1;
                        if (evalIn && innerEst > gap + t.len && innerEst > outerEst) target = d - 1;
                        urgency = 1;
                        if (target >= 0) dir = target == d + 1 ? 1 : -1;  // LaneChange::getDirection lanechange.cpp:104-113
                    }
                }
                lc.sigSend[vid] = 1;
                lc.sendTarget[vid] = target;
                lc.sendDir[vid] = (int8_t) dir;
                lc.sendUrg[vid] = (int8_t) urgency;
                isCand = target >= 0;
            }
        }
        // a candidate goes into its road's list (the schedule walk) and into the step's (its position in that walk)
        if (isCand) {
            const int road = c.n.laneRoad[d];
            const int i = atomicAdd(&lc.roadCand[road], 1);
            if (i < kLcRoadCand) lc.roadCandList[(size_t) road * kLcRoadCand + i] = make_int2(vid, s);
        }
        const int at = waveListAppend(lc.candAllCount, isCand);
        if (isCand) {
            lc.candAll[at] = vid;
            if (lc.roadsPerEnv < c.n.R) lc.candAllEnv[at] = c.n.laneRoad[d] / lc.roadsPerEnv;  // (batched environments only)
        }
    }
}

// The order of the reference's walk (engine.cpp:793-795): the candidates in creation order (ascending vid), then
// `std::sort` by urgency.  All urgencies are 1, so the comparator is always false — and libstdc++'s introsort still
// permutes: every partition step swaps the first element with the middle one and reverses the rest (__move_median_to_first
// + __unguarded_partition with a comparator that never fires), recursing on both halves while they hold more than 16
// elements; the final insertion sort moves nothing.  The walk position of the candidate with creation rank i among n is
// therefore a closed function of (i, n) — checked against std::sort itself for every n up to 3000.
__device__ inline int lcSortedPosition(int i, int n) {
    int f = 0, l = n, x = i;
    while (l - f > 16) {
        const int mid = f + (l - f) / 2;
        if (x == f) x = mid;
        else if (x == mid) x = f;
        if (x >= f + 1) x = (f + 1) + (l - 1) - x;
        const int cut = f + 1 + (l - f - 1) / 2;
        if (x >= cut) f = cut;
        else l = cut;
    }
    return x;
}

// ... taken by the road's wave in k_lc_schedule: creation rank of `me` among the step's candidates, all 64 lanes counting
// (batched environments: every environment is an Engine of its own — rank and count among ITS candidates)
__device__ __forceinline__ int lcWalkPosition(const LcDev &lc, int me, int env, int nAll, int tid) {
    if (env < 0) {  // one environment: every candidate of the step is one of its own
        int less = 0;
        for (int i = tid; i < nAll; i += 64) less += lc.candAll[i] < me;
        for (int off = 32; off > 0; off >>= 1) less += __shfl_down(less, off, 64);
        return lcSortedPosition(__shfl(less, 0, 64), nAll);
    }
    int less = 0, mine = 0;
    for (int i = tid; i < nAll; i += 64) {
        const bool same = lc.candAllEnv[i] == env;
        mine += same;
        less += same && lc.candAll[i] < me;
    }
    for (int off = 32; off > 0; off >>= 1) {
        less += __shfl_down(less, off, 64);
        mine += __shfl_down(mine, off, 64);
    }
    return lcSortedPosition(__shfl(less, 0, 64), __shfl(mine, 0, 64));
}

// What the schedule walk knows about "a vehicle in the target lane": an existing one (slot) or a shadow inserted earlier
// in this very walk (index into the road's local list).
struct LcNeighbour {
    int vid;      // vid, or -(record index + 2) for a shadow of this step, or -1 none
    double dis, len, speed, maxNegAcc;
};

// Engine::scheduleLaneChange engine.cpp:792-810 for ONE road: its candidates in the order of the reference's walk
// (k_lc_order; include/cityflow_amd.h "Lane change").  Roads are independent in this phase: a target lane is on the
// candidate's own road, and the laneLinks the leader search looks into do not change.
// The walk itself is strictly sequential — one chain of dependent accesses per candidate (its own fields, the target lane's
// segment lists searched twice, the neighbours found, the road's shadows so far).  One WAVE per road: its 64 lanes first
// copy what the walk will look at into LDS in a few coalesced rounds — every slot of the road's lanes {vehicle, distance,
// speed, segment, length and deceleration of its template}, the lanes' {start, count, segments}, the candidates with their
// walk positions — then lane 0 walks, and a step of the chain costs an LDS access instead of a trip to HBM.  A road with more
// slots than the stage holds is walked from global memory (same code, `staged` false).
constexpr int kLcSegItems = 64;
constexpr int kLcSchedStage = 384;  // slots of one road's lanes kept in LDS
constexpr int kLcSchedLanes = 8;    // lanes of one road whose layout is kept in LDS
constexpr int kLcSchedSegs = 24;    // segments per lane for which the lanes' segment runs are tabulated (more: searched per candidate)
__global__ __launch_bounds__(64) void k_lc_schedule(StepCtx c, DevScalars *sc, const int32_t *vPriority) {
    __shared__ int localRec[kLcRoadInserts];  // global record indices of this road's shadows so far ...
    __shared__ int insLane[kLcRoadInserts], insSeg[kLcRoadInserts], insAnchor[kLcRoadInserts], insParent[kLcRoadInserts];
    __shared__ double insDis[kLcRoadInserts], insSeq[kLcRoadInserts];  // ... and what later candidates read of them
    __shared__ int candVid[kLcRoadCand], candSlot[kLcRoadCand], candKey[kLcRoadCand];
    // what a candidate's turn reads of its OWN vehicle-table rows and nobody changes during the walk: all candidates' at once
    __shared__ int candTarget[kLcRoadCand], candPrio[kLcRoadCand];
    __shared__ int itemRef[kLcSegItems];  // >= 0 existing index in the lane; < 0: -(local shadow index + 1)
    __shared__ double itemDis[kLcSegItems];
    __shared__ double stDis[kLcSchedStage], stSpeed[kLcSchedStage], stLen[kLcSchedStage], stNegAcc[kLcSchedStage];
    __shared__ int stVid[kLcSchedStage], stSeg[kLcSchedStage], stDrv[kLcSchedStage];
    __shared__ int lnStart[kLcSchedLanes], lnCnt[kLcSchedLanes], lnSegs[kLcSchedLanes];
    // lnRun[lane][g]: the first list index of the lane whose segment number is < g (the lane's count if there is none)
    __shared__ int lnRun[kLcSchedLanes][kLcSchedSegs + 1];
    __shared__ int sRunsOff;
    const int road = blockIdx.x;
    if (road >= c.n.R) return;
    const LcDev &lc = c.lc;
    const bool oneEnv = lc.roadsPerEnv >= c.n.R;
    const int env = oneEnv ? -1 : road / lc.roadsPerEnv;  // (-1: lcWalkPosition counts all candidates)
    const int nListed = lc.roadCand[road];
    if (nListed == 0) return;
    KSTAMP(9, 0);
    KNOTE(9, 5, nListed);
    const int tid = threadIdx.x;
    const cfx_vehicle_template *tv = c.t.templ;
    const int l0 = lc.roadLaneStart[road], l1 = lc.roadLaneStart[road + 1];
    const int s0 = c.segStart[l0], s1 = c.segStart[l1 - 1] + cntNow(c, l1 - 1);
    const bool staged = s1 - s0 <= kLcSchedStage && l1 - l0 <= kLcSchedLanes;
    if (staged) {
        for (int i = tid; i < s1 - s0; i += 64) {
            const int q = s0 + i;
            const int v = c.s.vid[q];
            stVid[i] = v;
            stDrv[i] = c.s.drv[q];
            stDis[i] = c.s.dis[q];
            stSpeed[i] = c.s.speed[q];
            stSeg[i] = lc.segOfSlot[q];
            if (v >= 0) {  // (a lane's spare slot holds nothing)
                const cfx_vehicle_template &tq = tv[c.s.templ[q]];
                stLen[i] = tq.len;
                stNegAcc[i] = tq.max_neg_acc;
            }
        }
        if (tid < l1 - l0) {
            lnStart[tid] = c.segStart[l0 + tid];
            lnCnt[tid] = cntNow(c, l0 + tid);
            lnSegs[tid] = lc.laneNumSegs[l0 + tid];
        }
    }
    if (tid == 0) sRunsOff = 0;
    KSTAMP(9, 1);
    const bool tooMany = nListed > kLcRoadCand;
    const int nAll = *lc.candAllCount;
    KNOTE(9, 7, nAll);
    if (!tooMany && tid < nListed) {  // the road's candidates (sorted by walk position below)
        const int2 e = lc.roadCandList[(size_t) road * kLcRoadCand + tid];
        candVid[tid] = e.x;
        candSlot[tid] = e.y;
        candTarget[tid] = lc.sendTarget[e.x];
        candPrio[tid] = vPriority[e.x];
    }
    __syncthreads();
    // Where each segment's run begins in every lane's list (segment numbers never increase along a list, lcInitSegments): all
    // lanes of the wave, one (lane, segment) each — the walk below then finds "the vehicles of segments >= mine" and "... <=
    // mine" with one LDS read instead of a bisection per segment it looks into
    if (staged) {
        const int stride = lnSegs[0] + 1;
        if (stride - 1 > kLcSchedSegs) {
            if (tid == 0) sRunsOff = 1;
        } else {
            for (int e = tid; e < (l1 - l0) * stride; e += 64) {
                const int li = e / stride, g = e - li * stride;
                if (lnSegs[li] != stride - 1) sRunsOff = 1;  // (lanes of one road have the same number of segments; otherwise: the searches)
                const int base = lnStart[li] - s0;
                int lo = 0, hi = lnCnt[li];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (stSeg[base + mid] >= g) lo = mid + 1;
                    else hi = mid;
                }
                lnRun[li][g] = lo;
            }
        }
    }
    // Walk positions (the `std::sort` of scheduleLaneChange, lcSortedPosition): the creation rank of a candidate is the
    // number of the step's candidates with a smaller vehicle number — counted here, by the wave that needs it, over the
    // step's list (a few KB, read by every road's wave: it stays in L2).  k_lc_assign reads candPos of the shadows' parents.
    if (!tooMany && env < 0 && nAll <= 4 * 64) {
        // one environment, up to 256 candidates in the step (the bench workload: ~150): the step's list goes into registers
        // once — four vehicle numbers per lane — and every candidate of the road is counted from there
        int mine[4];
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 64 * q;
            mine[q] = i < nAll ? lc.candAll[i] : CFX_INT_MAX;
        }
        for (int j = 0; j < nListed; ++j) {
            const int me = candVid[j];
            int less = (mine[0] < me) + (mine[1] < me) + (mine[2] < me) + (mine[3] < me);
            for (int off = 32; off > 0; off >>= 1) less += __shfl_down(less, off, 64);
            if (tid == 0) {
                const int key = lcSortedPosition(less, nAll);
                candKey[j] = key;
                lc.candPos[me] = key;
            }
        }
    } else if (!tooMany) {
        for (int j = 0; j < nListed; ++j) {
            const int me = candVid[j];
            const int key = lcWalkPosition(lc, me, env, nAll, tid);
            if (tid == 0) {
                candKey[j] = key;
                lc.candPos[me] = key;
            }
        }
    } else {  // (slow, rare) every candidate on the road, found the way the walk below finds them
        for (int q = s0; q < s1; ++q) {
            const int w = c.s.vid[q];
            if (w < 0 || lc.ptype[w] == 2 || !lcPlanChange(lc, w, c.s.drv[q])) continue;
            const int key = lcWalkPosition(lc, w, env, nAll, tid);
            if (tid == 0) lc.candPos[w] = key;
        }
        __threadfence_block();
    }
    __syncthreads();
    KSTAMP(9, 2);
    if (tid != 0) return;
    lc.roadCand[road] = 0;
    // accessors: a slot of this road's lanes, a lane of this road
    auto slotVid = [&](int q) { return staged ? stVid[q - s0] : c.s.vid[q]; };
    auto slotDrv = [&](int q) { return staged ? stDrv[q - s0] : c.s.drv[q]; };
    auto slotDisOf = [&](int q) { return staged ? stDis[q - s0] : c.s.dis[q]; };
    auto slotSpeedOf = [&](int q) { return staged ? stSpeed[q - s0] : c.s.speed[q]; };
    auto slotSeg = [&](int q) { return staged ? stSeg[q - s0] : lc.segOfSlot[q]; };
    auto slotLen = [&](int q) { return staged ? stLen[q - s0] : tv[c.s.templ[q]].len; };
    auto slotNegAcc = [&](int q) { return staged ? stNegAcc[q - s0] : tv[c.s.templ[q]].max_neg_acc; };
    auto laneStart = [&](int lane) { return staged ? lnStart[lane - l0] : c.segStart[lane]; };
    auto laneCount = [&](int lane) { return staged ? lnCnt[lane - l0] : cntNow(c, lane); };
    auto laneSegs = [&](int lane) { return staged ? lnSegs[lane - l0] : lc.laneNumSegs[lane]; };
    int nLocal = 0;
    // the road's candidates in walk order (threadPlanLaneChange's buffer after the sort; the walk never creates new ones)
    int nCand = 0;
    if (!tooMany) {
        nCand = nListed;
        for (int j = 1; j < nCand; ++j) {  // insertion sort by walk position
            const int v = candVid[j], q = candSlot[j], key = candKey[j], tg = candTarget[j], pr = candPrio[j];
            int i = j;
            for (; i > 0 && candKey[i - 1] > key; --i) {
                candVid[i] = candVid[i - 1];
                candSlot[i] = candSlot[i - 1];
                candKey[i] = candKey[i - 1];
                candTarget[i] = candTarget[i - 1];
                candPrio[i] = candPrio[i - 1];
            }
            candVid[i] = v;
            candSlot[i] = q;
            candKey[i] = key;
            candTarget[i] = tg;
            candPrio[i] = pr;
        }
    }
    int lastKey = -1;
#if defined(CFX_TRACE) && CFX_TRACE_KERNEL == 9
    long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tMark = (long long) KCLOCK();
#define LC_SPAN(acc)                                \
    {                                               \
        const long long now_ = (long long) KCLOCK(); \
        acc += now_ - tMark;                        \
        tMark = now_;                               \
    }
#else
#define LC_SPAN(acc)
#endif
    for (int ci = 0;; ++ci) {
        int vid, s, myKey, target = -1, myPriority = 0;
        if (!tooMany) {
            if (ci >= nCand) break;
            vid = candVid[ci];
            s = candSlot[ci];
            myKey = candKey[ci];
            target = candTarget[ci];
            myPriority = candPrio[ci];
        } else {  // more candidates than the local list holds: pick the next one by scanning (slow, rare)
            vid = -1;
            s = -1;
            int best = CFX_INT_MAX;
            for (int q = s0; q < s1; ++q) {
                const int w = slotVid(q);
                if (w < 0 || lc.ptype[w] == 2 || !lcPlanChange(lc, w, slotDrv(q))) continue;
                const int key = lc.candPos[w];
                if (key <= lastKey || key >= best) continue;
                best = key;
                vid = w;
                s = q;
            }
            if (s < 0) break;
            lastKey = best;
            myKey = best;
        }
        const int d = slotDrv(s);
        if (tooMany) {
            target = lc.sendTarget[vid];
            myPriority = vPriority[vid];
        }
        const double dis = slotDisOf(s);
        const double myLen = slotLen(s), myNegAcc = slotNegAcc(s);
        // --- LaneChange::updateLeaderAndFollower lanechange.cpp:27-60 on the target lane as it is NOW: its segment lists,
        // walked with the candidate's own segment number, earlier shadows of this walk included (each sits in the segment
        // its parent's number names, at the place Segment::insertVehicle gave it, roadnet.cpp:943-947)
        LcNeighbour leader{-1, 0, 0, 0, 0}, follower{-1, 0, 0, 0, 0};
        const int tb = laneStart(target), tn = laneCount(target);
        LC_SPAN(tA)
        int followerAnchor = tn;  // lane-list position of the follower: existing index, or ...
        int followerRec = -1;     // ... the earlier shadow it is
        const int mySeg = slotSeg(s), nSeg = laneSegs(target);
        constexpr int kItems = kLcSegItems;
        auto buildSegment = [&](int i) {  // the sequence of segment i of the target lane
            int m = 0;
            // segment numbers never increase along the list (lcInitSegments): the members are one run
            int lo = 0, hi = tn;
            while (lo < hi) {  // first index whose segment number is <= i
                const int mid = (lo + hi) >> 1;
                if (slotSeg(tb + mid) > i) lo = mid + 1;
                else hi = mid;
            }
            for (int k = lo; k < tn && slotSeg(tb + k) == i; ++k) {
                if (m == kItems) {
                    sc->overflow = 6;
                    break;
                }
                itemRef[m] = k;
                itemDis[m++] = slotDisOf(tb + k);
            }
            for (int j = 0; j < nLocal; ++j) {
                if (insLane[j] != target || insSeg[j] != i) continue;
                const double rdis = insDis[j];
                int p = 0;
                while (p < m && itemDis[p] > rdis) ++p;  // before the first member that is not further ahead
                if (m == kItems) {
                    sc->overflow = 6;
                    break;
                }
                for (int q = m; q > p; --q) {
                    itemRef[q] = itemRef[q - 1];
                    itemDis[q] = itemDis[q - 1];
                }
                itemRef[p] = -(j + 1);
                itemDis[p] = rdis;
                ++m;
            }
            return m;
        };
        auto neighbourOf = [&](int ref) {
            if (ref >= 0) {
                const int ns = tb + ref;
                return LcNeighbour{slotVid(ns), slotDisOf(ns), slotLen(ns), slotSpeedOf(ns), slotNegAcc(ns)};
            }
            const int j = -ref - 1;
            const int ps = insParent[j];
            return LcNeighbour{-(localRec[j] + 2), insDis[j], slotLen(ps), slotSpeedOf(ps), slotNegAcc(ps)};
        };
        // No shadow of this walk in the target lane yet (the rule): the segment lists are runs of the lane's list, and walking
        // segments mySeg, mySeg + 1, ... each back to front IS walking the list from the end of run mySeg towards the front;
        // segments mySeg, mySeg - 1, ... each front to back is walking it from the start of run mySeg towards the tail.
        bool runs = staged && !sRunsOff && mySeg <= kLcSchedSegs && nSeg <= kLcSchedSegs;
        for (int j = 0; runs && j < nLocal; ++j)
            if (insLane[j] == target) runs = false;
        if (runs) {
            const int tl = target - l0;
            if (mySeg < nSeg)
                for (int k = lnRun[tl][mySeg] - 1; k >= 0; --k)
                    if (slotDisOf(tb + k) >= dis) {
                        leader = neighbourOf(k);
                        break;
                    }
            const int i0 = mySeg < nSeg ? mySeg : nSeg - 1;
            if (i0 >= 0)
                for (int k = lnRun[tl][i0 + 1]; k < tn; ++k)
                    if (slotDisOf(tb + k) < dis) {
                        follower = neighbourOf(k);
                        followerAnchor = k;
                        break;
                    }
        }
        for (int i = mySeg; !runs && i < nSeg && leader.vid == -1; ++i) {  // getVehicleAfterDistance: back to front
            const int m = buildSegment(i);
            for (int p = m - 1; p >= 0; --p)
                if (itemDis[p] >= dis) {
                    leader = neighbourOf(itemRef[p]);
                    break;
                }
        }
        for (int i = mySeg < nSeg ? mySeg : nSeg - 1; !runs && i >= 0 && follower.vid == -1; --i) {  // getVehicleBeforeDistance
            const int m = buildSegment(i);
            for (int p = 0; p < m; ++p)
                if (itemDis[p] < dis) {
                    follower = neighbourOf(itemRef[p]);
                    if (itemRef[p] >= 0) followerAnchor = itemRef[p];
                    else followerRec = -itemRef[p] - 1;  // (index in the road's local list)
                    break;
                }
        }
        LC_SPAN(tB)
        double leaderGap, followerGap = 1.7976931348623157e308;
        if (leader.vid == -1) {  // look into the laneLinks behind the target lane
            const double rest = c.n.drvLength[d] - dis;
            leaderGap = rest;
            double gap = 1.7976931348623157e308;
            for (int q = c.n.laneLLStart[target]; q < c.n.laneLLStart[target + 1]; ++q) {
                const int ll = c.n.L + c.n.laneLL[q];
                const int n = c.cnt[ll];
                if (n <= 0) continue;
                const int ls = c.segStart[ll] + n - 1;
                const double ld = c.s.dis[ls];
                if (ld + rest < gap) {
                    gap = ld + rest;
                    const cfx_vehicle_template &tl = tv[c.s.templ[ls]];
                    if (gap < tl.len) {
                        leader = LcNeighbour{c.s.vid[ls], ld, tl.len, c.s.speed[ls], tl.max_neg_acc};
                        leaderGap = rest - (tl.len - gap);
                    }
                }
            }
        } else {
            leaderGap = leader.dis - dis - leader.len;
        }
        LC_SPAN(tC)
        if (follower.vid != -1) followerGap = dis - follower.dis - myLen;
        lc.tLeader[vid] = leader.vid;
        lc.tFollower[vid] = follower.vid;
        for (int w = 0; w < 2; ++w) {  // shadows of this step get their numbers in k_lc_assign
            const int x = w ? follower.vid : leader.vid;
            if (x > -2) continue;
            const int fi = atomicAdd(lc.fixCount, 1);
            if (fi >= lc.fixCap) {
                sc->overflow = 6;
                continue;
            }
            lc.fixList[3 * fi] = vid;
            lc.fixList[3 * fi + 1] = w;
            lc.fixList[3 * fi + 2] = -x - 2;
        }
        lc.leaderGap[vid] = leaderGap;
        lc.followerGap[vid] = followerGap;
        // --- SimpleLaneChange::sendSignal lanechange.cpp:208-211 -> Vehicle::receiveSignal vehicle.cpp:391-401
        const LcNeighbour *nb[2] = {&leader, &follower};
        // (both neighbours' rows requested together: the walk is a chain of trips to memory, each one saved is half a microsecond)
        int nbChanging[2] = {0, 0}, nbFrom[2] = {-1, -1}, nbSig[2] = {0, 0}, nbPrio[2] = {0, 0};
        for (int i = 0; i < 2; ++i) {
            const int r = nb[i]->vid;
            if (r < 0) continue;
            nbChanging[i] = lc.changing[r];
            nbFrom[i] = lc.recvFrom[r];
            nbSig[i] = lc.sigSend[r];
            nbPrio[i] = vPriority[r];
        }
        if (leader.vid >= 0 && leader.vid == follower.vid) nbFrom[1] = -2;  // (cannot happen; re-read below if it ever does)
        for (int i = 0; i < 2; ++i) {
            const int r = nb[i]->vid;
            if (r == -1) continue;
            if (r <= -2) {  // a shadow of this step: not changing, sends nothing itself
                LcInsert &rec = lc.ins[-r - 2];
                const int cur = rec.recvFrom >= 0 ? vPriority[rec.recvFrom] : -1;
                if (rec.recvFrom < 0 || cur < myPriority) rec.recvFrom = vid;
                continue;
            }
            if (nbChanging[i]) continue;
            const int from = nbFrom[i] == -2 ? lc.recvFrom[r] : nbFrom[i];
            const int cur = from >= 0 ? vPriority[from] : -1;
            if ((from < 0 || cur < myPriority) && (!nbSig[i] || nbPrio[i] < myPriority)) lc.recvFrom[r] = vid;
        }
        LC_SPAN(tD)
        // --- insert a shadow? engine.cpp:800-806, LaneChange::isGapValid lanechange.h:80
        if (lcPlanChange(lc, vid, d) && lc.sigSend[vid] && lc.recvFrom[vid] < 0 && !lc.changing[vid] && d < c.n.L) {
            const double speed = slotSpeedOf(s);
            const double safeAfter = 0.5 * speed * speed / myNegAcc;
            const double safeBefore = follower.vid != -1 ? 0.5 * follower.speed * follower.speed / follower.maxNegAcc : 0.0;
            if (leaderGap >= safeAfter && followerGap >= safeBefore) {
                const int idx = atomicAdd(lc.insCount, 1);
                if (idx >= lc.insCap) {
                    sc->overflow = 5;  // more shadows in one step than priorities supplied
                } else if (nLocal >= kLcRoadInserts) {
                    sc->overflow = 6;  // more shadows on one road in one step than the walk keeps track of
                } else {
                    // the lane-list place: right before the follower (LaneChange::insertShadow lanechange.cpp:91-93)
                    int anchor = followerAnchor;
                    double seq;
                    if (followerRec >= 0) {  // before an earlier shadow: same anchor, between it and its predecessor
                        anchor = insAnchor[followerRec];
                        const double fseq = insSeq[followerRec];
                        double prev = fseq - 2.0;
                        for (int j = 0; j < nLocal; ++j)
                            if (insLane[j] == target && insAnchor[j] == anchor && insSeq[j] < fseq && insSeq[j] > prev) prev = insSeq[j];
                        seq = (prev + fseq) / 2;
                    } else {                 // before an existing vehicle (or at the end): behind the shadows already there
                        seq = 0.0;
                        for (int j = 0; j < nLocal; ++j)
                            if (insLane[j] == target && insAnchor[j] == anchor && insSeq[j] >= seq) seq = insSeq[j] + 1.0;
                    }
                    lc.ins[idx] = LcInsert{vid, s, target, -1, dis, lc.gap[vid], anchor, mySeg, seq};
                    // (a walk position beyond the field would spill into the next environment's range: code 12, never silent)
                    if (!oneEnv && myKey >= (1 << kLcEnvShift)) sc->overflow = 12;
                    lc.insKey[idx] = ((oneEnv ? 0 : env) << kLcEnvShift) + myKey;  // shadows are created, and numbered, in walk order (k_lc_insert)
                    // LaneChange::insertShadow lanechange.cpp:98-100: the follower's leader is the shadow from now on — a
                    // later candidate of this walk that copies itself (its own shadow) copies this gap too
                    if (follower.vid >= 0) lc.gap[follower.vid] = dis - myLen - follower.dis;
                    if (lc.insHead[target] < 0) lc.insLanes[atomicAdd(lc.insLaneCount, 1)] = target;
                    lc.insNext[idx] = lc.insHead[target];  // only this thread touches this road's lanes
                    lc.insHead[target] = idx;
                    localRec[nLocal] = idx;
                    insLane[nLocal] = target;
                    insSeg[nLocal] = mySeg;
                    insAnchor[nLocal] = anchor;
                    insParent[nLocal] = s;
                    insDis[nLocal] = dis;
                    insSeq[nLocal] = seq;
                    ++nLocal;
                    lc.changing[vid] = 1;  // LaneChange::insertShadow lanechange.cpp:71-76 (later candidates must see it)
                }
            }
        }
        LC_SPAN(tE)
    }
#if defined(CFX_TRACE) && CFX_TRACE_KERNEL == 9
    // (what the walk's time went into, 10 ns ticks: own fields | segment searches, then laneLinks | signals | insertion)
    KNOTE(9, 3, tA | (tB << 32));
    KNOTE(9, 6, tC | (tD << 20) | (tE << 40));
#endif
    KSTAMP(9, 4);
}

// Engine::insertShadow engine.cpp:812-820 + the Vehicle copy constructor vehicle.cpp:28-36 + LaneChange::insertShadow
// lanechange.cpp:83-95 for every lane that gets shadows in this step (~75 of 14 k at the benchmark's size), one 64-thread
// block per listed lane.
//  * Vehicle numbers and the supplied priorities go out in creation order = walk order: a shadow's number is the count of the
//    step's shadows whose parent comes earlier in the walk (LcDev::insKey).
//  * The lane's vehicles MOVE to fresh slots behind the layout's end — room for them and the shadows is taken from
//    segStart[D] itself, which every later kernel of the step reads as the number of slots — each behind the shadows that go
//    in front of it, and the shadows take the gaps: right before their target follower (LcInsert::anchor / seq), copied from
//    their parents' slots (a parent's lane may be moving at the same time: its old slot keeps everything but the vehicle
//    number).  The lane's old slots are left empty (vid -1) and its start points to the new ones until k_scan / k_scatter lay
//    the next generation out; nothing else of the layout moves, the step's admission stays the pending one (cntNow).
//    Stored blockers stay valid: they are slots of the PREVIOUS generation that go through oldToNew, and a moved vehicle's
//    entry there is found through newToOld (k_scatter).
//  * Report for cfx_lane_change_poll (pinned memory + event: the host waits for this kernel, not for the step).
__global__ __launch_bounds__(64) void k_lc_insert(StepCtx c, VidTable vt, DevScalars *sc,
                                                  int32_t *pollOut /*pinned: [0] count, [1] overflow code of the walk, [2..] parents*/,
                                                  int32_t *oldToNew, int32_t *segStartNow, int32_t *cntNow_, int slotCap) {
    const LcDev &lc = c.lc;
    __shared__ int sRec[kLcRoadInserts], sAnchor[kLcRoadInserts], sVid[kLcRoadInserts], sPool[kLcRoadInserts];
    __shared__ double sSeq[kLcRoadInserts];
    __shared__ int sM, sBase;
    const int nLanes = *lc.insLaneCount;
    int nIns = *lc.insCount;
    if (nIns > lc.insCap) nIns = lc.insCap;
    const int tid = threadIdx.x;
    const int D = c.n.L + c.n.K;
    // creation rank of record `rec` among the step's shadows, counted by the whole wave
    // (.x among all of them: the vehicle number; .y among its environment's: the priority it takes)
    const bool oneEnv = lc.roadsPerEnv >= c.n.R;
    auto rankOf = [&](int rec) {
        const int key = lc.insKey[rec];
        if (oneEnv) {
            int less = 0;
            for (int j = tid; j < nIns; j += 64) less += lc.insKey[j] < key;
            for (int off = 32; off > 0; off >>= 1) less += __shfl_down(less, off, 64);
            less = __shfl(less, 0, 64);
            return make_int2(less, less);
        }
        int less = 0, lessEnv = 0;
        for (int j = tid; j < nIns; j += 64) {
            const int kj = lc.insKey[j];
            less += kj < key;
            lessEnv += kj < key && (kj >> kLcEnvShift) == (key >> kLcEnvShift);
        }
        for (int off = 32; off > 0; off >>= 1) {
            less += __shfl_down(less, off, 64);
            lessEnv += __shfl_down(lessEnv, off, 64);
        }
        return make_int2(__shfl(less, 0, 64), __shfl(lessEnv, 0, 64));
    };
    for (int w = blockIdx.x; w < nLanes; w += gridDim.x) {
        const int d = lc.insLanes[w];
        const int base = c.segStart[d];
        const int live = c.cnt[d];
        const bool admitted = c.admitStep[d] == c.step;
        const int n = live + (admitted ? 1 : 0);
        __syncthreads();  // (the previous lane's lists are consumed)
        if (tid == 0) {
            int m = 0;
            for (int r = lc.insHead[d]; r >= 0; r = lc.insNext[r]) {
                if (m < kLcRoadInserts) {  // (a road's limit in k_lc_schedule, so a lane's too)
                    sRec[m] = r;
                    sAnchor[m] = lc.ins[r].anchor;
                    sSeq[m] = lc.ins[r].seq;
                }
                ++m;
            }
            sM = m;
            lc.insHead[d] = -1;
            const int nb = m <= kLcRoadInserts ? atomicAdd(&segStartNow[D], n + m) : -1;
            sBase = (nb >= 0 && nb + n + m <= slotCap) ? nb : -1;
        }
        __syncthreads();
        const int m = sM, nbase = sBase;
        if (nbase < 0) {  // too many shadows for one lane, or no room behind the layout (the host reserves a true bound, cfx_step)
            if (tid == 0) sc->overflow = m > kLcRoadInserts ? 12 : 11;
            continue;
        }
        for (int q = 0; q < m; ++q) {
            const int2 rank = rankOf(sRec[q]);
            if (tid == 0) {
                sVid[q] = lc.firstShadowVid + rank.x;
                sPool[q] = rank.y < lc.poolPerEnv ? (lc.insKey[sRec[q]] >> kLcEnvShift) * lc.poolPerEnv + rank.y : -1;
                if (sPool[q] < 0) sc->overflow = 5;  // an environment used up its share of the supplied priorities
            }
        }
        __syncthreads();
        for (int k = tid; k < n; k += 64) {
            const int s = base + k;
            int shift = 0;
            for (int j = 0; j < m; ++j) shift += sAnchor[j] <= k;
            const int ns = nbase + k + shift;
            const int vid = c.s.vid[s];
            c.s.vid[ns] = vid;
            c.s.drv[ns] = d;
            c.s.prevDrv[ns] = c.s.prevDrv[s];
            c.s.next[ns] = c.s.next[s];
            c.s.blocker[ns] = c.s.blocker[s];
            c.s.enterLLT[ns] = c.s.enterLLT[s];
            c.s.routePos[ns] = c.s.routePos[s];
            c.s.templ[ns] = c.s.templ[s];
            c.s.route[ns] = c.s.route[s];
            c.s.flags[ns] = c.s.flags[s];
            c.s.dis[ns] = c.s.dis[s];
            c.s.speed[ns] = c.s.speed[s];
            lc.slotOf[vid] = ns;
            if (!(admitted && k == live)) {  // (this step's admission has no previous slot)
                const int old = lc.newToOld[s];
                if (old >= 0) oldToNew[old] = ns;
            }
            c.s.vid[s] = -1;
            c.s.drv[s] = -1;
        }
        if (tid < m) {
            const int r = sRec[tid];
            const LcInsert rec = lc.ins[r];
            const int p = rec.parentVid, ps = rec.parentSlot, v = sVid[tid], rank = v - lc.firstShadowVid;
            // the new vehicle (Engine::insertShadow, Vehicle copy constructor, setShadow / setParent)
            vt.priority[v] = sPool[tid] >= 0 ? lc.pool[sPool[tid]] : 0;
            vt.templ[v] = vt.templ[p];
            vt.route[v] = vt.route[p];
            vt.enterTime[v] = vt.enterTime[p];
            vt.customSpeed[v] = vt.customSpeed[p];
            vt.pendingCustom[v] = 0;
            vt.nextWait[v] = -1;
            vt.state[v] = 1;
            lc.ptype[v] = 2;  // setParent
            lc.partner[v] = p;
            lc.offset[v] = lc.offset[p];
            lc.sigSend[v] = 0;
            lc.sendDir[v] = 0;
            lc.sendUrg[v] = 0;
            lc.lastDir[v] = 0;
            lc.changing[v] = 0;
            lc.lcFinished[v] = 0;
            lc.sendTarget[v] = -1;
            lc.recvFrom[v] = rec.recvFrom;
            lc.tLeader[v] = -1;
            lc.tFollower[v] = -1;
            lc.leaderGap[v] = 0.0;
            lc.followerGap[v] = 0.0;
            lc.lastChangeTime[v] = 0.0;
            lc.gap[v] = rec.gap;  // ControllerInfo is copied; the leader pass (k_action) refreshes it where a leader exists
            lc.ptype[p] = 1;  // setShadow
            lc.partner[p] = v;
            pollOut[2 + rank] = p;
            // ... and its place in the lane
            int before = rec.anchor;  // existing vehicles in front of it, and the shadows that go before it
            for (int q = 0; q < m; ++q)
                if (q != tid) before += (sAnchor[q] < rec.anchor) || (sAnchor[q] == rec.anchor && sSeq[q] < rec.seq);
            const int ns = nbase + before;
            const int route = c.s.route[ps], routePos = c.s.routePos[ps];
            lc.slotOf[v] = ns;
            c.s.vid[ns] = v;
            c.s.drv[ns] = d;
            c.s.prevDrv[ns] = c.s.prevDrv[ps];
            c.s.next[ns] = nextOf(c.n, c.t, d, route, routePos);
            c.s.blocker[ns] = -1;
            c.s.enterLLT[ns] = c.s.enterLLT[ps];
            c.s.routePos[ns] = routePos;
            c.s.templ[ns] = c.s.templ[ps];
            c.s.route[ns] = route;
            c.s.flags[ns] = c.s.flags[ps] & ~kFlagStateGap;  // (insertShadow computes the shadow's leader and gap)
            c.s.dis[ns] = rec.dis;
            c.s.speed[ns] = c.s.speed[ps];
        }
        if (tid == 0) {
            segStartNow[d] = nbase;
            cntNow_[d] = live + m;  // (the admission stays pending behind them: cntNow)
            c.laneTail[d] = nbase + n + m - 1;
        }
    }
    // shadows named provisionally (-(record + 2)) in the walk get their numbers (fixCount is cleared by k_scatter)
    int nFix = *lc.fixCount;
    if (nFix > lc.fixCap) nFix = lc.fixCap;
    for (int i0 = blockIdx.x * 64; i0 < nFix; i0 += gridDim.x * 64) {
        const int i = i0 + tid;
        if (i < nFix) {
            const int who = lc.fixList[3 * i], which = lc.fixList[3 * i + 1], key = lc.insKey[lc.fixList[3 * i + 2]];
            int rank = 0;
            for (int j = 0; j < nIns; ++j) rank += lc.insKey[j] < key;
            if (which) lc.tFollower[who] = lc.firstShadowVid + rank;
            else lc.tLeader[who] = lc.firstShadowVid + rank;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        sc->active += nIns;  // activeVehicleCount++ per shadow
        pollOut[0] = *lc.insCount;  // > insCap tells the host the supply was too small
        pollOut[1] = sc->overflow;  // a capacity of the schedule walk was exceeded: the step is not valid
        __threadfence_system();
    }
}

// The vehicles k_action / k_cross parked (finishAction), in the order of the reference's walk (creation order = ascending
// vid; a shadow is handled inside its real vehicle's turn, engine.cpp:195-205): yield as the tables stand NOW, the rest of
// getNextSpeed, the move; for a changing pair the common speed, the lateral offset and its end (finish), the shadow leaving
// its lane (abort) — engine.cpp:223-244.  The walk's order matters only along explicit dependencies (below).
__device__ inline double lcParkedSpeed(const StepCtx &c, int vid, int s, int turn) {
    const cfx_vehicle_template &t = c.t.templ[c.s.templ[s]];
    const int d = c.s.drv[s];
    const double speed = c.s.speed[s];
    double v = min2(c.lc.bSpeed[vid], lcYieldSpeed(c, vid, speed, t, turn));
    return speedTail(c, t, s, d, speed, c.s.dis[s], c.n.drvLength[d], c.s.next[s], v);
}

// The only thing an item needs from the walk order is: has the changing vehicle whose signal I (or my shadow) received —
// if it comes earlier in the walk — already been handled?  finishAction notes that vehicle when it parks the item
// (LcDev::parkDep).  Items that wait for nobody — nearly all — are independent of one another: k_lc_resolve does them with
// one thread each over the whole device; the few that wait run afterwards in one block, in rounds, an item in the round
// after the one it depends on (chains are short: one changing vehicle signalling the shadow of the next).
__device__ inline void lcResolveItem(const StepCtx &c, const ActionOut &o, const cfx_vehicle_template *tv, int p) {
    const LcDev &lc = c.lc;
    const int s = lc.slotOf[p];
    const int pd = c.s.drv[s];
    const cfx_vehicle_template &tp = tv[c.s.templ[s]];
    if (lc.ptype[p] != 1) {  // a single vehicle that was signalled by an earlier changing vehicle
        const double v = lcParkedSpeed(c, p, s, p);
        commitMove(c, o, s, pd, p, computeMove(c, tp, s, pd, c.s.speed[s], c.s.dis[s], c.n.drvLength[pd], c.s.next[s], v),
                   lc.bBlocker[p], true);
        return;
    }
    const int q = lc.partner[p];
    const int qs = lc.slotOf[q];
    const int qd = c.s.drv[qs];
    const cfx_vehicle_template &tq = tv[c.s.templ[qs]];
    const double ns = min2(lcParkedSpeed(c, p, s, p), lcParkedSpeed(c, q, qs, p));
    MoveOut mp = computeMove(c, tp, s, pd, c.s.speed[s], c.s.dis[s], c.n.drvLength[pd], c.s.next[s], ns);
    MoveOut mq = computeMove(c, tq, qs, qd, c.s.speed[qs], c.s.dis[qs], c.n.drvLength[qd], c.s.next[qs], ns);
    bool pCounted = true;
    // the real vehicle: lateral offset, LaneChange::finishChanging lanechange.cpp:115-127
    if (lc.changing[p]) {
        const int dir = lc.sigSend[p] ? lc.sendDir[p] : 0;
        const double maxOffset = (lc.laneWidth[lc.sendTarget[p]] + lc.laneWidth[pd]) / 2;
        double newOffset = fabs(lc.offset[p] + max2(0.2 * mp.v, 1) * c.interval * dir);
        newOffset = min2(newOffset, maxOffset);
        lc.offset[p] = newOffset * dir;
        if (newOffset >= maxOffset) {
            lc.changing[p] = 0;
            lc.lcFinished[p] = 1;
            lc.lastChangeTime[p] = c.step * c.interval;
            lc.ptype[q] = 0;  // the shadow is the vehicle from now on (the host moves the id with it)
            lc.offset[q] = 0.0;
            lc.partner[q] = -1;
            lc.partner[p] = -1;
            // clearSignal: LATER vehicles of the walk see the signal's neighbours cleared, earlier ones (whose items may
            // run at the same time) must not: lcYieldSpeed decides by lcFinished and the turn
            lc.lastDir[p] = lc.sigSend[p] ? lc.sendDir[p] : 0;
            lc.sigSend[p] = 0;
            lc.recvFrom[p] = -1;
            mp.newDrv = -2;  // Vehicle::finishChanging: setEnd(true)
            pCounted = false;
        }
    }
    // the shadow: leaving the target lane before the change is complete aborts it (vehicle.cpp:412-416)
    if (lc.ptype[q] == 2 && mq.newDrv >= 0) {
        mq.newDrv = -2;  // the shadow ends — and counts as a finished vehicle, as in the reference
        lc.changing[p] = 0;
        lc.ptype[p] = 0;
        lc.offset[p] = 0.0;
        lc.partner[p] = -1;
        lc.tLeader[q] = lc.tFollower[q] = -1;
        lc.lastDir[q] = 0;
        lc.recvFrom[q] = -1;
    }
    commitMove(c, o, s, pd, p, mp, lc.bBlocker[p], pCounted);
    commitMove(c, o, qs, qd, q, mq, lc.bBlocker[q], true);
}

// the items that wait for nobody: one thread each
__global__ void k_lc_resolve(StepCtx c, ActionOut o, int32_t *done /*[slot capacity] scratch*/) {
    const LcDev &lc = c.lc;
    const int n = *lc.parkCount;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (lc.parkDep[i] >= 0) {  // listed for k_lc_resolve_rest (in candAll: the schedule walk is done with it)
            done[i] = 0;
            lc.candAll[atomicAdd(&lc.parkCount[1], 1)] = i;
            continue;
        }
        lcResolveItem(c, o, c.t.templ, lc.parkList[i]);
        done[i] = 1;
    }
}

// ... and the ones that wait, in rounds (round 1 was k_lc_resolve); one block
__global__ void k_lc_resolve_rest(StepCtx c, ActionOut o, int32_t *done) {
    const LcDev &lc = c.lc;
    const int m = lc.parkCount[1];
    __shared__ int sLeft;
    for (int round = 2; m > 0; ++round) {
        if (threadIdx.x == 0) sLeft = 0;
        __syncthreads();
        int left = 0;
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            const int i = lc.candAll[j];
            if (done[i]) continue;
            const int dep = lc.parkIdx[lc.parkDep[i]];  // (a changing vehicle is parked in every step)
            if (!(done[dep] != 0 && done[dep] < round)) {
                left += 1;
                continue;
            }
            lcResolveItem(c, o, c.t.templ, lc.parkList[i]);
            __threadfence_block();
            done[i] = round;
        }
        if (left) atomicAdd(&sLeft, left);
        __syncthreads();
        if (sLeft == 0) break;
        if (round > 4096) {  // cannot happen (dependencies go to strictly smaller vids); never spin on corrupted state
            if (threadIdx.x == 0) o.sc->overflow = 7;
            break;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) lc.parkCount[0] = lc.parkCount[1] = 0;
}

}  // namespace cfxd
