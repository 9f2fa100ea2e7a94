// cfx C ABI (include/cityflow_amd.h) implemented on HIP for gfx950 / MI355X.
// Host-side bookkeeping of the device engine: buffer ownership and growth, launch sequencing, getters.
// The arithmetic lives in cfx_device.h / cfx_kernels.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>


#include "cfx_kernels.h"
#include "cfx_lc_kernels.h"
#include "cfx_ring_kernels.h"
#include "cfx_dense_kernels.h"

// k_cross2's grid on the dense layout: one block per kCross2Jobs slots of the layout's bound, at most this many (the
// kernel strides over the queue, so any grid is correct; blocks beyond the queue's length only pay the prologue)
#ifndef CFX_CROSS2_BLOCKS_PER_CU
#define CFX_CROSS2_BLOCKS_PER_CU 7  // k_cross2 on the dense layout: what its LDS lets a CU hold
#endif

using namespace cfxd;

namespace {

std::string g_createError;

#ifndef CFX_DENSE_FORM_DEFAULT
#define CFX_DENSE_FORM_DEFAULT 6  // what cfx_config::dense_form = 0 means (include/cityflow_amd.h): lanes-only admission, 1024 spawn records in its arguments
#endif
enum ProfKernel { PK_SPAWN = 0, PK_ADMIT, PK_ACTION, PK_CROSS, PK_SCAN, PK_SCATTER, PK_HALO_EXPORT, PK_HALO_IMPORT, PK_COMMIT, kNumProfKernels };
// names of the step's phases; the ring layout runs kr_admit / kr_action / k_cross<.., RingCtx> / kr_commit under the first
// four and the last name (it has no scan / scatter)
const char *const kProfNames[kNumProfKernels] = {"k_spawn_link", "k_admit", "k_action", "k_cross", "k_scan", "k_scatter",
                                                 "k_halo_export", "k_halo_import", "k_commit"};

#define HIP_TRY(call)                                                                                  \
    do {                                                                                               \
        hipError_t err__ = (call);                                                                     \
        if (err__ != hipSuccess) {                                                                     \
            fail(std::string(#call) + ": " + hipGetErrorString(err__));                                \
            return CFX_ERR_DEVICE;                                                                     \
        }                                                                                              \
    } while (0)

template <typename T> struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
};

inline int gridFor(size_t n) { return (int) std::max<size_t>(1, (n + kBlock - 1) / kBlock); }
// grid-stride kernels over slots: enough blocks to fill 256 CUs x 8, never more than needed
constexpr size_t kInitialVidCap = (size_t) 1 << 22;  // vehicle numbers the tables are sized for at creation (46 B each; 100 with lane change)
inline int gridStride(size_t n) { return (int) std::min<size_t>(std::max<size_t>(1, (n + kBlock - 1) / kBlock), 2048); }

}  // namespace

struct cfx_engine {
    int device = 0;
    int nCU = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    hipStream_t stream = nullptr;
    cfx_config cfg{};
    int R = 0, L = 0, K = 0, D = 0, I = 0, E = 0;
    std::string err;

    // ---- static network (device) ----
    DevNet net{};
    std::vector<void *> owned;  // every device allocation, for teardown

    // ---- tables ----
    std::vector<cfx_vehicle_template> hTempl;
    std::vector<int32_t> hRouteStart{0}, hRouteRoads, hNextStart{0}, hNextLL;
    DBuf<cfx_vehicle_template> dTempl;
    DBuf<int32_t> dRouteStart, dRouteRoads, dNextStart, dNextLL;
    bool tablesDirty = false;

    // ---- vehicle table ----
    VidTable vt{};
    LaneHistDev hist{};                // Lane::history, with cfx_config::lane_history (tiles: on the ring layout only)
    size_t vidCap = 0;
    int64_t spawned = 0;

    // ---- slots (two generations) ----
    SlotArrays gen[2]{};
    DBuf<int32_t> segStart[2], cnt[2];
    size_t slotCap = 0;
    int cur = 0;
    ActionBuf ab{};
    CompactScratch cs{};
    int32_t *oldToNew = nullptr;
    int32_t *finList = nullptr, *finTicket = nullptr, *crossJobs = nullptr, *jobCount = nullptr;
    int32_t *finCount = nullptr;  // [kFinShards * 32] counters of the finisher lists (FinMap)
    // getter scratch
    int32_t *viewLeader = nullptr;
    double *viewGap = nullptr;

    // ---- per-lane / per-laneLink / per-entry / per-intersection dynamic state ----
    int32_t *waitHead = nullptr, *admitStep = nullptr, *laneTail = nullptr, *curPhase = nullptr;
    int2 *admitRec = nullptr;
    int4 *llDyn = nullptr;
    int2 *llGate = nullptr;
    unsigned long long *interMask = nullptr;
    int nMaskWords = 0;
    double *remain = nullptr;
    unsigned long long *scanGranules = nullptr;
    int32_t *scanTicket = nullptr;
    int nScanBlocks = 0;
    // per-drivable arrays k_scan reads with 16-byte loads are padded to whole scan tiles
    size_t dPadded() const { return (size_t) ((D + kScanTile - 1) / kScanTile) * kScanTile; }
    int32_t *laneOut = nullptr;
    int32_t *hLaneOut = nullptr;  // pinned landing buffer of the per-lane getters (a D2H copy into pageable memory is staged twice)
    // The observation of a control loop (Engine::getLaneVehicleCount engine.cpp:628-634 read after every step): once the
    // caller has asked for the lane counts, every step's commit writes the counts it changes straight into this pinned host
    // array (kr_commit / k_scan, a few hundred 4-byte stores over PCIe), the step no longer defers its commit to the next
    // admission, and the getter is a wait for the stream plus a host memcpy — no launch, no copy engine.  `hCntValid`: the
    // array equals the device's counts as of the last step enqueued (every commit since it was filled has published);
    // `observing` is dropped again after kObserveIdle steps without a read.
    std::map<int32_t, double> futureCustom;  // cfx_set_vehicle_speed for vehicle numbers the next spawn records will create
    int64_t droppedFutureSpeeds = 0;         // ... of which the next step's records then did not create the vehicle
    std::vector<int32_t> phaseSeen;  // cfx_set_tl_phases: the call that last named each intersection (duplicates: last one wins)
    int32_t phaseCall = 0;
    int32_t *hCnt = nullptr;
    bool hCntValid = false, observing = false;
    int observeIdle = 0;
    static constexpr int kObserveIdle = 8;
    int32_t *publishTo() const { return (observing && hCntValid && !tiled) ? hCnt : nullptr; }
    int cross2 = -1;                // cross phase: 1 = k_cross2 (throughput), 0 = k_cross (latency), -1 = by size
    HostMirror *hMirror = nullptr;  // pinned; valid while the last thing that changed the scalars was a step
    bool mirrorValid = false;
    DevScalars *sc = nullptr;

    cfx_spawn *dRecs = nullptr;
    size_t recCap = 0;
    // pinned staging ring for the per-step spawn records (the caller may reuse `recs` immediately)
    static constexpr int kStages = 4;
    cfx_spawn *hStage[kStages] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stageEvent[kStages] = {nullptr, nullptr, nullptr, nullptr};
    bool stageBusy[kStages] = {false, false, false, false};
    size_t stageCap = 0;
    int stageIdx = 0;
    // ring layout, more spawn records than kr_admit's arguments hold: the batch's columns in pinned memory (SpawnBatchMem)
    int32_t *hMemStage[kStages] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t memStageEvent[kStages] = {nullptr, nullptr, nullptr, nullptr};
    bool memStageBusy[kStages] = {false, false, false, false};
    size_t memStageWords = 0;
    int memStageIdx = 0;
    std::vector<uint8_t> memSeen;
    std::vector<int32_t> memBucket, memCursor, memOrder;
    int32_t *hPhaseStage[kStages] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t phaseStageEvent[kStages] = {nullptr, nullptr, nullptr, nullptr};
    bool phaseStageBusy[kStages] = {false, false, false, false};
    size_t phaseStageCap = 0;
    int phaseStageIdx = 0;

    // ---- tiling (cfx_halo_config) ----
    bool tiled = false;
    bool mailboxesFineGrained = true;  // cfx_halo_mailbox_alloc never had to fall back to a plain allocation
    HaloDev halo{};
    std::vector<uint8_t> hLaneSpare;   // per lane; empty when not tiled
    std::vector<uint8_t> hLaneGhost;   // per lane; empty when not tiled
    int64_t spareTotal = 0;            // sum of spare slots over lanes
    char *dHaloSend = nullptr, *dHaloRecv = nullptr;
    char *hHaloSend = nullptr, *hHaloRecv = nullptr;  // pinned staging
    int haloSendBytes = 0, haloRecvBytes = 0;
    // mailbox exchange (cfx_halo_attach): per peer, device-visible addresses of the two mailboxes
    struct MailPeer {
        int sendBytes = 0, recvBytes = 0;
        void *sendHost = nullptr, *recvHost = nullptr;  // registered with hipHostRegister
        char *sendDev = nullptr, *recvDev = nullptr;
    };
    std::vector<MailPeer> mail;
    std::vector<void *> ipcOpened;     // peers' mailboxes opened with hipIpcOpenMemHandle
    std::vector<int32_t> hGhostSendOff, hGhostRecvOff, hImportSendOff, hImportRecvOff;
    int32_t *haloTicket = nullptr;
    double *finTerm = nullptr;  // [slot] travel times of the step's finishers in summation order
    HaloDev haloMail{};                // block addressing as (peer, offset inside the peer's message)
    uint32_t generation = 1;           // bumped by cfx_reset: epochs stay monotonic
    int64_t liveUpper = 0;             // upper bound of running vehicles (refreshed from the device when it runs out)
    std::vector<uint8_t> laneQueued;   // lanes that have ever had a vehicle queued (only they can admit)
    int64_t nQueueLanes = 0, spawnedHere = 0;  // ... their number; vehicles spawned onto this engine's lanes

    // ---- lane change (cfx_config::lane_change) ----
    LcDev lc{};                        // device tables (vid-indexed ones grow with the vehicle table)
    int32_t *oldToNew2 = nullptr;      // [slot] lane change: scratch of k_lc_resolve (which items are done)
    // dense layout, cfx_config::dense_form: which organisation of the step's kernels (cfx_dense_kernels.h); results never depend on it
    int denseForm = 0;
    bool laneAdmit() const { return (denseForm & 2) != 0 && useTails() && !tiled; }
    int32_t *gatePhase = nullptr;      // [2 I] laneAdmit(): the phase each intersection's gate records stand for (-1: none), by step parity
    int32_t *hPool = nullptr;          // ... pinned staging
    int32_t *hPoll = nullptr;          // pinned: [0] shadows created by the step, [1] overflow code, [2..] their parents in walk order
    hipEvent_t pollEvent = nullptr;    // the part of the step cfx_lane_change_poll has to wait for
    int poolN = 0;                     // priorities supplied for the next / current step
    bool pollPending = false;          // a lane-change step has run and was not polled yet

    // ---- dense layout with tail records (cfx_dense_kernels.h): engines without lane change and tiling ----
    TailRec *dTail[2] = {nullptr, nullptr}, *dTailNow = nullptr;
    int4 *dGate4 = nullptr;
    bool lcSegValid = false;           // lane change: segOfSlot holds every vehicle's own segment (k_scatter / k_lc_naive)
    bool tailsValid = false;           // the records describe the current generation (false after reset / load / resize)
    bool useTails() const { return !ring && !lc.on; }  // (tiles too since round 3: the halo kernels keep the cut lanes' records up)

    // ---- ring layout (cfx_ring_kernels.h): per-drivable ring segments, committed in place ----
    bool ring = false;                 // this engine uses it (decided at cfx_create; cfx_halo_config may still switch to dense)
    bool ringBuilt = false;
    std::vector<double> hDrvLength;    // host copy of cfx_net::drv_length (ring capacities)
    std::vector<int32_t> hLaneRoad, hLaneIndex;  // host copies of cfx_net::lane_road / lane_index (firstNextOf)
    // Router::getNextDrivable(0) of a vehicle waiting on `lane` with route `route`, as VidTable::firstNext keeps it (nextOf +
    // lastRoadBit of cfx_device.h on the host copies of the tables): looked up once per spawn record
    int32_t firstNextOf(int lane, int route) const {
        if (lane < 0 || lane >= L || route < 0 || route + 1 >= (int) hRouteStart.size()) return kFirstNextUnknown;
        const int road = hLaneRoad[lane], base = hRouteStart[route], len = hRouteStart[route + 1] - base;
        int p = 0;
        while (p < len && hRouteRoads[base + p] != road) ++p;
        int next = -1;
        if (p < len) {
            const int ll = hNextLL[hNextStart[base + p] + hLaneIndex[lane]];
            next = ll < 0 ? -1 : L + ll;
        }
        if (next >= 0) return next;
        return (len > 0 && hRouteRoads[base + len - 1] == road) ? -2 : -1;
    }
    std::vector<int2> hRingGeo;        // [D] {base, cap - 1}
    size_t ringSlots = 0;
    double ringMinLen = 0.0;           // shortest vehicle template the capacities were computed for
    int ringScale = 1;                 // doubled when a ring came close to full
    int ringG = 14, ringCalm = 0, ringHold = 0;      // lanes per block of the action kernel (adapted to the traffic) and steps without a dense block
    int2 *dRingGeo = nullptr;
    int32_t *rHead = nullptr, *rCnt = nullptr, *slotOf = nullptr;
    int4 *rScratch = nullptr;
    SlotArrays rs{};                   // per-slot state that changes only when a vehicle enters a drivable
    double2 *rKin[2] = {nullptr, nullptr};  // [slot] {dis, speed}: the two generations a step alternates
    int4 *rMeta = nullptr;                  // [slot] {template, next drivable, flags, enterLaneLinkTime}
    int rcur = 0;
    int2 *rBlk[2] = {nullptr, nullptr};  // by step parity
    TailRec *rTail[2] = {nullptr, nullptr}, *rTailNow = nullptr;  // [D] per-drivable tail records (by step parity; this step's view)
    MoverRec *rMovers = nullptr;
    long long *rFinKey = nullptr;
    int32_t *rFinVid = nullptr;
    double *rFinTerm = nullptr;
    int rFinCap = 0, rJobCap = 0;
    int32_t *rJobs = nullptr;
    RingJob *rJobRecs = nullptr;
    LLAux *rLLAux = nullptr;
    int4 *rLLGate = nullptr;
    int32_t *rGatePhase = nullptr;     // [2 I] RingCtx::gatePhase
    int4 *rList = nullptr;             // the step's vehicle list (kr_index -> kl_action), grown with the running vehicles
    size_t rListCap = 0;
    int32_t *rListCount = nullptr;     // [2] {entries, kr_index's tile ticket}
    unsigned long long *rIdxGranules = nullptr;  // [tiles of 256 drivables] {epoch, vehicles}
    RingDense rd{};                    // dense staging view (getters, archive, growth)
    size_t rdCap = 0;
    int32_t *rOff = nullptr;           // [D + 1] exclusive prefix sum of rCnt

    // Are the interval, every enter time seen so far and the travel-time sum multiples of 2^-10 below 2^42?  Then the
    // finish statistics need no order (exactFinishStatistics).  Sticky false once anything else shows up.
    bool timesDyadic = true;
    static bool dyadic(double x) { return x * 1024.0 == std::nearbyint(x * 1024.0) && std::fabs(x) < 4398046511104.0; }
    bool exactTimes() const {
        if (!timesDyadic) return false;
        // the running sum: as of the last step the device has finished (pinned mirror) it must be far below the bound
        const double cum = mirrorValid ? hMirror->sc.cumulativeTravelTime : cumLoaded;
        return dyadic(cum) && cum < 1099511627776.0;
    }
    double cumLoaded = 0.0;  // cumulativeTravelTime of the last reset / cfx_load_state (valid until a step has run)

    int64_t step = 0;
    int64_t finishedKnown = 0;  // lower bound of finished vehicles (refreshed on syncs)
    int64_t finishedOffset = 0; // finished vehicles that are not in the vid table (state loaded from an archive)

    // ---- optional per-kernel timing (HIP events on this engine's stream) ----
    bool profiling = false;
    std::vector<hipEvent_t> evPool;             // pairs: [2i] before, [2i+1] after
    std::vector<int> evKernel;                  // kernel id of pair i
    size_t evUsed = 0;                          // pairs in use
    double profMs[kNumProfKernels] = {};
    int64_t profLaunches[kNumProfKernels] = {};
    const char *slotSymbol[kNumProfKernels] = {};  // the kernel launched last in each timing slot (cfx_profile_kernel_symbol)
    // ---- where the host's time inside cfx_step goes (cfx_get_host_stats) ----
    cfx_host_stats hostStats{};
    int stallCause = 0;  // CFX_STALL_* bits raised by the running cfx_step call

    // Launch `kernel`; while profiling, with the dispatch's own start / stop timestamps (hipExtLaunchKernel records the
    // events at the kernel's begin and end, the same clock readings a rocprofv3 kernel trace reports — a pair of
    // hipEventRecord calls around the launch would add the marker packets' latency, ~3 us, to every kernel).
    template <typename K, typename... A>
    void launchNamed(int kernelId, const char *symbol, K kernel, dim3 grid, dim3 block, A... args) {
        slotSymbol[kernelId] = symbol;
        launch(kernelId, kernel, grid, block, args...);
    }
    template <typename K, typename... A>
    void launch(int kernelId, K kernel, dim3 grid, dim3 block, A... args) {
        if (profiling) {
            if (evUsed * 2 + 2 > evPool.size()) {
                for (int i = 0; i < 2; ++i) {
                    hipEvent_t ev = nullptr;
                    if (hipEventCreate(&ev) != hipSuccess) break;
                    evPool.push_back(ev);
                }
                evKernel.push_back(0);
            }
            if (evUsed * 2 + 2 <= evPool.size()) {
                const size_t pair = evUsed++;
                evKernel[pair] = kernelId;
                hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, evPool[2 * pair], evPool[2 * pair + 1], 0, args...);
                return;
            }
        }
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
    }
    void profCollect() {
        (void) hipStreamSynchronize(stream);
        for (size_t i = 0; i < evUsed; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, evPool[2 * i], evPool[2 * i + 1]) == hipSuccess) {
                profMs[evKernel[i]] += ms;
                profLaunches[evKernel[i]] += 1;
            }
        }
        evUsed = 0;
    }

    int fail(const std::string &m) {
        err = m;
        return CFX_ERR_DEVICE;
    }

    template <typename T> int allocRaw(T **p, size_t n) {
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
        *p = (T *) q;
        owned.push_back(q);
        return CFX_OK;
    }
    void forget(void *q) { owned.erase(std::remove(owned.begin(), owned.end(), q), owned.end()); }

    // Grow a device array preserving its first `keep` elements.
    template <typename T> int grow(T **p, size_t keep, size_t newCap) {
        T *np = nullptr;
        int rc = allocRaw(&np, newCap);
        if (rc) return rc;
        if (*p && keep) HIP_TRY(hipMemcpyAsync(np, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, stream));
        if (*p) {
            HIP_TRY(hipStreamSynchronize(stream));
            forget(*p);
            HIP_TRY(hipFree(*p));
        }
        *p = np;
        return CFX_OK;
    }
    // ... without draining the stream: the copy is ordered on the stream like everything else, the old array is freed later
    // (retired; hipFree waits for the device) — at the next cfx_sync / reset / destroy.  For the tables that grow with the
    // vehicles ever created: a step that doubles them does not stall (round 4: ten arrays x (malloc + drain + free) = 70 ms).
    std::vector<void *> retired;
    template <typename T> int growDeferred(T **p, size_t keep, size_t newCap) {
        T *np = nullptr;
        int rc = allocRaw(&np, newCap);
        if (rc) return rc;
        if (*p && keep) HIP_TRY(hipMemcpyAsync(np, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, stream));
        if (*p) retired.push_back(*p);  // (still in `owned`: cfx_destroy frees whatever is left)
        *p = np;
        return CFX_OK;
    }
    int freeRetired() {  // only where the stream is known to be idle
        for (void *q : retired) {
            forget(q);
            HIP_TRY(hipFree(q));
        }
        retired.clear();
        return CFX_OK;
    }
    template <typename T> int upload(T **dst, const T *src, size_t n) {
        int rc = allocRaw(dst, n);
        if (rc) return rc;
        if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
        return CFX_OK;
    }

    template <typename T> int uploadConst(const T *&field, const T *src, size_t n) {
        T *tmp = nullptr;
        int rc = upload(&tmp, src, n);
        field = tmp;
        return rc;
    }

    StepCtx ctx() const {
        StepCtx c{};
        c.n = net;
        c.t.templ = dTempl.p;
        c.t.nTempl = (int) hTempl.size();
        c.t.routeStart = dRouteStart.p;
        c.t.routeRoads = dRouteRoads.p;
        c.t.nextStart = dNextStart.p;
        c.t.nextLL = dNextLL.p;
        c.s = gen[cur];
        c.segStart = segStart[cur].p;
        c.cnt = cnt[cur].p;
        c.admitStep = admitStep;
        c.curPhase = curPhase;
        c.oldToNew = oldToNew;
        c.vPriority = vt.priority;
        c.vCustomSpeed = vt.customSpeed;
        c.vGapState = vt.gapState;
        c.llDyn = llDyn;
        c.llGate = llGate;
        c.laneTail = laneTail;
        c.admitRec = admitRec;
        c.interMask = interMask;
        c.step = (int32_t) step;
        c.interval = cfg.interval;
        c.lc = lc;
        if (dTailNow && useTails()) {
            c.tailR = dTail[(step + 1) & 1];  // written by step - 1
            c.tailW = dTail[step & 1];
            c.tailNow = dTailNow;
            c.llGate4 = dGate4;
            c.laneAdmit = laneAdmit() ? 1 : 0;
        }
        return c;
    }

    int ensureVidCap(size_t need) {
        if (need <= vidCap) return CFX_OK;
        stallCause |= CFX_STALL_VID_GROW;
        hostStats.table_grows_total += 1;
        // (cfx_config::ring_capacity_percent below 100 — the tests' "start small" knob — also starts the vehicle tables small,
        // so that their growth path runs)
        const size_t first = (cfg.ring_capacity_percent > 0 && cfg.ring_capacity_percent < 100) ? (size_t) 1 << 12 : kInitialVidCap;
        size_t nc = std::max<size_t>(need, std::max<size_t>(vidCap * 2, first));
        int rc;
        if (ring && (rc = growDeferred(&slotOf, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.priority, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.templ, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.route, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.nextWait, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.enterTime, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.state, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.customSpeed, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.gapState, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.pendingCustom, (size_t) spawned, nc))) return rc;
        if ((rc = growDeferred(&vt.firstNext, (size_t) spawned, nc))) return rc;
        if (lc.on) {
#define GROW_LC(f) if ((rc = growDeferred(&lc.f, (size_t) spawned, nc))) return rc;
            GROW_LC(ptype) GROW_LC(partner) GROW_LC(offset) GROW_LC(sigSend) GROW_LC(sendDir) GROW_LC(sendUrg) GROW_LC(lastDir)
            GROW_LC(changing) GROW_LC(lcFinished) GROW_LC(sendTarget) GROW_LC(recvFrom) GROW_LC(tLeader) GROW_LC(tFollower)
            GROW_LC(leaderGap) GROW_LC(followerGap) GROW_LC(lastChangeTime) GROW_LC(gap) GROW_LC(slotOf) GROW_LC(bSpeed)
            GROW_LC(bBlocker) GROW_LC(parkIdx) GROW_LC(candPos)
#undef GROW_LC
        }
        // nextWait of not-yet-used vids must read -1 (k_spawn_link relies on it)
        HIP_TRY(hipMemsetAsync(vt.nextWait + spawned, 0xFF, (nc - (size_t) spawned) * sizeof(int32_t), stream));
        if (ring) HIP_TRY(hipMemsetAsync(slotOf + spawned, 0xFF, (nc - (size_t) spawned) * sizeof(int32_t), stream));
        vidCap = nc;
        return CFX_OK;
    }

    int ensureSlotCap(size_t need) {
        if (need <= slotCap) return CFX_OK;
        stallCause |= CFX_STALL_SLOT_GROW;
        hostStats.table_grows_total += 1;
        // refresh the live bound first: maybe no growth is needed
        size_t nc = std::max<size_t>(need + need / 2, 1 << 14);
        // the current generation must be preserved; everything else is scratch
        SlotArrays &g = gen[cur];
        size_t keep = slotCap;
        int rc;
#define GROW_KEEP(f) if ((rc = grow(&g.f, keep, nc))) return rc;
        GROW_KEEP(vid) GROW_KEEP(drv) GROW_KEEP(prevDrv) GROW_KEEP(next) GROW_KEEP(blocker) GROW_KEEP(enterLLT) GROW_KEEP(routePos)
        GROW_KEEP(templ) GROW_KEEP(route) GROW_KEEP(flags) GROW_KEEP(dis) GROW_KEEP(speed)
#undef GROW_KEEP
        SlotArrays &o = gen[cur ^ 1];
#define GROW_SCRATCH(ptr) if ((rc = grow(&ptr, 0, nc))) return rc;
        GROW_SCRATCH(o.vid) GROW_SCRATCH(o.drv) GROW_SCRATCH(o.prevDrv) GROW_SCRATCH(o.next) GROW_SCRATCH(o.blocker) GROW_SCRATCH(o.enterLLT)
        GROW_SCRATCH(o.routePos) GROW_SCRATCH(o.templ) GROW_SCRATCH(o.route) GROW_SCRATCH(o.flags) GROW_SCRATCH(o.dis) GROW_SCRATCH(o.speed)
        GROW_SCRATCH(ab.dis) GROW_SCRATCH(ab.speed) GROW_SCRATCH(ab.drv) GROW_SCRATCH(ab.blocker)
        GROW_SCRATCH(cs.inNext) GROW_SCRATCH(finList) GROW_SCRATCH(finTerm) GROW_SCRATCH(viewLeader) GROW_SCRATCH(viewGap)
        if ((rc = grow(&crossJobs, 0, nc * kJobShards))) return rc;
#undef GROW_SCRATCH
        if ((rc = grow(&oldToNew, keep, nc))) return rc;  // committed blockers point through it
        if (lc.on && (rc = grow(&oldToNew2, 0, nc))) return rc;
        if (lc.on && (rc = grow(&lc.newToOld, keep, nc))) return rc;
        if (lc.on && (rc = grow(&lc.parkList, 0, nc))) return rc;
        if (lc.on && (rc = grow(&lc.parkDep, 0, nc))) return rc;
        if (lc.on && (rc = grow(&lc.candAll, 0, nc))) return rc;
        if (lc.on && (rc = grow(&lc.candAllEnv, 0, nc))) return rc;
        if (lc.on && (rc = grow(&lc.segOfSlot, keep, nc))) return rc;  // (k_scatter leaves the next step's input in it)
        slotCap = nc;
        return CFX_OK;
    }

    int syncTables() {
        if (!tablesDirty) return CFX_OK;
        auto up = [this](auto &dbuf, const auto &h) -> int {
            using T = typename std::remove_reference<decltype(*dbuf.p)>::type;
            if (h.size() > dbuf.cap) {
                size_t nc = std::max<size_t>(h.size() * 2, 64);
                int rc = grow(&dbuf.p, 0, nc);
                if (rc) return rc;
                dbuf.cap = nc;
            }
            if (!h.empty()) HIP_TRY(hipMemcpyAsync(dbuf.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
            return CFX_OK;
        };
        // kernels in flight may still read the old buffers: drain before (re)uploading
        stallCause |= CFX_STALL_TABLES;
        HIP_TRY(hipStreamSynchronize(stream));
        int rc;
        if ((rc = up(dTempl, hTempl))) return rc;
        if ((rc = up(dRouteStart, hRouteStart))) return rc;
        if ((rc = up(dRouteRoads, hRouteRoads))) return rc;
        if ((rc = up(dNextStart, hNextStart))) return rc;
        if ((rc = up(dNextLL, hNextLL))) return rc;
        HIP_TRY(hipStreamSynchronize(stream));  // sources are pageable host vectors
        tablesDirty = false;
        return CFX_OK;
    }

    int readScalars(DevScalars &out) {
        stallCause |= CFX_STALL_SCALARS;
        if (mirrorValid) {
            HIP_TRY(hipStreamSynchronize(stream));
            out = hMirror->sc;
        } else {
            HIP_TRY(hipMemcpyAsync(&out, sc, sizeof(DevScalars), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        finishedKnown = out.finishedCnt;
        if (out.overflow == 4) return fail("halo: a neighbour tile did not publish its step in time (cfx_halo_wait)");
        if (out.overflow == 3) return fail("halo: more vehicles crossed one cut lane in one step than CFX_HALO_MAX_MIGRANTS");
        if (out.overflow == 5) return fail("lane change: more shadows in one step than priorities supplied (cfx_lane_change_supply)");
        if (out.overflow == 6) return fail("lane change: more shadows on one road in one step than the schedule walk tracks");
        if (out.overflow == 7) return fail("lane change: inconsistent pair state (k_lc_resolve did not converge)");
        if (out.overflow == 8) return fail("ring layout: a drivable's ring of slots is full");
        if (out.overflow == 9) return fail("cross phase: job queue capacity exceeded");
        if (out.overflow == 10) return fail("action phase: more slots in use than the host's bound (internal error)");
        if (out.overflow == 11) return fail("lane change: no room behind the layout for the lanes that got shadows in one step");
        if (out.overflow == 12) return fail("lane change: more shadows on one lane in one step than k_lc_insert places (kLcRoadInserts)");
        if (out.overflow) return fail("device capacity overflow (finish list)");
        return CFX_OK;
    }


    // ------------------------------------------------------------------------------------------ ring layout
    // `atStep` / `atRcur`: the context of an EARLIER step (the deferred commit of step - 1, see settle)
    RingCtx rctx(bool stepping = false, int64_t atStep = -1, int atRcur = -1) const {
        const int64_t step = atStep >= 0 ? atStep : this->step;
        const int rcur = atRcur >= 0 ? atRcur : this->rcur;
        RingCtx c{};
        c.tailR = rTail[(step + 1) & 1];
        c.tailW = rTail[step & 1];
        c.tailNow = rTailNow;
        c.betweenSteps = stepping ? 0 : 1;
        c.n = net;
        c.t.templ = dTempl.p;
        c.t.nTempl = (int) hTempl.size();
        c.t.routeStart = dRouteStart.p;
        c.t.routeRoads = dRouteRoads.p;
        c.t.nextStart = dNextStart.p;
        c.t.nextLL = dNextLL.p;
        c.s = rs;
        c.kin = rKin[rcur];
        c.kinN = rKin[rcur ^ 1];
        c.meta = rMeta;
        c.blkR = rBlk[(step + 1) & 1];  // written by step - 1
        c.blkW = rBlk[step & 1];
        c.slotOf = slotOf;
        c.vState = vt.state;
        c.ringGeo = dRingGeo;
        c.head = rHead;
        c.cnt = rCnt;
        c.admitStep = admitStep;
        c.curPhase = curPhase;
        c.vPriority = vt.priority;
        c.vCustomSpeed = vt.customSpeed;
        c.vGapState = vt.gapState;
        c.llDyn = llDyn;
        c.interMask = interMask;
        c.llGate = rLLGate;
        c.gatePhase = rGatePhase;
        c.llAux = rLLAux;
        c.laneTail = laneTail;
        c.admitRec = admitRec;
        c.step = (int32_t) step;
        c.interval = cfg.interval;
        return c;
    }
    template <typename T> int freeRaw(T **p) {
        if (*p) {
            forget(*p);
            HIP_TRY(hipFree(*p));
            *p = nullptr;
        }
        return CFX_OK;
    }
    int ringFree() {
        int rc = 0;
        rc |= freeRaw(&rs.vid) | freeRaw(&rs.drv) | freeRaw(&rs.prevDrv) | freeRaw(&rs.routePos) | freeRaw(&rs.route) |
              freeRaw(&rBlk[0]) | freeRaw(&rBlk[1]) | freeRaw(&rKin[0]) | freeRaw(&rKin[1]) | freeRaw(&rMeta) | freeRaw(&rMovers) |
              freeRaw(&dRingGeo) | freeRaw(&rJobs) | freeRaw(&rJobRecs);
        return rc ? CFX_ERR_DEVICE : CFX_OK;
    }
    // Ring capacities: a drivable of length len holds at most ~len / (shortest vehicle) vehicles bumper to bumper; a few
    // more for the transient overlaps the reference allows (Lane::canEnter lets a vehicle in behind a moving tail,
    // roadnet.cpp:437-445), rounded up to a power of two.  kr_commit raises a flag well before a ring is full and the next
    // cfx_step doubles every capacity; a full ring is an error (DevScalars::overflow = 8), never silent.
    int ringAllocate(double minLen) {
        hRingGeo.resize((size_t) D);
        size_t run = 0;
        for (int d = 0; d < D; ++d) {
            double want = std::ceil(hDrvLength[d] / minLen) + 6.0;
            if (cfg.ring_capacity_percent > 0) want = std::max(2.0, want * cfg.ring_capacity_percent / 100.0);
            want *= ringScale;
            size_t cap = 8;
            while ((double) cap < want && cap < (1u << (kRingIdxBits - 1))) cap <<= 1;
            run = (run + cap - 1) & ~(cap - 1);
            hRingGeo[d] = make_int2((int) run, (int) cap - 1);
            run += cap;
            if (run > 0x7fff0000u) return fail("ring layout: more than 2^31 slots");
        }
        ringSlots = run;
        ringMinLen = minLen;
        int rc;
#define RALLOC(ptr) if ((rc = allocRaw(&ptr, ringSlots))) return rc;
        RALLOC(rs.vid) RALLOC(rs.drv) RALLOC(rs.prevDrv) RALLOC(rs.routePos) RALLOC(rs.route)
        RALLOC(rBlk[0]) RALLOC(rBlk[1]) RALLOC(rKin[0]) RALLOC(rKin[1]) RALLOC(rMeta) RALLOC(rMovers)
#undef RALLOC
        HIP_TRY(hipMemsetAsync(rBlk[0], 0xFF, ringSlots * sizeof(int2), stream));
        HIP_TRY(hipMemsetAsync(rBlk[1], 0xFF, ringSlots * sizeof(int2), stream));
        HIP_TRY(hipMemsetAsync(rMeta, 0, ringSlots * sizeof(int4), stream));
        if ((rc = upload(&dRingGeo, hRingGeo.data(), hRingGeo.size()))) return rc;
        rJobCap = (int) std::max<size_t>(4096, ringSlots / 8);
        if ((rc = allocRaw(&rJobs, (size_t) rJobCap * kJobShards))) return rc;
        if ((rc = allocRaw(&rJobRecs, (size_t) rJobCap * kJobShards))) return rc;
        if (!rHead) {
            if ((rc = allocRaw(&rHead, (size_t) D + 1))) return rc;
            if ((rc = allocRaw(&rCnt, (size_t) D + 1))) return rc;
            if ((rc = allocRaw(&rOff, (size_t) D + 2))) return rc;
            if ((rc = allocRaw(&rScratch, (size_t) D))) return rc;
            if ((rc = allocRaw(&rTail[0], (size_t) D))) return rc;
            if ((rc = allocRaw(&rTail[1], (size_t) D))) return rc;
            if ((rc = allocRaw(&rTailNow, (size_t) D))) return rc;
            if ((rc = allocRaw(&rLLAux, (size_t) K))) return rc;
            if ((rc = allocRaw(&rLLGate, (size_t) K))) return rc;
            if ((rc = allocRaw(&rGatePhase, (size_t) 2 * std::max(I, 1)))) return rc;
            HIP_TRY(hipMemsetAsync(rGatePhase, 0xFF, (size_t) 2 * std::max(I, 1) * sizeof(int32_t), stream));
            rFinCap = std::max(1 << 16, L * 8);
            if ((rc = allocRaw(&rFinKey, (size_t) rFinCap))) return rc;
            if ((rc = allocRaw(&rFinVid, (size_t) rFinCap))) return rc;
            if ((rc = allocRaw(&rFinTerm, (size_t) rFinCap))) return rc;
            HIP_TRY(hipMemsetAsync(rCnt, 0, ((size_t) D + 1) * sizeof(int32_t), stream));
        }
        return CFX_OK;
    }
    int ensureDense(size_t n) {
        if (n <= rdCap) return CFX_OK;
        const size_t nc = std::max<size_t>(n + n / 4, 1 << 14);
        int rc;
        HIP_TRY(hipStreamSynchronize(stream));
#define DGROW(ptr) if ((rc = freeRaw(&ptr)) || (rc = allocRaw(&ptr, nc))) return rc;
        DGROW(rd.vid) DGROW(rd.drv) DGROW(rd.prevDrv) DGROW(rd.blockerVid) DGROW(rd.enterLLT) DGROW(rd.routePos) DGROW(rd.leaderVid)
        DGROW(rd.flags) DGROW(rd.dis) DGROW(rd.speed) DGROW(rd.gap)
#undef DGROW
        rdCap = nc;
        return CFX_OK;
    }
    // ring order -> dense staging arrays (Drivable::vehicles order); returns the number of running vehicles
    int ringGather(bool wantLeader, int32_t *totalOut) {
        hipLaunchKernelGGL(kr_offsets, dim3(1), dim3(kOffBlock), 0, stream, (const int32_t *) rCnt, rOff, D + 1);
        int32_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, rOff + D, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int rc = ensureDense((size_t) total);
        if (rc) return rc;
        if (total) hipLaunchKernelGGL(kr_gather, dim3(gridFor(D)), dim3(kBlock), 0, stream, rctx(), (const int32_t *) rOff, rd, wantLeader ? 1 : 0);
        HIP_TRY(hipGetLastError());
        *totalOut = total;
        return CFX_OK;
    }
    // Records that are valid "for the step in their tag" must not survive a change of the step counter (reset, load) or of
    // the slots they name (rebuilt rings); the occupancy bits are rebuilt by kr_scatter_in.
    int ringClearStepTags() {
        HIP_TRY(hipMemsetAsync(rTailNow, 0xFF, (size_t) D * sizeof(TailRec), stream));
        HIP_TRY(hipMemsetAsync(interMask, 0, (size_t) std::max(nMaskWords, 1) * sizeof(unsigned long long), stream));
        return CFX_OK;
    }
    // ---- the commit of a step may ride with the NEXT step's admission (kr_admit<true>: one launch less per step).  Until
    //      then it is pending; anything else that looks at the state launches it on its own first.
    bool ringMerge = true;       // cfx_config::ring_lanes_per_wave / 10000 == 4 turns the deferral off (developer knob)
    bool commitPending = false;  // the commit of step - 1 has not been launched yet (its lights were advanced by kr_cross)
    size_t activeEstimate() const {
        const unsigned long long pr = __atomic_load_n(&hMirror->progress, __ATOMIC_RELAXED);
        return (size_t) (pr & 0xFFFFFFFFu) + (size_t) nQueueLanes * 4;
    }
    RingCommit commitArgs(size_t activeEst, bool lightsDone, int *nStatOut) {
        const int nStat = (int) std::min<size_t>(std::max<size_t>(1, activeEst >> 16), 64);
        *nStatOut = nStat;
        return RingCommit{rScratch, rMovers, waitHead, curPhase, remain, (int) cfg.rl_traffic_light, (int) nMaskWords,
                          sc, rFinKey, rFinVid, rFinTerm, rFinCap, jobCount,
                          // (a tile's `active` is final only after the halo import; its mirror serves as the stale estimate that
                          // sizes the next steps' grids — mirrorValid stays false, nothing else reads it)
                          hMirror, finTicket, nStat, vt.state, slotOf, exactTimes() ? 1 : 0,
                          lightsDone ? 1 : 0, publishTo(), finCount};
    }
    // tiling on the rings, mailbox transports: cfx_halo_wait leaves the import to the next step's admission launch (one launch
    // less per tile-step); anything else that looks at the state first sends it out as a kernel of its own
    bool haloImportPending = false;
    RingHaloIn pendingImport{};
    // Lane::history on the rings: the record of a step is taken by trailing blocks of the NEXT step's action launch (RingHist,
    // cfx_ring_kernels.h); `histPending`: the last submitted step's has not been.  settle() sends it out as a launch of its own —
    // except for the callers that only read what the history does not touch (the per-step getters of an agent's loop).
    bool histPending = false;
    RingHist takeHist(int firstBlock) {
        RingHist rh{};
        rh.firstBlock = 0x7fffffff;
        if (hist.num && ring && histPending) {  // (tiles too: the step's halo import runs in front of this launch)
            rh.h = hist;
            rh.firstBlock = firstBlock;
            histPending = false;
        }
        return rh;
    }
    int settle(bool withHistory = true) {
        if (haloImportPending) {
            haloImportPending = false;
            const int n = pendingImport.h.nGhost + pendingImport.h.nImport;
            launchNamed(PK_HALO_IMPORT, "kr_halo_import", kr_halo_import, dim3(gridFor(n)), dim3(kBlock), rctx(), pendingImport.h, pendingImport.io, vt,
                        sc, dHaloActiveOut);
            HIP_TRY(hipGetLastError());
        }
        if (commitPending) {
            commitPending = false;
            int nStat = 1;
            const RingCommit rk = commitArgs(activeEstimate(), true, &nStat);
            launchNamed(PK_COMMIT, "kr_commit", kr_commit, dim3(gridStride((size_t) std::max(D, std::max(I, nMaskWords))) + nStat), dim3(kBlock),
                   rctx(true, step - 1, rcur ^ 1), rk, vt, RingHalo{});
            HIP_TRY(hipGetLastError());
        }
        if (withHistory && histPending) {
            const RingHist rh = takeHist(0);
            if (rh.h.num) {
                hipLaunchKernelGGL(kr_lane_history, dim3(gridFor(L)), dim3(kBlock), 0, stream, rctx(), rh.h);
                HIP_TRY(hipGetLastError());
            }
            histPending = false;
        }
        return CFX_OK;
    }
    // (Re)build the rings: first use, a shorter vehicle template than the capacities were computed for, or growth.
    // ---- tiling on the rings: the export is part of the step's commit (RingHalo), the import a kernel of its own
    int32_t *dCutIndex = nullptr;     // [L] -1 / ghost lane i / nGhost + import lane j
    long long *dHaloActiveOut = nullptr;
    // the halo argument of the commit launched by the step that is being submitted (epoch = the step's number + 1: what
    // cfx_halo_post / cfx_halo_wait, called after cfx_step has returned, compute from the advanced step counter)
    RingHalo ringHalo() const {
        RingHalo rh{};
        if (!tiled || !ring) return rh;
        rh.on = 1;
        rh.cutIndex = dCutIndex;
        rh.activeOut = dHaloActiveOut;
        const unsigned long long epoch = ((unsigned long long) generation << 32) | (unsigned long long) (uint32_t) (step + 1);
        if (!mail.empty()) {
            rh.h = haloMail;
            const int par = (int) (epoch & 1ULL);
            for (size_t p = 0; p < mail.size(); ++p) {
                rh.io.send[p] = mail[p].sendDev + CFX_HALO_MAILBOX_HEADER + (size_t) par * mail[p].sendBytes;
                rh.io.signalFlag[p] = (unsigned long long *) mail[p].sendDev;
            }
            rh.io.nSignal = (int) mail.size();
            rh.io.ticket = haloTicket;
        } else {
            rh.h = halo;
            rh.io.send[0] = dHaloSend;  // the staged path: cfx_halo_export copies it out
        }
        rh.io.epoch = epoch;
        return rh;
    }
    bool ringGrowRequested = false;
    int ringEnsure() {
        double minLen = 1e300;
        for (const auto &t : hTempl) minLen = std::min(minLen, t.len);
        if (hTempl.empty()) minLen = 5.0;
        minLen = std::max(minLen, 0.25);
        if (ringBuilt && !(minLen < ringMinLen) && !ringGrowRequested) return CFX_OK;
        int rc;
        if ((rc = settle())) return rc;
        int32_t total = 0;
        if (ringBuilt) {  // carry the running vehicles over
            stallCause |= CFX_STALL_RING_REGROW;
            hostStats.ring_regrows_total += 1;
            if ((rc = ringGather(false, &total))) return rc;
            HIP_TRY(hipStreamSynchronize(stream));
            if ((rc = ringFree())) return rc;
            if (ringGrowRequested) ringScale *= 2;
            minLen = std::min(minLen, ringMinLen);
        }
        ringGrowRequested = false;
        if ((rc = ringAllocate(minLen))) return rc;
        ringBuilt = true;
        hipLaunchKernelGGL(kr_reset, dim3(gridFor(D)), dim3(kBlock), 0, stream, D, rHead, rCnt, rScratch);
        HIP_TRY(hipMemsetAsync(rTail[0], 0xFF, (size_t) D * sizeof(TailRec), stream));  // slot -1: empty
        HIP_TRY(hipMemsetAsync(rTail[1], 0xFF, (size_t) D * sizeof(TailRec), stream));
        if ((rc = ringClearStepTags())) return rc;
        if (total) {
            hipLaunchKernelGGL(kr_scatter_in, dim3(gridFor(D)), dim3(kBlock), 0, stream, rctx(), (const int32_t *) rOff, rd, vt);
            const int zero = 0;
            HIP_TRY(hipMemcpyAsync(&sc->ringNearFull, &zero, sizeof(int), hipMemcpyHostToDevice, stream));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        return CFX_OK;
    }

    int resetState() {
        // Lane::history outlives Engine::reset (Lane::reset roadnet.cpp:832-835) and a load that does not carry it: the last
        // step's record — and the commit it reads — first
        if (histPending && ringBuilt) {
            const int rc = settle();
            if (rc) return rc;
        }
        commitPending = false;  // (whatever a deferred commit would have written is overwritten below)
        haloImportPending = false;
        histPending = false;
        HIP_TRY(hipStreamSynchronize(stream));
        if (!retired.empty() && freeRetired()) return CFX_ERR_DEVICE;
        mirrorValid = false;
        hCntValid = false;
        futureCustom.clear();
        tailsValid = false;
        if (gatePhase) HIP_TRY(hipMemsetAsync(gatePhase, 0xFF, (size_t) 2 * std::max(I, 1) * sizeof(int32_t), stream));
        if (rGatePhase) HIP_TRY(hipMemsetAsync(rGatePhase, 0xFF, (size_t) 2 * std::max(I, 1) * sizeof(int32_t), stream));
        lcSegValid = false;
        if (hMirror) hMirror->progress = 0;  // the stream is idle: nothing is writing it
        pollPending = false;
        poolN = 0;
        if (lc.on) {
            HIP_TRY(hipMemsetAsync(lc.roadCand, 0, (size_t) std::max(R, 1) * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.insHead, 0xFF, (size_t) std::max(L, 1) * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.parkCount, 0, 4 * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.fixCount, 0, sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.insCount, 0, sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.candAllCount, 0, sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(lc.insLaneCount, 0, sizeof(int32_t), stream));
        }
        timesDyadic = dyadic(cfg.interval);
        cumLoaded = 0.0;
        laneQueued.assign((size_t) L, 0);
        nQueueLanes = 0;
        spawnedHere = 0;
        liveUpper = 0;
        generation += 1;
        cur = 0;
        step = 0;
        spawned = 0;
        finishedKnown = 0;
        finishedOffset = 0;
        if (!tiled) {
            hipLaunchKernelGGL(k_init_layout, dim3(gridFor(D + 1)), dim3(kBlock), 0, stream, D, L, segStart[0].p, cnt[0].p,
                               gen[0].vid, gen[0].drv);
        } else {  // lanes own laneSpare[l] empty slots each
            std::vector<int32_t> ss((size_t) D + 1);
            int32_t run = 0;
            for (int d = 0; d <= D; ++d) {
                ss[d] = run;
                if (d < L) run += hLaneSpare[d];
            }
            HIP_TRY(hipMemcpyAsync(segStart[0].p, ss.data(), ss.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemsetAsync(cnt[0].p, 0, (size_t) D * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(gen[0].vid, 0xFF, (size_t) run * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(gen[0].drv, 0xFF, (size_t) run * sizeof(int32_t), stream));
            HIP_TRY(hipMemsetAsync(gen[0].blocker, 0xFF, (size_t) run * sizeof(int32_t), stream));
            HIP_TRY(hipStreamSynchronize(stream));  // ss lives on this frame
            liveUpper = 2 * (int64_t) halo.nGhost;
        }
        if (ring && ringBuilt) {
            hipLaunchKernelGGL(kr_reset, dim3(gridFor(D)), dim3(kBlock), 0, stream, D, rHead, rCnt, rScratch);
            HIP_TRY(hipMemsetAsync(rTail[0], 0xFF, (size_t) D * sizeof(TailRec), stream));
            HIP_TRY(hipMemsetAsync(rTail[1], 0xFF, (size_t) D * sizeof(TailRec), stream));
            {
                const int rcClear = ringClearStepTags();
                if (rcClear) return rcClear;
            }
            // blocker records carry step numbers, which start over
            HIP_TRY(hipMemsetAsync(rBlk[0], 0xFF, ringSlots * sizeof(int2), stream));
            HIP_TRY(hipMemsetAsync(rBlk[1], 0xFF, ringSlots * sizeof(int2), stream));
            rcur = 0;
        }
        hipLaunchKernelGGL(k_init_lights, dim3(gridFor(I)), dim3(kBlock), 0, stream, net, curPhase, remain);
        HIP_TRY(hipMemsetAsync(waitHead, 0xFF, L * sizeof(int32_t), stream));
        HIP_TRY(hipMemsetAsync(admitStep, 0xFF, dPadded() * sizeof(int32_t), stream));
        HIP_TRY(hipMemsetAsync(interMask, 0, std::max(nMaskWords, 1) * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(sc, 0, sizeof(DevScalars), stream));
        HIP_TRY(hipMemsetAsync(scanGranules, 0, (size_t) nScanBlocks * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(scanTicket, 0, sizeof(int32_t), stream));
        if (rIdxGranules) {  // (the epochs are step numbers, which start over)
            HIP_TRY(hipMemsetAsync(rIdxGranules, 0, (size_t) gridFor(D) * sizeof(unsigned long long), stream));
            HIP_TRY(hipMemsetAsync(rListCount, 0, 2 * sizeof(int32_t), stream));
        }
        HIP_TRY(hipMemsetAsync(jobCount, 0, (size_t) kJobShards * kJobShardStride * sizeof(int32_t), stream));
        if (finCount) HIP_TRY(hipMemsetAsync(finCount, 0, (size_t) kFinShards * 32 * sizeof(int32_t), stream));
        if (finTicket) HIP_TRY(hipMemsetAsync(finTicket, 0, 4 * sizeof(int32_t), stream));
        if (dHaloActiveOut) HIP_TRY(hipMemsetAsync(dHaloActiveOut, 0, sizeof(long long), stream));
        HIP_TRY(hipMemsetAsync(oldToNew, 0xFF, slotCap * sizeof(int32_t), stream));
        if (vidCap) HIP_TRY(hipMemsetAsync(vt.nextWait, 0xFF, vidCap * sizeof(int32_t), stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        return CFX_OK;
    }
};

// =============================================================================================== C ABI
extern "C" {

int32_t cfx_abi_version(void) { return CFX_ABI_VERSION; }
const char *cfx_backend_name(void) { return "hip-gfx950"; }
const char *cfx_last_error(const cfx_engine *e) { return e ? e->err.c_str() : g_createError.c_str(); }

void cfx_destroy(cfx_engine *e) {
    if (!e) return;
    (void) hipSetDevice(e->device);
    if (e->stream) (void) hipStreamSynchronize(e->stream);
    for (void *p : e->owned) (void) hipFree(p);
    for (hipEvent_t ev : e->evPool) (void) hipEventDestroy(ev);
    for (int i = 0; i < cfx_engine::kStages; ++i) {
        if (e->hStage[i]) (void) hipHostFree(e->hStage[i]);
        if (e->hMemStage[i]) (void) hipHostFree(e->hMemStage[i]);
        if (e->memStageEvent[i]) (void) hipEventDestroy(e->memStageEvent[i]);
        if (e->stageEvent[i]) (void) hipEventDestroy(e->stageEvent[i]);
        if (e->hPhaseStage[i]) (void) hipHostFree(e->hPhaseStage[i]);
        if (e->phaseStageEvent[i]) (void) hipEventDestroy(e->phaseStageEvent[i]);
    }
    for (auto &m : e->mail) {
        if (m.sendHost) (void) hipHostUnregister(m.sendHost);
        if (m.recvHost) (void) hipHostUnregister(m.recvHost);
    }
    for (void *p : e->ipcOpened) (void) hipIpcCloseMemHandle(p);
    if (e->hLaneOut) (void) hipHostFree(e->hLaneOut);
    if (e->hCnt) (void) hipHostFree(e->hCnt);
    if (e->hMirror) (void) hipHostFree(e->hMirror);
    if (e->hHaloSend) (void) hipHostFree(e->hHaloSend);
    if (e->hHaloRecv) (void) hipHostFree(e->hHaloRecv);
    if (e->hPool) (void) hipHostFree(e->hPool);
    if (e->hPoll) (void) hipHostFree(e->hPoll);
    if (e->pollEvent) (void) hipEventDestroy(e->pollEvent);
    if (e->stream) (void) hipStreamDestroy(e->stream);
    delete e;
}

static int32_t createImpl(cfx_engine *e, const cfx_net *n, const cfx_config *cfg) {
    auto fail = [e](const std::string &m) { return e->fail(m); };
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return e->fail("no HIP device visible");
    e->device = cfg->device % ndev;
    HIP_TRY(hipSetDevice(e->device));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0) e->nCU = cus;
    }
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->cfg = *cfg;
    e->cross2 = cfg->cross_mode == CFX_CROSS_THROUGHPUT ? 1 : cfg->cross_mode == CFX_CROSS_LATENCY ? 0 : -1;
    if (cfg->layout == CFX_LAYOUT_RING && cfg->lane_change)
        return e->fail("cfx_create: layout ring does not run lane change (its mid-lane insertions use the dense layout)");
    // auto: the ring layout wherever it runs (not lane change; a tiled engine goes back to the dense layout in
    // cfx_halo_config).  It commits a step with a fraction of the dense layout's data movement; until round 5 its action
    // phase walked the lanes, which a large, sparsely filled network made slower than the dense layout (100x100: 227 vs 191
    // us) — with the list form of the action phase (kr_index + kl_action) it is ahead at every size measured: 100x100 with
    // 72 k / 277 k / 970 k running vehicles 54.7 / 86 / 141 us per step against 69.4 / 100 / 157.6 on the dense layout.
    e->ring = cfg->layout == CFX_LAYOUT_RING || (cfg->layout == CFX_LAYOUT_AUTO && !cfg->lane_change);
    e->ringMerge = (cfg->ring_lanes_per_wave / 10000) % 10 != 4;
    e->denseForm = cfg->dense_form ? (cfg->dense_form & 255) : CFX_DENSE_FORM_DEFAULT;
    e->hDrvLength.assign(n->drv_length, n->drv_length + n->n_lanes + n->n_lanelinks);
    e->hLaneRoad.assign(n->lane_road, n->lane_road + n->n_lanes);
    e->hLaneIndex.assign(n->lane_index, n->lane_index + n->n_lanes);
    e->timesDyadic = cfx_engine::dyadic(cfg->interval);
    e->R = n->n_roads;
    e->L = n->n_lanes;
    e->K = n->n_lanelinks;
    e->D = e->L + e->K;
    e->I = n->n_inters;
    e->E = n->n_xentries;
    DevNet &d = e->net;
    d.R = e->R;
    d.L = e->L;
    d.K = e->K;
    d.I = e->I;
    d.E = e->E;
    int rc;
#define UP(field, src, count) \
    if ((rc = e->uploadConst(d.field, src, (size_t) (count)))) return rc;
    UP(drvLength, n->drv_length, e->D)
    UP(drvMaxSpeed, n->drv_max_speed, e->D)
    UP(laneRoad, n->lane_road, e->L)
    UP(laneIndex, n->lane_index, e->L)
    UP(laneLLStart, n->lane_ll_start, e->L + 1)
    UP(laneLL, n->lane_ll, e->K)
    UP(llStartLane, n->ll_start_lane, e->K)
    UP(llEndLane, n->ll_end_lane, e->K)
    UP(llInter, n->ll_inter, e->K)
    UP(llRoadLink, n->ll_roadlink, e->K)
    UP(llType, n->ll_type, e->K)
    UP(llXStart, n->ll_x_start, e->K + 1)
    UP(xDist, n->x_dist, e->E)
    UP(xPeer, n->x_peer, e->E)
    UP(xLL, n->x_ll, e->E)
    UP(interVirtual, n->inter_virtual, e->I)
    UP(interNRL, n->inter_n_roadlinks, e->I)
    UP(interPhaseStart, n->inter_phase_start, e->I + 1)
    UP(interAvailStart, n->inter_avail_start, e->I)
    UP(phaseTime, n->phase_time, n->n_phases)
    UP(phaseAvail, n->phase_avail, n->n_avail)
#undef UP
    const size_t dPad = e->dPadded();
    for (int g = 0; g < 2; ++g) {
        if ((rc = e->allocRaw(&e->segStart[g].p, dPad + 8))) return rc;  // k_scan writes whole 8-entry groups
        if ((rc = e->allocRaw(&e->cnt[g].p, dPad))) return rc;
        HIP_TRY(hipMemset(e->cnt[g].p, 0, dPad * sizeof(int32_t)));
    }
    if ((rc = e->allocRaw(&e->cs.leaveCnt, dPad))) return rc;
    HIP_TRY(hipMemset(e->cs.leaveCnt, 0, dPad * sizeof(int32_t)));
    if ((rc = e->allocRaw(&e->cs.maxLeaveIdx, (size_t) e->D))) return rc;
    if ((rc = e->allocRaw(&e->cs.inCnt, dPad))) return rc;
    HIP_TRY(hipMemset(e->cs.inCnt, 0, dPad * sizeof(int32_t)));
    if ((rc = e->allocRaw(&e->cs.inHead, (size_t) e->D))) return rc;
    if ((rc = e->allocRaw(&e->waitHead, (size_t) e->L))) return rc;
    if ((rc = e->allocRaw(&e->admitStep, dPad))) return rc;  // lanes only are ever set; the rest stays -1 (k_scan reads 8 at a time)
    if ((rc = e->allocRaw(&e->laneOut, (size_t) e->L))) return rc;
    HIP_TRY(hipHostMalloc((void **) &e->hMirror, sizeof(HostMirror), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **) &e->hLaneOut, std::max<size_t>((size_t) e->L, 2) * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **) &e->hCnt, std::max<size_t>((size_t) e->L, 2) * sizeof(int32_t), hipHostMallocDefault));
    if ((rc = e->allocRaw(&e->finTicket, 4))) return rc;  // [0] ticket, [2..3] 64-bit total of exactFinishStatistics
    HIP_TRY(hipMemset(e->finTicket, 0, 4 * sizeof(int32_t)));
    if ((rc = e->allocRaw(&e->finCount, (size_t) kFinShards * 32))) return rc;
    HIP_TRY(hipMemset(e->finCount, 0, (size_t) kFinShards * 32 * sizeof(int32_t)));
    if ((rc = e->allocRaw(&e->llDyn, (size_t) e->K))) return rc;
    if ((rc = e->allocRaw(&e->llGate, (size_t) e->K))) return rc;
    if ((rc = e->allocRaw(&e->laneTail, (size_t) e->L))) return rc;
    if ((rc = e->allocRaw(&e->admitRec, dPad))) return rc;
    {
        // derived tables: index of each laneLink inside its intersection (laneLinks of one intersection are
        // contiguous in RoadNet::getLaneLinks() order), the peer's bit for every cross entry, mask offsets
        std::vector<int32_t> llLocal(e->K), xPeerBit(e->E), maskStart(e->I + 1, 0), nLL(e->I, 0), first(e->I, -1);
        for (int k = 0; k < e->K; ++k) {
            int in = n->ll_inter[k];
            if (first[in] < 0) first[in] = k;
            llLocal[k] = k - first[in];
            nLL[in] = std::max(nLL[in], llLocal[k] + 1);
        }
        for (int i = 0; i < e->I; ++i) maskStart[i + 1] = maskStart[i] + (nLL[i] + 63) / 64;
        for (int x = 0; x < e->E; ++x) xPeerBit[x] = llLocal[n->x_ll[n->x_peer[x]]];
        e->nMaskWords = maskStart[e->I];
        std::vector<double2> lm((size_t) e->D), xdd((size_t) e->E);
        std::vector<int4> xpack((size_t) e->E);
        for (int dv = 0; dv < e->D; ++dv) lm[dv] = make_double2(n->drv_length[dv], n->drv_max_speed[dv]);
        for (int x = 0; x < e->E; ++x) {
            int pe = n->x_peer[x], pll = n->x_ll[pe];
            xdd[x] = make_double2(n->x_dist[x], n->x_dist[pe]);
            xpack[x] = make_int4(pll, llLocal[pll], n->ll_type[pll], n->ll_roadlink[pll]);
        }
        if ((rc = e->uploadConst(d.drvLM, lm.data(), lm.size()))) return rc;
        if ((rc = e->uploadConst(d.xDD, xdd.data(), xdd.size()))) return rc;
        if ((rc = e->uploadConst(d.xPack, xpack.data(), xpack.size()))) return rc;
        std::vector<int4> llpack((size_t) e->K);
        for (int k = 0; k < e->K; ++k)
            llpack[k] = make_int4(n->ll_x_start[k], n->ll_x_start[k + 1], maskStart[n->ll_inter[k]], n->ll_type[k]);
        if ((rc = e->uploadConst(d.llPack, llpack.data(), llpack.size()))) return rc;
        std::vector<int4> ll4((size_t) e->L), le4((size_t) e->L);
        for (int l = 0; l < e->L; ++l) {
            const int b = n->lane_ll_start[l], cnt = n->lane_ll_start[l + 1] - b;
            int v[4] = {-1, -1, -1, -1}, en[4] = {-1, -1, -1, -1};
            for (int q = 0; q < cnt && q < 4; ++q) {
                v[q] = n->lane_ll[b + q];
                en[q] = n->ll_end_lane[v[q]];
            }
            ll4[l] = cnt > 4 ? make_int4(-2, -2, -2, -2) : make_int4(v[0], v[1], v[2], v[3]);
            le4[l] = cnt > 4 ? make_int4(-1, -1, -1, -1) : make_int4(en[0], en[1], en[2], en[3]);
        }
        if ((rc = e->uploadConst(d.laneLL4, ll4.data(), ll4.size()))) return rc;
        if ((rc = e->uploadConst(d.laneEnd4, le4.data(), le4.size()))) return rc;
        if ((rc = e->uploadConst(d.llLocal, llLocal.data(), llLocal.size()))) return rc;
        if ((rc = e->uploadConst(d.xPeerBit, xPeerBit.data(), xPeerBit.size()))) return rc;
        if ((rc = e->uploadConst(d.interMaskStart, maskStart.data(), maskStart.size()))) return rc;
        if ((rc = e->allocRaw(&e->interMask, (size_t) std::max(e->nMaskWords, 1)))) return rc;
    }
    if ((rc = e->allocRaw(&e->curPhase, (size_t) e->I))) return rc;
    if ((rc = e->allocRaw(&e->remain, (size_t) e->I))) return rc;
    e->nScanBlocks = (e->D + kScanTile - 1) / kScanTile;
    if ((rc = e->allocRaw(&e->scanGranules, (size_t) e->nScanBlocks))) return rc;
    if ((rc = e->allocRaw(&e->scanTicket, 1))) return rc;
    if ((rc = e->allocRaw(&e->jobCount, (size_t) kJobShards * kJobShardStride))) return rc;
    if ((rc = e->allocRaw(&e->sc, 1))) return rc;
    if (cfg->lane_history) {
        LaneHistDev &h = e->hist;
        const size_t nL = (size_t) std::max(e->L, 1);
        h.L = e->L;
        if ((rc = e->allocRaw(&h.num, nL * kLaneHistoryMax))) return rc;
        if ((rc = e->allocRaw(&h.avg, nL * kLaneHistoryMax))) return rc;
        if ((rc = e->allocRaw(&h.head, nL))) return rc;
        if ((rc = e->allocRaw(&h.len, nL))) return rc;
        if ((rc = e->allocRaw(&h.hNum, nL))) return rc;
        if ((rc = e->allocRaw(&h.hAvg, nL))) return rc;
        HIP_TRY(hipMemset(h.num, 0, nL * kLaneHistoryMax * sizeof(int32_t)));
        HIP_TRY(hipMemset(h.avg, 0, nL * kLaneHistoryMax * sizeof(double)));
        HIP_TRY(hipMemset(h.head, 0, nL * sizeof(int32_t)));
        HIP_TRY(hipMemset(h.len, 0, nL * sizeof(int32_t)));
        HIP_TRY(hipMemset(h.hNum, 0, nL * sizeof(int32_t)));
        HIP_TRY(hipMemset(h.hAvg, 0, nL * sizeof(double)));
    }
    if (cfg->lane_change) {
        LcDev &lc = e->lc;
        lc.on = 1;
        // batched environments (cfx_config::n_envs): E disjoint copies of one network, index spaces concatenated copy by copy
        const int nEnvs = cfg->n_envs > 1 ? cfg->n_envs : 1;
        if (e->R % nEnvs != 0 || nEnvs >= (1 << (31 - kLcEnvShift))) {
            e->err = "cfx_create: n_envs does not divide the number of roads (or is too large)";
            return CFX_ERR_INVALID;
        }
        lc.roadsPerEnv = std::max(e->R / nEnvs, 1);
        lc.poolPerEnv = 0;
        if ((rc = e->uploadConst(lc.laneWidth, n->lane_width, (size_t) e->L))) return rc;
        if ((rc = e->uploadConst(lc.roadLaneStart, n->road_lane_start, (size_t) e->R + 1))) return rc;
        if ((rc = e->uploadConst(lc.laneNumSegs, n->lane_n_segments, (size_t) e->L))) return rc;
        if ((rc = e->allocRaw(&lc.roadCand, (size_t) e->R))) return rc;
        HIP_TRY(hipMemset(lc.roadCand, 0, (size_t) std::max(e->R, 1) * sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.insHead, (size_t) e->L))) return rc;
        HIP_TRY(hipMemset(lc.insHead, 0xFF, (size_t) std::max(e->L, 1) * sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.insCount, 1))) return rc;
        HIP_TRY(hipMemset(lc.insCount, 0, sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.parkCount, 4))) return rc;
        HIP_TRY(hipMemset(lc.parkCount, 0, 4 * sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.roadCandList, (size_t) std::max(e->R, 1) * kLcRoadCand))) return rc;
        if ((rc = e->allocRaw(&lc.insLanes, (size_t) e->L))) return rc;
        if ((rc = e->allocRaw(&lc.insLaneCount, 1))) return rc;
        HIP_TRY(hipMemset(lc.insLaneCount, 0, sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.fixCount, 1))) return rc;
        HIP_TRY(hipMemset(lc.fixCount, 0, sizeof(int32_t)));
        if ((rc = e->allocRaw(&lc.candAllCount, 1))) return rc;
        HIP_TRY(hipMemset(lc.candAllCount, 0, sizeof(int32_t)));
        HIP_TRY(hipEventCreateWithFlags(&e->pollEvent, hipEventDisableTiming));
    }
    // Capacities for the run, taken once: slots for half of what the network holds bumper to bumper (5 m per vehicle; the
    // dense layout only — the ring layout has its own rings), vehicle numbers for kInitialVidCap vehicles between two resets.
    // Both still grow when a run outgrows them, the vehicle tables without draining the stream (growDeferred) — but a step of
    // an ordinary run never allocates (round 4's 70 ms hiccup in the 200-step windows was the vehicle table doubling from 64 k).
    {
        size_t slots = (size_t) e->L + 4096;
        if (!e->ring) {
            double jam = 0;
            for (int d = 0; d < e->D; ++d) jam += std::ceil(e->hDrvLength[d] / 5.0);
            slots = std::max(slots, std::min<size_t>((size_t) (jam / 2) + (size_t) e->L + 4096, (size_t) 1 << 24));
        }
        if ((rc = e->ensureSlotCap(slots))) return rc;
    }
    if ((rc = e->ensureVidCap(1))) return rc;
    return e->resetState();
}

int32_t cfx_create(const cfx_net *n, const cfx_config *cfg, cfx_engine **out) {
    if (!n || !cfg || !out) {
        g_createError = "cfx_create: null argument";
        return CFX_ERR_INVALID;
    }
    if (cfg->lane_change && (!n->lane_width || !n->road_lane_start || !n->lane_n_segments)) {
        g_createError = "cfx_create: lane_change needs cfx_net::lane_width, lane_n_segments and road_lane_start";
        return CFX_ERR_INVALID;
    }
    cfx_engine *e = new cfx_engine();
    int32_t rc = createImpl(e, n, cfg);
    if (rc != CFX_OK) {
        g_createError = e->err;
        cfx_destroy(e);
        *out = nullptr;
        return rc;
    }
    *out = e;
    return CFX_OK;
}

int32_t cfx_add_templates(cfx_engine *e, int32_t n, const cfx_vehicle_template *t) {
    if (!e || n < 0 || (n && !t)) return CFX_ERR_INVALID;
    e->hTempl.insert(e->hTempl.end(), t, t + n);
    e->tablesDirty = true;
    return CFX_OK;
}

int32_t cfx_add_routes(cfx_engine *e, int32_t nRoutes, const int32_t *routeStart, const int32_t *roads,
                       const int32_t *nextStart, const int32_t *nextLL) {
    if (!e || nRoutes < 0) return CFX_ERR_INVALID;
    int roadBase = (int) e->hRouteRoads.size();
    int nextBase = (int) e->hNextLL.size();
    int nPos = routeStart[nRoutes];
    for (int r = 1; r <= nRoutes; ++r) e->hRouteStart.push_back(roadBase + routeStart[r]);
    e->hRouteRoads.insert(e->hRouteRoads.end(), roads, roads + nPos);
    for (int p = 1; p <= nPos; ++p) e->hNextStart.push_back(nextBase + nextStart[p]);
    e->hNextLL.insert(e->hNextLL.end(), nextLL, nextLL + nextStart[nPos]);
    e->tablesDirty = true;
    return CFX_OK;
}

}  // extern "C"
static int32_t stepImpl(cfx_engine *e, const cfx_spawn *recs, int32_t n);
extern "C" {

int32_t cfx_step(cfx_engine *e, const cfx_spawn *recs, int32_t n) {
    if (!e || n < 0 || (n && !recs)) return CFX_ERR_INVALID;
    // (the host's own time in here is accounted: cfx_get_host_stats)
    e->stallCause = 0;
    const int64_t at = e->step;
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t rc = stepImpl(e, recs, n);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    cfx_host_stats &hs = e->hostStats;
    hs.step_calls += 1;
    hs.step_call_us_sum += us;
    if (us > 1000.0) hs.calls_over_1ms += 1;
    if (us > hs.worst_step_call_us) {
        hs.worst_step_call_us = us;
        hs.worst_step_call_at = at;
        hs.worst_step_call_cause = e->stallCause;
    }
    return rc;
}

int32_t cfx_get_host_stats(cfx_engine *e, cfx_host_stats *out, int32_t reset) {
    if (!e || !out) return CFX_ERR_INVALID;
    *out = e->hostStats;
    if (reset) {
        const int64_t rr = e->hostStats.ring_regrows_total, tg = e->hostStats.table_grows_total;
        e->hostStats = cfx_host_stats{};
        e->hostStats.ring_regrows_total = rr;
        e->hostStats.table_grows_total = tg;
    }
    return CFX_OK;
}

const char *cfx_profile_kernel_symbol(cfx_engine *e, int32_t k) {
    return (e && k >= 0 && k < kNumProfKernels && e->slotSymbol[k]) ? e->slotSymbol[k] : "";
}

}  // extern "C"
static int32_t stepImpl(cfx_engine *e, const cfx_spawn *recs, int32_t n) {
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = e->syncTables())) return rc;
    hipStream_t st = e->stream;
    // Fail at the step that failed (or the one after): every step leaves its scalars in pinned host memory, so an
    // overflow code raised by a step the device has already finished is seen here without waiting for anything.
    if (e->mirrorValid && __atomic_load_n(&e->hMirror->sc.overflow, __ATOMIC_RELAXED) != 0) {
        DevScalars s;
        if ((rc = e->readScalars(s))) return rc == CFX_ERR_DEVICE ? CFX_ERR_CAPACITY : rc;
    }
    // ---- the host's lead over the device is bounded.  A free-running caller submits a 30x30 step in ~20 us and the device takes
    //      ~40: unbounded, the stream's backlog grows by half a step per step, and the next call that has to wait for the device
    //      (a getter, the spawner's priority-collision query) pays for all of it — 4.5 ms behind a 200-step window, 2 ms with 48 steps in flight.  The step's
    //      commit publishes the number of completed steps in pinned host memory: wait (yielding) while more than kMaxLead steps
    //      are in flight.  The device is never idle for it — kMaxLead steps of work are queued behind the one it is executing —
    //      and a caller that synchronises every step never waits here.  (Built while looking for the 45-80 ms "stall" of the
    //      sustained windows of rounds 4-6, which it did NOT cure: that was the container's CPU quota — bench.py's header.)
    if (e->mirrorValid && !e->tiled) {
        constexpr int64_t kMaxLead = 16;  // (16 steps of a 30x30 network = 0.65 ms of queued work: what the next waiting call pays at most)
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const int64_t done = (int64_t) (__atomic_load_n(&e->hMirror->progress, __ATOMIC_RELAXED) >> 32);
            if (done <= 0 || done > e->step || e->step - done <= kMaxLead) break;
            if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;  // (never hang on it)
            std::this_thread::yield();
        }
    }
    if (e->ring) {
        if (e->mirrorValid && __atomic_load_n(&e->hMirror->sc.ringNearFull, __ATOMIC_RELAXED) != 0) e->ringGrowRequested = true;
        if ((rc = e->ringEnsure())) return rc;
    }
    if (e->observing && ++e->observeIdle > cfx_engine::kObserveIdle) {  // nobody has read the lane counts for a while
        e->observing = false;
        e->hCntValid = false;
    }

    if (e->lc.on) {
        if (e->pollPending) return e->fail("cfx_step: the previous lane-change step was not polled (cfx_lane_change_poll)");
        if (!e->hPoll) return e->fail("cfx_step: cfx_lane_change_supply must be called before a step with lane change");
        if ((rc = e->ensureVidCap((size_t) e->spawned + (size_t) n + (size_t) e->poolN))) return rc;
    }
    // ---- phase 0/1 tail: hand the spawn records to the device
    SpawnBatchBig batch;  // (the records go with the admission kernel's arguments: up to kAdmitRecs, kd_admit up to kAdmitRecsBig)
    batch.n = 0;
    batch.firstNewVid = 0;
    SpawnBatchMem memBatch{};  // ... or, more of them on the ring layout, in pinned memory
    bool inMem = false;
    int memStageUsed = -1;
    // (the ring layout too: more records than the small batch holds kept the step from riding its commit with the next
    // admission, and cost a launch of their own; a step with up to kAdmitRecs records runs the small instantiation as before)
    const int batchRoom = ((e->useTails() && (e->denseForm & 4)) || e->ring) ? kAdmitRecsBig : kAdmitRecs;
    auto smallBatch = [&batch]() {  // (the kernels that take at most kAdmitRecs records)
        SpawnBatch b;
        b.n = batch.n;
        b.firstNewVid = batch.firstNewVid;
        b.enterTime = batch.enterTime;
        for (int i = 0; i < batch.n && i < kAdmitRecs; ++i) {
            b.lane[i] = batch.lane[i];
            b.prevWait[i] = batch.prevWait[i];
            b.route[i] = batch.route[i];
            b.priority[i] = batch.priority[i];
            b.firstNext[i] = batch.firstNext[i];
            b.templ[i] = batch.templ[i];
            b.vidOff[i] = batch.vidOff[i];
        }
        return b;
    };
    if (n > 0) {
        // the batch carries the next n vehicle numbers, each once, in any order (checked in full by the CPU twin)
        if (recs[0].vid < e->spawned || recs[0].vid >= e->spawned + n)
            return e->fail("cfx_step: spawn records must continue the dense vid sequence");
        if ((rc = e->ensureVidCap((size_t) e->spawned + n))) return rc;
        // ring layout: a step's few records go with kr_admit's arguments (each lane's thread links its own): no launch.  They
        // fit if they are few, all for lanes of this engine, and all enter at the same time (what a host spawner produces:
        // Engine::getCurrentTime); record i of the batch is vehicle spawned + i, whatever order they came in
        bool inArgs = n <= batchRoom;  // (kr_admit / kd_admit / k_admit take the batch)
        // a custom speed waiting for one of these vehicles has to be in the vehicle table before the admission looks at it:
        // the records then take the k_spawn_link path and the speeds follow it on the stream (rare: push_vehicle + set_vehicle_speed)
        std::vector<std::pair<int32_t, double>> customNow;
        for (const auto &fc : e->futureCustom)
            if (fc.first < e->spawned + n) customNow.push_back(fc);
        if (!customNow.empty()) inArgs = false;
        if (inArgs) {
            batch.n = (int) n;
            batch.firstNewVid = (int) e->spawned;
            batch.enterTime = recs[0].enter_time;
            int order[kAdmitRecsBig];
            bool seen[kAdmitRecsBig] = {};
            for (int i = 0; inArgs && i < n; ++i) {
                const int64_t off = (int64_t) recs[i].vid - e->spawned;
                inArgs = off >= 0 && off < n && !seen[off] && recs[i].lane >= -1 && recs[i].enter_time == batch.enterTime &&
                         recs[i].templ < 32768;
                if (inArgs) seen[off] = true;
                order[i] = i;
            }
            if (inArgs) {
                std::sort(order, order + n, [recs](int x, int y) {
                    return recs[x].lane != recs[y].lane ? recs[x].lane < recs[y].lane : recs[x].vid < recs[y].vid;
                });
                for (int j = 0; j < n; ++j) {
                    const cfx_spawn &r = recs[order[j]];
                    batch.lane[j] = r.lane;
                    batch.prevWait[j] = r.prev_wait;
                    batch.route[j] = r.route;
                    batch.priority[j] = r.priority;
                    batch.firstNext[j] = e->firstNextOf(r.lane, r.route);
                    batch.templ[j] = (int16_t) r.templ;
                    batch.vidOff[j] = (int16_t) (r.vid - e->spawned);
                }
            } else {
                batch.n = 0;
            }
        }
        // ring layout, more records than the arguments hold (batched environments): the columns go into pinned memory, sorted
        // by lane, with the offsets of every block of lanes — kr_admit reads its own lanes' records from there (SpawnBatchMem)
        const int nLaneBlocks = (e->L + kBlock - 1) / kBlock;
        if (!inArgs && e->ring && n > batchRoom && customNow.empty() && !e->cfg.debug_sync) {
            inMem = true;
            const int64_t base = e->spawned;
            const double t0 = recs[0].enter_time;
            // a counting sort by block of lanes (bucket 0: the records without a lane here), then each block's few records by
            // lane — the kernel looks a lane's records up by bisection; a lane's own records may come in any order
            e->memSeen.assign((size_t) n, 0);
            e->memBucket.assign((size_t) nLaneBlocks + 2, 0);
            for (int i = 0; inMem && i < n; ++i) {
                const int64_t off = (int64_t) recs[i].vid - base;
                inMem = off >= 0 && off < n && !e->memSeen[(size_t) off] && recs[i].lane >= -1 && recs[i].lane < e->L &&
                        recs[i].enter_time == t0;
                if (inMem) {
                    e->memSeen[(size_t) off] = 1;  // (every vehicle number once)
                    e->memBucket[(size_t) (recs[i].lane < 0 ? 0 : 1 + recs[i].lane / kBlock) + 1] += 1;
                }
            }
        }
        if (inMem) {
            for (int q = 0; q <= nLaneBlocks; ++q) e->memBucket[(size_t) q + 1] += e->memBucket[(size_t) q];  // bucket q starts at [q]
            e->memOrder.resize((size_t) n);
            e->memCursor.assign(e->memBucket.begin(), e->memBucket.end() - 1);
            for (int i = 0; i < n; ++i) e->memOrder[(size_t) e->memCursor[(size_t) (recs[i].lane < 0 ? 0 : 1 + recs[i].lane / kBlock)]++] = i;
            for (int q = 1; q <= nLaneBlocks; ++q) {
                int32_t *const o = e->memOrder.data();
                const int from = e->memBucket[(size_t) q], to = e->memBucket[(size_t) q + 1];
                if (to - from > 24) {
                    std::stable_sort(o + from, o + to, [recs](int x, int y) { return recs[x].lane < recs[y].lane; });
                } else {
                    for (int j = from + 1; j < to; ++j) {
                        const int x = o[j], lx = recs[x].lane;
                        int i = j;
                        for (; i > from && recs[o[i - 1]].lane > lx; --i) o[i] = o[i - 1];
                        o[i] = x;
                    }
                }
            }
            const size_t words = (size_t) 7 * (size_t) n + (size_t) nLaneBlocks + 1;
            if (words > e->memStageWords) {
                e->stallCause |= CFX_STALL_STAGE_GROW;
                HIP_TRY(hipStreamSynchronize(st));
                const size_t nc = std::max<size_t>(words * 2, 4096);
                for (int i = 0; i < cfx_engine::kStages; ++i) {
                    if (e->hMemStage[i]) HIP_TRY(hipHostFree(e->hMemStage[i]));
                    e->hMemStage[i] = nullptr;
                    HIP_TRY(hipHostMalloc((void **) &e->hMemStage[i], nc * sizeof(int32_t), hipHostMallocDefault));
                    if (!e->memStageEvent[i]) HIP_TRY(hipEventCreateWithFlags(&e->memStageEvent[i], hipEventDisableTiming));
                    e->memStageBusy[i] = false;
                }
                e->memStageWords = nc;
            }
            const int si = e->memStageIdx;
            e->memStageIdx = (si + 1) % cfx_engine::kStages;
            if (e->memStageBusy[si]) HIP_TRY(hipEventSynchronize(e->memStageEvent[si]));
            int32_t *const w = e->hMemStage[si];
            int32_t *const cLane = w, *const cPrev = w + n, *const cRoute = w + 2 * (size_t) n, *const cPrio = w + 3 * (size_t) n,
                           *const cNext = w + 4 * (size_t) n, *const cTempl = w + 5 * (size_t) n, *const cOff = w + 6 * (size_t) n,
                           *const cBlock = w + 7 * (size_t) n;
            for (int j = 0; j < n; ++j) {
                const cfx_spawn &r = recs[e->memOrder[(size_t) j]];
                cLane[j] = r.lane;
                cPrev[j] = r.prev_wait;
                cRoute[j] = r.route;
                cPrio[j] = r.priority;
                cNext[j] = e->firstNextOf(r.lane, r.route);
                cTempl[j] = r.templ;
                cOff[j] = (int32_t) (r.vid - e->spawned);
            }
            for (int q = 0; q <= nLaneBlocks; ++q) cBlock[q] = e->memBucket[(size_t) q + 1];  // block q's records: bucket q + 1
            memBatch.n = (int) n;
            memBatch.firstNewVid = (int) e->spawned;
            memBatch.enterTime = recs[0].enter_time;
            memBatch.lane = cLane;
            memBatch.prevWait = cPrev;
            memBatch.route = cRoute;
            memBatch.priority = cPrio;
            memBatch.firstNext = cNext;
            memBatch.templ = cTempl;
            memBatch.vidOff = cOff;
            memBatch.blockOff = cBlock;
            memBatch.nLaneBlocks = nLaneBlocks;
            memStageUsed = si;
        } else if (inArgs) {
            // (nothing to launch)
        } else {
            if ((rc = e->settle(false))) return rc;  // (k_spawn_link looks at what the previous step's commit leaves)
            if ((size_t) n > e->recCap) {
                size_t nc = std::max<size_t>((size_t) n * 2, 1024);
                if ((rc = e->grow(&e->dRecs, 0, nc))) return rc;
                e->recCap = nc;
            }
            if ((size_t) n > e->stageCap) {
                e->stallCause |= CFX_STALL_STAGE_GROW;
                HIP_TRY(hipStreamSynchronize(st));
                size_t nc = std::max<size_t>((size_t) n * 2, 1024);
                for (int i = 0; i < cfx_engine::kStages; ++i) {
                    if (e->hStage[i]) HIP_TRY(hipHostFree(e->hStage[i]));
                    HIP_TRY(hipHostMalloc((void **) &e->hStage[i], nc * sizeof(cfx_spawn), hipHostMallocDefault));
                    if (!e->stageEvent[i]) HIP_TRY(hipEventCreateWithFlags(&e->stageEvent[i], hipEventDisableTiming));
                    e->stageBusy[i] = false;
                }
                e->stageCap = nc;
            }
            const int si = e->stageIdx;
            e->stageIdx = (si + 1) % cfx_engine::kStages;
            if (e->stageBusy[si]) HIP_TRY(hipEventSynchronize(e->stageEvent[si]));
            memcpy(e->hStage[si], recs, (size_t) n * sizeof(cfx_spawn));
            // the kernel reads the pinned (device-visible) staging buffer itself: no separate copy launch
            e->launchNamed(PK_SPAWN, "k_spawn_link", k_spawn_link, dim3(gridFor(n)), dim3(kBlock), (const cfx_spawn *) e->hStage[si], (int) n,
                      (int) e->spawned, e->vt, e->waitHead, e->lc);
            HIP_TRY(hipEventRecord(e->stageEvent[si], st));
            e->stageBusy[si] = true;
            if (!customNow.empty()) {  // Vehicle::setCustomSpeed on a vehicle still in its waiting buffer (cfx_set_vehicle_speed)
                const uint8_t one = 1;
                for (const auto &fc : customNow) {
                    HIP_TRY(hipMemcpyAsync(e->vt.customSpeed + fc.first, &fc.second, sizeof(double), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipMemcpyAsync(e->vt.pendingCustom + fc.first, &one, 1, hipMemcpyHostToDevice, st));
                }
                HIP_TRY(hipStreamSynchronize(st));  // (the sources live on this frame)
            }
        }
        e->spawned += n;
    }
    for (const auto &fc : e->futureCustom)  // (what this batch did not create was not a vehicle of this step: counted, not silent)
        if (fc.first >= e->spawned) e->droppedFutureSpeeds += 1;
    e->futureCustom.clear();
    // ---- slot capacity.  Two host-side upper bounds of the vehicles that can be running after this step:
    //   (a) spawned - finished (as of the last read)           — tight while nobody queues for long;
    //   (b) running (as of the last read) + what can have been admitted since: at most one vehicle per step on every
    //       lane that has ever had a vehicle queued, plus halo migrants — stays small when entry lanes saturate and the
    //       waiting queues grow without bound (they hold vehicle-table entries, not slots).
    // When the smaller of the two outgrows the buffers, the true count is read back and the buffers grow only if needed.
    for (int i = 0; i < n; ++i) {
        if (!cfx_engine::dyadic(recs[i].enter_time)) e->timesDyadic = false;
        const int lane = recs[i].lane;
        if (lane >= 0 && !e->laneQueued[lane]) {
            e->laneQueued[lane] = 1;
            e->nQueueLanes += 1;
        }
        e->spawnedHere += lane >= 0;
    }
    e->liveUpper += e->nQueueLanes;

    if (e->ring) {
        // ---- ring layout: admit, action (+ notify sources), cross, commit — no scan, no scatter
        RingCtx c = e->rctx(true);
        const bool dbg = e->cfg.debug_sync != 0;  // developer aid: name the kernel that faults
#define RING_CHECK(name)                                                                                          \
    if (dbg) {                                                                                                    \
        fprintf(stderr, "[cfx ring] step %lld: %s\n", (long long) e->step, name);                                 \
        hipError_t er = hipStreamSynchronize(st);                                                                 \
        if (er != hipSuccess) return e->fail(std::string("ring step: ") + name + ": " + hipGetErrorString(er));   \
    }
        RING_CHECK("before the step")
        const unsigned long long pr = __atomic_load_n(&e->hMirror->progress, __ATOMIC_RELAXED);
        // running vehicles as of the last step the device has completed (stale by the few steps the host runs ahead):
        // only sizes the cross phase's grid and picks its organisation
        const size_t activeEst = (size_t) (pr & 0xFFFFFFFFu) + (size_t) e->nQueueLanes * 4;
        const bool useBig = e->cross2 >= 0 ? e->cross2 == 1 : activeEst > 240000;  // which form of the cross phase (§4)
        // This step's commit rides with the next step's admission (one launch less per step) where the step runs kr_cross,
        // which then advances the lights; the previous step's, if it is still pending, goes with this step's admission.
        const bool deferCommit = e->ringMerge && !dbg && !e->tiled && !e->observing;  // (a caller that reads the lane counts after every step wants the commit now)
        // tiling: the previous step's halo import, if cfx_halo_wait left it to this launch
        const RingHaloIn hin = e->haloImportPending ? e->pendingImport : RingHaloIn{};
        e->haloImportPending = false;
        if (e->commitPending) {
            e->commitPending = false;
            int nStatPrev = 1;
            const RingCommit rkPrev = e->commitArgs(activeEst, true, &nStatPrev);
            if (inMem)
                e->launchNamed(PK_ADMIT, "kr_admit<true, kBlock, SpawnBatchMem>", kr_admit<true, kBlock, SpawnBatchMem>, dim3(gridFor(e->D) + nStatPrev), dim3(kBlock), e->rctx(true, e->step - 1, e->rcur ^ 1),
                          e->admitStep, e->waitHead, e->vt, e->sc, memBatch, rkPrev, RingHaloIn{});
            else if (batch.n > kAdmitRecs)
                e->launchNamed(PK_ADMIT, "kr_admit<true, kAdmitRecsBig>", kr_admit<true, kAdmitRecsBig>, dim3(gridFor(e->D) + nStatPrev), dim3(kBlock), e->rctx(true, e->step - 1, e->rcur ^ 1),
                          e->admitStep, e->waitHead, e->vt, e->sc, batch, rkPrev, RingHaloIn{});
            else
                e->launchNamed(PK_ADMIT, "kr_admit<true>", kr_admit<true>, dim3(gridFor(e->D) + nStatPrev), dim3(kBlock), e->rctx(true, e->step - 1, e->rcur ^ 1),
                          e->admitStep, e->waitHead, e->vt, e->sc, smallBatch(), rkPrev, RingHaloIn{});
        } else {
            if (inMem)
                e->launchNamed(PK_ADMIT, "kr_admit<false, kBlock, SpawnBatchMem>", kr_admit<false, kBlock, SpawnBatchMem>, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->sc, memBatch, RingCommit{}, hin);
            else if (batch.n > kAdmitRecs)
                e->launchNamed(PK_ADMIT, "kr_admit<false, kAdmitRecsBig>", kr_admit<false, kAdmitRecsBig>, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->sc, batch, RingCommit{}, hin);
            else
                e->launchNamed(PK_ADMIT, "kr_admit<false>", kr_admit<false>, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->sc, smallBatch(), RingCommit{}, hin);
        }
        if (memStageUsed >= 0) {  // (the kernel reads the pinned buffer itself: free again when this launch is through)
            HIP_TRY(hipEventRecord(e->memStageEvent[memStageUsed], st));
            e->memStageBusy[memStageUsed] = true;
        }
        RING_CHECK("kr_admit")
        RingJob *const jobRecs = useBig ? nullptr : e->rJobRecs;  // k_cross2 starts from the slots: no job records then
        RingOut ro{c.kinN, c.blkW, e->rScratch, e->rMovers, e->sc, e->rFinKey, e->rFinVid, e->rFinCap, e->finCount};
        JobQueue jq{e->jobCount, e->rJobs, e->rJobCap, &e->sc->overflow};
        {
            // One workgroup = B threads over G lanes (or B laneLinks).  G is picked so that a block's vehicles fit one pass
            // (B - 1) with room for uneven lanes; small networks take small blocks (every block resident at once, the step
            // is bound by the slowest block's chain), large ones full blocks (throughput).
            // cfx_config::ring_lanes_per_wave = G + 1000 * (B / 256) overrides both (developer knob; B = 256 or 512);
            // + 10000 forces the wave form of the action kernel (kw_action; the default above 240 k vehicles), + 20000 the
            // block form (kr_action; the default below).
            int G = e->ringG, Bsel = 256;
            const int form = (e->cfg.ring_lanes_per_wave / 10000) % 10;
            const bool blockForm = form == 2 || (form == 0 && !(useBig || activeEst > 240000));
            const int want = e->cfg.ring_lanes_per_wave % 10000;
            if (want > 0) {
                G = std::max(1, want % 1000);
                Bsel = want >= 2000 ? 512 : 256;
            } else if (useBig || activeEst > 240000) {
                G = 28;  // many rounds of blocks anyway: full blocks, a second chunk where the lanes are dense (throughput)
            } else if (e->mirrorValid) {
                // adapt to the traffic: the densest block of a recent step (pinned mirror, possibly a few steps old) should
                // fit one pass with some room; results do not depend on G
                const int maxT = __atomic_load_n(&e->hMirror->sc.actionMaxT, __ATOMIC_RELAXED);
                if (e->ringHold > 0) {
                    e->ringHold -= 1;  // (the mirror lags a few steps behind: let a change show before the next one)
                } else if (maxT > Bsel - 1) {
                    e->ringG = std::max(4, e->ringG - 1);
                    e->ringCalm = 0;
                    e->ringHold = 8;
                } else if (maxT == 0 && e->ringG < 16) {  // no block above 3/4 of a pass for a while: fuller blocks
                    if (++e->ringCalm >= 256) {
                        e->ringG += 1;
                        e->ringCalm = 0;
                    }
                } else {
                    e->ringCalm = 0;
                }
                G = e->ringG;
            }
            // + 30000: the list form (kr_index + kl_action; the default where the cross phase runs k_cross2)
            // (the lane-walking forms cost per lane, the list per vehicle plus 8 to 12 us for the list: 30x30, 10.8 k lanes, 90 k
            //  vehicles: 41.0 us per step in block form, 49.0 with the list; 100x100, 120 k lanes, 72 k vehicles: 97.1 and 54.7)
            const bool listForm = form == 3 || form == 6 || (form == 0 && (useBig || activeEst > 240000 || e->L > 20000));
            if (listForm) {
                // a TRUE bound of the vehicles this step can list (the list and the launch are sized by it): what the device
                // reported after the last step it has completed plus one admission per queueing lane and step since
                const int64_t done = (int64_t) (pr >> 32);
                // (a tile: plus what the halo can have brought in since — its count is as of the commit, before that step's import)
                if (done > 0 && done <= e->step)
                    e->liveUpper = std::min(e->liveUpper, (int64_t) (pr & 0xFFFFFFFFu) + e->nQueueLanes * (e->step + 1 - done) +
                                                              (e->tiled ? (int64_t) e->halo.nImport * CFX_HALO_MAX_MIGRANTS * (e->step + 2 - done) +
                                                                              2 * (int64_t) e->halo.nGhost : 0));
                const size_t listBound = (size_t) std::min<int64_t>(e->liveUpper, (int64_t) e->ringSlots);
                const size_t needList = (listBound + kListBlock - 1) / kListBlock * kListBlock + kListBlock;
                if (needList > e->rListCap) {
                    e->stallCause |= CFX_STALL_LIST_GROW;
                    e->hostStats.table_grows_total += 1;
                    const size_t nc = std::max(needList + needList / 4, e->rListCap * 2) / kListBlock * kListBlock;
                    int rc = e->growDeferred(&e->rList, 0, nc);  // (rebuilt every step: nothing to keep)
                    if (rc) return rc;
                    e->rListCap = nc;
                    if (!e->rListCount) {
                        if ((rc = e->allocRaw(&e->rListCount, 2)) || (rc = e->allocRaw(&e->rIdxGranules, (size_t) gridFor(e->D)))) return rc;
                        HIP_TRY(hipMemsetAsync(e->rIdxGranules, 0, (size_t) gridFor(e->D) * sizeof(unsigned long long), st));
                        HIP_TRY(hipMemsetAsync(e->rListCount, 0, 2 * sizeof(int32_t), st));
                    }
                }
                const int nVehBlocks = (int) (needList / kListBlock);  // (every entry a block of the launch reads exists)
                const int nLL = (e->K + kListBlock - 1) / kListBlock;
                const int nIdxTiles = (int) ((e->D + kIndexTile - 1) / kIndexTile);
                // (1024 threads: two blocks per CU.  Form 6 hands the tiles out by ticket whatever their number: the path of networks
                //  above half a million drivables, for the tests)
                int32_t *const idxTicket = (nIdxTiles > kScanResidentTiles || form == 6) ? e->rListCount + 1 : nullptr;
                e->launchNamed(PK_SCAN, "kr_index", kr_index, dim3(nIdxTiles), dim3(kIndexBlock), c, e->rIdxGranules, idxTicket, (unsigned) (e->step + 1),
                          // (the capacity the kernel checks is what THIS step's action launch covers, not the allocation — which has
                          //  25 % of headroom and never shrinks: entries beyond the launch would never take their step, unflagged)
                          e->rList, (int) std::min<size_t>(e->rListCap, (size_t) nVehBlocks * kListBlock), e->rListCount, e->sc);
                RING_CHECK("kr_index")
                const RingHist rh = e->takeHist(nVehBlocks + nLL);  // (the last step's Lane::history: trailing blocks of this launch)
                const int nHist = rh.h.num ? (e->L + kListBlock - 1) / kListBlock : 0;
                e->launchNamed(PK_ACTION, e->tiled ? "kl_action<true>" : "kl_action", e->tiled ? kl_action<true> : kl_action<false>, dim3(nVehBlocks + nLL + nHist), dim3(kListBlock), c, ro, jq, jobRecs, (const int4 *) e->rList,
                          (const int32_t *) e->rListCount, nVehBlocks, idxTicket, rh);
            } else {
            G = std::min(G, Bsel);
            const int nLaneBlocks = (e->L + G - 1) / G, nLLBlocks = (e->K + Bsel - 1) / Bsel;
            // (first form: as many blocks again at the end of the grid compute the laneLinks' notify sources)
            const RingHist rh = e->takeHist(nLaneBlocks + 2 * nLLBlocks);  // (the last step's Lane::history: trailing blocks of this launch)
            const dim3 grid(nLaneBlocks + 2 * nLLBlocks + (rh.h.num ? (e->L + Bsel - 1) / Bsel : 0)), block(Bsel);
            if (blockForm) {
                if (e->tiled) {  // (ghost lanes: the instantiations that know about frozen proxies)
                    if (Bsel == 256) e->launchNamed(PK_ACTION, "kr_action<256, true>", kr_action<256, true>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                    else e->launchNamed(PK_ACTION, "kr_action<512, true>", kr_action<512, true>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                } else if (Bsel == 256) e->launchNamed(PK_ACTION, "kr_action<256>", kr_action<256>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                else e->launchNamed(PK_ACTION, "kr_action<512>", kr_action<512>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
            } else {
                if (e->tiled) {
                    if (Bsel == 256) e->launchNamed(PK_ACTION, "kw_action<256, true>", kw_action<256, true>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                    else e->launchNamed(PK_ACTION, "kw_action<512, true>", kw_action<512, true>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                } else if (Bsel == 256) e->launchNamed(PK_ACTION, "kw_action<256>", kw_action<256>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
                else e->launchNamed(PK_ACTION, "kw_action<512>", kw_action<512>, grid, block, c, ro, jq, jobRecs, G, nLaneBlocks, nLLBlocks, rh);
            }
            }
        }
        RING_CHECK("kr_action")
        if (dbg) {
            HIP_TRY(hipMemsetAsync(e->laneOut, 0, 8 * sizeof(int32_t), st));
            hipLaunchKernelGGL(kr_validate, dim3(gridFor(std::max(e->D, 16))), dim3(kBlock), 0, st, c, jq, (int) e->spawned,
                               (int) e->hRouteStart.size() - 1, (int) e->ringSlots, e->laneOut);
            int32_t rep[5] = {0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(rep, e->laneOut, sizeof rep, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (rep[0]) {
                char buf[160];
                snprintf(buf, sizeof buf, "ring invariant %d violated at step %lld: %d %d %d %d", rep[0], (long long) e->step, rep[1], rep[2], rep[3], rep[4]);
                return e->fail(buf);
            }
        }
        if (useBig)
            e->launchNamed(PK_CROSS, "k_cross2<false, RingCtx, RingOut>", k_cross2<false, RingCtx, RingOut>,
                      dim3((int) std::min<size_t>(std::max<size_t>(64, (activeEst / 4 + 15) / 16), (size_t) CFX_RING_CROSS2_WAVES * e->nCU)),  // (5 blocks per CU: 96 registers)
                      dim3(kCross2Block), c, ro, jq, RingLights{e->curPhase, e->remain, (deferCommit && !e->cfg.rl_traffic_light) ? 1 : 0});
        else
        {
            // one 16-lane group per queued vehicle; sized by the job count of the last step the device has reported (every
            // block of the grid pays the prologue, and blocks that find no job still take a wavefront slot for it)
            size_t groups = activeEst / 4;
            if (e->mirrorValid) {
                const int lastJobs = __atomic_load_n(&e->hMirror->sc.nCrossJobs, __ATOMIC_RELAXED);
                if (lastJobs > 0) groups = std::min<size_t>(groups, (size_t) lastJobs + (size_t) lastJobs / 4 + 256);
            }
            e->launchNamed(PK_CROSS, "kr_cross", kr_cross,
                      dim3((int) std::min<size_t>(std::max<size_t>(64, (groups * 16 + kCrossBlock - 1) / kCrossBlock), 32768)),
                      dim3(kCrossBlock), c, ro, jq, (const RingJob *) e->rJobRecs,
                      RingLights{e->curPhase, e->remain, (deferCommit && !e->cfg.rl_traffic_light) ? 1 : 0});
        }
        RING_CHECK("k_cross")
        if (deferCommit) {
            e->commitPending = true;  // launched by the next cfx_step (with its admission) or by settle()
        } else {
            int nStat = 1;
            const RingCommit rk = e->commitArgs(activeEst, false, &nStat);
            e->launchNamed(PK_COMMIT, "kr_commit", kr_commit, dim3(gridStride((size_t) std::max(e->D, std::max(e->I, e->nMaskWords))) + nStat), dim3(kBlock), c, rk, e->vt,
                           e->ringHalo());
            RING_CHECK("kr_commit")
        }
#undef RING_CHECK
        HIP_TRY(hipGetLastError());
        e->rcur ^= 1;
        e->step += 1;
        e->mirrorValid = !e->tiled;
        e->histPending = e->hist.num != 0;  // (this step's Lane::history: with the next action launch, or settle() — on a tile behind the step's halo import)
        return CFX_OK;
    }
    const int64_t spare = e->tiled ? e->spareTotal : (int64_t) e->L;
    auto bound = [e]() {
        const int64_t a = e->spawnedHere - (e->finishedKnown - e->finishedOffset);
        return e->tiled ? e->liveUpper : std::min(a, e->liveUpper);
    };
    const int64_t shadowRoom = e->lc.on ? e->poolN : 0;  // this step's shadows
    // ... and the lanes that get them move behind the layout's end (k_lc_insert), each at most once per step, with the shadows
    // they receive: all running vehicles plus the step's shadows is a true bound of what can move (round 3 reserved a heuristic
    // quarter of the vehicles, which a dense jam in which every lane gets a shadow could exceed: ADVICE round 3)
    auto moveRoom = [e, &bound, shadowRoom]() { return e->lc.on ? bound() + shadowRoom + 64 : (int64_t) 0; };
    size_t need = (size_t) (bound() + spare + shadowRoom + moveRoom()) + 1;
    if (need > e->slotCap && !e->tiled) {
        // the device's own count as of the last step it has completed, read without waiting for it
        const unsigned long long pr = __atomic_load_n(&e->hMirror->progress, __ATOMIC_RELAXED);
        const int64_t done = (int64_t) (pr >> 32);
        if (done > 0 && done <= e->step) {
            // since then: at most one admission per queueing lane and step, and (lane change) that step's shadows
            e->liveUpper = std::min(e->liveUpper, (int64_t) (pr & 0xFFFFFFFFu) + (e->nQueueLanes + shadowRoom) * (e->step + 1 - done));
            need = (size_t) (bound() + spare + shadowRoom + moveRoom()) + 1;
        }
    }
    if (need > e->slotCap) {
        DevScalars s;
        if ((rc = e->readScalars(s))) return rc;  // refreshes finishedKnown too
        e->liveUpper = s.active + 2 * (int64_t) e->halo.nGhost + e->nQueueLanes;
        need = (size_t) (bound() + spare + shadowRoom + moveRoom()) + 1;
        if ((rc = e->ensureSlotCap(need))) return rc;
    }

    if (e->lc.on) {  // per-step fields of the lane-change context
        e->lc.firstShadowVid = (int) e->spawned;
        e->lc.pool = e->hPool;  // pinned: k_lc_insert reads the few priorities it hands out straight from the host's buffer
        e->lc.insCap = e->poolN;
    }
    const bool tails = e->useTails();
    if (tails && !e->dTailNow) {
        if ((rc = e->allocRaw(&e->dTail[0], (size_t) e->D))) return rc;
        if ((rc = e->allocRaw(&e->dTail[1], (size_t) e->D))) return rc;
        if ((rc = e->allocRaw(&e->dTailNow, (size_t) e->D))) return rc;
        if ((rc = e->allocRaw(&e->dGate4, (size_t) std::max(e->K, 1)))) return rc;
    }
    StepCtx c = e->ctx();
    if (tails && !e->tailsValid) {  // after a reset / cfx_load_state: the records of the generation the step starts from
        hipLaunchKernelGGL(kd_init_tails, dim3(gridFor(e->D)), dim3(kBlock), 0, st, c, e->dTail[0], e->dTail[1]);
        e->tailsValid = true;
    }
    // grids: the slots the layout can hold; with lane change the room reserved for moving lanes lies behind them and is
    // covered by the kernels' stride loops in the (rare) step that uses much of it
    const size_t slotBound = std::min(need - (size_t) moveRoom() + (e->lc.on ? (size_t) 4096 : 0), e->slotCap);
    // Two organisations of the cross walk: for latency (fewest dependent rounds per vehicle) and, for large networks, for
    // throughput (far fewer wave-rounds per vehicle).  They break even at ~220 k slots on the MI355X.
    const bool useBig = e->cross2 >= 0 ? e->cross2 == 1 : slotBound > 240000;
    if (e->lc.on && !e->lcSegValid) {  // after a reset / cfx_load_state
        hipLaunchKernelGGL(k_lc_naive, dim3(gridStride(slotBound)), dim3(kBlock), 0, st, c);
        e->lcSegValid = true;
    }
    if (tails && c.laneAdmit) {
        if (!e->gatePhase) {
            if ((rc = e->allocRaw(&e->gatePhase, (size_t) 2 * std::max(e->I, 1)))) return rc;
            HIP_TRY(hipMemsetAsync(e->gatePhase, 0xFF, (size_t) 2 * std::max(e->I, 1) * sizeof(int32_t), st));
        }
        if (batch.n > kAdmitRecs)
            e->launchNamed(PK_ADMIT, "kd_admit<true, kAdmitRecsBig>", kd_admit<true, kAdmitRecsBig>, dim3(gridFor(e->L)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->cs, batch, e->gatePhase);
        else
            e->launchNamed(PK_ADMIT, "kd_admit<true, kAdmitRecs>", kd_admit<true, kAdmitRecs>, dim3(gridFor(e->L)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->cs, smallBatch(), e->gatePhase);
    } else if (tails) {
        if (batch.n > kAdmitRecs)
            e->launchNamed(PK_ADMIT, "kd_admit<false, kAdmitRecsBig>", kd_admit<false, kAdmitRecsBig>, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->cs, batch, (int32_t *) nullptr);
        else
            e->launchNamed(PK_ADMIT, "kd_admit<false, kAdmitRecs>", kd_admit<false, kAdmitRecs>, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->cs, smallBatch(), (int32_t *) nullptr);
    } else e->launchNamed(PK_ADMIT, "k_admit", k_admit, dim3(gridFor(e->D)), dim3(kBlock), c, e->admitStep, e->waitHead, e->vt, e->cs, smallBatch());
    ActionOut ao{e->ab, e->cs, e->vt, e->sc, e->finList, (int) e->slotCap, e->finCount};
    if (e->lc.on) {
        // Engine::nextStep engine.cpp:571-575: initSegments, planLaneChange (+ scheduleLaneChange), and the order rebuilt
        // with the step's shadows in place (cfx_lc_kernels.h)
        const bool dbgSync = e->cfg.debug_sync != 0;  // developer aid: name the kernel that faults
#define LC_CHECK(name)                                                                                     \
    if (dbgSync) {                                                                                         \
        hipError_t er = hipStreamSynchronize(st);                                                          \
        if (er != hipSuccess) return e->fail(std::string("lane change: ") + name + ": " + hipGetErrorString(er)); \
    }
        hipLaunchKernelGGL(k_lc_plan, dim3(gridStride(slotBound)), dim3(kBlock), 0, st, c);
        LC_CHECK("k_lc_plan")
        hipLaunchKernelGGL(k_lc_schedule, dim3(e->R), dim3(64), 0, st, c, e->sc, (const int32_t *) e->vt.priority);
        LC_CHECK("k_lc_schedule")
        hipLaunchKernelGGL(k_lc_insert, dim3(256), dim3(64), 0, st, c, e->vt, e->sc, e->hPoll, e->oldToNew, e->segStart[e->cur].p,
                           e->cnt[e->cur].p, (int) e->slotCap);
        LC_CHECK("k_lc_insert")
        HIP_TRY(hipEventRecord(e->pollEvent, st));  // cfx_lane_change_poll waits for this, not for the whole step
        e->pollPending = true;
        HIP_TRY(hipGetLastError());
        c.admissionsVisible = 1;
        // the leader / gap pass between planLaneChange and getAction takes a history record too (engine.cpp:571-575, 429-442);
        // k_lc_insert has moved the lanes that got shadows: the context is taken anew
        if (e->hist.num && !e->tiled) hipLaunchKernelGGL(k_lane_history, dim3(gridFor(e->L)), dim3(kBlock), 0, st, e->ctx(), e->hist);
    }
    const int nxt = e->cur ^ 1;
    // Two organisations of the cross walk: for latency (fewest dependent rounds per vehicle) and, for large networks, for
    // throughput (far fewer wave-rounds per vehicle).  They break even at ~220 k slots on the MI355X.
    JobQueue jq{e->jobCount, e->crossJobs, (int) e->slotCap, &e->sc->overflow};
    {
        const int nVehBlocks = (int) std::min<size_t>(std::max<size_t>(1, (slotBound + kActBlock - 1) / kActBlock), 8192);
        const int nLLBlocks = (e->K + kActBlock - 1) / kActBlock;
        if (tails) {
            const int nv = (int) ((slotBound + kDenseActBlock - 1) / kDenseActBlock), nl = (e->K + kDenseActBlock - 1) / kDenseActBlock;
            e->launchNamed(PK_ACTION, "kd_action", kd_action, dim3(nv + nl), dim3(kDenseActBlock), c, ao, jq, nv);
        }
        else e->launchNamed(PK_ACTION, e->lc.on ? "k_action<true>" : "k_action<false>", e->lc.on ? k_action<true> : k_action<false>, dim3(nVehBlocks + nLLBlocks), dim3(kActBlock), c, ao, jq,
                       nVehBlocks);
    }
    if (useBig)
        // (the blocks the chip holds at once — 7 per CU: the kernel's LDS — or fewer for a short queue: the kernel sizes its batches)
        e->launchNamed(PK_CROSS, e->lc.on ? "k_cross2<true>" : "k_cross2<false>", e->lc.on ? k_cross2<true> : k_cross2<false>, dim3((int) std::min<size_t>(std::max<size_t>(1, (slotBound + 15) / 16), (size_t) CFX_CROSS2_BLOCKS_PER_CU * e->nCU)),
                  dim3(kCross2Block), c, ao, jq, RingLights{nullptr, nullptr, 0});
    else
        e->launchNamed(PK_CROSS, e->lc.on ? "k_cross<true>" : "k_cross<false>", e->lc.on ? k_cross<true> : k_cross<false>,
                  dim3((int) std::min<size_t>(std::max<size_t>(1, (slotBound * 16 + kCrossBlock - 1) / kCrossBlock), 32768)),
                  dim3(kCrossBlock), c, ao, jq);
    if (e->lc.on) {
        hipLaunchKernelGGL(k_lc_resolve, dim3((int) std::min<size_t>(std::max<size_t>(1, (slotBound / 8 + kBlock - 1) / kBlock), 1024)),
                           dim3(kBlock), 0, st, c, ao, e->oldToNew2);
        hipLaunchKernelGGL(k_lc_resolve_rest, dim3(1), dim3(kBlock), 0, st, c, ao, e->oldToNew2);
    }
    int32_t *const scanTicket = e->nScanBlocks > kScanResidentTiles ? e->scanTicket : nullptr;
    e->launchNamed(PK_SCAN, "k_scan", k_scan, dim3(e->nScanBlocks), dim3(kBlock), (int) e->D, (int) e->L, (const int32_t *) e->cnt[e->cur].p, e->cs,
              e->scanGranules, scanTicket, (unsigned) (e->step + 1), e->segStart[nxt].p, e->cnt[nxt].p, e->gen[nxt].vid,
              e->gen[nxt].drv, e->sc, e->net.laneSpare, (const int32_t *) e->admitStep, (int) e->step, e->waitHead, e->vt,
              e->net.laneGhost, (const int2 *) e->admitRec, e->publishTo(), e->lc.on ? 1 : 0);
    // finish statistics: one extra block per 64 k slots (a rank sort of the step's finishers, see finishStatistics)
    const int nStat = (int) std::min<size_t>(std::max<size_t>(1, slotBound >> 16), 64);
    e->launchNamed(PK_SCATTER, e->lc.on ? "k_scatter<true>" : "k_scatter<false>", e->lc.on ? k_scatter<true> : k_scatter<false>,
              dim3(gridStride(std::max<size_t>(slotBound, (size_t) std::max(e->I, e->nMaskWords))) + nStat), dim3(kBlock), c, e->ab,
              e->cs, e->gen[nxt], (const int32_t *) e->segStart[nxt].p, e->oldToNew, e->curPhase, e->remain,
              (int) e->cfg.rl_traffic_light, (int) e->nMaskWords, scanTicket, e->vt, e->sc, (const int32_t *) e->finList,
              e->finTerm, (int) e->slotCap, e->jobCount, e->tiled ? (HostMirror *) nullptr : e->hMirror, e->finTicket, nStat,
              e->exactTimes() ? 1 : 0, (const int32_t *) e->cnt[nxt].p, e->finCount);
    HIP_TRY(hipGetLastError());
    e->cur = nxt;
    e->step += 1;
    e->mirrorValid = !e->tiled;
    if (e->hist.num && !e->tiled) {  // Lane::updateHistory, on the committed state
        hipLaunchKernelGGL(k_lane_history, dim3(gridFor(e->L)), dim3(kBlock), 0, st, e->ctx(), e->hist);
        HIP_TRY(hipGetLastError());
    }
    return CFX_OK;
}
extern "C" {

// Lane::history as the ABI shows it: lane-major, oldest record first (the device keeps a ring per lane, record-major)
int32_t cfx_get_lane_history(cfx_engine *e, cfx_lane_history *out) {
    if (!e || !out) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (!e->hist.num) return e->fail("cfx_get_lane_history: the engine was created without cfx_config::lane_history"), CFX_ERR_STATE;
    if (out->n_lanes != e->L) return e->fail("cfx_get_lane_history: n_lanes"), CFX_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle()) return rcSettle;
    // (the rings are turned into the ABI's arrays on the device: 42 MB at 30x30, which a strided loop on the host took 20 ms over)
    const size_t L = (size_t) e->L, M = (size_t) kLaneHistoryMax;
    int32_t *dNum = nullptr;
    double *dAvg = nullptr;
    if (hipMalloc((void **) &dNum, M * L * sizeof(int32_t)) != hipSuccess || hipMalloc((void **) &dAvg, M * L * sizeof(double)) != hipSuccess) {
        (void) hipFree(dNum);
        return e->fail("cfx_get_lane_history: no device memory for the staging arrays"), CFX_ERR_DEVICE;
    }
    hipLaunchKernelGGL(k_hist_export, dim3((unsigned) ((M * L + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, e->hist, dNum, dAvg);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(out->vehicle_num, dNum, M * L * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out->average_speed, dAvg, M * L * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out->len, e->hist.len, L * 4, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out->history_vehicle_num, e->hist.hNum, L * 4, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out->history_average_speed, e->hist.hAvg, L * 8, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void) hipFree(dNum);
    (void) hipFree(dAvg);
    HIP_TRY(he);
    return CFX_OK;
}

int32_t cfx_set_lane_history(cfx_engine *e, const cfx_lane_history *in) {
    if (!e || !in) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (!e->hist.num) return e->fail("cfx_set_lane_history: the engine was created without cfx_config::lane_history"), CFX_ERR_STATE;
    if (in->n_lanes != e->L) return e->fail("cfx_set_lane_history: n_lanes"), CFX_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle()) return rcSettle;
    const size_t L = (size_t) e->L, M = (size_t) kLaneHistoryMax;
    for (size_t l = 0; l < L; ++l)
        if (in->len[l] < 0 || in->len[l] > kLaneHistoryMax) return e->fail("cfx_set_lane_history: len out of range"), CFX_ERR_INVALID;
    int32_t *dNum = nullptr;
    double *dAvg = nullptr;
    if (hipMalloc((void **) &dNum, M * L * sizeof(int32_t)) != hipSuccess || hipMalloc((void **) &dAvg, M * L * sizeof(double)) != hipSuccess) {
        (void) hipFree(dNum);
        return e->fail("cfx_set_lane_history: no device memory for the staging arrays"), CFX_ERR_DEVICE;
    }
    hipError_t he = hipMemcpyAsync(dNum, in->vehicle_num, M * L * sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(dAvg, in->average_speed, M * L * sizeof(double), hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(e->hist.len, in->len, L * 4, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(e->hist.hNum, in->history_vehicle_num, L * 4, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(e->hist.hAvg, in->history_average_speed, L * 8, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(k_hist_import, dim3((unsigned) ((M * L + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, e->hist, (const int32_t *) dNum,
                           (const double *) dAvg);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void) hipFree(dNum);
    (void) hipFree(dAvg);
    HIP_TRY(he);
    return CFX_OK;
}

int32_t cfx_sync(cfx_engine *e) {
    if (!e) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!e->retired.empty()) return e->freeRetired();  // (arrays a growing vehicle table left behind: the stream is idle now)
    return CFX_OK;
}

int32_t cfx_reset(cfx_engine *e) {
    if (!e) return CFX_ERR_INVALID;
    (void) hipSetDevice(e->device);
    return e->resetState();
}

int32_t cfx_set_tl_phase(cfx_engine *e, int32_t inter, int32_t phase) {
    if (!e) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (inter < 0 || inter >= e->I) {
        e->err = "cfx_set_tl_phase: intersection index out of range";
        return CFX_ERR_INVALID;
    }
    // range check against the host copy is the caller's job for speed; here only non-negativity
    if (phase < 0) {
        e->err = "cfx_set_tl_phase: negative phase";
        return CFX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    // TrafficLight::setPhase trafficlight.cpp:39-41 (remainDuration untouched); ordered on the stream
    HIP_TRY(hipMemcpyAsync(e->curPhase + inter, &phase, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CFX_OK;
}

int32_t cfx_set_tl_phases(cfx_engine *e, int32_t n, const int32_t *inters, const int32_t *phases) {
    if (!e || n < 0 || (n && (!inters || !phases))) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    for (int i = 0; i < n; ++i)
        if (inters[i] < 0 || inters[i] >= e->I || phases[i] < 0) {
            e->err = "cfx_set_tl_phases: index out of range";
            return CFX_ERR_INVALID;
        }
    // One pinned staging buffer per call slot (ring of kStages), uploaded asynchronously on the engine's stream and
    // scattered by a tiny kernel: no synchronisation, the RL loop keeps running ahead of the device.
    const size_t need = (size_t) n * 2;
    if (need > e->phaseStageCap) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        size_t nc = std::max<size_t>(need * 2, 4096);
        for (int i = 0; i < cfx_engine::kStages; ++i) {
            if (e->hPhaseStage[i]) HIP_TRY(hipHostFree(e->hPhaseStage[i]));
            HIP_TRY(hipHostMalloc((void **) &e->hPhaseStage[i], nc * sizeof(int32_t), hipHostMallocDefault));
            if (!e->phaseStageEvent[i]) HIP_TRY(hipEventCreateWithFlags(&e->phaseStageEvent[i], hipEventDisableTiming));
            e->phaseStageBusy[i] = false;
        }
        e->phaseStageCap = nc;
    }
    const int si = e->phaseStageIdx;
    e->phaseStageIdx = (si + 1) % cfx_engine::kStages;
    if (e->phaseStageBusy[si]) HIP_TRY(hipEventSynchronize(e->phaseStageEvent[si]));
    // An intersection named more than once keeps the LAST phase, as successive TrafficLight::setPhase calls would
    // (trafficlight.cpp:39-41; a host that collects set_tl_phase calls between two steps hands them over in one call): each
    // intersection goes to the device once — k_set_phases writes one thread per pair, and two threads writing one
    // intersection's phase would leave whichever came last in time, not in the list.
    if (e->phaseSeen.size() != (size_t) e->I) e->phaseSeen.assign((size_t) e->I, 0);
    e->phaseCall += 1;
    if (e->phaseCall == INT32_MAX) {
        e->phaseSeen.assign((size_t) e->I, 0);
        e->phaseCall = 1;
    }
    {
        int32_t *outI = e->hPhaseStage[si], *outP = e->hPhaseStage[si] + n;
        int m = 0;
        for (int i = n - 1; i >= 0; --i) {  // from the back: the first one seen is the one that wins
            if (e->phaseSeen[(size_t) inters[i]] == e->phaseCall) continue;
            e->phaseSeen[(size_t) inters[i]] = e->phaseCall;
            outI[m] = inters[i];
            outP[m] = phases[i];
            ++m;
        }
        if (m < n) memmove(outI + m, outP, (size_t) m * sizeof(int32_t));  // (the kernel reads pairs[i] and pairs[m + i])
        n = m;
    }
    if (n) hipLaunchKernelGGL(k_set_phases, dim3(gridFor(n)), dim3(kBlock), 0, e->stream, e->hPhaseStage[si], n, e->curPhase);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->phaseStageEvent[si], e->stream));
    e->phaseStageBusy[si] = true;
    return CFX_OK;
}

int32_t cfx_get_tl_state(cfx_engine *e, int32_t *phase, double *remain) {
    if (!e) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    if (phase) HIP_TRY(hipMemcpyAsync(phase, e->curPhase, e->I * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (remain) HIP_TRY(hipMemcpyAsync(remain, e->remain, e->I * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CFX_OK;
}

int32_t cfx_get_scalars(cfx_engine *e, cfx_scalars *out) {
    if (!e || !out) return CFX_ERR_INVALID;
    (void) hipSetDevice(e->device);
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    DevScalars s;
    int rc = e->readScalars(s);
    if (rc) return rc;
    out->step = e->step;
    out->active_vehicle_count = s.active;
    out->finished_vehicle_count = s.finishedCnt;
    out->spawned_vehicle_count = e->spawned;
    out->cumulative_travel_time = s.cumulativeTravelTime;
    out->live_enter_time_sum = 0.0;  // not maintained on the device path
    out->vehicle_steps = s.vehicleSteps;
    out->tie_events = s.tieEvents;
    for (int i = 0; i < 8; ++i) out->tie_drivables[i] = s.tieEvents > i ? s.tieDrv[i] : -1;
    out->diag_cross_jobs = s.nCrossJobs;
    out->dropped_future_speeds = (int32_t) std::min<int64_t>(e->droppedFutureSpeeds, INT32_MAX);
    return CFX_OK;
}

int32_t cfx_get_layout(cfx_engine *e) { return !e ? CFX_ERR_INVALID : (e->ring ? CFX_LAYOUT_RING : CFX_LAYOUT_DENSE); }
int32_t cfx_get_ring_info(cfx_engine *e, int64_t *slots, int32_t *scale) {
    if (!e) return CFX_ERR_INVALID;
    if (slots) *slots = e->ring ? (int64_t) e->ringSlots : 0;
    if (scale) *scale = e->ringScale;
    return CFX_OK;
}

int32_t cfx_get_lane_counts(cfx_engine *e, int32_t *out) {
    if (!e || !out) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    if (e->ring && !e->ringBuilt) {  // nothing has run yet
        memset(out, 0, e->L * sizeof(int32_t));
        return CFX_OK;
    }
    e->observeIdle = 0;
    if (!(e->hCntValid && e->observing)) {  // first read (or first after a load / reset / pause): fetch, and have the steps publish from now on
        HIP_TRY(hipMemcpyAsync(e->hCnt, e->ring ? e->rCnt : e->cnt[e->cur].p, e->L * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        e->hCntValid = !e->tiled;
        e->observing = !e->tiled;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(out, e->hCnt, e->L * sizeof(int32_t));
    return CFX_OK;
}

int32_t cfx_get_lane_waiting_counts(cfx_engine *e, int32_t *out) {
    if (!e || !out) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    int rc;
    if ((rc = e->syncTables())) return rc;
    if (e->ring && (rc = e->ringEnsure())) return rc;
    if (e->ring) hipLaunchKernelGGL(kr_lane_waiting, dim3(gridFor(e->L)), dim3(kBlock), 0, e->stream, e->rctx(), e->laneOut);
    else hipLaunchKernelGGL(k_lane_waiting, dim3(gridFor(e->L)), dim3(kBlock), 0, e->stream, e->ctx(), e->laneOut);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->hLaneOut, e->laneOut, e->L * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(out, e->hLaneOut, e->L * sizeof(int32_t));
    return CFX_OK;
}

int32_t cfx_get_vehicles(cfx_engine *e, cfx_vehicle_view *view) {
    if (!e || !view) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    int rc;
    if ((rc = e->syncTables())) return rc;
    if (e->ring) {
        // the ring order as dense arrays (kr_gather), then only the columns the caller asked for
        if ((rc = e->ringEnsure())) return rc;
        int32_t n = 0;
        if ((rc = e->ringGather(view->leader_vid || view->gap, &n))) return rc;
        view->count = n;
        if (n > view->capacity) {
            e->err = "cfx_get_vehicles: capacity too small";
            return CFX_ERR_CAPACITY;
        }
        auto dl = [&](void *dst, const void *src, size_t bytes) {
            return (dst && bytes) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream) : hipSuccess;
        };
        const size_t sn = (size_t) n;
        HIP_TRY(dl(view->vid, e->rd.vid, sn * 4));
        HIP_TRY(dl(view->drivable, e->rd.drv, sn * 4));
        HIP_TRY(dl(view->prev_drivable, e->rd.prevDrv, sn * 4));
        HIP_TRY(dl(view->leader_vid, e->rd.leaderVid, sn * 4));
        HIP_TRY(dl(view->blocker_vid, e->rd.blockerVid, sn * 4));
        HIP_TRY(dl(view->enter_ll_time, e->rd.enterLLT, sn * 4));
        HIP_TRY(dl(view->route_pos, e->rd.routePos, sn * 4));
        HIP_TRY(dl(view->dis, e->rd.dis, sn * 8));
        HIP_TRY(dl(view->speed, e->rd.speed, sn * 8));
        HIP_TRY(dl(view->gap, e->rd.gap, sn * 8));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int i = 0; i < n; ++i) {  // no lane change on this layout
            if (view->lc_partner_vid) view->lc_partner_vid[i] = -1;
            if (view->lc_flags) view->lc_flags[i] = 0;
            if (view->lc_offset) view->lc_offset[i] = 0.0;
            if (view->lc_last_dir) view->lc_last_dir[i] = 0;
            if (view->lc_target_lane) view->lc_target_lane[i] = -1;
            if (view->lc_direction) view->lc_direction[i] = 0;
            if (view->lc_last_change_time) view->lc_last_change_time[i] = 0.0;
            if (view->lc_waiting_time) view->lc_waiting_time[i] = 0.0;
        }
        return CFX_OK;
    }
    // number of slots of the current generation
    int32_t S = 0;
    if (e->mirrorValid) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        S = e->hMirror->slots;
    } else {
        HIP_TRY(hipMemcpyAsync(&S, e->segStart[e->cur].p + e->D, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    // only what the caller asked for is computed and copied (the string getters want vid + one column)
    const bool wantLeader = view->leader_vid || view->gap, wantBlocker = view->blocker_vid != nullptr;
    if (wantLeader) {
        StepCtx c = e->ctx();
        hipLaunchKernelGGL(k_leader_view, dim3(gridStride(S)), dim3(kBlock), 0, e->stream, c, e->viewLeader, e->viewGap);
        HIP_TRY(hipGetLastError());
    }
    const SlotArrays &g = e->gen[e->cur];
    std::vector<int32_t> vid(S), drv, prev, blk, ellt, rpos, lead, o2n;
    std::vector<double> dis, speed, gap;
    auto dl = [&](void *dst, const void *src, size_t bytes) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream);
    };
    if (S) {
        HIP_TRY(dl(vid.data(), g.vid, S * 4));
#define WANT(field, vec, src, bytes) \
    if (field) {                      \
        vec.resize(S);                \
        HIP_TRY(dl(vec.data(), src, (size_t) S * bytes)); \
    }
        WANT(view->drivable, drv, g.drv, 4)
        WANT(view->prev_drivable, prev, g.prevDrv, 4)
        WANT(wantBlocker, blk, g.blocker, 4)
        WANT(view->enter_ll_time, ellt, g.enterLLT, 4)
        WANT(view->route_pos, rpos, g.routePos, 4)
        WANT(view->leader_vid, lead, e->viewLeader, 4)
        WANT(view->dis, dis, g.dis, 8)
        WANT(view->speed, speed, g.speed, 8)
        WANT(view->gap, gap, e->viewGap, 8)
#undef WANT
        if (wantBlocker) {
            o2n.resize(e->slotCap);
            HIP_TRY(dl(o2n.data(), e->oldToNew, e->slotCap * 4));
        }
    }
    // lane change: per-vehicle state lives in vid-indexed tables
    const bool wantLc = e->lc.on && (view->lc_partner_vid || view->lc_flags || view->lc_offset || view->lc_last_dir ||
                                     view->lc_target_lane || view->lc_direction || view->lc_last_change_time || view->gap);
    std::vector<int8_t> lcType, lcChanging, lcLastDir, lcSendDir;
    std::vector<int32_t> lcPartner, lcTarget;
    std::vector<double> lcOffset, lcLastTime, lcGap;
    if (wantLc) {
        const size_t nv = (size_t) e->spawned;
        lcType.resize(nv); lcChanging.resize(nv); lcLastDir.resize(nv); lcSendDir.resize(nv);
        lcPartner.resize(nv); lcTarget.resize(nv); lcOffset.resize(nv); lcLastTime.resize(nv); lcGap.resize(nv);
        if (nv) {
            HIP_TRY(dl(lcType.data(), e->lc.ptype, nv));
            HIP_TRY(dl(lcChanging.data(), e->lc.changing, nv));
            HIP_TRY(dl(lcLastDir.data(), e->lc.lastDir, nv));
            HIP_TRY(dl(lcSendDir.data(), e->lc.sendDir, nv));
            HIP_TRY(dl(lcPartner.data(), e->lc.partner, nv * 4));
            HIP_TRY(dl(lcTarget.data(), e->lc.sendTarget, nv * 4));
            HIP_TRY(dl(lcOffset.data(), e->lc.offset, nv * 8));
            HIP_TRY(dl(lcLastTime.data(), e->lc.lastChangeTime, nv * 8));
            HIP_TRY(dl(lcGap.data(), e->lc.gap, nv * 8));
        }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    int n = 0;
    for (int s = 0; s < S; ++s) n += vid[s] >= 0;
    view->count = n;
    if (n > view->capacity) {
        e->err = "cfx_get_vehicles: capacity too small";
        return CFX_ERR_CAPACITY;
    }
    int i = 0;
    for (int s = 0; s < S; ++s) {
        if (vid[s] < 0) continue;
        if (view->vid) view->vid[i] = vid[s];
        if (view->drivable) view->drivable[i] = drv[s];
        if (view->prev_drivable) view->prev_drivable[i] = prev[s];
        if (view->leader_vid) view->leader_vid[i] = lead[s] >= 0 ? vid[lead[s]] : -1;
        if (view->blocker_vid) {
            if (blk[s] <= -2) {  // kept by vehicle id (a proxy on a ghost lane, see k_cross)
                view->blocker_vid[i] = -blk[s] - 2;
            } else {
                int b = blk[s] >= 0 ? o2n[blk[s]] : -1;
                int bv = b >= 0 ? vid[b] : -1;
                view->blocker_vid[i] = bv <= -2 ? -bv - 2 : bv;  // tombstone of a vehicle that migrated to a neighbour
            }
        }
        if (view->enter_ll_time) view->enter_ll_time[i] = ellt[s];
        if (view->route_pos) view->route_pos[i] = rpos[s];
        if (view->dis) view->dis[i] = dis[s];
        if (view->speed) view->speed[i] = speed[s];
        if (view->gap) view->gap[i] = gap[s];
        const int v = vid[s];
        if (wantLc && view->gap && lead[s] < 0) view->gap[i] = lcGap[v];  // ControllerInfo::gap keeps its last value
        const bool chg = wantLc && lcChanging[v];
        if (view->lc_partner_vid) view->lc_partner_vid[i] = wantLc ? lcPartner[v] : -1;
        if (view->lc_flags)
            view->lc_flags[i] = wantLc ? (uint8_t) ((lcType[v] == 2 ? CFX_LC_SHADOW : 0) | (lcType[v] == 1 ? CFX_LC_PARENT : 0) |
                                                   (chg ? CFX_LC_CHANGING : 0))
                                       : 0;
        if (view->lc_offset) view->lc_offset[i] = wantLc ? lcOffset[v] : 0.0;
        if (view->lc_last_dir) view->lc_last_dir[i] = wantLc ? lcLastDir[v] : 0;
        if (view->lc_target_lane) view->lc_target_lane[i] = chg ? lcTarget[v] : -1;
        if (view->lc_direction) view->lc_direction[i] = chg ? lcSendDir[v] : 0;
        if (view->lc_last_change_time) view->lc_last_change_time[i] = wantLc ? lcLastTime[v] : 0.0;
        if (view->lc_waiting_time) view->lc_waiting_time[i] = 0.0;  // LaneChange::waitingTime is write-only in the reference; not kept
        ++i;
    }
    return CFX_OK;
}

int32_t cfx_get_vehicle_status(cfx_engine *e, int32_t first, int32_t n, uint8_t *out) {
    if (!e || !out || first < 0 || n < 0 || (int64_t) first + n > e->spawned) {
        if (e) e->err = "cfx_get_vehicle_status: range out of bounds";
        return CFX_ERR_INVALID;
    }
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    const auto q0 = std::chrono::steady_clock::now();
    auto qus = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    const auto q1 = std::chrono::steady_clock::now();
    // A few bytes (the spawner's priority-collision query asks for ONE vehicle, in the middle of a run): through the engine's
    // pinned scratch — a device-to-host copy into the caller's pageable memory makes the runtime stage or pin it per call.
    // The call is timed (cfx_get_host_stats): behind a free-running host its cost is the wait for the stream's backlog.
    const size_t scratchBytes = std::max<size_t>((size_t) e->L, 2) * sizeof(int32_t);
    if (n && (size_t) n <= scratchBytes) {
        HIP_TRY(hipMemcpyAsync(e->hLaneOut, e->vt.state + first, (size_t) n, hipMemcpyDeviceToHost, e->stream));
        const auto q2 = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamSynchronize(e->stream));
        const auto q3 = std::chrono::steady_clock::now();
        memcpy(out, e->hLaneOut, (size_t) n);
        cfx_host_stats &hs = e->hostStats;
        hs.status_queries += 1;
        if (qus(q0, q3) > hs.worst_status_query_us) {
            hs.worst_status_query_us = qus(q0, q3);
            hs.worst_status_query_settle_us = qus(q0, q1);
            hs.worst_status_query_copy_us = qus(q1, q2);
            hs.worst_status_query_wait_us = qus(q2, q3);
        }
        return CFX_OK;
    }
    if (n) HIP_TRY(hipMemcpyAsync(out, e->vt.state + first, (size_t) n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CFX_OK;
}

int32_t cfx_get_waiting(cfx_engine *e, int32_t capacity, int32_t *vid, int32_t *lane, int32_t *nOut) {
    if (!e || !nOut) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    // The waiting FIFOs are linked lists through the vid table; walk them on the host (debug / API path).
    std::vector<int32_t> head(e->L), next((size_t) e->spawned);
    HIP_TRY(hipMemcpyAsync(head.data(), e->waitHead, e->L * 4, hipMemcpyDeviceToHost, e->stream));
    if (e->spawned)
        HIP_TRY(hipMemcpyAsync(next.data(), e->vt.nextWait, (size_t) e->spawned * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    int i = 0;
    for (int l = 0; l < e->L; ++l)
        for (int v = head[l]; v >= 0; v = next[v]) {
            if (i >= capacity) {
                e->err = "cfx_get_waiting: capacity too small";
                return CFX_ERR_CAPACITY;
            }
            if (vid) vid[i] = v;
            if (lane) lane[i] = l;
            ++i;
        }
    *nOut = i;
    return CFX_OK;
}

int32_t cfx_set_vehicle_speed(cfx_engine *e, int32_t vid, double speed) {
    if (!e) return CFX_ERR_INVALID;
    if (vid < 0 || vid >= e->spawned + CFX_FUTURE_SPEED_WINDOW) {
        e->err = "cfx_set_vehicle_speed: no such vehicle";
        return CFX_ERR_INVALID;
    }
    if (vid >= e->spawned) {
        // a vehicle the next spawn records will create (pushed since the last step): kept until then, see cfx_step — which
        // counts the speeds whose vehicle its records did not create (cfx_scalars::dropped_future_speeds)
        e->futureCustom[vid] = speed;
        return CFX_OK;
    }
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    uint8_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, e->vt.state + vid, 1, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (st == 2) {
        e->err = "cfx_set_vehicle_speed: vehicle already finished";
        return CFX_ERR_INVALID;
    }
    HIP_TRY(hipMemcpyAsync(e->vt.customSpeed + vid, &speed, sizeof(double), hipMemcpyHostToDevice, e->stream));
    if (st == 0) {
        uint8_t one = 1;
        HIP_TRY(hipMemcpyAsync(e->vt.pendingCustom + vid, &one, 1, hipMemcpyHostToDevice, e->stream));
    } else {
        int rc = e->syncTables();
        if (rc) return rc;
        if (e->ring && (rc = e->ringEnsure())) return rc;
        if (e->ring) hipLaunchKernelGGL(kr_set_speed, dim3(1), dim3(1), 0, e->stream, e->rctx(), vid);
        else hipLaunchKernelGGL(k_set_speed, dim3(gridStride(e->slotCap)), dim3(kBlock), 0, e->stream, e->ctx(), vid);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(e->stream));  // the two sources above live on this stack frame
    return CFX_OK;
}

int32_t cfx_set_vehicle_route(cfx_engine *e, int32_t vid, int32_t route) {
    if (!e || vid < 0 || vid >= e->spawned || route < 0 || route + 1 >= (int32_t) e->hRouteStart.size()) {
        if (e) e->err = "cfx_set_vehicle_route: bad vehicle or route";
        return CFX_ERR_INVALID;
    }
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    int rc = e->syncTables();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(e->vt.route + vid, &route, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) (e->vt.firstNext + vid), kFirstNextUnknown, 1, e->stream));  // (a waiting vehicle: its admission walks the new route)
    if (e->ring) {
        if ((rc = e->ringEnsure())) return rc;
        uint8_t st = 0;
        HIP_TRY(hipMemcpyAsync(&st, e->vt.state + vid, 1, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (st == 1) hipLaunchKernelGGL(kr_set_route, dim3(1), dim3(1), 0, e->stream, e->rctx(), vid, route);
    } else {
        hipLaunchKernelGGL(k_set_route, dim3(gridStride(e->slotCap)), dim3(kBlock), 0, e->stream, e->ctx(), vid, route);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CFX_OK;
}

int32_t cfx_get_vehicle(cfx_engine *e, int32_t vid, int32_t *state, int32_t *drivable, int32_t *routePos, int32_t *route) {
    if (!e || vid < 0 || vid >= e->spawned) {
        if (e) e->err = "cfx_get_vehicle: vid out of range";
        return CFX_ERR_INVALID;
    }
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    int rc = e->syncTables();
    if (rc) return rc;
    uint8_t st = 0;
    int32_t r = -1, found[2] = {-1, -1};
    HIP_TRY(hipMemcpyAsync(&st, e->vt.state + vid, 1, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(&r, e->vt.route + vid, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemsetAsync(e->laneOut, 0xFF, 8, e->stream));
    if (e->ring) {
        if ((rc = e->ringEnsure())) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (st == 1) hipLaunchKernelGGL(kr_find_vehicle, dim3(1), dim3(1), 0, e->stream, e->rctx(), vid, e->laneOut);
    } else {
        hipLaunchKernelGGL(k_find_vehicle, dim3(gridStride(e->slotCap)), dim3(kBlock), 0, e->stream, e->ctx(), vid, e->laneOut);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(found, e->laneOut, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (state) *state = st;
    if (drivable) *drivable = st == 1 ? found[0] : -1;
    if (routePos) *routePos = st == 1 ? found[1] : -1;
    if (route) *route = r;
    return CFX_OK;
}

// Lane change: the priorities the step's shadows will get, and who got one (include/cityflow_amd.h "Lane change")
int32_t cfx_lane_change_supply(cfx_engine *e, int32_t n, const int32_t *priorities) {
    if (!e || n < 0 || (n && !priorities)) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (!e->lc.on) return e->fail("cfx_lane_change_supply: this engine was created without lane change"), CFX_ERR_STATE;
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if (e->pollPending) return e->fail("cfx_lane_change_supply: the previous lane-change step was not polled"), CFX_ERR_STATE;
    const int nEnvs = e->cfg.n_envs > 1 ? e->cfg.n_envs : 1;
    if (n % nEnvs != 0) return e->fail("cfx_lane_change_supply: n must be a multiple of cfx_config::n_envs"), CFX_ERR_INVALID;
    e->lc.poolPerEnv = n / nEnvs;
    if (n > e->lc.insCap || !e->hPool) {  // (re)allocate pool, records and the landing buffer of the poll
        HIP_TRY(hipStreamSynchronize(e->stream));
        const size_t cap = std::max<size_t>((size_t) n, 1024);
        if ((rc = e->grow(&e->lc.ins, 0, cap))) return rc;
        if ((rc = e->grow(&e->lc.insNext, 0, cap))) return rc;
        if ((rc = e->grow(&e->lc.insKey, 0, cap))) return rc;
        if ((rc = e->grow(&e->lc.fixList, 0, 3 * 8 * cap))) return rc;
        e->lc.fixCap = (int) (8 * cap);
        if (e->hPool) HIP_TRY(hipHostFree(e->hPool));
        if (e->hPoll) HIP_TRY(hipHostFree(e->hPoll));
        HIP_TRY(hipHostMalloc((void **) &e->hPool, cap * sizeof(int32_t), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **) &e->hPoll, (cap + 2) * sizeof(int32_t), hipHostMallocDefault));
        e->lc.insCap = (int) cap;
    }
    // (the previous step was polled, so the device is done with the pinned staging buffer)
    memcpy(e->hPool, priorities, (size_t) n * sizeof(int32_t));
    e->poolN = n;
    return CFX_OK;
}

int32_t cfx_lane_change_poll(cfx_engine *e, int32_t capacity, int32_t *parent_vid, int32_t *n) {
    if (!e || !n || capacity < 0 || (capacity && !parent_vid)) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (!e->lc.on) return e->fail("cfx_lane_change_poll: this engine was created without lane change"), CFX_ERR_STATE;
    *n = 0;
    if (!e->pollPending) return CFX_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->pollEvent));  // plan + schedule + assign of the step; the rest keeps running
    e->pollPending = false;
    const int k = e->hPoll[0];
    if (e->hPoll[1] == 6) {
        e->err = "lane change: a capacity of the schedule walk was exceeded (shadows per road / members of a segment / "
                 "provisional neighbours in one step)";
        return CFX_ERR_CAPACITY;
    }
    if (k > e->poolN) {
        e->err = "lane change: more shadows in one step than priorities supplied (cfx_lane_change_supply)";
        return CFX_ERR_CAPACITY;
    }
    if (k > capacity) {
        e->err = "cfx_lane_change_poll: capacity too small";
        return CFX_ERR_CAPACITY;
    }
    for (int i = 0; i < k; ++i) parent_vid[i] = e->hPoll[2 + i];
    *n = k;
    e->spawned += k;  // shadows are vehicles: the next step's spawn records are numbered after them
    e->spawnedHere += k;
    e->liveUpper += k;
    return CFX_OK;
}

int32_t cfx_get_custom_speeds(cfx_engine *e, int32_t capacity, double *out) {
    if (!e || !out) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (int rcSettle = e->settle(false)) return rcSettle;  // (ring layout: a commit deferred to the next step's admission; Lane::history is not read here)
    if (e->ring) {
        int rc;
        if ((rc = e->syncTables()) || (rc = e->ringEnsure())) return rc;
        int32_t n = 0;
        if ((rc = e->ringGather(false, &n))) return rc;
        if (n > capacity) return CFX_ERR_CAPACITY;
        std::vector<int32_t> vid((size_t) n);
        std::vector<uint8_t> flags((size_t) n);
        std::vector<double> cs((size_t) e->spawned);
        if (n) {
            HIP_TRY(hipMemcpyAsync(vid.data(), e->rd.vid, (size_t) n * 4, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync(flags.data(), e->rd.flags, (size_t) n, hipMemcpyDeviceToHost, e->stream));
        }
        if (e->spawned) HIP_TRY(hipMemcpyAsync(cs.data(), e->vt.customSpeed, (size_t) e->spawned * 8, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int i = 0; i < n; ++i) out[i] = (flags[i] & 1) ? cs[vid[i]] : __builtin_nan("");
        return CFX_OK;
    }
    int32_t S = 0;
    HIP_TRY(hipMemcpyAsync(&S, e->segStart[e->cur].p + e->D, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<int32_t> vid(S);
    std::vector<uint8_t> flags(S);
    std::vector<double> cs((size_t) e->spawned);
    if (S) {
        HIP_TRY(hipMemcpyAsync(vid.data(), e->gen[e->cur].vid, S * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(flags.data(), e->gen[e->cur].flags, S, hipMemcpyDeviceToHost, e->stream));
    }
    if (e->spawned) HIP_TRY(hipMemcpyAsync(cs.data(), e->vt.customSpeed, (size_t) e->spawned * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    int i = 0;
    for (int s = 0; s < S; ++s) {
        if (vid[s] < 0) continue;
        if (i >= capacity) return CFX_ERR_CAPACITY;
        out[i++] = (flags[s] & 1) ? cs[vid[s]] : __builtin_nan("");
    }
    return CFX_OK;
}

// Archive::resume (archive.cpp:73-126): replace the whole dynamic state.
int32_t cfx_load_state(cfx_engine *e, const cfx_state *s) {
    if (!e || !s) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = e->syncTables())) return rc;
    if ((rc = e->resetState())) return rc;
    const int L = e->L, D = e->D, nV = s->n_vehicles, nR = s->n_running;
    if ((rc = e->ensureVidCap((size_t) nV + 1))) return rc;
    if (e->ring && (rc = e->ringEnsure())) return rc;
    if (!e->ring && (rc = e->ensureSlotCap((size_t) nR + (e->tiled ? (size_t) e->spareTotal : (size_t) L) + 1))) return rc;
    // ---- layout: vehicles of drivable d, then one spare slot for lanes
    std::vector<int32_t> cnt(D, 0), segStart(D + 1, 0);
    for (int i = 0; i < nR; ++i) {
        int d = s->r_drivable[i];
        if (d < 0 || d >= D || (i && d < s->r_drivable[i - 1])) return e->fail("cfx_load_state: running vehicles must be grouped by drivable, ascending");
        cnt[d]++;
    }
    auto up = [&](void *dst, const void *src, size_t bytes) {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream) : hipSuccess;
    };
    std::vector<double> customR(std::max(nV, 1), 0.0);
    // ControllerInfo::gap as the state carries it: used instead of the derived gap in the first step (kFlagStateGap, cfx_device.h)
    std::vector<double> gapState(std::max(nV, 1), 0.0);
    auto stateGapOf = [&](int i, int v) {
        if (!s->r_gap || !(s->r_gap[i] == s->r_gap[i])) return 0;
        gapState[v] = s->r_gap[i];
        return kFlagStateGap;
    };
    if (e->ring) {
        // the caller's arrays ARE the dense staging view (Drivable::vehicles order); kr_scatter_in (below, once the vehicle
        // table is on the device) puts them on the rings
        // rings that are too small for the archive (capacities from a fraction of the bound, or an archive from a denser
        // state than this engine has seen): double every capacity until it fits — the growth path of a running engine
        for (;;) {
            bool fits = true;
            for (int d = 0; d < D && fits; ++d) fits = cnt[d] + std::min(8, (e->hRingGeo[d].y + 1) / 2) <= e->hRingGeo[d].y;
            if (fits) break;
            if (e->ringScale >= 4096) return e->fail("cfx_load_state: more vehicles on one drivable than its ring can hold");
            e->ringGrowRequested = true;
            if ((rc = e->ringEnsure())) return rc;
        }
        for (int d = 0; d < D; ++d) {
            segStart[d + 1] = segStart[d] + cnt[d];
            if (cnt[d] > e->hRingGeo[d].y) return e->fail("cfx_load_state: more vehicles on one drivable than its ring holds");
        }
        if ((rc = e->ensureDense((size_t) nR))) return rc;
        std::vector<int32_t> blk((size_t) std::max(nR, 1), -1);
        std::vector<uint8_t> flags((size_t) std::max(nR, 1), 0);
        for (int i = 0; i < nR; ++i) {
            const int v = s->r_vid[i];
            if (v < 0 || v >= nV) return e->fail("cfx_load_state: running vid out of range");
            const int b = s->r_blocker_vid[i];
            blk[i] = (b >= 0 && b < nV) ? b : -1;
            if (s->r_custom_speed && s->r_custom_speed[i] == s->r_custom_speed[i]) {
                flags[i] = 1;
                customR[v] = s->r_custom_speed[i];
            }
            flags[i] |= (uint8_t) stateGapOf(i, v);
        }
        HIP_TRY(up(e->rOff, segStart.data(), ((size_t) D + 1) * 4));
        HIP_TRY(up(e->rd.vid, s->r_vid, (size_t) nR * 4));
        HIP_TRY(up(e->rd.prevDrv, s->r_prev_drivable, (size_t) nR * 4));
        HIP_TRY(up(e->rd.blockerVid, blk.data(), (size_t) nR * 4));
        HIP_TRY(up(e->rd.enterLLT, s->r_enter_ll_time, (size_t) nR * 4));
        HIP_TRY(up(e->rd.routePos, s->r_route_pos, (size_t) nR * 4));
        HIP_TRY(up(e->rd.flags, flags.data(), (size_t) nR));
        HIP_TRY(up(e->rd.dis, s->r_dis, (size_t) nR * 8));
        HIP_TRY(up(e->rd.speed, s->r_speed, (size_t) nR * 8));
        HIP_TRY(hipStreamSynchronize(e->stream));  // blk / flags die with this scope
    }
    // (a tile's lanes own laneSpare[l] empty slots each: room for a step's migrants on import lanes)
    for (int d = 0; d < D && !e->ring; ++d) segStart[d + 1] = segStart[d] + cnt[d] + (d < L ? (e->tiled ? (int) e->hLaneSpare[d] : 1) : 0);
    const int S = e->ring ? 0 : segStart[D];
    std::vector<int32_t> vid(S, -1), drv(S, -1), prev(S, -1), next(S, -1), blk(S, -1), ellt(S, CFX_INT_MAX), rpos(S, 0),
        templ(S, 0), route(S, 0), slotOfVid(std::max(nV, 1), -1);
    std::vector<uint8_t> flags(S, 0);
    std::vector<double> dis(S, 0.0), speed(S, 0.0), custom(std::max(nV, 1), 0.0);
    if (!e->ring) {
        std::vector<int32_t> fill(D, 0);
        for (int i = 0; i < nR; ++i) {
            int d = s->r_drivable[i];
            int sl = segStart[d] + fill[d]++;
            int v = s->r_vid[i];
            if (v < 0 || v >= nV) return e->fail("cfx_load_state: running vid out of range");
            slotOfVid[v] = sl;
            vid[sl] = v;
            drv[sl] = d;
            prev[sl] = s->r_prev_drivable[i];
            ellt[sl] = s->r_enter_ll_time[i];
            rpos[sl] = s->r_route_pos[i];
            templ[sl] = s->v_templ[v];
            route[sl] = s->v_route[v];
            dis[sl] = s->r_dis[i];
            speed[sl] = s->r_speed[i];
            if (s->r_custom_speed && s->r_custom_speed[i] == s->r_custom_speed[i]) {
                flags[sl] = 1;
                custom[v] = s->r_custom_speed[i];
            }
            flags[sl] |= (uint8_t) stateGapOf(i, v);
        }
        for (int i = 0; i < nR; ++i) {  // blockers: vid -> slot (oldToNew is reset to identity below)
            int b = s->r_blocker_vid[i];
            int bs = (b >= 0 && b < nV) ? slotOfVid[b] : -1;
            // tiling: a blocker that is the proxy on a ghost lane is kept by vehicle id (keepBlocker, cfx_kernels.h)
            if (bs >= 0 && e->tiled && drv[bs] < L && e->hLaneGhost[drv[bs]]) bs = -(b + 2);
            blk[slotOfVid[s->r_vid[i]]] = bs;
        }
    }
    // ---- uploads
    SlotArrays &g = e->gen[e->cur];
    if (e->ring) custom = customR;
    if (!e->ring) {
    HIP_TRY(up(e->segStart[e->cur].p, segStart.data(), (D + 1) * 4));
    HIP_TRY(up(e->cnt[e->cur].p, cnt.data(), D * 4));
    HIP_TRY(up(g.vid, vid.data(), S * 4));
    HIP_TRY(up(g.drv, drv.data(), S * 4));
    HIP_TRY(up(g.prevDrv, prev.data(), S * 4));
    HIP_TRY(up(g.blocker, blk.data(), S * 4));
    HIP_TRY(up(g.enterLLT, ellt.data(), S * 4));
    HIP_TRY(up(g.routePos, rpos.data(), S * 4));
    HIP_TRY(up(g.templ, templ.data(), S * 4));
    HIP_TRY(up(g.route, route.data(), S * 4));
    HIP_TRY(up(g.flags, flags.data(), S));
    HIP_TRY(up(g.dis, dis.data(), S * 8));
    HIP_TRY(up(g.speed, speed.data(), S * 8));
    std::vector<int32_t> ident(S);
    for (int i = 0; i < S; ++i) ident[i] = i;
    HIP_TRY(up(e->oldToNew, ident.data(), S * 4));
    if (e->lc.on) HIP_TRY(up(e->lc.newToOld, ident.data(), S * 4));
    HIP_TRY(hipStreamSynchronize(e->stream));
    }
    // vehicle table + waiting FIFOs
    std::vector<int32_t> nextWait(std::max(nV, 1), -1), waitHead(L, -1);
    for (int i = 0; i < s->n_waiting; ++i) {
        int v = s->w_vid[i], lane = s->w_lane[i];
        if (v < 0 || v >= nV || lane < 0 || lane >= L) return e->fail("cfx_load_state: waiting entry out of range");
        if (i > 0 && s->w_lane[i - 1] == lane) nextWait[s->w_vid[i - 1]] = v;
        else waitHead[lane] = v;
    }
    HIP_TRY(up(e->vt.priority, s->v_priority, (size_t) nV * 4));
    HIP_TRY(up(e->vt.templ, s->v_templ, (size_t) nV * 4));
    HIP_TRY(up(e->vt.route, s->v_route, (size_t) nV * 4));
    HIP_TRY(up(e->vt.enterTime, s->v_enter_time, (size_t) nV * 8));
    HIP_TRY(up(e->vt.state, s->v_state, (size_t) nV));
    HIP_TRY(up(e->vt.customSpeed, custom.data(), (size_t) nV * 8));
    HIP_TRY(up(e->vt.gapState, gapState.data(), (size_t) nV * 8));
    HIP_TRY(hipMemsetAsync(e->vt.pendingCustom, 0, std::max(nV, 1), e->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) e->vt.firstNext, kFirstNextUnknown, (size_t) std::max(nV, 1), e->stream));  // (waiting vehicles: the walk)
    HIP_TRY(up(e->vt.nextWait, nextWait.data(), (size_t) nV * 4));
    HIP_TRY(up(e->waitHead, waitHead.data(), (size_t) L * 4));
    HIP_TRY(up(e->curPhase, s->tl_phase, (size_t) e->I * 4));
    HIP_TRY(up(e->remain, s->tl_remain, (size_t) e->I * 8));
    DevScalars sc{};
    sc.active = nR;
    if (e->tiled)  // the vehicle on a ghost lane is the frozen proxy of the owner's tail: not one of this tile's vehicles
        for (int i = 0; i < nR; ++i) sc.active -= s->r_drivable[i] < L && e->hLaneGhost[s->r_drivable[i]];
    sc.finishedCnt = s->finished_vehicle_count;
    sc.cumulativeTravelTime = s->cumulative_travel_time;
    sc.vehicleSteps = s->vehicle_steps;
    e->cumLoaded = s->cumulative_travel_time;
    for (int v = 0; v < nV; ++v)
        if (s->v_state[v] != 2 && !cfx_engine::dyadic(s->v_enter_time[v])) e->timesDyadic = false;
    HIP_TRY(up(e->sc, &sc, sizeof sc));
    e->step = s->step;
    e->spawned = nV;
    e->spawnedHere = nV;
    e->liveUpper = nR + 2 * (int64_t) e->halo.nGhost;
    for (int i = 0; i < s->n_waiting; ++i)
        if (!e->laneQueued[s->w_lane[i]]) {
            e->laneQueued[s->w_lane[i]] = 1;
            e->nQueueLanes += 1;
        }
    e->finishedKnown = s->finished_vehicle_count;
    {
        int64_t inTable = 0;
        for (int v = 0; v < nV; ++v) inTable += s->v_state[v] == 2;
        e->finishedOffset = s->finished_vehicle_count - inTable;
    }
    if (e->lc.on) {
        // lane-change tables: defaults for every vehicle number, then what the archive carries for the running ones (the
        // part of LaneChange / LaneChangeInfo that outlives a step, include/cityflow_amd.h cfx_state)
        const size_t nv = (size_t) std::max(nV, 1);
        std::vector<int8_t> ptype(nv, 0), sig(nv, 0), sdir(nv, 0), surg(nv, 0), ldir(nv, 0), chg(nv, 0), fin(nv, 0);
        std::vector<int32_t> partner(nv, -1), target(nv, -1), none(nv, -1);
        std::vector<double> offset(nv, 0.0), lastTime(nv, 0.0), gapv(nv, 0.0), zero(nv, 0.0);
        for (int i = 0; i < nR; ++i) {
            const int v = s->r_vid[i];
            if (s->r_gap) gapv[v] = s->r_gap[i];
            if (s->r_lc_flags) {
                const uint8_t f = s->r_lc_flags[i];
                ptype[v] = (f & CFX_LC_SHADOW) ? 2 : ((f & CFX_LC_PARENT) ? 1 : 0);
                chg[v] = (f & CFX_LC_CHANGING) ? 1 : 0;
            }
            if (s->r_lc_partner_vid) partner[v] = s->r_lc_partner_vid[i];
            if (s->r_lc_offset) offset[v] = s->r_lc_offset[i];
            if (s->r_lc_last_dir) ldir[v] = (int8_t) s->r_lc_last_dir[i];
            if (s->r_lc_last_change_time) lastTime[v] = s->r_lc_last_change_time[i];
            if (chg[v] && s->r_lc_target_lane && s->r_lc_direction) {  // the signal of a change in progress
                sig[v] = 1;
                surg[v] = 1;
                target[v] = s->r_lc_target_lane[i];
                sdir[v] = (int8_t) s->r_lc_direction[i];
            }
        }
        const LcDev &lc = e->lc;
        HIP_TRY(up(lc.ptype, ptype.data(), nv));
        HIP_TRY(up(lc.sigSend, sig.data(), nv));
        HIP_TRY(up(lc.sendDir, sdir.data(), nv));
        HIP_TRY(up(lc.sendUrg, surg.data(), nv));
        HIP_TRY(up(lc.lastDir, ldir.data(), nv));
        HIP_TRY(up(lc.changing, chg.data(), nv));
        HIP_TRY(up(lc.lcFinished, fin.data(), nv));
        HIP_TRY(up(lc.partner, partner.data(), nv * 4));
        HIP_TRY(up(lc.sendTarget, target.data(), nv * 4));
        HIP_TRY(up(lc.recvFrom, none.data(), nv * 4));
        HIP_TRY(up(lc.tLeader, none.data(), nv * 4));
        HIP_TRY(up(lc.tFollower, none.data(), nv * 4));
        HIP_TRY(up(lc.slotOf, slotOfVid.data(), nv * 4));
        HIP_TRY(up(lc.offset, offset.data(), nv * 8));
        HIP_TRY(up(lc.lastChangeTime, lastTime.data(), nv * 8));
        HIP_TRY(up(lc.gap, gapv.data(), nv * 8));
        HIP_TRY(up(lc.leaderGap, zero.data(), nv * 8));
        HIP_TRY(up(lc.followerGap, zero.data(), nv * 8));
        HIP_TRY(hipStreamSynchronize(e->stream));  // the staging vectors die with this scope
    }
    if (e->ring) {  // (needs the vehicle table and e->step: the blockers' validity tag is "set in the previous step")
        if (e->spawned) HIP_TRY(hipMemsetAsync(e->slotOf, 0xFF, (size_t) e->spawned * sizeof(int32_t), e->stream));
        if ((rc = e->ringClearStepTags())) return rc;
        hipLaunchKernelGGL(kr_scatter_in, dim3(gridFor(D)), dim3(kBlock), 0, e->stream, e->rctx(), (const int32_t *) e->rOff, e->rd, e->vt);
    } else
    // cached next drivable of every slot (uses the device copies of the route tables)
    hipLaunchKernelGGL(k_refresh_next, dim3(gridStride(std::max(S, 1))), dim3(kBlock), 0, e->stream, e->ctx());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CFX_OK;
}

// ---------------------------------------------------------------------------------------------- tiling
int32_t cfx_halo_config(cfx_engine *e, const cfx_halo_layout *h) {
    if (!e || !h || h->n_ghost < 0 || h->n_import < 0) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (e->tiled || e->step != 0 || e->spawned != 0) {
        e->err = "cfx_halo_config: must be called once, right after cfx_create";
        return CFX_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e->device));
    // Tiles run on either layout.  On the rings (round 6; `layout: ring`, and what `auto` resolves to unless the developer knob
    // cfx_config::ring_lanes_per_wave / 10000 == 7 asks for the dense tiles of rounds 2-5) the halo export is part of the step's
    // commit and the import one small kernel: 5 launches per tile-step instead of 7.
    if (e->ring && e->cfg.layout == CFX_LAYOUT_AUTO && (e->cfg.ring_lanes_per_wave / 10000) % 10 == 7) e->ring = false;
    std::vector<uint8_t> ghost((size_t) e->L, 0);
    e->hLaneSpare.assign((size_t) e->L, 1);
    for (int i = 0; i < h->n_ghost; ++i) {
        if (h->ghost_lane[i] < 0 || h->ghost_lane[i] >= e->L) return CFX_ERR_INVALID;
        ghost[h->ghost_lane[i]] = 1;
    }
    for (int i = 0; i < h->n_import; ++i) {
        if (h->import_lane[i] < 0 || h->import_lane[i] >= e->L) return CFX_ERR_INVALID;
        e->hLaneSpare[h->import_lane[i]] = 1 + CFX_HALO_MAX_MIGRANTS;
    }
    e->hLaneGhost = ghost;
    e->spareTotal = 0;
    for (uint8_t v : e->hLaneSpare) e->spareTotal += v;
    int rc;
    if ((rc = e->uploadConst(e->net.laneGhost, ghost.data(), ghost.size()))) return rc;
    if (e->ring) {
        std::vector<int32_t> cut((size_t) e->L, -1);
        for (int i = 0; i < h->n_ghost; ++i) cut[(size_t) h->ghost_lane[i]] = i;
        for (int j = 0; j < h->n_import; ++j) {
            if (cut[(size_t) h->import_lane[j]] >= 0) return e->fail("cfx_halo_config: a lane is both a ghost and an import lane");
            cut[(size_t) h->import_lane[j]] = h->n_ghost + j;
        }
        if ((rc = e->upload(&e->dCutIndex, cut.data(), cut.size()))) return rc;
        if ((rc = e->allocRaw(&e->dHaloActiveOut, 1))) return rc;
        HIP_TRY(hipMemset(e->dHaloActiveOut, 0, sizeof(long long)));
    }
    if ((rc = e->uploadConst(e->net.laneSpare, e->hLaneSpare.data(), e->hLaneSpare.size()))) return rc;
    HaloDev &d = e->halo;
    d.nGhost = h->n_ghost;
    d.nImport = h->n_import;
    if ((rc = e->uploadConst(d.ghostLane, h->ghost_lane, (size_t) h->n_ghost))) return rc;
    if ((rc = e->uploadConst(d.ghostSendOff, h->ghost_send_off, (size_t) h->n_ghost))) return rc;
    if ((rc = e->uploadConst(d.ghostRecvOff, h->ghost_recv_off, (size_t) h->n_ghost))) return rc;
    if ((rc = e->uploadConst(d.importLane, h->import_lane, (size_t) h->n_import))) return rc;
    if ((rc = e->uploadConst(d.importRecvOff, h->import_recv_off, (size_t) h->n_import))) return rc;
    if ((rc = e->uploadConst(d.importSendOff, h->import_send_off, (size_t) h->n_import))) return rc;
    if ((rc = e->uploadConst(d.llGlobal, h->lanelink_global, (size_t) e->K))) return rc;
    if ((rc = e->uploadConst(d.llLocalOfGlobal, h->lanelink_local, (size_t) h->n_global_lanelinks))) return rc;
    if ((rc = e->allocRaw(&d.ghostHadEntrants, (size_t) h->n_ghost))) return rc;
    {
        std::vector<int32_t> zeros((size_t) std::max(h->n_ghost, h->n_import) + 1, 0);
        if ((rc = e->uploadConst(d.ghostPeer, zeros.data(), (size_t) h->n_ghost))) return rc;
        if ((rc = e->uploadConst(d.importPeer, zeros.data(), (size_t) h->n_import))) return rc;
    }
    e->hGhostSendOff.assign(h->ghost_send_off, h->ghost_send_off + h->n_ghost);
    e->hGhostRecvOff.assign(h->ghost_recv_off, h->ghost_recv_off + h->n_ghost);
    e->hImportSendOff.assign(h->import_send_off, h->import_send_off + h->n_import);
    e->hImportRecvOff.assign(h->import_recv_off, h->import_recv_off + h->n_import);
    e->haloSendBytes = h->send_bytes;
    e->haloRecvBytes = h->recv_bytes;
    if ((rc = e->allocRaw(&e->dHaloSend, (size_t) h->send_bytes))) return rc;
    if ((rc = e->allocRaw(&e->dHaloRecv, (size_t) h->recv_bytes))) return rc;
    HIP_TRY(hipHostMalloc((void **) &e->hHaloSend, std::max(h->send_bytes, 1), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **) &e->hHaloRecv, std::max(h->recv_bytes, 1), hipHostMallocDefault));
    e->tiled = true;
    if ((rc = e->ensureSlotCap((size_t) e->spareTotal + 4096))) return rc;
    return e->resetState();
}

int32_t cfx_halo_export(cfx_engine *e, void *sendHost) {
    if (!e || !e->tiled) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    const int n = e->halo.nGhost + e->halo.nImport;
    HaloIO io{};
    io.send[0] = e->dHaloSend;
    // (ring layout: the step's commit has written the messages already — RingHalo)
    if (n && !e->ring) hipLaunchKernelGGL(k_halo_export, dim3(gridFor(n)), dim3(kBlock), 0, e->stream, e->ctx(), e->cnt[e->cur].p, e->halo,
                                          e->cs.inCnt, io, e->sc);
    HIP_TRY(hipGetLastError());
    if (e->haloSendBytes && sendHost) HIP_TRY(hipMemcpyAsync(e->hHaloSend, e->dHaloSend, (size_t) e->haloSendBytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // (device-to-device transports run on another stream: the message is complete)
    if (e->haloSendBytes && sendHost) memcpy(sendHost, e->hHaloSend, (size_t) e->haloSendBytes);
    return CFX_OK;
}

int32_t cfx_halo_import(cfx_engine *e, const void *recvHost) {
    if (!e || !e->tiled) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));  // the pinned staging buffer of the previous import may still be read
    if (e->haloRecvBytes && recvHost) {
        memcpy(e->hHaloRecv, recvHost, (size_t) e->haloRecvBytes);
        HIP_TRY(hipMemcpyAsync(e->dHaloRecv, e->hHaloRecv, (size_t) e->haloRecvBytes, hipMemcpyHostToDevice, e->stream));
    }
    const int n = e->halo.nGhost + e->halo.nImport;
    HaloIO io{};
    io.recv[0] = e->dHaloRecv;
    if (n && e->ring) {
        hipLaunchKernelGGL(kr_halo_import, dim3(gridFor(n)), dim3(kBlock), 0, e->stream, e->rctx(), e->halo, io, e->vt, e->sc, e->dHaloActiveOut);
        e->hCntValid = false;
    } else if (n) hipLaunchKernelGGL(k_halo_import, dim3(gridFor(n)), dim3(kBlock), 0, e->stream, e->ctx(), e->cnt[e->cur].p, e->halo,
                              io, e->vt, e->sc);
    HIP_TRY(hipGetLastError());
    e->liveUpper += (int64_t) e->halo.nImport * CFX_HALO_MAX_MIGRANTS;
    return CFX_OK;
}

int32_t cfx_halo_attach(cfx_engine *e, int32_t nPeers, const cfx_halo_peer *peers) {
    if (!e || !e->tiled || nPeers < 0 || (nPeers && !peers)) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    if (nPeers > CFX_HALO_MAX_PEERS) {
        e->err = "cfx_halo_attach: more neighbour tiles than CFX_HALO_MAX_PEERS";
        return CFX_ERR_CAPACITY;
    }
    if (!e->mail.empty()) {
        e->err = "cfx_halo_attach: already attached";
        return CFX_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e->device));
    e->mail.resize((size_t) nPeers);
    for (int p = 0; p < nPeers; ++p) {
        cfx_engine::MailPeer &m = e->mail[p];
        m.sendBytes = peers[p].send_bytes;
        m.recvBytes = peers[p].recv_bytes;
        if (!peers[p].send_mailbox || !peers[p].recv_mailbox) return CFX_ERR_INVALID;
        if (peers[p].device_memory) {
            // mailboxes in HBM: mine (recv) on this GPU, the peer's (send) on its GPU — opened through IPC by the caller, or a
            // plain pointer of another GPU of this process, which needs peer access
            m.sendDev = (char *) peers[p].send_mailbox;
            m.recvDev = (char *) peers[p].recv_mailbox;
            hipPointerAttribute_t attr{};
            if (hipPointerGetAttributes(&attr, m.sendDev) == hipSuccess && attr.device != e->device) {
                hipError_t pe = hipDeviceEnablePeerAccess(attr.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return e->fail("cfx_halo_attach: no peer access to the neighbour tile's GPU");
                (void) hipGetLastError();
            }
            continue;
        }
        m.sendHost = peers[p].send_mailbox;
        m.recvHost = peers[p].recv_mailbox;
        HIP_TRY(hipHostRegister(m.sendHost, CFX_HALO_MAILBOX_BYTES(m.sendBytes), hipHostRegisterMapped | hipHostRegisterPortable));
        HIP_TRY(hipHostRegister(m.recvHost, CFX_HALO_MAILBOX_BYTES(m.recvBytes), hipHostRegisterMapped | hipHostRegisterPortable));
        HIP_TRY(hipHostGetDevicePointer((void **) &m.sendDev, m.sendHost, 0));
        HIP_TRY(hipHostGetDevicePointer((void **) &m.recvDev, m.recvHost, 0));
    }
    // address every block as (peer, offset inside that peer's message)
    auto split = [&](const std::vector<int32_t> &off, bool send, std::vector<int32_t> &peer, std::vector<int32_t> &rel) {
        peer.resize(off.size());
        rel.resize(off.size());
        for (size_t i = 0; i < off.size(); ++i) {
            int found = -1;
            for (int p = 0; p < nPeers; ++p) {
                const int lo = send ? peers[p].send_off : peers[p].recv_off;
                const int n = send ? peers[p].send_bytes : peers[p].recv_bytes;
                if (off[i] >= lo && off[i] < lo + n) {
                    found = p;
                    rel[i] = off[i] - lo;
                }
            }
            if (found < 0) return false;
            peer[i] = found;
        }
        return true;
    };
    std::vector<int32_t> gp, gs, gp2, gr, ip, is, ip2, ir;
    if (!split(e->hGhostSendOff, true, gp, gs) || !split(e->hGhostRecvOff, false, gp2, gr) ||
        !split(e->hImportSendOff, true, ip, is) || !split(e->hImportRecvOff, false, ip2, ir) || gp != gp2 || ip != ip2) {
        e->err = "cfx_halo_attach: peer slices do not cover the halo layout";
        return CFX_ERR_INVALID;
    }
    HaloDev &d = e->haloMail;
    d = e->halo;  // lanes, maps, ghostHadEntrants are shared with the staged path
    int rc;
    if ((rc = e->uploadConst(d.ghostPeer, gp.data(), gp.size()))) return rc;
    if ((rc = e->uploadConst(d.ghostSendOff, gs.data(), gs.size()))) return rc;
    if ((rc = e->uploadConst(d.ghostRecvOff, gr.data(), gr.size()))) return rc;
    if ((rc = e->uploadConst(d.importPeer, ip.data(), ip.size()))) return rc;
    if ((rc = e->uploadConst(d.importSendOff, is.data(), is.size()))) return rc;
    if ((rc = e->uploadConst(d.importRecvOff, ir.data(), ir.size()))) return rc;
    if ((rc = e->allocRaw(&e->haloTicket, 1))) return rc;
    HIP_TRY(hipMemset(e->haloTicket, 0, sizeof(int32_t)));
    return CFX_OK;
}

int32_t cfx_halo_mailbox_alloc(cfx_engine *e, int32_t messageBytes, void **devicePtr, uint8_t *handle) {
    if (!e || !e->tiled || messageBytes < 0 || !devicePtr || !handle) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    static_assert(sizeof(hipIpcMemHandle_t) <= CFX_IPC_HANDLE_BYTES, "IPC handle size");
    const size_t bytes = CFX_HALO_MAILBOX_BYTES(messageBytes);
    void *p = nullptr;
    // fine-grained device memory first (stores of a kernel on ANOTHER GPU become visible while both kernels run); a plain
    // allocation if the platform cannot export that
    hipIpcMemHandle_t h;
    bool ok = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess && hipIpcGetMemHandle(&h, p) == hipSuccess;
    if (!ok) {
        (void) hipGetLastError();
        if (p) (void) hipFree(p);
        p = nullptr;
        e->mailboxesFineGrained = false;
        if (hipMalloc(&p, bytes) != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) {
            (void) hipGetLastError();
            if (p) (void) hipFree(p);
            e->err = "cfx_halo_mailbox_alloc: device memory cannot be shared with other processes on this platform";
            return CFX_ERR_STATE;
        }
    }
    e->owned.push_back(p);
    HIP_TRY(hipMemset(p, 0, bytes));  // epoch 0 = nothing published
    memset(handle, 0, CFX_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof h);
    *devicePtr = p;
    return CFX_OK;
}

int32_t cfx_halo_mailbox_fine_grained(cfx_engine *e) { return e && e->mailboxesFineGrained ? 1 : 0; }

int32_t cfx_device_memory(cfx_engine *e, int64_t *free_bytes, int64_t *total_bytes) {
    if (!e) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t) f;
    if (total_bytes) *total_bytes = (int64_t) t;
    return CFX_OK;
}

int32_t cfx_device_identity(cfx_engine *e, char *buf, int32_t capacity) {
    if (!e || !buf || capacity < 2) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipDeviceGetPCIBusId(buf, capacity, e->device));
    buf[capacity - 1] = 0;
    return CFX_OK;
}

int32_t cfx_halo_mailbox_open(cfx_engine *e, const uint8_t *handle, void **devicePtr) {
    if (!e || !e->tiled || !handle || !devicePtr) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void) hipGetLastError();
        e->err = "cfx_halo_mailbox_open: hipIpcOpenMemHandle failed";
        return CFX_ERR_STATE;
    }
    e->ipcOpened.push_back(p);
    *devicePtr = p;
    return CFX_OK;
}

int32_t cfx_halo_device_buffers(cfx_engine *e, void **sendDev, void **recvDev) {
    if (!e || !e->tiled) return CFX_ERR_INVALID;
    if (sendDev) *sendDev = e->dHaloSend;
    if (recvDev) *recvDev = e->dHaloRecv;
    return CFX_OK;
}

// epoch of the step that has just been launched: (generation, step) packed, monotonic over resets
static inline unsigned long long haloEpoch(const cfx_engine *e) {
    return ((unsigned long long) e->generation << 32) | (unsigned long long) (uint32_t) e->step;
}

int32_t cfx_halo_post(cfx_engine *e) {
    if (!e || !e->tiled) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    const unsigned long long epoch = haloEpoch(e);
    const int par = (int) (epoch & 1ULL);
    HaloIO io{};
    const int nPeers = (int) e->mail.size();
    for (int p = 0; p < nPeers; ++p) {
        io.send[p] = e->mail[p].sendDev + CFX_HALO_MAILBOX_HEADER + (size_t) par * e->mail[p].sendBytes;
        io.signalFlag[p] = (unsigned long long *) e->mail[p].sendDev;
    }
    io.nSignal = nPeers;
    io.ticket = e->haloTicket;
    io.epoch = epoch;
    const int n = e->halo.nGhost + e->halo.nImport;  // every peer implies at least one cut lane, so n > 0 with peers
    if (n && !e->ring) {  // (ring layout: the step's commit wrote the messages and published the epoch — RingHalo)
        e->launchNamed(PK_HALO_EXPORT, "k_halo_export", k_halo_export, dim3(gridFor(n)), dim3(kBlock), e->ctx(), e->cnt[e->cur].p, e->haloMail,
                  (const int32_t *) e->cs.inCnt, io, e->sc);
    }
    HIP_TRY(hipGetLastError());
    return CFX_OK;
}

int32_t cfx_halo_wait(cfx_engine *e) {
    if (!e || !e->tiled) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    const unsigned long long epoch = haloEpoch(e);
    const int par = (int) (epoch & 1ULL);
    HaloIO io{};
    const int nPeers = (int) e->mail.size();
    for (int p = 0; p < nPeers; ++p) {
        io.recv[p] = e->mail[p].recvDev + CFX_HALO_MAILBOX_HEADER + (size_t) par * e->mail[p].recvBytes;
        io.waitFlag[p] = (const unsigned long long *) e->mail[p].recvDev;
    }
    io.nWait = nPeers;
    io.epoch = epoch;
    const int n = e->halo.nGhost + e->halo.nImport;
    if (n && e->ring) {
        // (left to the next step's admission launch — kr_admit, RingHaloIn; a getter in between sends it out itself: settle())
        e->pendingImport = RingHaloIn{1, e->dCutIndex, e->haloMail, io, e->dHaloActiveOut};
        e->haloImportPending = true;
    } else if (n) {  // includes the wait for the neighbours' epochs
        e->launchNamed(PK_HALO_IMPORT, "k_halo_import", k_halo_import, dim3(gridFor(n)), dim3(kBlock), e->ctx(), e->cnt[e->cur].p, e->haloMail, io,
                  e->vt, e->sc);
    }
    HIP_TRY(hipGetLastError());
    e->liveUpper += (int64_t) e->halo.nImport * CFX_HALO_MAX_MIGRANTS;
    return CFX_OK;
}

#ifdef CFX_TRACE
// developer build: dump the action kernel's per-block phase stamps of the LAST step to a file (raw int64[blocks][8])
int32_t cfx_trace_dump(const char *path, int32_t blocks) {
    static long long *buf = nullptr;
    if (!buf) {
        (void) hipMalloc((void **) &buf, (size_t) 65536 * 8 * sizeof(long long));
        (void) hipMemset(buf, 0, (size_t) 65536 * 8 * sizeof(long long));
        (void) hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof buf);
        return 0;
    }
    (void) hipDeviceSynchronize();
    if (blocks < 0) {  // clear the stamps (a kernel whose blocks do not all stamp in every step)
        (void) hipMemset(buf, 0, (size_t) 65536 * 8 * sizeof(long long));
        return 0;
    }
    std::vector<long long> h((size_t) blocks * 8);
    (void) hipMemcpy(h.data(), buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(h.data(), sizeof(long long), h.size(), f);
    fclose(f);
    return 0;
}
#endif

__global__ void k_device_spin(long long ticks, double *sink) {  // wall_clock64: 100 MHz on gfx950
    const long long t0 = (long long) wall_clock64();
    double x = 1.0 + threadIdx.x * 1e-9;
    while ((long long) wall_clock64() - t0 < ticks)
        for (int i = 0; i < 64; ++i) x = x * 1.0000001 + 1e-12;
    if (x == 0.12345) *sink = x;
}

int32_t cfx_device_spin(cfx_engine *e, int64_t microseconds) {
    if (!e || microseconds < 0) return CFX_ERR_INVALID;
    auto fail = [e](const std::string &m) { return e->fail(m); };
    HIP_TRY(hipSetDevice(e->device));
    if (microseconds > 2000000) microseconds = 2000000;
    hipLaunchKernelGGL(k_device_spin, dim3(1024), dim3(256), 0, e->stream, (long long) microseconds * 100LL, (double *) e->sc);
    HIP_TRY(hipGetLastError());
    return CFX_OK;
}

int32_t cfx_profile_kernel_count(void) { return kNumProfKernels; }
const char *cfx_profile_kernel_name(int32_t k) { return (k >= 0 && k < kNumProfKernels) ? kProfNames[k] : ""; }

int32_t cfx_profile_enable(cfx_engine *e, int32_t on) {
    if (!e) return CFX_ERR_INVALID;
    (void) hipSetDevice(e->device);
    if (e->profiling && !on) e->profCollect();
    e->profiling = on != 0;
    return CFX_OK;
}

int32_t cfx_profile_read(cfx_engine *e, double *totalMs, int64_t *launches) {
    if (!e) return CFX_ERR_INVALID;
    (void) hipSetDevice(e->device);
    e->profCollect();
    for (int k = 0; k < kNumProfKernels; ++k) {
        if (totalMs) totalMs[k] = e->profMs[k];
        if (launches) launches[k] = e->profLaunches[k];
        e->profMs[k] = 0;
        e->profLaunches[k] = 0;
    }
    return CFX_OK;
}

}  // extern "C"
