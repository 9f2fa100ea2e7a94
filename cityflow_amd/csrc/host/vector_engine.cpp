#include "vector_engine.h"

#include <algorithm>
#include <chrono>

#include <iostream>
#include <stdexcept>

namespace cfa {

namespace {

// R disjoint copies of a flat network, index spaces concatenated replica by replica.
struct ReplicatedNet {
    cfx_net flat{};
    std::vector<double> drvLength, drvMaxSpeed, xDist, phaseTime;
    std::vector<int32_t> laneRoad, laneIndex, laneLLStart, laneLL, roadLaneStart, llStartLane, llEndLane, llInter,
        llRoadLink, llType, llXStart, xPeer, xLL, interVirtual, interNRL, interPhaseStart, interAvailStart;
    std::vector<uint8_t> phaseAvail;
    std::vector<double> laneWidth;          // lane change only
    std::vector<int32_t> laneNSegments;

    void build(const cfx_net &n, int R) {
        const int L = n.n_lanes, K = n.n_lanelinks, Rd = n.n_roads, I = n.n_inters, E = n.n_xentries, P = n.n_phases,
                  A = n.n_avail;
        auto rep = [R](auto &dst, const auto *src, int count, int offsetPerReplica) {
            dst.resize((size_t) count * R);
            for (int r = 0; r < R; ++r)
                for (int i = 0; i < count; ++i) dst[(size_t) r * count + i] = src[i] + offsetPerReplica * r;
        };
        // per drivable: all lanes of all replicas first, then all laneLinks
        drvLength.resize((size_t) (L + K) * R);
        drvMaxSpeed.resize((size_t) (L + K) * R);
        for (int r = 0; r < R; ++r) {
            for (int l = 0; l < L; ++l) {
                drvLength[(size_t) r * L + l] = n.drv_length[l];
                drvMaxSpeed[(size_t) r * L + l] = n.drv_max_speed[l];
            }
            for (int k = 0; k < K; ++k) {
                drvLength[(size_t) R * L + (size_t) r * K + k] = n.drv_length[L + k];
                drvMaxSpeed[(size_t) R * L + (size_t) r * K + k] = n.drv_max_speed[L + k];
            }
        }
        rep(laneRoad, n.lane_road, L, Rd);
        rep(laneIndex, n.lane_index, L, 0);
        rep(laneLL, n.lane_ll, K, K);
        rep(llStartLane, n.ll_start_lane, K, L);
        rep(llEndLane, n.ll_end_lane, K, L);
        rep(llInter, n.ll_inter, K, I);
        rep(llRoadLink, n.ll_roadlink, K, 0);
        rep(llType, n.ll_type, K, 0);
        rep(xPeer, n.x_peer, E, E);
        rep(xLL, n.x_ll, E, K);
        rep(interVirtual, n.inter_virtual, I, 0);
        rep(interNRL, n.inter_n_roadlinks, I, 0);
        rep(interAvailStart, n.inter_avail_start, I, A);
        xDist.resize((size_t) E * R);
        phaseTime.resize((size_t) P * R);
        phaseAvail.resize((size_t) A * R);
        for (int r = 0; r < R; ++r) {
            std::copy(n.x_dist, n.x_dist + E, xDist.begin() + (size_t) r * E);
            std::copy(n.phase_time, n.phase_time + P, phaseTime.begin() + (size_t) r * P);
            std::copy(n.phase_avail, n.phase_avail + A, phaseAvail.begin() + (size_t) r * A);
        }
        // CSR arrays: [count*R + 1]
        auto repCsr = [R](std::vector<int32_t> &dst, const int32_t *src, int count) {
            const int total = src[count];
            dst.resize((size_t) count * R + 1);
            for (int r = 0; r < R; ++r)
                for (int i = 0; i < count; ++i) dst[(size_t) r * count + i] = src[i] + total * r;
            dst[(size_t) count * R] = total * R;
        };
        repCsr(laneLLStart, n.lane_ll_start, L);
        repCsr(roadLaneStart, n.road_lane_start, Rd);
        repCsr(llXStart, n.ll_x_start, K);
        repCsr(interPhaseStart, n.inter_phase_start, I);

        flat = cfx_net{};
        flat.n_roads = Rd * R;
        flat.n_lanes = L * R;
        flat.n_lanelinks = K * R;
        flat.n_inters = I * R;
        flat.n_xentries = E * R;
        flat.n_phases = P * R;
        flat.n_avail = A * R;
        flat.drv_length = drvLength.data();
        flat.drv_max_speed = drvMaxSpeed.data();
        flat.lane_road = laneRoad.data();
        flat.lane_index = laneIndex.data();
        flat.lane_ll_start = laneLLStart.data();
        flat.lane_ll = laneLL.data();
        flat.road_lane_start = roadLaneStart.data();
        flat.ll_start_lane = llStartLane.data();
        flat.ll_end_lane = llEndLane.data();
        flat.ll_inter = llInter.data();
        flat.ll_roadlink = llRoadLink.data();
        flat.ll_type = llType.data();
        flat.ll_x_start = llXStart.data();
        flat.x_dist = xDist.data();
        flat.x_peer = xPeer.data();
        flat.x_ll = xLL.data();
        flat.inter_virtual = interVirtual.data();
        flat.inter_n_roadlinks = interNRL.data();
        flat.inter_phase_start = interPhaseStart.data();
        flat.inter_avail_start = interAvailStart.data();
        flat.phase_time = phaseTime.data();
        flat.phase_avail = phaseAvail.data();
        if (n.lane_width && n.lane_n_segments) {  // lane change
            laneWidth.resize((size_t) L * R);
            for (int r = 0; r < R; ++r) std::copy(n.lane_width, n.lane_width + L, laneWidth.begin() + (size_t) r * L);
            rep(laneNSegments, n.lane_n_segments, L, 0);
            flat.lane_width = laneWidth.data();
            flat.lane_n_segments = laneNSegments.data();
        }
    }
};

}  // namespace

VectorEngineHost::VectorEngineHost(const std::string &configFile, int numEnvs, int threadNum, const std::string &backendLib)
    : R_(numEnvs) {
    if (numEnvs < 1) throw std::runtime_error("VectorEngine: num_envs must be >= 1");
    EngineConfig cfg = readEngineConfig(configFile);
    laneChange_ = cfg.laneChange;
    interval_ = cfg.interval;
    rlTrafficLight_ = cfg.rlTrafficLight;
    hostThreads_ = cfg.hostThreads;
    aheadEnabled_ = cfg.spawnAhead && !cfg.laneChange;
    try {
        net_->load(cfg.dir + cfg.roadnetFile);
        for (int r = 0; r < R_; ++r) {
            spawners_.emplace_back(new Spawner());
            spawners_.back()->init(net_.get(), interval_, threadNum, cfg.seed + r);
            spawners_.back()->exactPeekOnly = cfg.exactShadowPeek;
            spawners_.back()->loadFlows(cfg.dir + cfg.flowFile);
        }
    } catch (const std::exception &e) {
        throw std::runtime_error(std::string("load config failed! ") + e.what());
    }
    L_ = (int) net_->lanes.size();
    K_ = (int) net_->laneLinks.size();
    I_ = (int) net_->inters.size();
    localToGlobal_.resize(R_);

    ReplicatedNet rn;
    rn.build(net_->flat(), R_);
    be_.open(backendLib.empty() ? defaultBackendPath() : backendLib);
    cfx_config cc{};
    cfg.apply(cc);
    cc.n_envs = R_;  // (lane change: every environment has its own schedule walk and priority stream)
    int32_t rc = be_.cfx_create(&rn.flat, &cc, &dev_);
    if (rc != CFX_OK || !dev_) {
        const char *msg = be_.cfx_last_error(nullptr);
        throw std::runtime_error(std::string("cityflow_amd: cfx_create failed in ") + be_.path + ": " +
                                 (msg ? msg : "unknown error") + " (there is no CPU fallback)");
    }
    for (int r = 0; r < R_; ++r) {
        Spawner *sp = spawners_[r].get();
        sp->setFinishedQuery([this, r](int localVid) {
            // (a batch prepared ahead: "finished" means finished by the steps before it — wait until they are all submitted)
            while (submitted_.load(std::memory_order_acquire) < (uint64_t) preparing_.load(std::memory_order_acquire)) std::this_thread::yield();
            std::lock_guard<std::mutex> guard(queryMutex_);  // rare (priority collision); the ABI is not re-entrant
            uint8_t st = 0;
            check(be_.cfx_get_vehicle_status(dev_, localToGlobal_[r][localVid], 1, &st), "cfx_get_vehicle_status");
            return st == 2;
        });
    }
    // templates are shared; each environment gets its own copy of the route tables with shifted indices
    const Spawner &s0 = *spawners_[0];
    check(be_.cfx_add_templates(dev_, (int) s0.templates.size(), s0.templates.data()), "cfx_add_templates");
    const RouteTable &rt = s0.routes;
    routesPerEnv_ = rt.count();
    const int Rd = (int) net_->roads.size();
    for (int r = 0; r < R_; ++r) {
        std::vector<int32_t> roads(rt.roads), nextLL(rt.nextLL);
        for (auto &x : roads) x += r * Rd;
        for (auto &x : nextLL)
            if (x >= 0) x += r * K_;
        check(be_.cfx_add_routes(dev_, routesPerEnv_, rt.routeStart.data(), roads.data(), rt.nextStart.data(), nextLL.data()),
              "cfx_add_routes");
    }
}

VectorEngineHost::~VectorEngineHost() {
    if (aheadThread_.joinable()) {
        {
            std::lock_guard<std::mutex> guard(aheadMutex_);
            aheadStop_ = true;
        }
        aheadCv_.notify_all();
        submitted_.store(~0ull, std::memory_order_release);  // (a query waiting for a submission that will not come)
        aheadThread_.join();
    }
    {
        std::lock_guard<std::mutex> guard(poolMutex_);
        poolStop_ = true;
    }
    poolCv_.notify_all();
    for (std::thread &t : workers_) t.join();
    if (dev_) be_.cfx_destroy(dev_);
}

// ---------------------------------------------------------------- host threads for the R spawners
void VectorEngineHost::runEnvs() {
    for (;;) {
        const int r = nextEnv_.fetch_add(1, std::memory_order_acq_rel);  // pairs with the release store that opens a batch
        if (r >= R_) return;
        try {
            (this->*poolFn_)(r);
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> guard(poolMutex_);
            if (poolError_.empty()) poolError_ = e.what();
        }
        envsDone_.fetch_add(1, std::memory_order_release);
    }
}

void VectorEngineHost::workerLoop() {
    uint64_t seen = 0;
    for (;;) {
        // While the engine is being stepped the next batch arrives within microseconds: poll for a moment before
        // going to sleep on the condition variable (a wake-up through the kernel costs tens of microseconds).
        bool got = false;
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
        while (std::chrono::steady_clock::now() < until) {
            if (poolGeneration_.load(std::memory_order_acquire) != seen) {
                got = true;
                break;
            }
            std::this_thread::yield();
        }
        if (!got) {
            std::unique_lock<std::mutex> lock(poolMutex_);
            poolCv_.wait(lock, [&] { return poolStop_ || poolGeneration_.load(std::memory_order_acquire) != seen; });
            if (poolStop_) return;
        }
        seen = poolGeneration_.load(std::memory_order_acquire);
        runEnvs();
    }
}

void VectorEngineHost::forEachEnv(void (VectorEngineHost::*fn)(int)) {
    // Whether waking threads pays depends on the work per environment, not on their number: 16 replicas of a city-scale
    // network spend ~13 us each in their spawner (200 us serial), 16 replicas of a 6x6 grid 1.6 us (a wake-up costs more).
    // Auto (cfx.hostThreads < 0): the first kAutoProbe batches run serially and are timed; from then on the pool is used if
    // a serial batch took more than kAutoSerialUs.  cfx.hostThreads > 0 asks for the pool, 0 for the serial loop.
    constexpr int kAutoProbe = 24;
    constexpr double kAutoSerialUs = 40.0;
    bool serial = hostThreads_ == 0 || R_ < 2;
    if (!serial && hostThreads_ < 0) {
        if (R_ >= 32) {
            serial = false;
        } else if (R_ < 4) {
            serial = true;
        } else if (autoBatches_ < kAutoProbe) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R_; ++r) (this->*fn)(r);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            autoSerialUs_ = std::max(autoSerialUs_, 0.0) + us;
            if (++autoBatches_ == kAutoProbe) autoUsePool_ = autoSerialUs_ / kAutoProbe > kAutoSerialUs;
            return;
        } else {
            serial = !autoUsePool_;
        }
    }
    if (serial) {
        for (int r = 0; r < R_; ++r) (this->*fn)(r);
        return;
    }
    if (workers_.empty()) {
        unsigned hw = std::thread::hardware_concurrency();
        int n = (int) std::min<unsigned>(std::min<unsigned>(hw > 2 ? hw / 2 : 1, 16u), (unsigned) std::max(R_ / 2, 1));
        if (hostThreads_ > 0) n = std::min(hostThreads_, R_);
        for (int i = 0; i + 1 < n; ++i) workers_.emplace_back([this] { workerLoop(); });
    }
    poolFn_ = fn;
    envsDone_.store(0, std::memory_order_relaxed);
    nextEnv_.store(0, std::memory_order_release);  // a straggler of the previous batch may pick work up from here on
    {
        std::lock_guard<std::mutex> guard(poolMutex_);
        poolGeneration_.fetch_add(1, std::memory_order_release);
    }
    poolCv_.notify_all();
    runEnvs();  // the calling thread works too
    while (envsDone_.load(std::memory_order_acquire) < R_) std::this_thread::yield();
    if (!poolError_.empty()) {
        std::string msg = poolError_;
        poolError_.clear();
        throw std::runtime_error(msg);
    }
}

// phases 0-1 of one environment (its own mt19937 and flows)
void VectorEngineHost::spawnEnv(int r) {
    if (journalling_) spawners_[r]->beginAhead();
    spawners_[r]->step((size_t) preparing_.load(std::memory_order_relaxed), envRecs_[r]);
}

// an environment's records, renumbered into the device engine's index spaces, at their place in the batch
void VectorEngineHost::translateEnv(int r) {
    int32_t g = batchFirstVid_ + envBase_[r];
    cfx_spawn *out = recsNext_.data() + envBase_[r];
    std::vector<int32_t> &l2g = localToGlobal_[r];
    const int32_t firstLocal = (int32_t) l2g.size();  // a batch holds the next dense run of local vids, in any order
    l2g.resize(l2g.size() + envRecs_[r].size());
    for (cfx_spawn s : envRecs_[r]) {
        const int32_t gv = g + (s.vid - firstLocal);
        l2g[(size_t) s.vid] = gv;
        globalToLocal_[gv] = std::make_pair((int32_t) r, s.vid);
        s.vid = gv;
        s.lane += r * L_;
        s.route += r * routesPerEnv_;
        s.prev_wait = s.prev_wait >= 0 ? l2g[s.prev_wait] : -1;
        *out++ = s;
    }
}

void VectorEngineHost::check(int32_t rc, const char *what) {
    if (rc == CFX_OK) return;
    const char *msg = be_.cfx_last_error(dev_);
    throw std::runtime_error(std::string("cityflow_amd: ") + what + " failed (" + std::to_string(rc) + "): " + (msg ? msg : ""));
}

// Lane change: which vehicles got a shadow in the last step (EngineHost::settleLaneChange for R environments).  The device lists
// the parents environment by environment, each environment's in the order of its own schedule walk; shadow i of the list has
// the global number (vehicles before the step's batch) + (the batch) + i, and each spawner numbers ITS shadows in that order.
void VectorEngineHost::settleLaneChange() {
    if (!lcPollPending_) return;
    lcPollPending_ = false;
    shadowParents_.resize((size_t) shadowPoolPerEnv_ * R_);
    int32_t k = 0;
    check(be_.cfx_lane_change_poll(dev_, (int32_t) shadowParents_.size(), shadowParents_.data(), &k), "cfx_lane_change_poll");
    std::vector<std::vector<int32_t>> perEnv((size_t) R_);
    int lastEnv = 0;
    for (int i = 0; i < k; ++i) {
        const auto gl = globalToLocal_.at((size_t) shadowParents_[(size_t) i]);
        if (gl.first < lastEnv) throw std::runtime_error("VectorEngine: lane-change poll is not in environment order");
        lastEnv = gl.first;
        Spawner &sp = *spawners_[(size_t) gl.first];
        const int32_t localShadow = (int32_t) (sp.vehicles.size() + perEnv[(size_t) gl.first].size());
        perEnv[(size_t) gl.first].push_back(gl.second);
        const int32_t globalShadow = (int32_t) globalToLocal_.size();
        globalToLocal_.emplace_back(gl.first, localShadow);
        localToGlobal_[(size_t) gl.first].push_back(globalShadow);
    }
    int most = 0;
    for (int r = 0; r < R_; ++r) {
        spawners_[(size_t) r]->commitShadows(perEnv[(size_t) r]);
        most = std::max(most, (int) perEnv[(size_t) r].size());
    }
    if (4 * most > shadowPoolPerEnv_) shadowPoolPerEnv_ = 8 * most;  // stay well clear of a step's demand
}

// an environment's share of the priorities this step's shadows would draw, after the step's own spawn draws
void VectorEngineHost::peekEnv(int r) {
    std::vector<int32_t> &tmp = envPeek_[(size_t) r];
    spawners_[(size_t) r]->peekShadowPriorities(shadowPoolPerEnv_, tmp);
    std::copy(tmp.begin(), tmp.end(), shadowPool_.begin() + (size_t) r * shadowPoolPerEnv_);
}

// One step's batch: the R spawners, the numbering of the new vehicles (environment order), the translation into the device
// engine's index spaces — into recsNext_.
void VectorEngineHost::prepareBatch(size_t step, double *spawnSec, double *translateSec) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    preparing_.store((uint64_t) step, std::memory_order_release);
    envRecs_.resize((size_t) R_);
    envBase_.resize((size_t) R_);
    if (journalling_) {
        l2gMark_ = globalToLocal_.size();
        l2gEnvMark_.resize((size_t) R_);
        for (int r = 0; r < R_; ++r) l2gEnvMark_[(size_t) r] = localToGlobal_[(size_t) r].size();
    }
    forEachEnv(&VectorEngineHost::spawnEnv);
    const auto t1 = clk::now();
    int32_t total = 0;
    for (int r = 0; r < R_; ++r) {  // the device numbers vehicles in environment order
        const Spawner &sp = *spawners_[r];
        if (sp.routes.count() != routesPerEnv_ || sp.templates.size() != spawners_[0]->templates.size())
            throw std::runtime_error("VectorEngine: dynamic routes/templates are not supported");
        envBase_[r] = total;
        total += (int32_t) envRecs_[r].size();
    }
    batchFirstVid_ = (int32_t) globalToLocal_.size();
    globalToLocal_.resize(globalToLocal_.size() + (size_t) total);
    recsNext_.resize((size_t) total);
    forEachEnv(&VectorEngineHost::translateEnv);
    const auto t2 = clk::now();
    if (spawnSec) *spawnSec += std::chrono::duration<double>(t1 - t0).count();
    if (translateSec) *translateSec += std::chrono::duration<double>(t2 - t1).count();
}

void VectorEngineHost::aheadLoop() {
    for (;;) {
        size_t step;
        {
            // While the engine is being stepped the next request arrives within tens of microseconds: poll for a moment
            // before going to sleep on the condition variable (a wake-up through the kernel costs 10-30 us, as long as the job)
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
            while (aheadKicks_.load(std::memory_order_acquire) == aheadSeen_ && std::chrono::steady_clock::now() < until)
                __builtin_ia32_pause();  // (no system call in the loop: a yield per iteration costs the stepping thread its core's attention)
            std::unique_lock<std::mutex> lock(aheadMutex_);
            aheadCv_.wait(lock, [&] { return aheadStop_ || aheadState_ == kAheadWorking; });
            if (aheadStop_) return;
            step = aheadStep_;
            aheadSeen_ = aheadKicks_.load(std::memory_order_acquire);
        }
        std::string error;
        const auto t0 = std::chrono::steady_clock::now();
        try {
            prepareBatch(step, nullptr, nullptr);
        } catch (const std::exception &e) {
            error = e.what()[0] ? e.what() : "unknown error";
        }
        hostAheadSec_.store(hostAheadSec_.load(std::memory_order_relaxed) +
                                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> guard(aheadMutex_);
            aheadError_ = error;
            aheadState_ = error.empty() ? kAheadReady : kAheadFailed;
            aheadBusy_.store(false, std::memory_order_release);
        }
        aheadCv_.notify_all();
    }
}

void VectorEngineHost::kickAhead(size_t step) {
    if (!aheadThread_.joinable()) aheadThread_ = std::thread([this] { aheadLoop(); });
    {
        std::lock_guard<std::mutex> guard(aheadMutex_);
        aheadStep_ = step;
        aheadState_ = kAheadWorking;
        aheadBusy_.store(true, std::memory_order_release);
        aheadKicks_.fetch_add(1, std::memory_order_release);
    }
    aheadCv_.notify_all();
}

void VectorEngineHost::waitAhead() {
    {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        while (aheadBusy_.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
    }
    std::unique_lock<std::mutex> lock(aheadMutex_);
    aheadCv_.wait(lock, [&] { return aheadState_ != kAheadWorking; });
}

// Whatever was prepared for a step that is not going to be taken as it stands: every spawner goes back to where the last
// step that WAS taken left it (generator, flows, priority set, vehicle tables), and so does the numbering.
void VectorEngineHost::dropAhead() {
    if (!aheadEnabled_) return;
    waitAhead();
    if (aheadState_ == kAheadReady || aheadState_ == kAheadFailed) {
        for (auto &sp : spawners_) sp->rollbackAhead();
        if (l2gEnvMark_.size() == (size_t) R_) {
            globalToLocal_.resize(std::min(globalToLocal_.size(), l2gMark_));
            for (int r = 0; r < R_; ++r)
                localToGlobal_[(size_t) r].resize(std::min(localToGlobal_[(size_t) r].size(), l2gEnvMark_[(size_t) r]));
        }
    }
    {
        std::lock_guard<std::mutex> guard(aheadMutex_);  // (the ahead thread reads it under the mutex)
        aheadState_ = kAheadIdle;
    }
    journalling_ = false;
}

void VectorEngineHost::nextStep() {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    settleLaneChange();  // every generator must be past the last step's shadow draws before this step's spawns
    double spawnSec = 0, translateSec = 0;
    bool taken = false;
    if (aheadEnabled_) {
        waitAhead();
        if (aheadState_ == kAheadFailed) {  // what the ahead thread ran into is raised by the step it belonged to
            const std::string msg = aheadError_;
            dropAhead();
            throw std::runtime_error(msg);
        }
        if (aheadState_ == kAheadReady && aheadStep_ == step_) {
            for (auto &sp : spawners_) sp->commitAhead();
            {
                std::lock_guard<std::mutex> guard(aheadMutex_);  // (the ahead thread reads it under the mutex)
                aheadState_ = kAheadIdle;
            }
            taken = true;
            spawnSec = std::chrono::duration<double>(clk::now() - t0).count();  // (the wait is what the caller paid)
        } else if (aheadState_ == kAheadReady) {
            dropAhead();
        }
    }
    if (!taken) {
        journalling_ = false;
        prepareBatch(step_, &spawnSec, &translateSec);
    }
    recs_.swap(recsNext_);
    if (laneChange_) {
        shadowPool_.resize((size_t) shadowPoolPerEnv_ * R_);
        envPeek_.resize((size_t) R_);
        forEachEnv(&VectorEngineHost::peekEnv);
        check(be_.cfx_lane_change_supply(dev_, (int32_t) shadowPool_.size(), shadowPool_.data()), "cfx_lane_change_supply");
    }
    if (aheadEnabled_) {  // the next step's batch, beside this step's submission and the device's work
        journalling_ = true;
        kickAhead(step_ + 1);
    }
    const auto t2 = clk::now();
    int32_t rcStep;
    {
        std::lock_guard<std::mutex> guard(queryMutex_);  // (a priority collision on the ahead thread asks the device too)
        rcStep = be_.cfx_step(dev_, recs_.data(), (int32_t) recs_.size());
    }
    // (published whatever the outcome: a query of the ahead thread that waits for this submission must not wait for ever)
    submitted_.store((uint64_t) step_ + 1, std::memory_order_release);
    check(rcStep, "cfx_step");
    lcPollPending_ = laneChange_;
    const auto t3 = clk::now();
    hostSpawnSec_ += spawnSec;
    hostTranslateSec_ += translateSec;
    hostSubmitSec_ += std::chrono::duration<double>(t3 - t2).count();
    step_ += 1;
}

void VectorEngineHost::reset(bool resetRnd) {
    dropAhead();
    settleLaneChange();  // (without a reseed the generators go on from behind the last step's shadow draws)
    {
        std::lock_guard<std::mutex> guard(queryMutex_);
        check(be_.cfx_reset(dev_), "cfx_reset");
    }
    for (auto &sp : spawners_) sp->reset(resetRnd);
    for (auto &v : localToGlobal_) v.clear();
    globalToLocal_.clear();
    step_ = 0;
    submitted_.store(0, std::memory_order_release);
    hostSpawnSec_ = hostTranslateSec_ = hostSubmitSec_ = 0;
    hostAheadSec_.store(0.0, std::memory_order_relaxed);
}

std::vector<int32_t> VectorEngineHost::laneVehicleCounts() {
    std::lock_guard<std::mutex> guard(queryMutex_);  // (the ABI is not re-entrant: the ahead thread may be asking the device)
    std::vector<int32_t> out((size_t) R_ * L_);
    check(be_.cfx_get_lane_counts(dev_, out.data()), "cfx_get_lane_counts");
    return out;
}

std::vector<int32_t> VectorEngineHost::laneWaitingVehicleCounts() {
    std::lock_guard<std::mutex> guard(queryMutex_);  // (the ABI is not re-entrant: the ahead thread may be asking the device)
    std::vector<int32_t> out((size_t) R_ * L_);
    check(be_.cfx_get_lane_waiting_counts(dev_, out.data()), "cfx_get_lane_waiting_counts");
    return out;
}

cfx_scalars VectorEngineHost::scalars() {
    std::lock_guard<std::mutex> guard(queryMutex_);  // (the ABI is not re-entrant: the ahead thread may be asking the device)
    cfx_scalars s{};
    check(be_.cfx_get_scalars(dev_, &s), "cfx_get_scalars");
    return s;
}

int64_t VectorEngineHost::totalVehicleCount() { return scalars().active_vehicle_count; }

void VectorEngineHost::sync() {
    std::lock_guard<std::mutex> guard(queryMutex_);
    check(be_.cfx_sync(dev_), "cfx_sync");
}

void VectorEngineHost::profileEnable(bool on) {
    std::lock_guard<std::mutex> guard(queryMutex_);
    check(be_.cfx_profile_enable(dev_, on ? 1 : 0), "cfx_profile_enable");
}

std::map<std::string, std::pair<double, int64_t>> VectorEngineHost::profileRead() {
    std::lock_guard<std::mutex> guard(queryMutex_);  // (the ABI is not re-entrant: the ahead thread may be asking the device)
    int n = be_.cfx_profile_kernel_count();
    std::vector<double> ms(n > 0 ? n : 1);
    std::vector<int64_t> cnt(n > 0 ? n : 1);
    check(be_.cfx_profile_read(dev_, ms.data(), cnt.data()), "cfx_profile_read");
    std::map<std::string, std::pair<double, int64_t>> out;
    for (int k = 0; k < n; ++k) out[be_.cfx_profile_kernel_name(k)] = std::make_pair(ms[k], cnt[k]);
    return out;
}

void VectorEngineHost::setTrafficLightPhases(const std::vector<int32_t> &phases) {
    if (!rlTrafficLight_) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    if ((int) phases.size() != R_ * I_) throw std::runtime_error("set_tl_phases: expected num_envs * num_intersections phases");
    std::vector<int32_t> inters, ph;
    for (int r = 0; r < R_; ++r)
        for (int i = 0; i < I_; ++i) {
            const HostInter &in = net_->inters[i];
            if (in.isVirtual) continue;
            int p = phases[(size_t) r * I_ + i];
            if (p < 0 || p >= (int) in.phases.size())
                throw std::out_of_range("set_tl_phases: phase out of range for intersection '" + in.id + "'");
            inters.push_back(r * I_ + i);
            ph.push_back(p);
        }
    std::lock_guard<std::mutex> guard(queryMutex_);
    check(be_.cfx_set_tl_phases(dev_, (int32_t) inters.size(), inters.data(), ph.data()), "cfx_set_tl_phases");
}

std::vector<std::string> VectorEngineHost::laneIds() const {
    std::vector<std::string> ids(L_);
    for (int l = 0; l < L_; ++l) ids[l] = net_->laneId(l);
    return ids;
}

std::vector<std::string> VectorEngineHost::intersectionIds() const {
    std::vector<std::string> ids(I_);
    for (int i = 0; i < I_; ++i) ids[i] = net_->inters[i].id;
    return ids;
}

std::map<std::string, int> VectorEngineHost::getLaneVehicleCount(int env) {
    if (env < 0 || env >= R_) throw std::out_of_range("env index out of range");
    std::vector<int32_t> all = laneVehicleCounts();
    std::map<std::string, int> ret;
    for (int l = 0; l < L_; ++l) ret.emplace(net_->laneId(l), all[(size_t) env * L_ + l]);
    return ret;
}

std::map<std::string, double> VectorEngineHost::getVehicleSpeed(int env) {
    if (env < 0 || env >= R_) throw std::out_of_range("env index out of range");
    if (aheadEnabled_) waitAhead();  // (the numbering tables are the ahead thread's while it works)
    settleLaneChange();  // (the shadows of the last step have their numbers then)
    int cap = (int) scalars().active_vehicle_count + 16;
    std::vector<int32_t> vid(cap), drv(cap);
    std::vector<double> speed(cap);
    std::vector<uint8_t> lcFlags(laneChange_ ? cap : 0);
    cfx_vehicle_view v{};
    v.capacity = cap;
    v.vid = vid.data();
    v.drivable = drv.data();
    v.speed = speed.data();
    if (laneChange_) v.lc_flags = lcFlags.data();
    {
        std::lock_guard<std::mutex> guard(queryMutex_);
        check(be_.cfx_get_vehicles(dev_, &v), "cfx_get_vehicles");
    }
    std::map<std::string, double> ret;
    for (int i = 0; i < v.count; ++i) {
        const auto &gl = globalToLocal_[vid[i]];
        if (laneChange_ && (lcFlags[i] & CFX_LC_SHADOW)) continue;  // Engine::getRunningVehicles lists real vehicles only
        if (gl.first == env) ret.emplace(spawners_[env]->vehicleId(gl.second), speed[i]);
    }
    return ret;
}

}  // namespace cfa
