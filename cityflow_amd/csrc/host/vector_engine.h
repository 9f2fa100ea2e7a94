// VectorEngine: R independent copies ("environments") of one scenario advanced in lock-step by ONE device
// engine.  The device sees a single road network made of R disjoint replicas (lane r*L+l, laneLink r*K+k, ...),
// so every kernel launch of a step covers all environments: launch latency is paid once per step instead of
// once per environment, and the car-following kernel finally gets enough vehicles per launch to approach the
// HBM roofline.  Each environment has its own flows and its own std::mt19937 (seed + env index)
// and evolves exactly like a standalone Engine built from the same config with that seed
// (tests/test_vector_engine.py) — with laneChange: true as well: the device takes each environment's lane-change
// schedule walk and priority stream separately (cfx_config::n_envs).
#pragma once

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine_host.h"

namespace cfa {

class VectorEngineHost {
public:
    VectorEngineHost(const std::string &configFile, int numEnvs, int threadNum, const std::string &backendLib = "");
    ~VectorEngineHost();
    VectorEngineHost(const VectorEngineHost &) = delete;
    VectorEngineHost &operator=(const VectorEngineHost &) = delete;

    int numEnvs() const { return R_; }
    int numLanes() const { return L_; }
    int numIntersections() const { return I_; }
    void nextStep();
    void reset(bool resetRnd);
    double getCurrentTime() const { return step_ * interval_; }
    std::vector<int32_t> laneVehicleCounts();         // [R * L], env-major
    std::vector<int32_t> laneWaitingVehicleCounts();  // [R * L]
    int64_t totalVehicleCount();
    // phases: [R * I] (entries of virtual intersections are ignored); requires rlTrafficLight
    void setTrafficLightPhases(const std::vector<int32_t> &phases);
    std::vector<std::string> laneIds() const;
    std::vector<std::string> intersectionIds() const;
    cfx_scalars scalars();
    void sync();
    void profileEnable(bool on);
    std::map<std::string, std::pair<double, int64_t>> profileRead();
    std::string backendName() const { return be_.cfx_backend_name(); }
    // cumulative host wall seconds since the last reset, ON THE CALLER'S PATH: {spawners (with the batch prepared a step ahead:
    // the wait for it), record translation, cfx_step (launches + back-pressure)}, then the ahead thread's own busy time
    std::vector<double> hostSeconds() const {
        return {hostSpawnSec_, hostTranslateSec_, hostSubmitSec_, hostAheadSec_.load(std::memory_order_relaxed)};
    }
    // per-environment id-keyed view, for parity tests against a standalone Engine
    std::map<std::string, double> getVehicleSpeed(int env);
    std::map<std::string, int> getLaneVehicleCount(int env);

private:
    void check(int32_t rc, const char *what);

    std::shared_ptr<HostRoadNet> net_ = std::make_shared<HostRoadNet>();
    std::vector<std::unique_ptr<Spawner>> spawners_;
    std::vector<std::vector<int32_t>> localToGlobal_;  // [env][local vid] -> global vid
    std::vector<std::pair<int32_t, int32_t>> globalToLocal_;  // global vid -> (env, local vid)
    Backend be_;
    cfx_engine *dev_ = nullptr;
    int R_ = 1, L_ = 0, K_ = 0, I_ = 0, routesPerEnv_ = 0;
    int hostThreads_ = -1;  // config "cfx": {"hostThreads": n}: -1 auto, 0 serial
    int autoBatches_ = 0;          // auto: serial batches timed so far (forEachEnv)
    double autoSerialUs_ = 0.0;    //       their total time
    bool autoUsePool_ = false;     //       the verdict once enough of them were seen
    double interval_ = 1.0;
    bool rlTrafficLight_ = false;
    size_t step_ = 0;
    std::vector<cfx_spawn> recs_;       // the batch being handed to the device
    std::vector<cfx_spawn> recsNext_;   // the batch being prepared (swapped with recs_ when a step takes it)
    std::vector<std::vector<cfx_spawn>> envRecs_;  // [env] this step's records in the environment's own numbering

    // The R spawners are independent (own mt19937, own flows): phases 0-1 of all environments run on a small pool of
    // host threads, the numbering of the new vehicles is then assigned serially in environment order.
    void forEachEnv(void (VectorEngineHost::*fn)(int));  // fn(env) for every environment, on the pool from 32 envs on
    void spawnEnv(int r);
    void translateEnv(int r);
    void peekEnv(int r);
    // lane change (laneChange: true): as EngineHost, per environment (include/cityflow_amd.h "Lane change", n_envs)
    void settleLaneChange();
    bool laneChange_ = false, lcPollPending_ = false;
    int shadowPoolPerEnv_ = 256;
    std::vector<int32_t> shadowPool_, shadowParents_;
    std::vector<std::vector<int32_t>> envPeek_;
    void workerLoop();
    void runEnvs();
    void (VectorEngineHost::*poolFn_)(int) = nullptr;
    std::vector<int32_t> envBase_;  // [env] offset of the environment's records in this step's batch
    int32_t batchFirstVid_ = 0;
    double hostSpawnSec_ = 0, hostTranslateSec_ = 0, hostSubmitSec_ = 0;
    std::atomic<double> hostAheadSec_{0.0};  // (written by the ahead thread)
    // ---- the batch of step t+1 is prepared while step t is submitted and runs (config "cfx": {"spawnAhead": false} turns it
    //      off; not with lane change, whose shadows draw from the generators after the device has scheduled them).  The
    //      reference's Flow::nextStep / planRoute (flow.cpp:6-22, engine.cpp:450-470) depend on nothing a step computes
    //      except through the rare priority collision, which asks the device — after step t has been handed over, as always.
    //      One more host thread runs the R spawners and the translation (on the pool above when that pays); every spawner
    //      journals the step (Spawner::beginAhead), so reset() can take it back.  nextStep() only waits for the batch.
    void prepareBatch(size_t step, double *spawnSec, double *translateSec);  // spawn + number + translate into recsNext_
    void aheadLoop();
    void kickAhead(size_t step);
    void waitAhead();                 // until the ahead thread is not working (rethrows what it threw)
    void dropAhead();                 // ... and take a prepared batch back (rollback of every spawner's journal)
    bool aheadEnabled_ = false;
    bool journalling_ = false;        // the batch being prepared is one a reset may have to take back
    enum AheadState { kAheadIdle, kAheadWorking, kAheadReady, kAheadFailed };
    AheadState aheadState_ = kAheadIdle;
    size_t aheadStep_ = 0;
    std::string aheadError_;
    size_t l2gMark_ = 0;              // globalToLocal_.size() before the prepared batch
    std::vector<size_t> l2gEnvMark_;  // localToGlobal_[r].size() before it
    std::thread aheadThread_;
    std::mutex aheadMutex_;
    std::condition_variable aheadCv_;
    bool aheadStop_ = false;
    std::atomic<uint64_t> aheadKicks_{0};  // requests so far (the ahead thread polls it before it sleeps)
    uint64_t aheadSeen_ = 0;
    std::atomic<bool> aheadBusy_{false};   // a request is being worked on (the caller polls it before it sleeps)
    std::atomic<uint64_t> preparing_{0};  // the step whose batch is being prepared
    std::atomic<uint64_t> submitted_{0};  // steps handed to the device so far (a priority-collision query of step s waits for s)
    std::vector<std::thread> workers_;
    std::mutex poolMutex_, queryMutex_;
    std::condition_variable poolCv_;
    std::atomic<uint64_t> poolGeneration_{0};
    bool poolStop_ = false;
    std::atomic<int> nextEnv_{0}, envsDone_{0};
    std::string poolError_;
};

}  // namespace cfa
