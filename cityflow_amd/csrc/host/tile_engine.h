// One road network over several device engines (reference: none — CityFlow's only parallelism is the
// thread pool of Engine::startThread engine.cpp:19-31, which splits vehicles/roads/intersections over threads that
// share one address space).  Here the network is cut into TILES of intersections; every tile is a complete cfx
// engine on its own sub-network and GPU, and the tiles exchange a one-lane halo once per step (cfx_halo_*).
//
// Ownership: a tile owns a set of intersections, all their laneLinks, and every lane that ENDS in one of them
// (lanes into a virtual border intersection belong to the tile of the intersection they start from).  A cut
// lane — upstream intersection in tile A, downstream in tile B — is owned by B; A keeps it as a GHOST lane that
// only carries a frozen proxy of the lane's current tail vehicle, which is all that phases 2-7 of A ever read
// from it (Lane::canEnter roadnet.cpp:437-445, the leader search vehicle.cpp:157-196, the `u` source of
// threadNotifyCross engine.cpp:331-342).  Per step and cut lane the halo carries: A -> B the vehicles that crossed
// (with their committed state), B -> A the lane's tail.  Every process runs the same host spawner (same mt19937
// stream), so static per-vehicle data and the waiting queues need no communication.
//
// Precondition (checked): cut lanes are longer than any vehicle's look-ahead, so nothing in A reads past the tail
// of a ghost lane and nothing in B reads upstream of an import lane.
#pragma once

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "archive.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <tuple>

#include "engine_host.h"
#include "replay.h"

namespace cfa {

struct AheadAbandoned {};  // thrown through the spawner by a priority-collision query on the ahead thread (TiledEngineHost)

// rows x cols blocks over the (sorted distinct) coordinates of the non-virtual intersections; virtual
// intersections follow their only neighbour.  Returns the owning tile per intersection.
std::vector<int> gridPartition(const HostRoadNet &net, int rows, int cols);

struct TilePeer {
    int rank = -1;
    int sendOff = 0, sendBytes = 0, recvOff = 0, recvBytes = 0;
};

// Sub-network of one tile + the halo layout agreed with its neighbours.
struct TileNet {
    int rank = 0;
    std::vector<int32_t> laneL2G, llL2G, interL2G, roadL2G;  // local -> global index
    std::vector<int32_t> laneG2L, llG2L, interG2L, roadG2L;  // global -> local index or -1
    std::vector<uint8_t> laneGhost;                          // per local lane
    std::vector<int32_t> ghostLane, ghostSendOff, ghostRecvOff, importLane, importRecvOff, importSendOff;
    std::vector<TilePeer> peers;  // ascending rank
    int sendBytes = 0, recvBytes = 0;
    cfx_net flat{};

    void build(const HostRoadNet &net, const std::vector<int> &owner, int rank);

private:
    std::vector<double> drvLength_, drvMaxSpeed_, xDist_, phaseTime_;
    std::vector<int32_t> laneRoad_, laneIndex_, laneLLStart_, laneLL_, roadLaneStart_, llStartLane_, llEndLane_, llInter_,
        llRoadLink_, llType_, llXStart_, xPeer_, xLL_, interVirtual_, interNRL_, interPhaseStart_, interAvailStart_;
    std::vector<uint8_t> phaseAvail_;
};

// One tile: a cfx engine on the tile's sub-network.  Speaks global indices to its caller.
class TileEngine {
public:
    TileEngine(std::shared_ptr<HostRoadNet> net, const std::vector<int> &owner, int rank, const EngineConfig &cfg,
               Backend *be, int device);
    ~TileEngine();
    TileEngine(const TileEngine &) = delete;
    TileEngine &operator=(const TileEngine &) = delete;

    void uploadTables(const Spawner &sp);
    void step(const std::vector<cfx_spawn> &globalRecs);
    void haloExport();  // -> send
    void haloImport();  // <- recv
    // the same with the messages left in / taken from the engine's device-resident buffers (a device-to-device transport
    // such as RCCL send / recv moves them); the export returns when the message is complete
    void haloExportDevice();
    void haloImportDevice();
    void deviceBuffers(void **send, void **recv);
    // device-initiated exchange through shared-memory mailboxes named <prefix>_<from>_<to> (POSIX shm)
    void attachMailboxes(const std::string &prefix);
    void unlinkMailboxes();  // remove the names once every process has mapped them
    // ... or through mailboxes in the RECEIVER's device memory (peer HBM over xGMI; hipIpc between processes).  Two phases
    // with a barrier between them: every tile allocates the mailboxes of its incoming messages and publishes their handles
    // (tiny shm records <prefix>_h_<from>_<to>), then opens the ones it sends into.  false: not possible here.
    bool allocDeviceMailboxes(const std::string &prefix);
    bool mailboxesFineGrained() const { return be_->cfx_halo_mailbox_fine_grained(dev_) != 0; }
    std::string deviceIdentity() const {
        char buf[64] = {0};
        return be_->cfx_device_identity(dev_, buf, (int32_t) sizeof buf) == CFX_OK ? std::string(buf) : std::string("?");
    }
    bool attachDeviceMailboxes(const std::string &prefix);
    const char *mailboxKind() const { return !mailboxes_ ? "none" : (deviceMailboxes_ ? "device" : "host"); }
    void haloPost();
    void haloWait();
    bool hasMailboxes() const { return mailboxes_; }
    void reset();
    void sync();
    void profileEnable(bool on);
    void deviceSpin(long long microseconds) { check(be_->cfx_device_spin(dev_, microseconds), "cfx_device_spin"); }
    std::string layoutName() {
        const int l = be_->cfx_get_layout(dev_);
        return l == CFX_LAYOUT_RING ? "ring" : (l == CFX_LAYOUT_DENSE ? "dense" : "n/a");
    }
    std::map<std::string, std::pair<double, int64_t>> profileRead();  // kernel -> (total ms, launches)

    void addLaneCounts(std::vector<int32_t> &global, bool waiting);  // writes the lanes this tile owns
    cfx_scalars scalars();
    void mergeStatus(int first, int n, uint8_t *inout);              // inout[i] = max(inout[i], local state)
    void setPhases(const std::vector<int32_t> &globalInter, const std::vector<int32_t> &phase);  // owned ones are applied
    // owned running vehicles, global drivable ids (unsorted across tiles); customSpeed: their pending custom speeds (NaN none)
    void appendVehicles(VehicleSnapshot &out, std::vector<double> *customSpeed = nullptr);
    // vehicles queued on the lanes this tile owns, queue by queue in FIFO order; lanes: their (global) lanes
    void appendWaiting(std::vector<int32_t> &vids, std::vector<int32_t> *lanes = nullptr);
    void setVehicleSpeed(int vid, double speed);
    void setVehicleRoute(int vid, int route);  // on every tile: a vehicle takes its route along when it migrates
    // traffic lights of the intersections this tile owns, into the network-wide arrays
    void trafficLights(const std::vector<int> &owner, std::vector<int32_t> &phase, std::vector<double> &remain);
    // Archive::resume for this tile's part of a network-wide archive (archive.cpp:73-126): the vehicles on the drivables it
    // owns, the proxy (= the tail) of every ghost lane, the waiting buffers of its lanes (owned and mirrored), its lights.
    // The job's totals (finished vehicles, travel-time sum, vehicle steps) go to ONE tile (takesTotals): the job sums them.
    void loadState(const Archive &a, bool takesTotals);
    // Lane::history (roadnet.cpp:900-915; cfx_get / cfx_set_lane_history) of the lanes this tile OWNS, in the network-wide arrays
    // of an archive (a ghost lane's rows here record its proxy: the owner has the lane's)
    void laneHistoryInto(DeviceState &d);
    void setLaneHistory(const DeviceState &d);

    const TileNet &tile() const { return tn_; }
    std::vector<char> send, recv;

private:
    void check(int32_t rc, const char *what);
    std::shared_ptr<HostRoadNet> net_;
    TileNet tn_;
    Backend *be_;
    cfx_engine *dev_ = nullptr;
    int templatesUploaded_ = 0, routesUploaded_ = 0;
    std::vector<cfx_spawn> recs_;
    bool mailboxes_ = false, deviceMailboxes_ = false;
    std::vector<void *> recvBoxes_;  // device mailboxes of the incoming messages, by peer
    struct Mapping {
        std::string name;
        void *ptr;
        size_t bytes;
    };
    std::vector<Mapping> maps_;
};

// The tiles this process runs (all of them: one process drives the whole network, e.g. several tiles on one GPU for
// tests; or one: the torch.distributed launch, one process per GPU) behind the reference's Engine methods.
class TiledEngineHost {
public:
    // localTiles empty => all tiles are local
    TiledEngineHost(const std::string &configFile, int rows, int cols, const std::vector<int> &localTiles,
                    const std::string &backendLib = "");

    // ---- stepping.  All tiles local: nextStep() does everything.  Otherwise the caller moves the halo between
    //      stepBegin() and stepEnd(): send/recv buffers of local tile i via sendBuffer(i)/recvBuffer(i), peers(i).
    void nextStep();
    void stepBegin();
    void stepEnd();
    void stepBeginDevice();  // ... with the halo of every local tile left in its device buffers (haloDeviceBuffers)
    void stepEndDevice();
    std::tuple<uintptr_t, int, uintptr_t, int> haloDeviceBuffers(int i);
    // Switch the halo to device-initiated mailboxes (all tiles of the job must share one node).  `jobId` must be
    // unique per job and identical on every process; call unlinkMailboxes() after all processes have enabled.
    void enableMailboxes(const std::string &jobId);
    void unlinkMailboxes();
    // The same with the mailboxes in device memory.  All tiles local: enableDeviceMailboxes() does both phases (false:
    // not possible, nothing changed — use enableMailboxes).  One tile per process: phase 1, a barrier of the caller's, phase
    // 2, and every process must have succeeded in both.
    bool enableDeviceMailboxes(const std::string &jobId);
    bool deviceMailboxPhase(const std::string &jobId, int phase);
    std::string haloTransport() const;
    int nTiles() const { return nTiles_; }
    int nLocal() const { return (int) tiles_.size(); }
    int localRank(int i) const { return localRanks_[i]; }
    std::vector<char> &sendBuffer(int i) { return tiles_[i]->send; }
    std::vector<char> &recvBuffer(int i) { return tiles_[i]->recv; }
    const std::vector<TilePeer> &peers(int i) const { return tiles_[i]->tile().peers; }

    // ---- reference API subset (values of the lanes / vehicles of the LOCAL tiles; callers in a multi-process
    //      launch reduce them: counts by sum, status by max)
    std::vector<int32_t> laneVehicleCountArray();
    std::vector<int32_t> laneWaitingVehicleCountArray();
    std::map<std::string, int> getLaneVehicleCount();
    std::map<std::string, int> getLaneWaitingVehicleCount();
    size_t getVehicleCount();
    cfx_scalars scalars();                     // sums over local tiles (step / spawned are global)
    double getCurrentTime() const { return step_ * cfg_.interval; }
    void setTrafficLightPhase(const std::string &id, int phaseIndex);
    void setTrafficLightPhases(const std::vector<int32_t> &phases);  // [n_intersections]
    void reset(bool resetRnd);
    // the rest of the reference's query / control API, over the vehicles of the local tiles (all of them when every
    // tile is local): same semantics as EngineHost (engine.cpp:615-720,827-850)
    std::vector<std::string> getVehicles(bool includeWaiting);
    std::map<std::string, std::vector<std::string>> getLaneVehicles();
    std::map<std::string, double> getVehicleSpeed();
    std::map<std::string, double> getVehicleDistance();
    std::string getLeader(const std::string &vehicleId);
    std::map<std::string, std::string> getVehicleInfo(const std::string &vehicleId);
    double getAverageTravelTime();
    // several processes: the pieces the caller reduces over the ranks (cityflow_amd/tiled.py: DistributedEngine)
    std::vector<std::pair<int32_t, std::string>> pendingPushedKeyed() const;  // pushed since the last step (same on every rank)
    bool isPendingPushed(const std::string &id) const;
    std::vector<std::pair<int32_t, std::string>> vehiclesKeyed(bool includeWaiting);  // local {priority, id}
    bool runsHere(const std::string &vehicleId);                                       // on a drivable one of the local tiles owns
    std::vector<uint8_t> localStatus();                                                // per vehicle number, merged over the local tiles
    double averageTravelTimeFrom(double cumulative, int64_t finished, const std::vector<uint8_t> &status) const;
    void pushVehicle(const std::map<std::string, double> &info, const std::vector<std::string> &roads);
    void setVehicleSpeed(const std::string &id, double speed);
    void setRandomSeed(int seed) {
        dropAhead();
        spawner_.seed(seed);
    }
    void snapshotVehicles(VehicleSnapshot &out);  // local tiles, sorted by global drivable
    // ---- archive (reference src/engine/archive.cpp; EngineHost::snapshot / load / loadFromFile): every process loads the
    //      same archive and keeps its tiles' part.  A snapshot is assembled from one PART per process (opaque bytes: the
    //      state of the local tiles); with every tile local snapshot() does both steps.
    std::string snapshotPart();
    Archive snapshotFromParts(const std::vector<std::string> &parts, bool keepRoutePositions = false);
    Archive snapshot();
    void load(const Archive &a);
    // Forget the finished vehicles (the reference frees a vehicle when it finishes, engine.cpp:296-310; EngineHost::compactVehicles,
    // archive.cpp): the vehicles that still wait or run renumbered 0 .. n-1 in their old order and loaded back through load().
    // `parts`: snapshotPart() of every process, in rank order — every process calls it with the same parts between the same two
    // steps (every process runs the whole spawner, so wantsCompaction() answers alike everywhere).  Automatic with every tile in
    // this process; cityflow_amd/tiled.py does the gather otherwise.  "cfx": {"compactVehicles": N} as for Engine.
    void compactFromParts(const std::vector<std::string> &parts);
    void compactVehicles();
    // (the vehicle numbers out as of the batch the last step took: the step-ahead thread may be adding the next step's while
    //  this is asked, and how far it has come differs from rank to rank)
    bool wantsCompaction() const { return compactAt_ > 0 && numbersOut_ >= nextCompactAt_; }
    bool keepsLaneHistory() const { return keepsHistory_; }
    int64_t vehicleCompactions() const { return vehicleCompactions_; }
    int64_t vehicleTableSize() const { return (int64_t) numbersOut_; }
    void loadFromFile(const std::string &path);
    // Engine::setRoute engine.cpp:852-866.  Several processes: the status reducer also merges the vehicle's position.
    bool setRoute(const std::string &vehicleId, const std::vector<std::string> &anchorIds);
    // replay (engine.cpp:518-554,773-790): written by the process that runs every tile
    void setReplayLogFile(const std::string &logFile);
    void setSaveReplay(bool open);
    // one tile per process: after every step each process's replayPart() goes to the process of tile 0, which writes the line
    bool wantsReplay() const { return saveReplay_; }
    std::string replayPart();
    void replayWrite(const std::vector<std::string> &parts);
    void sync();
    // cumulative host wall time of this process since the last reset: {inside the spawner, submitting the steps (kernel
    // launches; includes back-pressure waits when the device is the bottleneck)}
    std::pair<double, double> hostSeconds() const { return std::make_pair(hostSpawnSec_, hostSubmitSec_); }
    // the step-ahead thread: {batches it prepared that a step took, batches a step had to make itself (the first step, after a
    // call that took a prepared step back, a priority collision), seconds the thread was busy}
    std::tuple<int64_t, int64_t, double> aheadStats() const { return std::make_tuple(aheadTaken_, aheadRedone_, aheadBusySec_.load()); }
    void profileEnable(int localTile, bool on) { tiles_.at(localTile)->profileEnable(on); }
    void deviceSpin(long long microseconds) { for (auto &t : tiles_) t->deviceSpin(microseconds); }
    std::string backendName() const { return be_.cfx_backend_name ? be_.cfx_backend_name() : "?"; }
    bool deviceMailboxesFineGrained() const {
        for (auto &t : tiles_)
            if (!t->mailboxesFineGrained()) return false;
        return true;
    }
    // the physical devices of this process's tiles (equal strings = one device), in tile order
    std::vector<std::string> deviceIdentities() const {
        std::vector<std::string> out;
        for (auto &t : tiles_) out.push_back(t->deviceIdentity());
        return out;
    }
    std::string layoutName() { return tiles_.empty() ? "n/a" : tiles_.front()->layoutName(); }
    std::map<std::string, std::pair<double, int64_t>> profileRead(int localTile) { return tiles_.at(localTile)->profileRead(); }
    std::vector<int> owner() const { return owner_; }
    std::vector<std::string> laneIds() const;
    const HostRoadNet &net() const { return *net_; }
    std::string vehicleId(int vid) {
        waitAhead();  // (the vehicle table may be growing on the ahead thread)
        return spawner_.vehicleId(vid);
    }
    // max-reduction of a vehicle status over all processes; identity when every tile is local
    void setStatusReducer(std::function<int(int)> r) { reduceStatus_ = std::move(r); }

private:
    EngineConfig cfg_;
    std::shared_ptr<HostRoadNet> net_ = std::make_shared<HostRoadNet>();
    Spawner spawner_;
    Backend be_;
    std::vector<int> owner_, localRanks_;
    std::vector<std::unique_ptr<TileEngine>> tiles_;
    int nTiles_ = 1;
    bool allLocal_ = true, mailboxes_ = false;
    bool keepsHistory_ = false;  // Lane::history kept by the tiles (ring layout; EngineHost's default: networks up to 20 k lanes)
    size_t numbersOut_ = 0;  // vehicle numbers handed out as of the last batch taken / load / compaction
    size_t compactAt_ = 3500000, nextCompactAt_ = 3500000;  // (EngineHost's policy: 3.5 M vehicle numbers, then 32 x the vehicles alive)
    bool compactAuto_ = true;
    int64_t vehicleCompactions_ = 0;
    std::map<int32_t, double> waitingCustom_;  // set_vehicle_speed on vehicles that were waiting (or not yet numbered) then
    size_t step_ = 0;
    std::vector<cfx_spawn> spawnBuf_;
    double hostSpawnSec_ = 0, hostSubmitSec_ = 0;  // wall time of this process inside the spawner / the ABI calls of a step
    std::vector<int32_t> pendingInter_, pendingPhase_;
    std::function<int(int)> reduceStatus_;
    ReplayWriter replay_;
    bool saveReplay_ = false, saveReplayInConfig_ = false, replayWriter_ = true;
    void updateLog();
    void flushPhases();
    int statusOf(int vid);  // merged over the local tiles (and the reducer)
    // ---- the spawner of step t+1 runs on a host thread of its own while step t is submitted and runs (every rank runs the
    //      whole spawner: at 30x30 it was two thirds of a rank's host time per step and the tiles were host-bound).  As in
    //      EngineHost / VectorEngineHost: Flow::nextStep + planRoute (flow.cpp:6-22, engine.cpp:450-470) depend on nothing a
    //      step computes; the step ahead is journalled (Spawner::beginAhead) and any call that could see or change the
    //      spawner's state takes it back first (dropAhead).  A priority collision — the one thing that asks the devices, and
    //      over several ranks a collective — is never answered on the ahead thread: it abandons the step, which nextStep()
    //      then takes plainly, on every rank alike.  `"cfx": {"spawnAhead": false}` turns it off; off with saveReplay.
    void takeBatch();   // this step's spawn records into spawnBuf_ (prepared ahead, or now)
    void kickAhead();   // start the spawner of step_ + 1
    void waitAhead();
    void dropAhead();
    void aheadLoop();
    size_t committedVehicleCount();  // vehicles created by the steps that were TAKEN
    bool aheadEnabled_ = false;
    enum AheadState { kAheadIdle, kAheadWorking, kAheadReady, kAheadAbandoned, kAheadFailed };
    AheadState aheadState_ = kAheadIdle;
    size_t aheadStep_ = 0;
    std::vector<cfx_spawn> aheadBuf_;
    std::string aheadError_;
    std::thread aheadThread_;
    std::mutex aheadMutex_;
    std::condition_variable aheadCv_;
    bool aheadStop_ = false;
    std::atomic<uint64_t> aheadKicks_{0};  // requests so far (the ahead thread polls it before it sleeps)
    uint64_t aheadSeen_ = 0;
    std::atomic<bool> aheadBusy_{false};   // a request is being worked on (the caller polls it before it sleeps)
    std::atomic<bool> onAheadThread_{false};
    int64_t aheadTaken_ = 0, aheadRedone_ = 0;
    std::atomic<double> aheadBusySec_{0.0};
public:
    ~TiledEngineHost();
};

}  // namespace cfa
