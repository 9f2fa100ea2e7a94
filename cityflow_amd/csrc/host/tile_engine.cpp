#include "tile_engine.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <iostream>
#include <numeric>
#include <stdexcept>

#include "json.h"

namespace cfa {

// ---------------------------------------------------------------- partition
std::vector<int> gridPartition(const HostRoadNet &net, int rows, int cols) {
    if (rows < 1 || cols < 1) throw std::runtime_error("tiling: rows and cols must be positive");
    const int I = (int) net.inters.size();
    std::vector<double> xs, ys;
    for (const HostInter &in : net.inters)
        if (!in.isVirtual) {
            xs.push_back(in.point.x);
            ys.push_back(in.point.y);
        }
    auto distinct = [](std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    };
    distinct(xs);
    distinct(ys);
    if ((int) xs.size() < cols || (int) ys.size() < rows)
        throw std::runtime_error("tiling: more tiles than intersection rows/columns");
    auto block = [](const std::vector<double> &v, double x, int parts) {
        int idx = (int) (std::lower_bound(v.begin(), v.end(), x) - v.begin());
        return (int) ((long long) idx * parts / (long long) v.size());
    };
    std::vector<int> owner(I, -1);
    for (int i = 0; i < I; ++i) {
        const HostInter &in = net.inters[i];
        if (!in.isVirtual) owner[i] = block(ys, in.point.y, rows) * cols + block(xs, in.point.x, cols);
    }
    for (int i = 0; i < I; ++i) {
        if (owner[i] >= 0) continue;
        for (int r : net.inters[i].roads) {
            int other = net.roads[r].startInter == i ? net.roads[r].endInter : net.roads[r].startInter;
            if (other >= 0 && owner[other] >= 0) {
                owner[i] = owner[other];
                break;
            }
        }
        if (owner[i] < 0) owner[i] = 0;  // isolated virtual intersection
    }
    return owner;
}

// ---------------------------------------------------------------- sub-network
void TileNet::build(const HostRoadNet &net, const std::vector<int> &owner, int rk) {
    rank = rk;
    const cfx_net &g = net.flat();
    const int L = g.n_lanes, K = g.n_lanelinks, I = g.n_inters, R = g.n_roads;
    auto upOf = [&](int lane) { return owner[net.roads[g.lane_road[lane]].startInter]; };
    auto downOf = [&](int lane) { return owner[net.roads[g.lane_road[lane]].endInter]; };

    laneG2L.assign(L, -1);
    llG2L.assign(K, -1);
    interG2L.assign(I, -1);
    roadG2L.assign(R, -1);
    laneL2G.clear();
    llL2G.clear();
    interL2G.clear();
    roadL2G.clear();
    laneGhost.clear();
    for (int l = 0; l < L; ++l) {
        const bool owned = downOf(l) == rank;
        if (!owned && upOf(l) != rank) continue;
        laneG2L[l] = (int32_t) laneL2G.size();
        laneL2G.push_back(l);
        laneGhost.push_back(owned ? 0 : 1);
        const int r = g.lane_road[l];
        if (roadG2L[r] < 0) {  // lanes are grouped by road and roads ascend with lanes
            roadG2L[r] = (int32_t) roadL2G.size();
            roadL2G.push_back(r);
        }
    }
    for (int k = 0; k < K; ++k)
        if (owner[g.ll_inter[k]] == rank) {
            llG2L[k] = (int32_t) llL2G.size();
            llL2G.push_back(k);
        }
    for (int i = 0; i < I; ++i)
        if (owner[i] == rank) {
            interG2L[i] = (int32_t) interL2G.size();
            interL2G.push_back(i);
        }
    const int nL = (int) laneL2G.size(), nK = (int) llL2G.size(), nI = (int) interL2G.size(), nR = (int) roadL2G.size();

    drvLength_.resize(nL + nK);
    drvMaxSpeed_.resize(nL + nK);
    laneRoad_.resize(nL);
    laneIndex_.resize(nL);
    laneLLStart_.assign(nL + 1, 0);
    laneLL_.clear();
    roadLaneStart_.assign(nR + 1, 0);
    for (int l = 0; l < nL; ++l) {
        const int gl = laneL2G[l];
        drvLength_[l] = g.drv_length[gl];
        drvMaxSpeed_[l] = g.drv_max_speed[gl];
        laneRoad_[l] = roadG2L[g.lane_road[gl]];
        laneIndex_[l] = g.lane_index[gl];
        laneLLStart_[l] = (int32_t) laneLL_.size();
        for (int q = g.lane_ll_start[gl]; q < g.lane_ll_start[gl + 1]; ++q) {
            int k = llG2L[g.lane_ll[q]];
            if (k >= 0) laneLL_.push_back(k);
        }
        roadLaneStart_[laneRoad_[l] + 1] = l + 1;
    }
    laneLLStart_[nL] = (int32_t) laneLL_.size();
    for (int r = 0; r < nR; ++r) roadLaneStart_[r + 1] = std::max(roadLaneStart_[r + 1], roadLaneStart_[r]);

    llStartLane_.resize(nK);
    llEndLane_.resize(nK);
    llInter_.resize(nK);
    llRoadLink_.resize(nK);
    llType_.resize(nK);
    llXStart_.assign(nK + 1, 0);
    xDist_.clear();
    xLL_.clear();
    xPeer_.clear();
    std::vector<int32_t> eG2L((size_t) g.n_xentries, -1), eL2G;
    for (int k = 0; k < nK; ++k) {
        const int gk = llL2G[k];
        drvLength_[nL + k] = g.drv_length[L + gk];
        drvMaxSpeed_[nL + k] = g.drv_max_speed[L + gk];
        llStartLane_[k] = laneG2L[g.ll_start_lane[gk]];
        llEndLane_[k] = laneG2L[g.ll_end_lane[gk]];
        if (llStartLane_[k] < 0 || llEndLane_[k] < 0) throw std::runtime_error("tiling: laneLink with a lane outside its tile");
        llInter_[k] = interG2L[g.ll_inter[gk]];
        llRoadLink_[k] = g.ll_roadlink[gk];
        llType_[k] = g.ll_type[gk];
        llXStart_[k] = (int32_t) xDist_.size();
        for (int e = g.ll_x_start[gk]; e < g.ll_x_start[gk + 1]; ++e) {
            eG2L[e] = (int32_t) xDist_.size();
            eL2G.push_back(e);
            xDist_.push_back(g.x_dist[e]);
            xLL_.push_back(k);
        }
    }
    llXStart_[nK] = (int32_t) xDist_.size();
    xPeer_.resize(xDist_.size());
    for (size_t e = 0; e < eL2G.size(); ++e) {
        xPeer_[e] = eG2L[g.x_peer[eL2G[e]]];
        if (xPeer_[e] < 0) throw std::runtime_error("tiling: cross between laneLinks of different tiles");
    }

    interVirtual_.resize(nI);
    interNRL_.resize(nI);
    interPhaseStart_.assign(nI + 1, 0);
    interAvailStart_.resize(nI);
    phaseTime_.clear();
    phaseAvail_.clear();
    for (int i = 0; i < nI; ++i) {
        const int gi = interL2G[i];
        interVirtual_[i] = g.inter_virtual[gi];
        interNRL_[i] = g.inter_n_roadlinks[gi];
        interPhaseStart_[i] = (int32_t) phaseTime_.size();
        interAvailStart_[i] = (int32_t) phaseAvail_.size();
        const int np = g.inter_phase_start[gi + 1] - g.inter_phase_start[gi];
        for (int p = 0; p < np; ++p) phaseTime_.push_back(g.phase_time[g.inter_phase_start[gi] + p]);
        const uint8_t *av = g.phase_avail + g.inter_avail_start[gi];
        phaseAvail_.insert(phaseAvail_.end(), av, av + (size_t) np * g.inter_n_roadlinks[gi]);
    }
    interPhaseStart_[nI] = (int32_t) phaseTime_.size();

    flat = cfx_net{};
    flat.n_roads = nR;
    flat.n_lanes = nL;
    flat.n_lanelinks = nK;
    flat.n_inters = nI;
    flat.n_xentries = (int32_t) xDist_.size();
    flat.n_phases = (int32_t) phaseTime_.size();
    flat.n_avail = (int32_t) phaseAvail_.size();
    flat.drv_length = drvLength_.data();
    flat.drv_max_speed = drvMaxSpeed_.data();
    flat.lane_road = laneRoad_.data();
    flat.lane_index = laneIndex_.data();
    flat.lane_ll_start = laneLLStart_.data();
    flat.lane_ll = laneLL_.data();
    flat.road_lane_start = roadLaneStart_.data();
    flat.ll_start_lane = llStartLane_.data();
    flat.ll_end_lane = llEndLane_.data();
    flat.ll_inter = llInter_.data();
    flat.ll_roadlink = llRoadLink_.data();
    flat.ll_type = llType_.data();
    flat.ll_x_start = llXStart_.data();
    flat.x_dist = xDist_.data();
    flat.x_peer = xPeer_.data();
    flat.x_ll = xLL_.data();
    flat.inter_virtual = interVirtual_.data();
    flat.inter_n_roadlinks = interNRL_.data();
    flat.inter_phase_start = interPhaseStart_.data();
    flat.inter_avail_start = interAvailStart_.data();
    flat.phase_time = phaseTime_.data();
    flat.phase_avail = phaseAvail_.data();

    // ---- halo layout.  The message between two tiles lists their cut lanes in ascending global lane id; each
    //      lane contributes a migrant block in the upstream -> downstream message and a tail block the other way.
    struct Cut {
        int lane, peer;
        bool up;  // this tile is the upstream side
    };
    std::vector<Cut> cuts;
    for (int l = 0; l < L; ++l) {
        const int u = upOf(l), d = downOf(l);
        if (u == d) continue;
        if (u == rank) cuts.push_back({l, d, true});
        else if (d == rank) cuts.push_back({l, u, false});
    }
    peers.clear();
    {
        std::vector<int> ranks;
        for (const Cut &c : cuts) ranks.push_back(c.peer);
        std::sort(ranks.begin(), ranks.end());
        ranks.erase(std::unique(ranks.begin(), ranks.end()), ranks.end());
        for (int r : ranks) {
            TilePeer p;
            p.rank = r;
            peers.push_back(p);
        }
    }
    auto peerOf = [&](int r) -> TilePeer & {
        for (TilePeer &p : peers)
            if (p.rank == r) return p;
        throw std::logic_error("tiling: unknown peer");
    };
    for (const Cut &c : cuts) {
        TilePeer &p = peerOf(c.peer);
        p.sendBytes += c.up ? CFX_HALO_MIG_BYTES : CFX_HALO_TAIL_BYTES;
        p.recvBytes += c.up ? CFX_HALO_TAIL_BYTES : CFX_HALO_MIG_BYTES;
    }
    sendBytes = recvBytes = 0;
    for (TilePeer &p : peers) {
        p.sendOff = sendBytes;
        p.recvOff = recvBytes;
        sendBytes += p.sendBytes;
        recvBytes += p.recvBytes;
    }
    ghostLane.clear();
    ghostSendOff.clear();
    ghostRecvOff.clear();
    importLane.clear();
    importRecvOff.clear();
    importSendOff.clear();
    std::vector<int> sendCur(peers.size()), recvCur(peers.size());
    for (size_t i = 0; i < peers.size(); ++i) {
        sendCur[i] = peers[i].sendOff;
        recvCur[i] = peers[i].recvOff;
    }
    for (const Cut &c : cuts) {
        size_t pi = 0;
        while (peers[pi].rank != c.peer) ++pi;
        if (c.up) {
            ghostLane.push_back(laneG2L[c.lane]);
            ghostSendOff.push_back(sendCur[pi]);
            ghostRecvOff.push_back(recvCur[pi]);
            sendCur[pi] += CFX_HALO_MIG_BYTES;
            recvCur[pi] += CFX_HALO_TAIL_BYTES;
        } else {
            importLane.push_back(laneG2L[c.lane]);
            importSendOff.push_back(sendCur[pi]);
            importRecvOff.push_back(recvCur[pi]);
            sendCur[pi] += CFX_HALO_TAIL_BYTES;
            recvCur[pi] += CFX_HALO_MIG_BYTES;
        }
    }
}

// ---------------------------------------------------------------- one tile
TileEngine::TileEngine(std::shared_ptr<HostRoadNet> net, const std::vector<int> &owner, int rank, const EngineConfig &cfg,
                       Backend *be, int device)
    : net_(std::move(net)), be_(be) {
    tn_.build(*net_, owner, rank);
    cfx_config cc{};
    cfg.apply(cc);
    cc.lane_change = 0;
    cc.device = device;
    int32_t rc = be_->cfx_create(&tn_.flat, &cc, &dev_);
    if (rc != CFX_OK || !dev_) {
        const char *msg = be_->cfx_last_error(nullptr);
        throw std::runtime_error(std::string("cityflow_amd: cfx_create (tile ") + std::to_string(rank) + ") failed in " +
                                 be_->path + ": " + (msg ? msg : "unknown error") + " (there is no CPU fallback)");
    }
    cfx_halo_layout h{};
    h.n_ghost = (int32_t) tn_.ghostLane.size();
    h.ghost_lane = tn_.ghostLane.data();
    h.ghost_send_off = tn_.ghostSendOff.data();
    h.ghost_recv_off = tn_.ghostRecvOff.data();
    h.n_import = (int32_t) tn_.importLane.size();
    h.import_lane = tn_.importLane.data();
    h.import_recv_off = tn_.importRecvOff.data();
    h.import_send_off = tn_.importSendOff.data();
    h.send_bytes = tn_.sendBytes;
    h.recv_bytes = tn_.recvBytes;
    h.n_global_lanelinks = (int32_t) tn_.llG2L.size();
    h.lanelink_global = tn_.llL2G.data();
    h.lanelink_local = tn_.llG2L.data();
    check(be_->cfx_halo_config(dev_, &h), "cfx_halo_config");
    send.assign((size_t) tn_.sendBytes, 0);
    recv.assign((size_t) tn_.recvBytes, 0);
}

TileEngine::~TileEngine() {
    if (dev_) be_->cfx_destroy(dev_);  // unregisters the mailboxes before they are unmapped
    for (Mapping &m : maps_) munmap(m.ptr, m.bytes);
}

static void *mapShared(const std::string &name, size_t bytes) {
    int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) throw std::runtime_error("tiling: shm_open(" + name + ") failed: " + strerror(errno));
    if (ftruncate(fd, (off_t) bytes) != 0) {
        close(fd);
        throw std::runtime_error("tiling: ftruncate(" + name + ") failed: " + strerror(errno));
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) throw std::runtime_error("tiling: mmap(" + name + ") failed: " + strerror(errno));
    return p;
}

void TileEngine::attachMailboxes(const std::string &prefix) {
    std::vector<cfx_halo_peer> peers;
    for (const TilePeer &p : tn_.peers) {
        cfx_halo_peer hp{};
        hp.send_off = p.sendOff;
        hp.send_bytes = p.sendBytes;
        hp.recv_off = p.recvOff;
        hp.recv_bytes = p.recvBytes;
        // both ends create-or-open the same name with the same size (fresh shm is zero: epoch 0 = nothing published)
        Mapping out{prefix + "_" + std::to_string(tn_.rank) + "_" + std::to_string(p.rank), nullptr,
                    CFX_HALO_MAILBOX_BYTES(p.sendBytes)};
        out.ptr = mapShared(out.name, out.bytes);
        maps_.push_back(out);
        Mapping in{prefix + "_" + std::to_string(p.rank) + "_" + std::to_string(tn_.rank), nullptr,
                   CFX_HALO_MAILBOX_BYTES(p.recvBytes)};
        in.ptr = mapShared(in.name, in.bytes);
        maps_.push_back(in);
        hp.send_mailbox = out.ptr;
        hp.recv_mailbox = in.ptr;
        peers.push_back(hp);
    }
    check(be_->cfx_halo_attach(dev_, (int32_t) peers.size(), peers.data()), "cfx_halo_attach");
    mailboxes_ = true;
}

namespace {
struct BoxRecord {  // what the receiver of a message publishes about its mailbox
    long long pid;
    unsigned long long ptr;
    uint8_t handle[CFX_IPC_HANDLE_BYTES];
    int32_t ready;
};
}  // namespace

bool TileEngine::allocDeviceMailboxes(const std::string &prefix) {
    if (!be_->cfx_halo_mailbox_alloc) return false;
    recvBoxes_.clear();
    for (const TilePeer &p : tn_.peers) {
        void *ptr = nullptr;
        BoxRecord rec{};
        if (be_->cfx_halo_mailbox_alloc(dev_, p.recvBytes, &ptr, rec.handle) != CFX_OK) return false;
        recvBoxes_.push_back(ptr);
        rec.pid = (long long) getpid();
        rec.ptr = (unsigned long long) (uintptr_t) ptr;
        rec.ready = 1;
        Mapping m{prefix + "_h_" + std::to_string(p.rank) + "_" + std::to_string(tn_.rank), nullptr, sizeof(BoxRecord)};
        m.ptr = mapShared(m.name, m.bytes);
        memcpy(m.ptr, &rec, sizeof rec);
        maps_.push_back(m);
    }
    return true;
}

bool TileEngine::attachDeviceMailboxes(const std::string &prefix) {
    std::vector<cfx_halo_peer> peers;
    size_t i = 0;
    for (const TilePeer &p : tn_.peers) {
        Mapping m{prefix + "_h_" + std::to_string(tn_.rank) + "_" + std::to_string(p.rank), nullptr, sizeof(BoxRecord)};
        m.ptr = mapShared(m.name, m.bytes);
        maps_.push_back(m);
        BoxRecord rec;
        memcpy(&rec, m.ptr, sizeof rec);
        if (!rec.ready) return false;  // the neighbour did not get this far
        void *sendBox = nullptr;
        if (rec.pid == (long long) getpid()) sendBox = (void *) (uintptr_t) rec.ptr;  // a tile of this very process
        else if (be_->cfx_halo_mailbox_open(dev_, rec.handle, &sendBox) != CFX_OK) return false;
        cfx_halo_peer hp{};
        hp.send_off = p.sendOff;
        hp.send_bytes = p.sendBytes;
        hp.recv_off = p.recvOff;
        hp.recv_bytes = p.recvBytes;
        hp.send_mailbox = sendBox;
        hp.recv_mailbox = recvBoxes_[i++];
        hp.device_memory = 1;
        peers.push_back(hp);
    }
    if (be_->cfx_halo_attach(dev_, (int32_t) peers.size(), peers.data()) != CFX_OK) return false;
    mailboxes_ = deviceMailboxes_ = true;
    return true;
}

void TileEngine::unlinkMailboxes() {
    for (Mapping &m : maps_) shm_unlink(m.name.c_str());  // the mappings stay valid; only the names go away
}

void TileEngine::haloPost() { check(be_->cfx_halo_post(dev_), "cfx_halo_post"); }
void TileEngine::haloWait() { check(be_->cfx_halo_wait(dev_), "cfx_halo_wait"); }

void TileEngine::check(int32_t rc, const char *what) {
    if (rc == CFX_OK) return;
    const char *msg = be_->cfx_last_error(dev_);
    throw std::runtime_error(std::string("cityflow_amd: ") + what + " failed on tile " + std::to_string(tn_.rank) + " (" +
                             std::to_string(rc) + "): " + (msg ? msg : ""));
}

void TileEngine::uploadTables(const Spawner &sp) {
    const int nt = (int) sp.templates.size();
    if (nt > templatesUploaded_) {
        // the halo is one lane deep: nothing may look or move past a cut lane within one step
        double minCut = 1e300;
        for (int l : tn_.ghostLane) minCut = std::min(minCut, tn_.flat.drv_length[l]);
        for (int l : tn_.importLane) minCut = std::min(minCut, tn_.flat.drv_length[l]);
        for (int t = templatesUploaded_; t < nt; ++t)
            if (sp.templates[t].approach_dist + sp.templates[t].len >= minCut)
                throw std::runtime_error("cityflow_amd: tiling needs every cut lane to be longer than a vehicle's look-ahead (" +
                                         std::to_string(sp.templates[t].approach_dist) + " m); shortest cut lane is " +
                                         std::to_string(minCut) + " m");
        check(be_->cfx_add_templates(dev_, nt - templatesUploaded_, sp.templates.data() + templatesUploaded_), "cfx_add_templates");
        templatesUploaded_ = nt;
    }
    const RouteTable &rt = sp.routes;
    const int nr = rt.count();
    if (nr > routesUploaded_) {
        std::vector<int32_t> routeStart{0}, roads, nextStart{0}, nextLL;
        for (int r = routesUploaded_; r < nr; ++r) {
            for (int p = rt.routeStart[r]; p < rt.routeStart[r + 1]; ++p) {
                const int lroad = tn_.roadG2L[rt.roads[p]];
                roads.push_back(lroad);
                if (lroad >= 0)
                    for (int q = rt.nextStart[p]; q < rt.nextStart[p + 1]; ++q)
                        nextLL.push_back(rt.nextLL[q] >= 0 ? tn_.llG2L[rt.nextLL[q]] : -1);
                nextStart.push_back((int32_t) nextLL.size());
            }
            routeStart.push_back((int32_t) roads.size());
        }
        check(be_->cfx_add_routes(dev_, nr - routesUploaded_, routeStart.data(), roads.data(), nextStart.data(), nextLL.data()),
              "cfx_add_routes");
        routesUploaded_ = nr;
    }
}

void TileEngine::step(const std::vector<cfx_spawn> &globalRecs) {
    recs_ = globalRecs;
    for (cfx_spawn &r : recs_) r.lane = tn_.laneG2L[r.lane];
    check(be_->cfx_step(dev_, recs_.data(), (int32_t) recs_.size()), "cfx_step");
}

void TileEngine::haloExport() { check(be_->cfx_halo_export(dev_, send.data()), "cfx_halo_export"); }
void TileEngine::haloImport() { check(be_->cfx_halo_import(dev_, recv.data()), "cfx_halo_import"); }
void TileEngine::haloExportDevice() { check(be_->cfx_halo_export(dev_, nullptr), "cfx_halo_export"); }
void TileEngine::haloImportDevice() { check(be_->cfx_halo_import(dev_, nullptr), "cfx_halo_import"); }
void TileEngine::deviceBuffers(void **s, void **r) { check(be_->cfx_halo_device_buffers(dev_, s, r), "cfx_halo_device_buffers"); }
void TileEngine::reset() { check(be_->cfx_reset(dev_), "cfx_reset"); }
void TileEngine::sync() { check(be_->cfx_sync(dev_), "cfx_sync"); }
void TileEngine::profileEnable(bool on) { check(be_->cfx_profile_enable(dev_, on ? 1 : 0), "cfx_profile_enable"); }

std::map<std::string, std::pair<double, int64_t>> TileEngine::profileRead() {
    int n = be_->cfx_profile_kernel_count();
    std::vector<double> ms(n > 0 ? n : 1);
    std::vector<int64_t> cnt(n > 0 ? n : 1);
    check(be_->cfx_profile_read(dev_, ms.data(), cnt.data()), "cfx_profile_read");
    std::map<std::string, std::pair<double, int64_t>> out;
    for (int k = 0; k < n; ++k) out[be_->cfx_profile_kernel_name(k)] = std::make_pair(ms[k], cnt[k]);
    return out;
}

void TileEngine::addLaneCounts(std::vector<int32_t> &global, bool waiting) {
    std::vector<int32_t> local(tn_.laneL2G.size());
    if (waiting) check(be_->cfx_get_lane_waiting_counts(dev_, local.data()), "cfx_get_lane_waiting_counts");
    else check(be_->cfx_get_lane_counts(dev_, local.data()), "cfx_get_lane_counts");
    for (size_t l = 0; l < local.size(); ++l)
        if (!tn_.laneGhost[l]) global[tn_.laneL2G[l]] = local[l];
}

void TileEngine::laneHistoryInto(DeviceState &d) {
    const size_t nL = tn_.laneL2G.size(), M = CFX_LANE_HISTORY_MAX;
    std::vector<int32_t> len(nL), num(nL * M), hNum(nL);
    std::vector<double> avg(nL * M), hAvg(nL);
    cfx_lane_history h{(int32_t) nL, len.data(), num.data(), avg.data(), hNum.data(), hAvg.data()};
    check(be_->cfx_get_lane_history(dev_, &h), "cfx_get_lane_history");
    for (size_t l = 0; l < nL; ++l) {
        if (tn_.laneGhost[l]) continue;
        const size_t g = (size_t) tn_.laneL2G[l];
        d.hLen[g] = len[l];
        d.hHistoryVehicleNum[g] = hNum[l];
        d.hHistoryAverageSpeed[g] = hAvg[l];
        std::copy(num.begin() + (ptrdiff_t) (l * M), num.begin() + (ptrdiff_t) ((l + 1) * M), d.hVehicleNum.begin() + (ptrdiff_t) (g * M));
        std::copy(avg.begin() + (ptrdiff_t) (l * M), avg.begin() + (ptrdiff_t) ((l + 1) * M), d.hAverageSpeed.begin() + (ptrdiff_t) (g * M));
    }
}

void TileEngine::setLaneHistory(const DeviceState &d) {
    const size_t nL = tn_.laneL2G.size(), M = CFX_LANE_HISTORY_MAX;
    std::vector<int32_t> len(nL, 0), num(nL * M, 0), hNum(nL, 0);
    std::vector<double> avg(nL * M, 0.0), hAvg(nL, 0.0);
    for (size_t l = 0; l < nL; ++l) {  // (a ghost lane's rows are nobody's: the lane's own history for them, harmless)
        const size_t g = (size_t) tn_.laneL2G[l];
        len[l] = d.hLen[g];
        hNum[l] = d.hHistoryVehicleNum[g];
        hAvg[l] = d.hHistoryAverageSpeed[g];
        std::copy(d.hVehicleNum.begin() + (ptrdiff_t) (g * M), d.hVehicleNum.begin() + (ptrdiff_t) ((g + 1) * M), num.begin() + (ptrdiff_t) (l * M));
        std::copy(d.hAverageSpeed.begin() + (ptrdiff_t) (g * M), d.hAverageSpeed.begin() + (ptrdiff_t) ((g + 1) * M), avg.begin() + (ptrdiff_t) (l * M));
    }
    cfx_lane_history h{(int32_t) nL, len.data(), num.data(), avg.data(), hNum.data(), hAvg.data()};
    check(be_->cfx_set_lane_history(dev_, &h), "cfx_set_lane_history");
}

cfx_scalars TileEngine::scalars() {
    cfx_scalars s{};
    check(be_->cfx_get_scalars(dev_, &s), "cfx_get_scalars");
    return s;
}

void TileEngine::mergeStatus(int first, int n, uint8_t *inout) {
    std::vector<uint8_t> st((size_t) std::max(n, 1));
    check(be_->cfx_get_vehicle_status(dev_, first, n, st.data()), "cfx_get_vehicle_status");
    for (int i = 0; i < n; ++i) inout[i] = std::max(inout[i], st[i]);
}

void TileEngine::setPhases(const std::vector<int32_t> &globalInter, const std::vector<int32_t> &phase) {
    std::vector<int32_t> in, ph;
    for (size_t i = 0; i < globalInter.size(); ++i) {
        int li = tn_.interG2L[globalInter[i]];
        if (li < 0) continue;
        in.push_back(li);
        ph.push_back(phase[i]);
    }
    if (!in.empty()) check(be_->cfx_set_tl_phases(dev_, (int32_t) in.size(), in.data(), ph.data()), "cfx_set_tl_phases");
}

void TileEngine::appendVehicles(VehicleSnapshot &out, std::vector<double> *customSpeed) {
    const int cap = (int) scalars().active_vehicle_count + 2 * (int) tn_.ghostLane.size() + 16;
    std::vector<int32_t> vid(cap), drv(cap), prev(cap), lead(cap), blk(cap), ellt(cap), rpos(cap);
    std::vector<double> dis(cap), speed(cap), gap(cap);
    cfx_vehicle_view v{};
    v.capacity = cap;
    v.vid = vid.data();
    v.drivable = drv.data();
    v.prev_drivable = prev.data();
    v.leader_vid = lead.data();
    v.blocker_vid = blk.data();
    v.enter_ll_time = ellt.data();
    v.route_pos = rpos.data();
    v.dis = dis.data();
    v.speed = speed.data();
    v.gap = gap.data();
    check(be_->cfx_get_vehicles(dev_, &v), "cfx_get_vehicles");
    std::vector<double> custom;
    if (customSpeed) {  // same order as cfx_get_vehicles
        custom.resize((size_t) std::max(v.count, 1));
        if (v.count) check(be_->cfx_get_custom_speeds(dev_, v.count, custom.data()), "cfx_get_custom_speeds");
    }
    const int nL = (int) tn_.laneL2G.size(), gL = (int) tn_.laneG2L.size();
    auto toGlobal = [&](int d) {
        if (d <= -2) return gL + (-d - 2);  // a migrant's laneLink of origin (kept as global id)
        if (d < 0) return -1;
        return d < nL ? tn_.laneL2G[d] : gL + tn_.llL2G[d - nL];
    };
    for (int i = 0; i < v.count; ++i) {
        if (drv[i] < nL && tn_.laneGhost[drv[i]]) continue;  // proxy of a neighbour's vehicle
        out.vid.push_back(vid[i]);
        out.drivable.push_back(toGlobal(drv[i]));
        out.prevDrivable.push_back(toGlobal(prev[i]));
        out.leader.push_back(lead[i]);
        out.blocker.push_back(blk[i]);
        out.enterLLTime.push_back(ellt[i]);
        out.routePos.push_back(rpos[i]);
        out.dis.push_back(dis[i]);
        out.speed.push_back(speed[i]);
        out.gap.push_back(gap[i]);
        out.count += 1;
        if (customSpeed) customSpeed->push_back(custom[i]);
    }
}

void TileEngine::appendWaiting(std::vector<int32_t> &vids, std::vector<int32_t> *lanes) {
    cfx_scalars sc = scalars();
    int cap = (int) sc.spawned_vehicle_count + 16;
    std::vector<int32_t> v(cap), l(cap);
    int32_t n = 0;
    check(be_->cfx_get_waiting(dev_, cap, v.data(), l.data(), &n), "cfx_get_waiting");
    for (int i = 0; i < n; ++i)
        if (!tn_.laneGhost[l[i]]) {  // ghost lanes only mirror their owner's queue
            vids.push_back(v[i]);
            if (lanes) lanes->push_back(tn_.laneL2G[l[i]]);
        }
}

void TileEngine::setVehicleRoute(int vid, int route) {
    check(be_->cfx_set_vehicle_route(dev_, vid, route), "cfx_set_vehicle_route");
}

void TileEngine::trafficLights(const std::vector<int> &owner, std::vector<int32_t> &phase, std::vector<double> &remain) {
    const size_t nI = tn_.interL2G.size();
    std::vector<int32_t> ph(std::max<size_t>(nI, 1));
    std::vector<double> rem(std::max<size_t>(nI, 1));
    check(be_->cfx_get_tl_state(dev_, ph.data(), rem.data()), "cfx_get_tl_state");
    for (size_t i = 0; i < nI; ++i) {
        const int g = tn_.interL2G[i];
        if (owner[g] != tn_.rank) continue;
        phase[g] = ph[i];
        remain[g] = rem[i];
    }
}

void TileEngine::loadState(const Archive &a, bool takesTotals) {
    const DeviceState &d = a.dev;
    const int nV = (int) a.host.vehicles.size();
    const int nL = (int) tn_.laneL2G.size(), nK = (int) tn_.llL2G.size(), gL = (int) tn_.laneG2L.size();
    std::vector<int32_t> prio(std::max(nV, 1)), templ(std::max(nV, 1)), route(std::max(nV, 1));
    std::vector<double> enter(std::max(nV, 1));
    std::vector<uint8_t> vstate(std::max(nV, 1), 0);
    for (int v = 0; v < nV; ++v) {
        prio[v] = a.host.vehicles[v].priority;
        templ[v] = a.host.vehicles[v].templ;
        route[v] = a.host.vehicles[v].route;
        enter[v] = a.host.vehicles[v].enterTime;
        if (d.vState[v] == 2 && takesTotals) vstate[v] = 2;  // (the job takes the maximum over the tiles)
    }
    auto localOf = [&](int g) {  // global drivable -> this tile's, or -1
        if (g < 0) return -1;
        if (g < gL) return (int) tn_.laneG2L[g];
        const int k = tn_.llG2L[g - gL];
        return k < 0 ? -1 : nL + k;
    };
    // the archive lists the running vehicles drivable by drivable, front to back: bucket them by local drivable
    std::vector<std::vector<int>> on((size_t) nL + nK);
    for (size_t i = 0; i < d.rVid.size(); ++i) {
        const int ld = localOf(d.rDrivable[i]);
        if (ld < 0) continue;
        if (ld < nL && tn_.laneGhost[ld]) on[ld].assign(1, (int) i);  // a ghost lane keeps its tail only (the proxy)
        else on[ld].push_back((int) i);
    }
    std::vector<int32_t> rVid, rDrv, rPrev, rBlk, rEllt, rPos;
    std::vector<double> rDis, rSpeed, rCustom, rGap;
    const double nan = __builtin_nan("");
    const bool haveGap = d.rGap.size() == d.rVid.size();
    for (int ld = 0; ld < nL + nK; ++ld)
        for (int i : on[ld]) {
            const bool proxy = ld < nL && tn_.laneGhost[ld];
            const int vid = d.rVid[i];
            rVid.push_back(vid);
            rDrv.push_back(ld);
            const int gp = d.rPrevDrivable[i];
            int lp = localOf(gp);
            if (lp < 0 && gp >= gL) lp = -((gp - gL) + 2);  // a laneLink of another tile, kept as its global id (cfx_halo_import)
            rPrev.push_back(lp);
            rBlk.push_back(proxy ? -1 : d.rBlocker[i]);
            rEllt.push_back(proxy ? INT_MAX : d.rEnterLLTime[i]);
            rPos.push_back(proxy ? 0 : d.rRoutePos[i]);
            rDis.push_back(d.rDis[i]);
            rSpeed.push_back(d.rSpeed[i]);
            rCustom.push_back(proxy || d.rCustomSpeed.empty() ? nan : d.rCustomSpeed[i]);
            rGap.push_back(proxy || !haveGap ? nan : d.rGap[i]);  // ControllerInfo::gap is state (cfx_state::r_gap)
            if (!proxy) vstate[vid] = 1;
        }
    std::vector<int32_t> wVid, wLane;
    for (size_t i = 0; i < d.wVid.size(); ++i) {
        const int ll = tn_.laneG2L[d.wLane[i]];
        if (ll < 0) continue;
        wVid.push_back(d.wVid[i]);
        wLane.push_back(ll);
    }
    const size_t nI = tn_.interL2G.size();
    std::vector<int32_t> tlPhase(std::max<size_t>(nI, 1));
    std::vector<double> tlRemain(std::max<size_t>(nI, 1));
    for (size_t i = 0; i < nI; ++i) {
        tlPhase[i] = d.tlPhase[tn_.interL2G[i]];
        tlRemain[i] = d.tlRemain[tn_.interL2G[i]];
    }
    cfx_state st{};
    st.step = d.step;
    st.finished_vehicle_count = takesTotals ? d.finished : 0;
    st.vehicle_steps = takesTotals ? d.vehicleSteps : 0;
    st.cumulative_travel_time = takesTotals ? d.cumulativeTravelTime : 0.0;
    st.n_vehicles = nV;
    st.v_priority = prio.data();
    st.v_templ = templ.data();
    st.v_route = route.data();
    st.v_enter_time = enter.data();
    st.v_state = vstate.data();
    st.n_running = (int) rVid.size();
    st.r_vid = rVid.data();
    st.r_drivable = rDrv.data();
    st.r_prev_drivable = rPrev.data();
    st.r_blocker_vid = rBlk.data();
    st.r_enter_ll_time = rEllt.data();
    st.r_route_pos = rPos.data();
    st.r_dis = rDis.data();
    st.r_speed = rSpeed.data();
    st.r_custom_speed = rCustom.data();
    st.r_gap = rGap.data();
    st.n_waiting = (int) wVid.size();
    st.w_vid = wVid.data();
    st.w_lane = wLane.data();
    st.tl_phase = tlPhase.data();
    st.tl_remain = tlRemain.data();
    check(be_->cfx_load_state(dev_, &st), "cfx_load_state");
}

void TileEngine::setVehicleSpeed(int vid, double speed) {
    // Harmless where the vehicle is not running: a pending custom speed is only consulted at admission, a proxy is frozen.
    check(be_->cfx_set_vehicle_speed(dev_, vid, speed), "cfx_set_vehicle_speed");
}

// ---------------------------------------------------------------- the tiles of this process
TiledEngineHost::TiledEngineHost(const std::string &configFile, int rows, int cols, const std::vector<int> &localTiles,
                                 const std::string &backendLib) {
    cfg_ = readEngineConfig(configFile);
    try {
        net_->load(cfg_.dir + cfg_.roadnetFile);
        spawner_.init(net_.get(), cfg_.interval, 1, cfg_.seed);
        spawner_.loadFlows(cfg_.dir + cfg_.flowFile);
    } catch (const JsonError &e) {
        throw std::runtime_error(std::string("load config failed! ") + e.what());
    }
    // (what it takes — two all-gathers per step for the walk order and the shadows' numbers, a mid-step tail update across the cut,
    //  lane-change state in the migrants' records — is written down in DESIGN.md section 7)
    if (cfg_.laneChange)
        throw std::runtime_error("TiledEngine: laneChange=true is not implemented for tiles (Engine and VectorEngine have it; DESIGN.md section 7 "
                                 "lists what lane change over tiles needs)");
    // Lane::history: EngineHost's default (kept where it is nearly free: the ring layout, networks up to 20 k lanes); the tiles
    // take a step's record behind the step's halo import, when the vehicles that entered a cut lane in the step are on it
    if (cfg_.laneHistory < 0) cfg_.laneHistory = (net_->lanes.size() <= 20000 && cfg_.layout != CFX_LAYOUT_DENSE) ? 1 : 0;
    nTiles_ = rows * cols;
    owner_ = gridPartition(*net_, rows, cols);
    be_.open(backendLib.empty() ? defaultBackendPath() : backendLib);
    localRanks_ = localTiles;
    if (localRanks_.empty()) {
        localRanks_.resize(nTiles_);
        std::iota(localRanks_.begin(), localRanks_.end(), 0);
    }
    allLocal_ = (int) localRanks_.size() == nTiles_;
    cfx_config probe{};
    cfg_.apply(probe);
    const int baseDevice = probe.device;
    for (size_t i = 0; i < localRanks_.size(); ++i) {
        if (localRanks_[i] < 0 || localRanks_[i] >= nTiles_) throw std::runtime_error("tiling: local tile out of range");
        // several local tiles spread over the visible devices (the engine takes device % device count)
        tiles_.emplace_back(new TileEngine(net_, owner_, localRanks_[i], cfg_, &be_, baseDevice + (int) i));
    }
    keepsHistory_ = cfg_.laneHistory > 0;
    for (auto &t : tiles_)
        if (t->layoutName() == "dense") keepsHistory_ = false;  // (dense tiles — a developer's choice — do not keep it)
    aheadEnabled_ = cfg_.spawnAhead && !cfg_.laneChange && !cfg_.saveReplay;
    compactAt_ = nextCompactAt_ = cfg_.compactVehicles < 0 ? (size_t) 3500000 : (size_t) cfg_.compactVehicles;
    compactAuto_ = cfg_.compactVehicles < 0;
    spawner_.setFinishedQuery([this](int vid) {
        // (never from the ahead thread: the devices are the caller's, and over several ranks the answer is a collective)
        if (onAheadThread_.load(std::memory_order_relaxed)) throw AheadAbandoned();
        uint8_t st = 0;
        for (auto &t : tiles_) t->mergeStatus(vid, 1, &st);
        int s = st;
        if (reduceStatus_) s = reduceStatus_(s);
        return s == 2;
    });
    for (auto &t : tiles_) t->uploadTables(spawner_);
    saveReplayInConfig_ = saveReplay_ = cfg_.saveReplay;
    // Engine::setLogFile engine.cpp:773-778.  One tile per process: the process of tile 0 is the writer; every process hands
    // the caller its part of a step (replayPart) and the caller gives the writer all of them (replayWrite).
    replayWriter_ = std::find(localRanks_.begin(), localRanks_.end(), 0) != localRanks_.end();
    if (saveReplay_ && replayWriter_) {
        if (!writeRoadnetLog(*net_, cfg_.dir + cfg_.roadnetLogFile)) std::cerr << "write roadnet log file error" << std::endl;
        replay_.open(cfg_.dir + cfg_.replayLogFile);
    }
}

TiledEngineHost::~TiledEngineHost() {
    if (aheadThread_.joinable()) {
        {
            std::lock_guard<std::mutex> guard(aheadMutex_);
            aheadStop_ = true;
        }
        aheadCv_.notify_all();
        aheadThread_.join();
    }
}

void TiledEngineHost::aheadLoop() {
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
        AheadState result = kAheadReady;
        std::string error;
        onAheadThread_.store(true, std::memory_order_relaxed);
        const auto busy0 = std::chrono::steady_clock::now();
        try {
            spawner_.beginAhead();
            spawner_.step(step, aheadBuf_);
        } catch (const AheadAbandoned &) {
            result = kAheadAbandoned;
        } catch (const std::exception &e) {
            result = kAheadFailed;
            error = e.what();
        }
        onAheadThread_.store(false, std::memory_order_relaxed);
        aheadBusySec_.store(aheadBusySec_.load(std::memory_order_relaxed) +
                                std::chrono::duration<double>(std::chrono::steady_clock::now() - busy0).count(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> guard(aheadMutex_);
            aheadError_ = error;
            aheadState_ = result;
            aheadBusy_.store(false, std::memory_order_release);
        }
        aheadCv_.notify_all();
    }
}

void TiledEngineHost::kickAhead() {
    if (!aheadEnabled_) return;
    if (!aheadThread_.joinable()) aheadThread_ = std::thread([this] { aheadLoop(); });
    {
        std::lock_guard<std::mutex> guard(aheadMutex_);
        aheadStep_ = step_ + 1;
        aheadState_ = kAheadWorking;
        aheadBusy_.store(true, std::memory_order_release);
        aheadKicks_.fetch_add(1, std::memory_order_release);
    }
    aheadCv_.notify_all();
}

void TiledEngineHost::waitAhead() {
    if (!aheadEnabled_) return;
    {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        while (aheadBusy_.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
    }
    std::unique_lock<std::mutex> lock(aheadMutex_);
    aheadCv_.wait(lock, [&] { return aheadState_ != kAheadWorking; });
}

// Whatever was prepared for a step that is not going to be taken as it stands: the spawner goes back to where the last step
// that WAS taken left it.
void TiledEngineHost::dropAhead() {
    if (!aheadEnabled_) return;
    waitAhead();
    if (aheadState_ != kAheadIdle) spawner_.rollbackAhead();
    {
        std::lock_guard<std::mutex> guard(aheadMutex_);  // (the ahead thread reads it under the mutex)
        aheadState_ = kAheadIdle;
    }
}

size_t TiledEngineHost::committedVehicleCount() {
    waitAhead();
    return spawner_.committedVehicleCount();
}

void TiledEngineHost::takeBatch() {
    if (aheadEnabled_) {
        waitAhead();
        if (aheadState_ == kAheadReady && aheadStep_ == step_) {
            spawner_.commitAhead();
            spawnBuf_.swap(aheadBuf_);
            aheadTaken_ += 1;
            {
                std::lock_guard<std::mutex> guard(aheadMutex_);  // (the ahead thread reads it under the mutex)
                aheadState_ = kAheadIdle;
            }
            numbersOut_ = spawner_.vehicles.size();
            return;
        }
        // abandoned (a priority collision), failed (raised below, where it belongs), or prepared for another step
        dropAhead();
    }
    aheadRedone_ += 1;
    spawner_.step(step_, spawnBuf_);
    numbersOut_ = spawner_.vehicles.size();  // (the spawner is at rest here: the ahead thread is started after the batch is taken)
}

void TiledEngineHost::flushPhases() {
    if (pendingInter_.empty()) return;
    for (auto &t : tiles_) t->setPhases(pendingInter_, pendingPhase_);
    pendingInter_.clear();
    pendingPhase_.clear();
}

void TiledEngineHost::stepBegin() {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    flushPhases();
    takeBatch();
    const auto t1 = clk::now();
    hostSpawnSec_ += std::chrono::duration<double>(t1 - t0).count();
    struct Acc {
        double &a;
        clk::time_point t;
        ~Acc() { a += std::chrono::duration<double>(clk::now() - t).count(); }
    } acc{hostSubmitSec_, t1};
    for (auto &t : tiles_) t->uploadTables(spawner_);  // (the spawner is at rest: the ahead thread starts below)
    kickAhead();
    for (auto &t : tiles_) t->step(spawnBuf_);
    if (mailboxes_) {  // device-initiated: export + publish now, the import kernel of stepEnd() does the waiting
        for (auto &t : tiles_) t->haloPost();
        return;
    }
    for (auto &t : tiles_) t->haloExport();
    // messages between two local tiles never leave the process
    for (size_t a = 0; a < tiles_.size(); ++a)
        for (const TilePeer &p : tiles_[a]->tile().peers)
            for (size_t b = 0; b < tiles_.size(); ++b) {
                if (localRanks_[b] != p.rank) continue;
                for (const TilePeer &q : tiles_[b]->tile().peers)
                    if (q.rank == localRanks_[a]) {
                        if (q.recvBytes != p.sendBytes) throw std::logic_error("tiling: halo layouts of two tiles disagree");
                        std::copy(tiles_[a]->send.begin() + p.sendOff, tiles_[a]->send.begin() + p.sendOff + p.sendBytes,
                                  tiles_[b]->recv.begin() + q.recvOff);
                    }
            }
}

void TiledEngineHost::stepBeginDevice() {
    if (mailboxes_) throw std::runtime_error("tiling: step_begin_device() is the staged exchange; mailboxes are enabled");
    flushPhases();
    takeBatch();
    for (auto &t : tiles_) t->uploadTables(spawner_);
    kickAhead();
    for (auto &t : tiles_) t->step(spawnBuf_);
    for (auto &t : tiles_) t->haloExportDevice();
}

void TiledEngineHost::stepEndDevice() {
    for (auto &t : tiles_) t->haloImportDevice();
    if (saveReplay_ && allLocal_) updateLog();
    step_ += 1;
    if (allLocal_ && wantsCompaction()) compactVehicles();
}

std::tuple<uintptr_t, int, uintptr_t, int> TiledEngineHost::haloDeviceBuffers(int i) {
    void *s = nullptr, *r = nullptr;
    tiles_.at((size_t) i)->deviceBuffers(&s, &r);
    return {(uintptr_t) s, (int) tiles_[i]->send.size(), (uintptr_t) r, (int) tiles_[i]->recv.size()};
}

void TiledEngineHost::stepEnd() {
    const auto t0 = std::chrono::steady_clock::now();
    struct Acc {
        double &a;
        std::chrono::steady_clock::time_point t;
        ~Acc() { a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
    } acc{hostSubmitSec_, t0};
    for (auto &t : tiles_) {
        if (mailboxes_) t->haloWait();
        else t->haloImport();
    }
    if (saveReplay_ && allLocal_) updateLog();
    step_ += 1;
    // (several processes: cityflow_amd/tiled.py gathers the parts when wantsCompaction() says so — on every rank alike)
    if (allLocal_ && wantsCompaction()) compactVehicles();
}

void TiledEngineHost::enableMailboxes(const std::string &jobId) {
    if (mailboxes_) return;
    if (step_ != 0) throw std::runtime_error("tiling: enable the mailboxes before the first step");
    std::string prefix = "/cfx_" + jobId;
    for (char &c : prefix)
        if (!(isalnum((unsigned char) c) || c == '_' || c == '/')) c = '_';
    for (auto &t : tiles_) t->attachMailboxes(prefix);
    mailboxes_ = true;
    if (allLocal_) unlinkMailboxes();
}

void TiledEngineHost::unlinkMailboxes() {
    for (auto &t : tiles_) t->unlinkMailboxes();
}

static std::string shmPrefix(const std::string &jobId) {
    std::string prefix = "/cfx_" + jobId;
    for (char &c : prefix)
        if (!(isalnum((unsigned char) c) || c == '_' || c == '/')) c = '_';
    return prefix;
}

bool TiledEngineHost::deviceMailboxPhase(const std::string &jobId, int phase) {
    if (step_ != 0) throw std::runtime_error("tiling: enable the mailboxes before the first step");
    if (mailboxes_) return true;
    const std::string prefix = shmPrefix(jobId);
    bool ok = true;
    for (auto &t : tiles_) ok = (phase == 1 ? t->allocDeviceMailboxes(prefix) : t->attachDeviceMailboxes(prefix)) && ok;
    if (phase == 2 && ok) mailboxes_ = true;
    return ok;
}

bool TiledEngineHost::enableDeviceMailboxes(const std::string &jobId) {
    if (!allLocal_) throw std::runtime_error("tiling: enable_device_mailboxes() needs every tile in this process; use the two phases");
    if (mailboxes_) return true;
    const bool ok = deviceMailboxPhase(jobId, 1) && deviceMailboxPhase(jobId, 2);
    unlinkMailboxes();
    return ok;
}

std::string TiledEngineHost::haloTransport() const {
    if (tiles_.empty() || !mailboxes_) return "staged";
    return std::string(tiles_[0]->mailboxKind()) + " mailboxes";
}

void TiledEngineHost::nextStep() {
    if (!allLocal_) throw std::runtime_error("tiling: next_step() needs every tile in this process; use step_begin / step_end");
    stepBegin();
    stepEnd();
}

std::vector<int32_t> TiledEngineHost::laneVehicleCountArray() {
    std::vector<int32_t> out(net_->lanes.size(), 0);
    for (auto &t : tiles_) t->addLaneCounts(out, false);
    return out;
}

std::vector<int32_t> TiledEngineHost::laneWaitingVehicleCountArray() {
    std::vector<int32_t> out(net_->lanes.size(), 0);
    for (auto &t : tiles_) t->addLaneCounts(out, true);
    return out;
}

std::vector<std::string> TiledEngineHost::laneIds() const {
    std::vector<std::string> ids(net_->lanes.size());
    for (size_t l = 0; l < ids.size(); ++l) ids[l] = net_->laneId((int) l);
    return ids;
}

std::map<std::string, int> TiledEngineHost::getLaneVehicleCount() {
    std::vector<int32_t> cnt = laneVehicleCountArray();
    std::map<std::string, int> ret;
    for (size_t l = 0; l < cnt.size(); ++l) ret.emplace(net_->laneId((int) l), cnt[l]);
    return ret;
}

std::map<std::string, int> TiledEngineHost::getLaneWaitingVehicleCount() {
    std::vector<int32_t> cnt = laneWaitingVehicleCountArray();
    std::map<std::string, int> ret;
    for (size_t l = 0; l < cnt.size(); ++l) ret.emplace(net_->laneId((int) l), cnt[l]);
    return ret;
}

cfx_scalars TiledEngineHost::scalars() {
    cfx_scalars sum{};
    for (int &d : sum.tie_drivables) d = -1;  // (tile-local indices: not reported)
    for (auto &t : tiles_) {
        cfx_scalars s = t->scalars();
        sum.active_vehicle_count += s.active_vehicle_count;
        sum.finished_vehicle_count += s.finished_vehicle_count;
        sum.cumulative_travel_time += s.cumulative_travel_time;
        sum.vehicle_steps += s.vehicle_steps;
        sum.tie_events += s.tie_events;
    }
    sum.step = (int64_t) step_;
    sum.spawned_vehicle_count = (int64_t) committedVehicleCount();
    return sum;
}

size_t TiledEngineHost::getVehicleCount() { return (size_t) scalars().active_vehicle_count; }

void TiledEngineHost::setTrafficLightPhase(const std::string &id, int phaseIndex) {
    if (!cfg_.rlTrafficLight) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    auto it = net_->interIndex.find(id);
    if (it == net_->interIndex.end()) throw std::runtime_error("Intersection '" + id + "' not found");
    const HostInter &in = net_->inters[it->second];
    if (in.isVirtual || phaseIndex < 0 || phaseIndex >= (int) in.phases.size())
        throw std::out_of_range("phase index " + std::to_string(phaseIndex) + " out of range for intersection '" + id + "'");
    pendingInter_.push_back(it->second);
    pendingPhase_.push_back(phaseIndex);
}

void TiledEngineHost::setTrafficLightPhases(const std::vector<int32_t> &phases) {
    if (!cfg_.rlTrafficLight) {
        std::cerr << "please set rlTrafficLight to true to enable traffic light control" << std::endl;
        return;
    }
    if (phases.size() != net_->inters.size()) throw std::runtime_error("set_tl_phases: expected one phase per intersection");
    for (size_t i = 0; i < phases.size(); ++i) {
        const HostInter &in = net_->inters[i];
        if (in.isVirtual) continue;
        if (phases[i] < 0 || phases[i] >= (int) in.phases.size())
            throw std::out_of_range("set_tl_phases: phase out of range for intersection '" + in.id + "'");
        pendingInter_.push_back((int32_t) i);
        pendingPhase_.push_back(phases[i]);
    }
}

void TiledEngineHost::reset(bool resetRnd) {
    dropAhead();
    pendingInter_.clear();
    pendingPhase_.clear();
    for (auto &t : tiles_) t->reset();
    spawner_.reset(resetRnd);
    step_ = 0;
    hostSpawnSec_ = hostSubmitSec_ = 0;
    waitingCustom_.clear();
    nextCompactAt_ = compactAt_;
    numbersOut_ = 0;
}

void TiledEngineHost::sync() {
    for (auto &t : tiles_) t->sync();
}

void TiledEngineHost::snapshotVehicles(VehicleSnapshot &out) {
    dropAhead();
    VehicleSnapshot all;
    for (auto &t : tiles_) t->appendVehicles(all);
    // every drivable belongs to exactly one tile and a tile lists it front to back: a stable sort by global
    // drivable gives Drivable::vehicles order over the whole network
    std::vector<int> idx((size_t) all.count);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return all.drivable[a] < all.drivable[b]; });
    out = VehicleSnapshot{};
    out.count = all.count;
    for (int i : idx) {
        out.vid.push_back(all.vid[i]);
        out.drivable.push_back(all.drivable[i]);
        out.prevDrivable.push_back(all.prevDrivable[i]);
        out.leader.push_back(all.leader[i]);
        out.blocker.push_back(all.blocker[i]);
        out.enterLLTime.push_back(all.enterLLTime[i]);
        out.routePos.push_back(all.routePos[i]);
        out.dis.push_back(all.dis[i]);
        out.speed.push_back(all.speed[i]);
        out.gap.push_back(all.gap[i]);
    }
}

}  // namespace cfa

namespace cfa {

int TiledEngineHost::statusOf(int vid) {
    dropAhead();
    uint8_t st = 0;
    for (auto &t : tiles_) t->mergeStatus(vid, 1, &st);
    int s = st;
    if (reduceStatus_) s = reduceStatus_(s);
    return s;
}

// getVehicles engine.cpp:619-626 — vehiclePool (priority) order
std::vector<std::pair<int32_t, std::string>> TiledEngineHost::vehiclesKeyed(bool includeWaiting) {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::vector<std::pair<int32_t, int32_t>> byPriority;
    for (int i = 0; i < s.count; ++i) byPriority.emplace_back(spawner_.vehicles[s.vid[i]].priority, s.vid[i]);
    if (includeWaiting) {
        std::vector<int32_t> wv;
        for (auto &t : tiles_) t->appendWaiting(wv);
        for (int32_t v : wv) byPriority.emplace_back(spawner_.vehicles[v].priority, v);
    }
    std::sort(byPriority.begin(), byPriority.end());
    std::vector<std::pair<int32_t, std::string>> ret;
    for (auto &p : byPriority) ret.emplace_back(p.first, spawner_.vehicleId(p.second));
    return ret;
}

// Vehicles pushed (push_vehicle) since the last step: in the reference's vehiclePool from the moment of the call
// (EngineHost::isPendingPushed, engine_host.cpp); every rank's spawner holds the same ones.
std::vector<std::pair<int32_t, std::string>> TiledEngineHost::pendingPushedKeyed() const {
    const_cast<TiledEngineHost *>(this)->dropAhead();
    std::vector<std::pair<int32_t, std::string>> pushed;
    spawner_.pendingPushed(pushed);
    return pushed;
}

bool TiledEngineHost::isPendingPushed(const std::string &id) const {
    const_cast<TiledEngineHost *>(this)->dropAhead();
    for (const auto &p : pendingPushedKeyed())
        if (p.second == id) return true;
    return false;
}

std::vector<std::string> TiledEngineHost::getVehicles(bool includeWaiting) {
    dropAhead();
    std::vector<std::pair<int32_t, std::string>> keyed = vehiclesKeyed(includeWaiting);
    if (includeWaiting) {
        for (auto &p : pendingPushedKeyed()) keyed.push_back(std::move(p));
        std::sort(keyed.begin(), keyed.end());  // (priorities are unique)
    }
    std::vector<std::string> ret;
    for (auto &p : keyed) ret.push_back(std::move(p.second));
    return ret;
}

bool TiledEngineHost::runsHere(const std::string &vehicleId) {
    dropAhead();
    const int vid = spawner_.vidOfId(vehicleId);
    if (vid < 0) return false;
    VehicleSnapshot s;
    snapshotVehicles(s);
    for (int i = 0; i < s.count; ++i)
        if (s.vid[i] == vid) return true;
    return false;
}

std::vector<uint8_t> TiledEngineHost::localStatus() {
    dropAhead();
    const int total = (int) spawner_.vehicles.size();
    std::vector<uint8_t> st((size_t) total, 0);
    if (total)
        for (auto &t : tiles_) t->mergeStatus(0, total, st.data());
    return st;
}

std::map<std::string, std::vector<std::string>> TiledEngineHost::getLaneVehicles() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    const int L = (int) net_->lanes.size();
    std::vector<std::vector<std::string>> perLane(L);
    for (int i = 0; i < s.count; ++i)
        if (s.drivable[i] < L) perLane[s.drivable[i]].push_back(spawner_.vehicleId(s.vid[i]));
    std::map<std::string, std::vector<std::string>> ret;
    for (int l = 0; l < L; ++l) ret.emplace(net_->laneId(l), std::move(perLane[l]));
    return ret;
}

std::map<std::string, double> TiledEngineHost::getVehicleSpeed() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::map<std::string, double> ret;
    for (int i = 0; i < s.count; ++i) ret.emplace(spawner_.vehicleId(s.vid[i]), s.speed[i]);
    return ret;
}

std::map<std::string, double> TiledEngineHost::getVehicleDistance() {
    dropAhead();
    VehicleSnapshot s;
    snapshotVehicles(s);
    std::map<std::string, double> ret;
    for (int i = 0; i < s.count; ++i) ret.emplace(spawner_.vehicleId(s.vid[i]), s.dis[i]);
    return ret;
}

std::string TiledEngineHost::getLeader(const std::string &vehicleId) {
    dropAhead();
    int vid = spawner_.vidOfId(vehicleId);
    if (vid < 0 && isPendingPushed(vehicleId)) return "";
    int st = vid >= 0 ? statusOf(vid) : 2;
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + vehicleId + "' not found");
    if (st == 0) return "";
    VehicleSnapshot s;
    snapshotVehicles(s);
    for (int i = 0; i < s.count; ++i)
        if (s.vid[i] == vid) return s.leader[i] >= 0 ? spawner_.vehicleId(s.leader[i]) : "";
    return "";
}

std::map<std::string, std::string> TiledEngineHost::getVehicleInfo(const std::string &vehicleId) {
    dropAhead();
    int vid = spawner_.vidOfId(vehicleId);
    if (vid < 0 && isPendingPushed(vehicleId)) return {{"running", "0"}};
    int st = vid >= 0 ? statusOf(vid) : 2;
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + vehicleId + "' not found");
    std::map<std::string, std::string> info;
    info["running"] = std::to_string(st == 1);
    if (st != 1) return info;
    VehicleSnapshot s;
    snapshotVehicles(s);
    for (int i = 0; i < s.count; ++i) {
        if (s.vid[i] != vid) continue;
        info["distance"] = std::to_string(s.dis[i]);
        info["speed"] = std::to_string(s.speed[i]);
        info["drivable"] = net_->drivableId(s.drivable[i]);
        if (s.drivable[i] < (int) net_->lanes.size()) {
            const HostRoad &road = net_->roads[net_->lanes[s.drivable[i]].road];
            info["road"] = road.id;
            info["intersection"] = net_->inters[road.endInter].id;
        }
        const RouteTable &rt = spawner_.routes;
        int r = spawner_.vehicles[vid].route;
        std::string route;
        for (int p = rt.routeStart[r] + s.routePos[i]; p < rt.routeStart[r + 1]; ++p) route += net_->roads[rt.roads[p]].id + " ";
        info["route"] = route;
    }
    return info;
}

// getAverageTravelTime engine.cpp:682-691 (finished part summed per tile)
double TiledEngineHost::getAverageTravelTime() {
    dropAhead();
    const cfx_scalars sc = scalars();
    return averageTravelTimeFrom(sc.cumulative_travel_time, sc.finished_vehicle_count, localStatus());
}

double TiledEngineHost::averageTravelTimeFrom(double tt, int64_t n, const std::vector<uint8_t> &st) const {
    const int total = (int) spawner_.vehicles.size();
    if ((int) st.size() != total) throw std::runtime_error("tiling: vehicle status of a different vehicle table");
    std::vector<std::pair<int32_t, double>> live;
    for (int v = 0; v < total; ++v)
        if (st[v] != 2) live.emplace_back(spawner_.vehicles[v].priority, spawner_.vehicles[v].enterTime);
    std::sort(live.begin(), live.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) {
        return a.first < b.first;
    });
    const double now = getCurrentTime();
    for (auto &p : live) {
        tt += now - p.second;
        n++;
    }
    n += (int64_t) pendingPushedKeyed().size();  // (entered now: + 0.0 each, EngineHost::getAverageTravelTime)
    return n == 0 ? 0 : tt / n;
}

void TiledEngineHost::pushVehicle(const std::map<std::string, double> &info, const std::vector<std::string> &roads) {
    dropAhead();
    auto get = [&info](const char *k, double d) {
        auto it = info.find(k);
        return it == info.end() ? d : it->second;
    };
    cfx_vehicle_template t = spawner_.makeTemplate(get("length", 5), get("width", 2), get("maxPosAcc", 4.5), get("maxNegAcc", 4.5),
                                                   get("usualPosAcc", 2.5), get("usualNegAcc", 2.5), get("minGap", 2),
                                                   get("maxSpeed", 16.66667), get("headwayTime", 1), get("speed", 0));
    std::vector<int> anchors;
    for (auto &r : roads) {
        auto it = net_->roadIndex.find(r);
        if (it == net_->roadIndex.end()) throw std::runtime_error("Road '" + r + "' not found");
        anchors.push_back(it->second);
    }
    if (anchors.empty()) throw std::runtime_error("push_vehicle: empty route");
    spawner_.pushManual(spawner_.addTemplate(t), anchors, step_);
}

void TiledEngineHost::setVehicleSpeed(const std::string &id, double speed) {
    dropAhead();
    int vid = spawner_.vidOfId(id);
    if (vid < 0) {  // pushed since the last step (EngineHost::setVehicleSpeed): every tile keeps the speed for the number to come
        const int future = spawner_.pendingPushedVid(id);
        if (future == -2) return;
        if (future >= 0) {
            for (auto &t : tiles_) t->setVehicleSpeed(future, speed);
            waitingCustom_[future] = speed;
            return;
        }
    }
    int st = vid >= 0 ? statusOf(vid) : 2;
    if (vid < 0 || st == 2) throw std::runtime_error("Vehicle '" + id + "' not found");
    for (auto &t : tiles_) t->setVehicleSpeed(vid, speed);
    if (st == 0) waitingCustom_[vid] = speed;  // (still in its lane's waiting buffer: compactFromParts hands it over)
}

}  // namespace cfa

// ---------------------------------------------------------------- archive, routes, replay
namespace cfa {

namespace {
// a snapshot PART on the wire: counted runs of plain values, in the order snapshotPart() writes them
struct PartWriter {
    std::string out;
    template <typename T> void scalar(T v) { out.append((const char *) &v, sizeof v); }
    template <typename T> void vec(const std::vector<T> &v) {
        scalar<uint64_t>(v.size());
        if (!v.empty()) out.append((const char *) v.data(), v.size() * sizeof(T));
    }
};
struct PartReader {
    const std::string &in;
    size_t at = 0;
    explicit PartReader(const std::string &s) : in(s) {}
    template <typename T> T scalar() {
        if (at + sizeof(T) > in.size()) throw std::runtime_error("tiling: truncated snapshot part");
        T v;
        memcpy(&v, in.data() + at, sizeof v);
        at += sizeof v;
        return v;
    }
    template <typename T> std::vector<T> vec() {
        const uint64_t n = scalar<uint64_t>();
        if (at + n * sizeof(T) > in.size()) throw std::runtime_error("tiling: truncated snapshot part");
        std::vector<T> v((size_t) n);
        if (n) memcpy(v.data(), in.data() + at, (size_t) n * sizeof(T));
        at += (size_t) n * sizeof(T);
        return v;
    }
};
}  // namespace

std::string TiledEngineHost::snapshotPart() {
    dropAhead();
    flushPhases();
    PartWriter w;
    const cfx_scalars sc = scalars();  // sums over the local tiles
    w.scalar<int64_t>((int64_t) step_);
    w.scalar<int64_t>(sc.finished_vehicle_count);
    w.scalar<int64_t>(sc.vehicle_steps);
    w.scalar<double>(sc.cumulative_travel_time);
    const int nV = (int) spawner_.vehicles.size();
    std::vector<uint8_t> vState((size_t) nV, 0);
    if (nV)
        for (auto &t : tiles_) t->mergeStatus(0, nV, vState.data());
    w.vec(vState);
    VehicleSnapshot all;
    std::vector<double> custom;
    for (auto &t : tiles_) t->appendVehicles(all, &custom);
    w.vec(all.vid);
    w.vec(all.drivable);
    w.vec(all.prevDrivable);
    w.vec(all.leader);
    w.vec(all.blocker);
    w.vec(all.enterLLTime);
    w.vec(all.routePos);
    w.vec(all.dis);
    w.vec(all.speed);
    w.vec(all.gap);
    w.vec(custom);
    std::vector<int32_t> wVid, wLane;
    for (auto &t : tiles_) t->appendWaiting(wVid, &wLane);
    w.vec(wVid);
    w.vec(wLane);
    const size_t nI = net_->inters.size();
    std::vector<int32_t> phase(nI, -1);  // -1: not one of this process's intersections
    std::vector<double> remain(nI, 0.0);
    for (auto &t : tiles_) t->trafficLights(owner_, phase, remain);
    w.vec(phase);
    w.vec(remain);
    // Lane::history of the lanes this process's tiles own: {global lane, records held, the two aggregates}, then the records
    std::vector<int32_t> hLane, hLen, hHistNum, hNum;
    std::vector<double> hHistAvg, hAvg;
    if (keepsHistory_) {
        const size_t nL = net_->lanes.size(), M = CFX_LANE_HISTORY_MAX;
        DeviceState d;
        d.hLen.assign(nL, -1);
        d.hVehicleNum.assign(nL * M, 0);
        d.hAverageSpeed.assign(nL * M, 0.0);
        d.hHistoryVehicleNum.assign(nL, 0);
        d.hHistoryAverageSpeed.assign(nL, 0.0);
        for (auto &t : tiles_) t->laneHistoryInto(d);
        for (size_t g = 0; g < nL; ++g) {
            if (d.hLen[g] < 0) continue;  // (another process's lane)
            hLane.push_back((int32_t) g);
            hLen.push_back(d.hLen[g]);
            hHistNum.push_back(d.hHistoryVehicleNum[g]);
            hHistAvg.push_back(d.hHistoryAverageSpeed[g]);
            hNum.insert(hNum.end(), d.hVehicleNum.begin() + (ptrdiff_t) (g * M), d.hVehicleNum.begin() + (ptrdiff_t) (g * M + (size_t) d.hLen[g]));
            hAvg.insert(hAvg.end(), d.hAverageSpeed.begin() + (ptrdiff_t) (g * M), d.hAverageSpeed.begin() + (ptrdiff_t) (g * M + (size_t) d.hLen[g]));
        }
    }
    w.vec(hLane);
    w.vec(hLen);
    w.vec(hHistNum);
    w.vec(hHistAvg);
    w.vec(hNum);
    w.vec(hAvg);
    return std::move(w.out);
}

Archive TiledEngineHost::snapshotFromParts(const std::vector<std::string> &parts, bool keepRoutePositions) {
    dropAhead();
    Archive a;
    a.host = spawner_.saveState();
    a.net = net_;
    a.templates = spawner_.templates;
    a.routeStart = spawner_.routes.routeStart;
    a.routeRoads = spawner_.routes.roads;
    for (const HostFlow &f : spawner_.flows) a.flowIds.push_back(f.id);
    DeviceState &d = a.dev;
    d.step = (int64_t) step_;
    const int nV = (int) spawner_.vehicles.size();
    d.vState.assign((size_t) nV, 0);
    const size_t nI = net_->inters.size();
    d.tlPhase.assign(nI, 0);
    d.tlRemain.assign(nI, 0.0);
    VehicleSnapshot all;
    std::vector<double> custom;
    std::vector<int32_t> wVid, wLane;
    for (const std::string &blob : parts) {
        PartReader r(blob);
        if (r.scalar<int64_t>() != d.step) throw std::runtime_error("tiling: snapshot parts of different steps");
        d.finished += r.scalar<int64_t>();
        d.vehicleSteps += r.scalar<int64_t>();
        d.cumulativeTravelTime += r.scalar<double>();
        const std::vector<uint8_t> vs = r.vec<uint8_t>();
        if ((int) vs.size() != nV) throw std::runtime_error("tiling: snapshot part of a different vehicle table");
        for (int v = 0; v < nV; ++v) d.vState[v] = std::max(d.vState[v], vs[v]);
        auto app = [](auto &dst, const auto &src) { dst.insert(dst.end(), src.begin(), src.end()); };
        app(all.vid, r.vec<int32_t>());
        app(all.drivable, r.vec<int32_t>());
        app(all.prevDrivable, r.vec<int32_t>());
        app(all.leader, r.vec<int32_t>());
        app(all.blocker, r.vec<int32_t>());
        app(all.enterLLTime, r.vec<int32_t>());
        app(all.routePos, r.vec<int32_t>());
        app(all.dis, r.vec<double>());
        app(all.speed, r.vec<double>());
        app(all.gap, r.vec<double>());
        app(custom, r.vec<double>());
        app(wVid, r.vec<int32_t>());
        app(wLane, r.vec<int32_t>());
        const std::vector<int32_t> ph = r.vec<int32_t>();
        const std::vector<double> rem = r.vec<double>();
        if (ph.size() != nI || rem.size() != nI) throw std::runtime_error("tiling: snapshot part of a different road network");
        for (size_t i = 0; i < nI; ++i)
            if (ph[i] >= 0) {
                d.tlPhase[i] = ph[i];
                d.tlRemain[i] = rem[i];
            }
        const std::vector<int32_t> hLane = r.vec<int32_t>(), hLen = r.vec<int32_t>(), hHistNum = r.vec<int32_t>();
        const std::vector<double> hHistAvg = r.vec<double>();
        const std::vector<int32_t> hNum = r.vec<int32_t>();
        const std::vector<double> hAvg = r.vec<double>();
        if (!hLane.empty()) {
            const size_t nL = net_->lanes.size(), M = CFX_LANE_HISTORY_MAX;
            if (d.hLen.empty()) {
                d.hLen.assign(nL, 0);
                d.hVehicleNum.assign(nL * M, 0);
                d.hAverageSpeed.assign(nL * M, 0.0);
                d.hHistoryVehicleNum.assign(nL, 0);
                d.hHistoryAverageSpeed.assign(nL, 0.0);
            }
            size_t at = 0;
            for (size_t i = 0; i < hLane.size(); ++i) {
                const size_t g = (size_t) hLane[i], n = (size_t) hLen[i];
                if (g >= nL || n > M || at + n > hNum.size()) throw std::runtime_error("tiling: snapshot part with a broken lane history");
                d.hLen[g] = hLen[i];
                d.hHistoryVehicleNum[g] = hHistNum[i];
                d.hHistoryAverageSpeed[g] = hHistAvg[i];
                std::copy(hNum.begin() + (ptrdiff_t) at, hNum.begin() + (ptrdiff_t) (at + n), d.hVehicleNum.begin() + (ptrdiff_t) (g * M));
                std::copy(hAvg.begin() + (ptrdiff_t) at, hAvg.begin() + (ptrdiff_t) (at + n), d.hAverageSpeed.begin() + (ptrdiff_t) (g * M));
                at += n;
            }
        }
    }
    // every drivable belongs to exactly one tile, which lists it front to back: a stable sort by global drivable gives
    // Drivable::vehicles order over the whole network (the same for the waiting buffers, lane by lane)
    std::vector<int> idx(all.vid.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return all.drivable[x] < all.drivable[y]; });
    for (int i : idx) {
        d.rVid.push_back(all.vid[i]);
        d.rDrivable.push_back(all.drivable[i]);
        d.rPrevDrivable.push_back(all.prevDrivable[i]);
        d.rLeader.push_back(all.leader[i]);
        d.rBlocker.push_back(all.blocker[i]);
        d.rEnterLLTime.push_back(all.enterLLTime[i]);
        d.rRoutePos.push_back(keepRoutePositions ? all.routePos[i] : 0);  // (Router copy constructor: see EngineHost::snapshot, archive.cpp; compactFromParts is not a load)
        d.rDis.push_back(all.dis[i]);
        d.rSpeed.push_back(all.speed[i]);
        d.rGap.push_back(all.gap[i]);
        d.rCustomSpeed.push_back(custom[i]);
    }
    idx.resize(wVid.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return wLane[x] < wLane[y]; });
    for (int i : idx) {
        d.wVid.push_back(wVid[i]);
        d.wLane.push_back(wLane[i]);
    }
    return a;
}

Archive TiledEngineHost::snapshot() {
    dropAhead();
    if (!allLocal_) throw std::runtime_error("tiling: snapshot() needs every tile in this process; gather snapshot_part() of every process");
    return snapshotFromParts({snapshotPart()});
}

// Forget the finished vehicles: EngineHost::compactVehicles (archive.cpp) over tiles.  The state as the parts hold it, the
// vehicles that still wait or run renumbered 0 .. n-1 in their old order (creation order: what breaks exact-distance ties),
// every process loading the whole and keeping its tiles' part — nothing a caller can see changes.
void TiledEngineHost::compactFromParts(const std::vector<std::string> &parts) {
    Archive a = snapshotFromParts(parts, /*keepRoutePositions=*/true);
    const int nV = (int) spawner_.vehicles.size();
    DeviceState &d = a.dev;
    std::vector<int32_t> newOfOld((size_t) nV, -1);
    int nLive = 0;
    for (int v = 0; v < nV; ++v)
        if (d.vState[(size_t) v] != 2) newOfOld[(size_t) v] = nLive++;
    auto renumber = [&](std::vector<int32_t> &vids) {
        for (int32_t &v : vids) v = v >= 0 && v < nV ? newOfOld[(size_t) v] : -1;
    };
    renumber(d.rVid);
    renumber(d.rBlocker);
    renumber(d.rLeader);
    renumber(d.wVid);
    for (int32_t v : d.rVid)
        if (v < 0) throw std::logic_error("compact_vehicles: a running vehicle counted as finished");
    for (int32_t v : d.wVid)
        if (v < 0) throw std::logic_error("compact_vehicles: a waiting vehicle counted as finished");
    std::vector<uint8_t> state((size_t) nLive);
    for (int v = 0; v < nV; ++v)
        if (newOfOld[(size_t) v] >= 0) state[(size_t) newOfOld[(size_t) v]] = d.vState[(size_t) v];
    d.vState.swap(state);
    a.host = spawner_.compactedState(newOfOld, nLive);
    // custom speeds of vehicles that are STILL waiting (cfx_state carries those of running vehicles only)
    std::map<int32_t, double> stillWaiting;
    for (const auto &kv : waitingCustom_)
        if (kv.first >= 0 && kv.first < nV && newOfOld[(size_t) kv.first] >= 0 && d.vState[(size_t) newOfOld[(size_t) kv.first]] == 0)
            stillWaiting[newOfOld[(size_t) kv.first]] = kv.second;
    load(a);
    for (const auto &kv : stillWaiting)
        for (auto &t : tiles_) t->setVehicleSpeed(kv.first, kv.second);
    waitingCustom_.swap(stillWaiting);
    vehicleCompactions_ += 1;
    const size_t alive = spawner_.vehicles.size();
    nextCompactAt_ = compactAuto_ ? std::max(compactAt_, 32 * alive) : alive + compactAt_;
}

void TiledEngineHost::compactVehicles() {
    if (!allLocal_) throw std::runtime_error("tiling: compact_vehicles() needs every tile in this process; gather _snapshot_part() of every process for _compact_from_parts()");
    compactFromParts({snapshotPart()});
}

void TiledEngineHost::load(const Archive &a) {
    dropAhead();
    waitingCustom_.clear();  // (compactFromParts puts back what it carries over)
    if (!a.dev.rLcFlags.empty()) throw std::runtime_error("TiledEngine.load: the archive carries lane-change state");
    if (a.net.get() != net_.get() && a.net->lanes.size() != net_->lanes.size())
        throw std::runtime_error("TiledEngine.load: archive belongs to a different road network");
    pendingInter_.clear();
    pendingPhase_.clear();
    spawner_.loadState(a.host);
    for (auto &t : tiles_) {
        t->uploadTables(spawner_);
        t->loadState(a, t->tile().rank == 0);
        // Archive::resume archive.cpp:107-109 (an archive without it: the lanes keep the history they have)
        if (keepsHistory_ && a.dev.hLen.size() == net_->lanes.size()) t->setLaneHistory(a.dev);
    }
    numbersOut_ = spawner_.vehicles.size();
    step_ = (size_t) a.dev.step;
}

void TiledEngineHost::loadFromFile(const std::string &path) {
    dropAhead();
    load(readArchiveFile(path, net_, spawner_, false));
}

// Engine::setRoute engine.cpp:852-866 + Router::setRoute router.cpp:245-264 (EngineHost::setRoute)
bool TiledEngineHost::setRoute(const std::string &vehicleId, const std::vector<std::string> &anchorIds) {
    dropAhead();
    const int vid = spawner_.vidOfId(vehicleId);
    if (vid < 0) return false;
    const int state = statusOf(vid);
    if (state == 2) return false;
    std::vector<int> anchors;
    for (const auto &id : anchorIds) {
        auto it = net_->roadIndex.find(id);
        if (it == net_->roadIndex.end()) return false;
        anchors.push_back(it->second);
    }
    const int L = (int) net_->lanes.size();
    int drivable = -1, routePos = -1;
    if (state == 0) {  // still in a waiting buffer: its drivable is its first lane, iCurRoad = begin
        drivable = spawner_.vehicles[vid].firstLane;
        routePos = 0;
    } else {  // where it runs: known to the process that has its tile
        VehicleSnapshot s;
        snapshotVehicles(s);
        for (int i = 0; i < s.count; ++i)
            if (s.vid[i] == vid) {
                drivable = s.drivable[i];
                routePos = s.routePos[i];
            }
        if (reduceStatus_) {
            drivable = reduceStatus_(drivable + 1) - 1;
            routePos = reduceStatus_(routePos + 1) - 1;
        }
        if (drivable < 0) return false;
    }
    if (drivable >= L) return false;  // on a laneLink (router.cpp:246)
    // (the start road of the new route: EngineHost::setRoute, engine_host.cpp — the cursor's road where the path from there
    // leads through the vehicle's lane's road, that road itself otherwise)
    const int laneRoad = net_->lanes[drivable].road;
    const RouteTable &rt = spawner_.routes;
    const int cursorRoad = rt.roads[rt.routeStart[spawner_.vehicles[vid].route] + routePos];
    std::vector<int> seq;
    auto plan = [&](int from) {
        std::vector<int> newAnchors{from};
        newAnchors.insert(newAnchors.end(), anchors.begin(), anchors.end());
        return spawner_.expandRoute(newAnchors, seq);
    };
    auto positionOf = [&](int road) {
        for (size_t i = 0; i < seq.size(); ++i)
            if (seq[i] == road) return (int) i;
        return -1;
    };
    if (!plan(cursorRoad)) return false;
    int pos = positionOf(laneRoad);
    if (pos < 0) {
        if (!plan(laneRoad)) return false;
        pos = 0;
    }
    const int newRoute = spawner_.internRoute(seq);
    // Router::onValidLane (router.h:66-68) under the new route: a next drivable exists or this is the last road
    const RouteTable &rt2 = spawner_.routes;
    const int laneIdx = net_->lanes[drivable].index;
    const bool hasNext = rt2.nextLL[rt2.nextStart[rt2.routeStart[newRoute] + pos] + laneIdx] >= 0;
    const bool lastRoad = seq.back() == laneRoad;
    if (!hasNext && !lastRoad) return false;
    for (auto &t : tiles_) {
        t->uploadTables(spawner_);
        t->setVehicleRoute(vid, newRoute);
    }
    spawner_.setVehicleRoute(vid, newRoute);
    return true;
}

// Engine::updateLog (engine.cpp:518-554): the state after the step, lights after TrafficLight::passTime
void TiledEngineHost::updateLog() { replayWrite({replayPart()}); }

// what the replay line of the step that has just finished needs from this process: its vehicles' {number, drivable,
// distance} and the phases of its intersections
std::string TiledEngineHost::replayPart() {
    dropAhead();
    PartWriter w;
    VehicleSnapshot s;
    for (auto &t : tiles_) t->appendVehicles(s);
    w.vec(s.vid);
    w.vec(s.drivable);
    w.vec(s.dis);
    std::vector<int32_t> phase(net_->inters.size(), -1);
    std::vector<double> remain(net_->inters.size(), 0.0);
    for (auto &t : tiles_) t->trafficLights(owner_, phase, remain);
    w.vec(phase);
    return std::move(w.out);
}

void TiledEngineHost::replayWrite(const std::vector<std::string> &parts) {
    dropAhead();
    if (!replayWriter_) return;
    VehicleSnapshot s;
    std::vector<int32_t> phase(net_->inters.size(), 0);
    for (const std::string &blob : parts) {
        PartReader r(blob);
        auto app = [](auto &dst, const auto &src) { dst.insert(dst.end(), src.begin(), src.end()); };
        app(s.vid, r.vec<int32_t>());
        app(s.drivable, r.vec<int32_t>());
        app(s.dis, r.vec<double>());
        const std::vector<int32_t> ph = r.vec<int32_t>();
        for (size_t i = 0; i < ph.size() && i < phase.size(); ++i)
            if (ph[i] >= 0) phase[i] = ph[i];
    }
    s.count = (int) s.vid.size();
    replay_.writeStep(*net_, spawner_, s, phase);  // (orders the vehicles itself: vehiclePool order)
}

void TiledEngineHost::setReplayLogFile(const std::string &logFile) {
    dropAhead();
    if (!saveReplayInConfig_) {
        std::cerr << "saveReplay is not set to true in config file!" << std::endl;
        return;
    }
    if (replayWriter_) replay_.open(cfg_.dir + logFile);
}

void TiledEngineHost::setSaveReplay(bool open) {
    dropAhead();
    if (!saveReplayInConfig_) {
        std::cerr << "saveReplay is not set to true in config file!" << std::endl;
        return;
    }
    saveReplay_ = open;
}

}  // namespace cfa
