/*
 * cityflow_amd.h — C ABI of the MI355X-native CityFlow step engine ("cfx").
 *
 * This is the drop-in boundary between the C++ host (JSON loading, flow spawning / RNG, string ids,
 * the pybind11 `cityflow.Engine` class) and the device engine that owns all vehicle state in HBM and
 * advances it with hand-written HIP kernels.  The reference has no internal FFI seam (its only
 * boundary is the pybind11 class, reference src/cityflow.cpp:10-47); the entry points below are what a
 * maintainer of the reference would bind from `CityFlow::Engine` to off-load `Engine::nextStep`
 * (reference src/engine/engine.cpp:566-594) and the RL getters around it (engine.cpp:615-760).
 * INTEGRATION.md shows that binding.
 *
 * Conventions: opaque handle; plain pointers and sizes; caller-allocated outputs; every function
 * returns 0 on success or a negative cfx_status, with a message available from cfx_last_error();
 * no exceptions cross the ABI; one HIP stream per engine; getters are synchronous on return,
 * cfx_step() is asynchronous (ordered on the engine's stream).  Strings never cross this ABI:
 * lanes, laneLinks, intersections, roads, routes, templates and vehicles are dense int32 indices.
 *
 * Index spaces
 *   lane      0..n_lanes-1        reference order RoadNet::getLanes()      (roadnet.cpp:314-318)
 *   laneLink  0..n_lanelinks-1    reference order RoadNet::getLaneLinks()  (roadnet.cpp:319-323)
 *   drivable  lane l -> l ; laneLink k -> n_lanes + k   (== RoadNet::getDrivables())
 *   vid       host-assigned vehicle number, dense, monotone since the last reset
 */
#ifndef CITYFLOW_AMD_H
#define CITYFLOW_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFX_ABI_VERSION 9

typedef enum cfx_status {
    CFX_OK = 0,
    CFX_ERR_INVALID = -1,   /* bad argument / index out of range                    */
    CFX_ERR_DEVICE = -2,    /* HIP runtime failure (no device, launch error, OOM)   */
    CFX_ERR_CAPACITY = -3,  /* an internal fixed-capacity table overflowed          */
    CFX_ERR_STATE = -4      /* call not valid in the current state                  */
} cfx_status;

typedef struct cfx_engine cfx_engine; /* opaque */

/* Flattened, immutable road network (replaces RoadNet / Road / Lane / LaneLink / RoadLink / Cross /
 * Intersection / LightPhase, reference src/roadnet/roadnet.h:64-481, trafficlight.h:15-50).
 * All arrays are copied by cfx_create(); the caller may free them afterwards. */
typedef struct cfx_net {
    int32_t n_roads, n_lanes, n_lanelinks, n_inters;
    int32_t n_xentries;  /* 2 * number of Cross objects: one entry per (laneLink, cross) */
    int32_t n_phases;    /* total LightPhase count over all intersections */
    int32_t n_avail;     /* total size of phase_avail */

    /* per drivable [n_lanes + n_lanelinks] */
    const double *drv_length;    /* Drivable::length (roadnet.cpp:252,502) */
    const double *drv_max_speed; /* lanes: JSON maxSpeed; laneLinks: 10000 (roadnet.h:456) */

    /* per lane [n_lanes] */
    const int32_t *lane_road;     /* owning road */
    const int32_t *lane_index;    /* Lane::laneIndex */
    const int32_t *lane_ll_start; /* [n_lanes+1] CSR offsets into lane_ll */
    const int32_t *lane_ll;       /* [n_lanelinks] laneLink ids in Lane::laneLinks order */

    /* per road [n_roads] : lanes of a road are contiguous */
    const int32_t *road_lane_start; /* [n_roads+1] */

    /* per laneLink [n_lanelinks] */
    const int32_t *ll_start_lane;
    const int32_t *ll_end_lane;
    const int32_t *ll_inter;    /* owning intersection */
    const int32_t *ll_roadlink; /* RoadLink::index inside that intersection (roadnet.h:417) */
    const int32_t *ll_type;     /* RoadLinkType: 1 turn_right, 2 turn_left, 3 go_straight (roadnet.h:401-403) */
    const int32_t *ll_x_start;  /* [n_lanelinks+1] CSR offsets into the x_* arrays */

    /* per cross entry [n_xentries]; a laneLink's entries are sorted ascending by x_dist exactly as
     * Intersection::initCrosses sorts LaneLink::crosses (roadnet.cpp:568-575) */
    const double *x_dist;   /* Cross::distanceOnLane for the owning laneLink */
    const int32_t *x_peer;  /* entry index of the same Cross seen from the other laneLink */
    const int32_t *x_ll;    /* owning laneLink */

    /* per intersection [n_inters] */
    const int32_t *inter_virtual;
    const int32_t *inter_n_roadlinks;
    const int32_t *inter_phase_start; /* [n_inters+1] offsets into phase_time */
    const int32_t *inter_avail_start; /* [n_inters]   offset of this intersection's block in phase_avail */
    const double *phase_time;         /* [n_phases] LightPhase::time */
    const uint8_t *phase_avail;       /* per intersection: n_phases_i x n_roadlinks_i row-major 0/1 */

    /* lane change only (may be NULL when cfx_config::lane_change == 0), per lane [n_lanes] */
    const double *lane_width;         /* Lane::width (roadnet.h:303): lateral offset that completes a change */
    const int32_t *lane_n_segments;   /* Lane::segments.size() (Road::buildSegmentationByInterval roadnet.cpp:687-691) */
} cfx_net;

typedef struct cfx_config {
    double interval;          /* Engine::interval */
    int32_t rl_traffic_light; /* Engine::rlTrafficLight: lights advance only via cfx_set_tl_phase */
    int32_t lane_change;      /* Engine::laneChange (engine.cpp:53): see "Lane change" below */
    int32_t device;           /* HIP device ordinal (ignored by CPU implementations) */
    /* Implementation choices that never change results (all 0 = let the engine decide; CPU implementations ignore them).
     * The host reads them from an optional "cfx" object of the config file, which the reference ignores. */
    int32_t cross_mode;       /* cross walk: CFX_CROSS_AUTO by size, CFX_CROSS_LATENCY (k_cross), CFX_CROSS_THROUGHPUT (k_cross2) */
    int32_t layout;           /* vehicle order in HBM: CFX_LAYOUT_AUTO, CFX_LAYOUT_DENSE (rebuilt every step),
                               * CFX_LAYOUT_RING (per-drivable ring segments, committed in place; not with lane_change) */
    int32_t debug_sync;       /* synchronise after every kernel of a step and name the one that faulted (developer aid) */
    int32_t ring_lanes_per_wave; /* ring layout, developer knob (0 = the engine decides everything): digits v = G + 1000 * (B / 256)
                                  * + 10000 * F.  G: lanes one workgroup of the action kernel owns (0 = adaptive); B: its
                                  * workgroup size (256 / 512 / 1024); F: form of the step — 0 by size, 1 wave-granular action
                                  * kernel, 2 block-granular, 3 one thread per entry of the step's vehicle list (kr_index + kl_action; 6: its tiles handed out by
                                  * ticket, as on networks above 524 k drivables), 4 as 0 with the commit
                                  * as a launch of its own.  Results never depend on it (tests/test_parity_pins.py) */
    int32_t ring_capacity_percent; /* ring layout: initial ring capacities as a percentage of the bumper-to-bumper bound
                                    * (0 = 100).  Small values make the growth path run (tests) — of the rings and, below 100,
                                    * of the vehicle tables too (they start at 4 k instead of 4 M vehicle numbers); results
                                    * never depend on it */
    int32_t lane_history;     /* keep Lane::history (roadnet.cpp:900-915; see cfx_get_lane_history), numbers only Archive dumps
                               * show.  Ring layout: the record of a step is taken by spare blocks of the NEXT step's action launch
                               * (or by a launch of its own when cfx_get/set_lane_history, a reset or a load come first).  A tile
                               * (ring layout) takes the step's record behind the step's halo import: the rows of the lanes it
                               * owns are the lanes' history, a ghost lane's rows record its proxy */
    int32_t n_envs;           /* 0 or 1: one simulation.  E > 1: the network is E disjoint copies of one network with their
                               * index spaces concatenated copy by copy (roads, lanes, laneLinks, intersections: n_roads etc. are
                               * multiples of E) — E independent simulations advanced by one engine (batched environments).  Only
                               * lane change needs to know: each environment has its own schedule walk and its own priority
                               * stream (see "Lane change" below) */
    int32_t dense_form;       /* dense layout, developer knob (0 = the engine decides): 256 + a bit mask of how the step's kernels
                               * are organised — 2: the admission kernel over the lanes only (a laneLink's gate record rewritten
                               * only when its intersection's phase has changed, laneLink tails read from the committed
                               * records); 4: up to 1024 (instead of 128) spawn records of a step travel in the admission
                               * kernel's arguments.  256 = both off.  Results never depend on it (tests/test_dense_forms.py) */
} cfx_config;
#define CFX_CROSS_AUTO 0
#define CFX_CROSS_LATENCY 1
#define CFX_CROSS_THROUGHPUT 2
#define CFX_LAYOUT_AUTO 0
#define CFX_LAYOUT_DENSE 1
#define CFX_LAYOUT_RING 2

/* VehicleInfo (reference src/vehicle/vehicle.h:31-45) + the two per-template constants derived from it. */
typedef struct cfx_vehicle_template {
    double len, width, max_pos_acc, max_neg_acc, usual_pos_acc, usual_neg_acc;
    double min_gap, max_speed, headway_time, yield_distance, turn_speed;
    /* maxSpeed^2/usualNegAcc/2 + maxSpeed*interval*2 : ControllerInfo::approachingIntersectionDistance
     * (vehicle.cpp:42-44) and the head-of-lane leader search bound (vehicle.cpp:190-192). */
    double approach_dist;
    double initial_speed; /* VehicleInfo::speed (vehicle.h:32) a vehicle enters its first lane with; only
                           * push_vehicle can set it (engine.cpp:696), flows always create vehicles at rest */
} cfx_vehicle_template;

/* One vehicle entering the simulation this step (Flow::nextStep flow.cpp:6-22 + Engine::planRoute
 * engine.cpp:450-470, both evaluated on the host because they own the mt19937 stream). */
typedef struct cfx_spawn {
    int32_t vid;        /* dense id; must equal the number of vehicles spawned since the last reset */
    int32_t priority;   /* Vehicle::priority (vehicle.cpp:45) */
    int32_t templ;      /* template index */
    int32_t route;      /* route index */
    int32_t lane;       /* first lane (Router::getFirstDrivable router.cpp:23-37) */
    int32_t prev_wait;  /* vid previously pushed on this lane's waitingBuffer since the last reset, or -1 */
    double enter_time;  /* Vehicle::enterTime (vehicle.cpp:46) */
} cfx_spawn;

typedef struct cfx_scalars {
    int64_t step;                  /* Engine::step */
    int64_t active_vehicle_count;  /* Engine::activeVehicleCount (running vehicles) */
    int64_t finished_vehicle_count;/* Engine::finishedVehicleCnt */
    int64_t spawned_vehicle_count; /* vehicles handed to cfx_step since the last reset */
    double cumulative_travel_time; /* Engine::cumulativeTravelTime */
    double live_enter_time_sum;    /* sum of enter_time over spawned-and-not-finished vehicles (0 if not kept) */
    int64_t vehicle_steps;         /* sum over executed steps of the vehicles that took the step (ran phase 4) */
    /* pairs of vehicles that entered the same drivable in one step with EXACTLY equal distances, since the last reset /
     * load.  Engine::updateLocation orders the entrants with an unstable std::sort on distance alone (engine.cpp:480), so
     * the reference's own order of such a pair depends on heap addresses and thread timing; this ABI breaks the tie by
     * vehicle number.  A comparison with the reference is well defined only while this counter stands still. */
    int64_t tie_events;
    /* the drivables (index space above) that received the most recent tie events: event number i (counting from 0 since
     * the last reset / load) is kept at index i % 8; -1 = none yet, or not known (a tiled engine does not report them).
     * A checker uses them to see WHERE two correct engines may differ from then on. */
    int32_t tie_drivables[8];
    /* diagnostic of the last step (0 where an implementation does not count it; never part of a result): vehicles handed to
     * the cross phase */
    int32_t diag_cross_jobs;
    /* cfx_set_vehicle_speed calls for a vehicle number the NEXT cfx_step was expected to create (see there) whose step then did
     * not create it, since the engine was created: such a speed is dropped, and counted here */
    int32_t dropped_future_speeds;
} cfx_scalars;

/* Full per-vehicle state of every running vehicle, caller-allocated SoA (any pointer may be NULL).
 * Order: by drivable (index space above), then front (furthest ahead) to back inside a drivable,
 * i.e. the order of Drivable::vehicles (roadnet.h:245). */
typedef struct cfx_vehicle_view {
    int32_t capacity;       /* in: number of elements each array can hold */
    int32_t count;          /* out: number of running vehicles */
    int32_t *vid;
    int32_t *drivable;
    int32_t *prev_drivable; /* -1 if none */
    int32_t *leader_vid;    /* -1 if none */
    int32_t *blocker_vid;   /* -1 if none */
    int32_t *enter_ll_time; /* INT32_MAX when not on a laneLink */
    int32_t *route_pos;     /* Router::iCurRoad as an index into the route */
    double *dis;
    double *speed;
    double *gap;            /* ControllerInfo::gap: refreshed only where leader_vid >= 0, otherwise the value it last had */
    /* lane change (zero / -1 when it is off) */
    int32_t *lc_partner_vid; /* LaneChangeInfo::partner */
    uint8_t *lc_flags;       /* CFX_LC_* bits */
    double *lc_offset;       /* LaneChangeInfo::offset */
    int32_t *lc_last_dir;    /* LaneChange::lastDir (what the replay log prints, engine.cpp:524) */
    int32_t *lc_target_lane; /* signalSend->target while CFX_LC_CHANGING (the only signal that outlives a step), else -1 */
    int32_t *lc_direction;   /* signalSend->direction, same condition, else 0 */
    double *lc_last_change_time; /* LaneChange::lastChangeTime */
    double *lc_waiting_time;     /* LaneChange::waitingTime */
} cfx_vehicle_view;
#define CFX_LC_SHADOW 1u   /* partnerType == 2: a shadow; its id is "<parent id>_shadow" until the change finishes */
#define CFX_LC_PARENT 2u   /* partnerType == 1: the real vehicle of a changing pair */
#define CFX_LC_CHANGING 4u /* LaneChange::changing */

int32_t cfx_abi_version(void);
int32_t cfx_create(const cfx_net *net, const cfx_config *cfg, cfx_engine **out);
void cfx_destroy(cfx_engine *e);
const char *cfx_last_error(const cfx_engine *e); /* e may be NULL: error of the last failed cfx_create */
const char *cfx_backend_name(void);             /* "hip-gfx950" for the product library */

/* Append vehicle templates / routes (flows at load time; push_vehicle / set_vehicle_route later).
 * A route is its road sequence plus, for every position p and every lane j of road[p], the laneLink
 * Router::getNextDrivable would choose from that lane (router.cpp:49-76,96-129), or -1.
 *   route_start[n_routes+1]           offsets into `roads`
 *   next_start[route_start[n]+1]      offsets into `next_ll`, one block per route position */
int32_t cfx_add_templates(cfx_engine *e, int32_t n, const cfx_vehicle_template *t);
int32_t cfx_add_routes(cfx_engine *e, int32_t n_routes, const int32_t *route_start, const int32_t *roads,
                       const int32_t *next_start, const int32_t *next_ll);

/* One Engine::nextStep (engine.cpp:566-594): enqueue `recs` on their lanes' waiting buffers, admit,
 * notify crosses, get action, update location, commit, leader/gap, traffic lights, step += 1. */
int32_t cfx_step(cfx_engine *e, const cfx_spawn *recs, int32_t n);
int32_t cfx_sync(cfx_engine *e);

/* Engine::reset (engine.cpp:744-760): drop all vehicles and waiting buffers, lights to phase 0, step 0.
 * Templates and routes are kept. */
int32_t cfx_reset(cfx_engine *e);

/* TrafficLight::setPhase (trafficlight.cpp:39-41).  Unlike the reference this is range-checked. */
int32_t cfx_set_tl_phase(cfx_engine *e, int32_t inter, int32_t phase);
/* same for n (intersection, phase) pairs in one call (an RL agent sets every signal each step) */
int32_t cfx_set_tl_phases(cfx_engine *e, int32_t n, const int32_t *inters, const int32_t *phases);
int32_t cfx_get_tl_state(cfx_engine *e, int32_t *cur_phase /*[n_inters]*/, double *remain /*[n_inters]*/);

int32_t cfx_get_scalars(cfx_engine *e, cfx_scalars *out);
/* the vehicle layout this engine runs on: CFX_LAYOUT_DENSE or CFX_LAYOUT_RING (what cfx_config::layout = AUTO resolved to;
 * CPU implementations report CFX_LAYOUT_AUTO) */
int32_t cfx_get_layout(cfx_engine *e);
/* ring layout: total slots of all rings and how often every capacity has been doubled so far (1 = never; 0 slots on other
 * layouts / implementations) */
int32_t cfx_get_ring_info(cfx_engine *e, int64_t *slots, int32_t *capacity_scale);
int32_t cfx_get_lane_counts(cfx_engine *e, int32_t *out /*[n_lanes]*/);         /* getLaneVehicleCount */
int32_t cfx_get_lane_waiting_counts(cfx_engine *e, int32_t *out /*[n_lanes]*/); /* speed < 0.1 (engine.cpp:641) */
int32_t cfx_get_vehicles(cfx_engine *e, cfx_vehicle_view *view);
/* per-vid life-cycle state for vids [first, first+n): 0 waiting, 1 running, 2 finished */
int32_t cfx_get_vehicle_status(cfx_engine *e, int32_t first_vid, int32_t n, uint8_t *out);
/* vids still sitting in lanes' waiting buffers, lane by lane, FIFO order; returns count via *n */
int32_t cfx_get_waiting(cfx_engine *e, int32_t capacity, int32_t *vid, int32_t *lane, int32_t *n);

/* Vehicle::setCustomSpeed (vehicle.h:128-131, used by getCarFollowSpeed vehicle.cpp:214,220-221): overrides
 * the car-following target for the vehicle's next step only (Vehicle::update clears it, vehicle.cpp:120-122).
 * A vid at or beyond the vehicles created so far names a vehicle the NEXT cfx_step's spawn records will create (a vehicle
 * pushed since the last step, which Engine::setVehicleSpeed engine.cpp:827-834 already finds): the speed is kept and is in
 * place before that step's admission; it is dropped if that step does not create the vehicle — counted in
 * cfx_scalars::dropped_future_speeds.  Only the next CFX_FUTURE_SPEED_WINDOW vehicle numbers are accepted that way; anything
 * beyond (or negative) is CFX_ERR_INVALID. */
#define CFX_FUTURE_SPEED_WINDOW 65536
int32_t cfx_set_vehicle_speed(cfx_engine *e, int32_t vid, double speed);
/* Switch a waiting or running vehicle to route `route` (already added with cfx_add_routes) whose first road is the
 * road the vehicle is on; Router::iCurRoad restarts at 0 (Router::setRoute router.cpp:245-264 after its checks,
 * which the host performs). */
int32_t cfx_set_vehicle_route(cfx_engine *e, int32_t vid, int32_t route);
/* state: 0 waiting, 1 running, 2 finished; drivable / route_pos / route are -1 unless running (route: any state) */
int32_t cfx_get_vehicle(cfx_engine *e, int32_t vid, int32_t *state, int32_t *drivable, int32_t *route_pos,
                        int32_t *route);

/* Complete dynamic state, for Archive-style snapshot / restore (reference src/engine/archive.cpp:9-126).
 * Reading it back is done with the getters above; cfx_load_state replaces the engine's whole dynamic state
 * (templates and routes referenced by it must already have been added). */
typedef struct cfx_state {
    int64_t step, finished_vehicle_count, vehicle_steps;
    double cumulative_travel_time;
    /* vehicle table, index = vid */
    int32_t n_vehicles;
    const int32_t *v_priority, *v_templ, *v_route;
    const double *v_enter_time;
    const uint8_t *v_state;          /* 0 waiting, 1 running, 2 finished */
    /* running vehicles in Drivable::vehicles order: by drivable, front to back */
    int32_t n_running;
    const int32_t *r_vid, *r_drivable, *r_prev_drivable, *r_blocker_vid, *r_enter_ll_time, *r_route_pos;
    const double *r_dis, *r_speed;
    const double *r_custom_speed;    /* NaN where no custom speed is pending; may be NULL */
    /* ControllerInfo::gap is STORED state in the reference: Vehicle::getCarFollowSpeed (vehicle.cpp:212-238) reads what
     * updateLeaderAndGap left at the end of the previous step, and Archive::resume restores it with the vehicle.  An engine
     * that derives leader and gap from the order at the start of a step uses r_gap[i] INSTEAD of the derived gap in the first
     * step after the load, for every vehicle that has a leader then (NaN, or r_gap == NULL: the derived one).  The two are the
     * same number unless the archive came through a file whose `dis` / `gap` literals were not read back exactly — the
     * reference's JSON reader is not correctly rounded (csrc/host/json_number.h).  Lane change: makeSignal reads it too. */
    const double *r_gap;
    /* lane change (all may be NULL = no lane-change state): what of LaneChange / LaneChangeInfo outlives a step —
     * signals received, target leader / follower are cleared by every step's clearSignal (engine.cpp:424) */
    const int32_t *r_lc_partner_vid, *r_lc_last_dir, *r_lc_target_lane, *r_lc_direction;
    const uint8_t *r_lc_flags;       /* CFX_LC_* */
    const double *r_lc_offset, *r_lc_last_change_time, *r_lc_waiting_time;
    /* waiting buffers, lane by lane, FIFO order */
    int32_t n_waiting;
    const int32_t *w_vid, *w_lane;
    /* traffic lights [n_inters] */
    const int32_t *tl_phase;
    const double *tl_remain;
} cfx_state;
int32_t cfx_load_state(cfx_engine *e, const cfx_state *s);
/* pending custom speeds of the running vehicles, same order as cfx_get_vehicles (NaN = none) */
int32_t cfx_get_custom_speeds(cfx_engine *e, int32_t capacity, double *out);

/* ---- Lane::history (reference src/roadnet/roadnet.h:305-316, Lane::updateHistory roadnet.cpp:900-915, called for every lane at
 * the end of every step, engine.cpp:429-442): the last steps' {vehicle count, mean speed} per lane — up to
 * CFX_LANE_HISTORY_MAX records, the list is trimmed to 240 BEFORE a step's record is appended — and the two running aggregates
 * historyVehicleNum / historyAverageSpeed, kept up with the reference's own recurrence (the sum of the speeds is taken in the
 * lane's list order; the popped records' share is subtracted from historyVehicleNum * historyAverageSpeed, a product formed anew
 * every step).  It feeds Road::getAverageSpeed / the DURATION router, which nothing in the reference can select (router.h:42);
 * what makes it visible is Archive::dump (archive.cpp:286-294).  Kept only with cfx_config::lane_history; Engine::reset does
 * NOT clear it (Lane::reset roadnet.cpp:832-835), cfx_load_state leaves it alone — Archive::resume's part is cfx_set_lane_history. */
#define CFX_LANE_HISTORY_MAX 241
typedef struct cfx_lane_history {
    int32_t n_lanes;                /* in: must equal the engine's */
    int32_t *len;                   /* [n_lanes] records held */
    int32_t *vehicle_num;           /* [n_lanes * CFX_LANE_HISTORY_MAX] lane-major, oldest record first */
    double *average_speed;          /* [n_lanes * CFX_LANE_HISTORY_MAX] */
    int32_t *history_vehicle_num;   /* [n_lanes] Lane::historyVehicleNum */
    double *history_average_speed;  /* [n_lanes] Lane::historyAverageSpeed */
} cfx_lane_history;
int32_t cfx_get_lane_history(cfx_engine *e, cfx_lane_history *out);       /* CFX_ERR_STATE: the engine does not keep it */
int32_t cfx_set_lane_history(cfx_engine *e, const cfx_lane_history *in);

/* ---- Lane change (cfx_config::lane_change = 1; reference src/vehicle/lanechange.cpp, engine.cpp:374-400,792-820) ----
 * A vehicle that starts to change lane gets a SHADOW in the target lane: a new vehicle (`new Vehicle(*v, id + "_shadow")`,
 * engine.cpp:812-820) that draws its priority from the engine's mt19937 — the stream the host's spawner owns.  So
 *   before cfx_step   cfx_lane_change_supply(e, n, priorities): the next n priorities the generator WOULD hand out
 *                     (already filtered for collisions with live vehicles); the step uses the first k of them, in the
 *                     order the shadows are created;
 *   after cfx_step    cfx_lane_change_poll(e, cap, parent_vid, &k): the vehicles that got a shadow in that step, in
 *                     creation order; shadow i has vid = (vehicles known before the step's spawn records) + (number of
 *                     spawn records) + i.  The host advances its generator past the draws that produced the first k
 *                     priorities and numbers the NEXT step's spawn records after the shadows.
 * cfx_lane_change_poll waits only for the part of the step that creates shadows, not for the whole step.
 * Order: the reference walks the candidates in `std::set<Vehicle*>` (heap address) order, which is not a function of
 * the simulation state, and then sorts them by urgency with std::sort — all urgencies are equal, yet
 * libstdc++'s introsort permutes more than 16 equal elements.  This ABI fixes the walk to: candidates in creation order
 * (ascending vid = the address order of a heap that never reuses memory), put through that very permutation (a closed
 * function of the candidate count; `lcSortedPosition`, csrc/hip/cfx_lc_kernels.h).  Shadows are numbered, and take the
 * supplied priorities, in walk order.  More than n shadows in one step: CFX_ERR_CAPACITY from the poll.
 * Batched environments (cfx_config::n_envs = E > 1): every environment is its own Engine in the reference, so the walk above is
 * taken per environment (its candidates in ascending vid, put through the permutation of ITS candidate count), the
 * environments one after the other; n must be a multiple of E and environment e takes its priorities from
 * priorities[e * n / E ...]; the poll lists the parents environment by environment, shadows are numbered in that order. */
int32_t cfx_lane_change_supply(cfx_engine *e, int32_t n, const int32_t *priorities);
int32_t cfx_lane_change_poll(cfx_engine *e, int32_t capacity, int32_t *parent_vid, int32_t *n);

/* ---- Tiling one road network over several engines / GPUs -----------------------------------
 * An engine may be created on a SUB-network: the intersections one tile owns, their laneLinks, every lane ending
 * in them, plus the "ghost" lanes it feeds that end in a foreign tile.  Vehicles on ghost lanes are frozen proxies
 * of the owner's state.  After every cfx_step the caller performs ONE exchange:
 *   cfx_halo_export(e, send)   fills `send` (host memory, cfx_halo_layout::send_bytes) with
 *                              - for every ghost lane (this tile is upstream): the vehicles that entered it this step
 *                                (migrants), and keeps only the last of them as the lane's proxy;
 *                              - for every import lane (this tile is downstream): the record of the lane's tail;
 *   ... the caller moves each neighbour's slice to that neighbour (RCCL / host copy) ...
 *   cfx_halo_import(e, recv)   appends received migrants to the import lanes and refreshes the proxies of ghost
 *                              lanes that had no entrant of their own this step.
 * Block formats (little endian, offsets are byte offsets into the send / recv buffers):
 *   migrant block (CFX_HALO_MIG_BYTES): int32 count, int32 pad, then CFX_HALO_MAX_MIGRANTS records
 *       {int32 vid, int32 route_pos, int32 prev_lanelink (global id), int32 pad, double dis, double speed}
 *   tail block (CFX_HALO_TAIL_BYTES):   {int32 vid (-1: lane empty), int32 prev_lanelink (global id or -1),
 *                                        double dis, double speed}
 * cfx_load_state on a tile (Archive::resume, archive.cpp:73-126, for one tile of a network-wide archive): the running
 * vehicles of the drivables the tile owns; for every ghost lane at most ONE vehicle, the lane's tail — it becomes the proxy
 * (not counted as running here, never stepped); r_prev_drivable of a vehicle that came from another tile's laneLink as
 * -(global laneLink id + 2), as migrants carry it; waiting buffers of owned AND ghost lanes (a ghost lane mirrors its
 * owner's queue); finished_vehicle_count / cumulative_travel_time / vehicle_steps on one tile of the job only. */
#define CFX_HALO_MAX_MIGRANTS 8
#define CFX_HALO_MIG_BYTES (8 + CFX_HALO_MAX_MIGRANTS * 32)
#define CFX_HALO_TAIL_BYTES 24
typedef struct cfx_halo_layout {
    int32_t n_ghost;                /* ghost lanes (local lane ids), this tile upstream */
    const int32_t *ghost_lane;
    const int32_t *ghost_send_off;  /* migrant block in the send buffer */
    const int32_t *ghost_recv_off;  /* tail block in the recv buffer */
    int32_t n_import;               /* owned lanes fed by a foreign tile */
    const int32_t *import_lane;
    const int32_t *import_recv_off; /* migrant block in the recv buffer */
    const int32_t *import_send_off; /* tail block in the send buffer */
    int32_t send_bytes, recv_bytes;
    int32_t n_global_lanelinks;     /* size of the two maps below */
    const int32_t *lanelink_global; /* [n_lanelinks of this engine] local -> global laneLink id */
    const int32_t *lanelink_local;  /* [n_global_lanelinks] global -> local laneLink id or -1 */
} cfx_halo_layout;
int32_t cfx_halo_config(cfx_engine *e, const cfx_halo_layout *layout);
int32_t cfx_halo_export(cfx_engine *e, void *send_host);      /* NULL: see cfx_halo_device_buffers */
int32_t cfx_halo_import(cfx_engine *e, const void *recv_host);

/* Device-initiated exchange (no host round trip).  Every directed neighbour message gets a MAILBOX in host memory that
 * both engines can reach (the caller maps it into both processes, e.g. POSIX shared memory; the HIP engine registers
 * it with the device): a 128-byte header whose first 8 bytes are the epoch flag, then two message buffers used
 * alternately (a sender is never more than one step ahead of its receiver, because its own next import needs the
 * receiver's next export).  After cfx_halo_attach the per-step exchange is
 *     cfx_halo_post(e)   export straight into the peers' mailboxes, then publish the step's epoch (system-scope release)
 *     cfx_halo_wait(e)   wait until every peer has published this epoch (system-scope acquire), then import
 * both asynchronous on the engine's stream for the HIP engine (the waiting is done by the import kernel, bounded), so a
 * step never synchronises with the host; CPU implementations block in cfx_halo_wait.  Call post on every local engine
 * before wait on any.  Epochs grow monotonically over cfx_reset, so mailboxes are never cleared. */
#define CFX_HALO_MAILBOX_HEADER 128
#define CFX_HALO_MAILBOX_BYTES(message_bytes) (CFX_HALO_MAILBOX_HEADER + 2 * (size_t) (message_bytes))
#define CFX_HALO_MAX_PEERS 16
typedef struct cfx_halo_peer {
    int32_t send_off, send_bytes; /* this peer's slice of the send layout of cfx_halo_config */
    int32_t recv_off, recv_bytes; /* ... and of the recv layout */
    void *send_mailbox;           /* CFX_HALO_MAILBOX_BYTES(send_bytes), written by this engine */
    void *recv_mailbox;           /* CFX_HALO_MAILBOX_BYTES(recv_bytes), written by the peer */
    int32_t device_memory;        /* 0: host memory both processes map (registered with the device by the engine);
                                   * 1: device pointers from cfx_halo_mailbox_alloc / _open — the mailbox lives in the
                                   *    RECEIVER's HBM and the sender's export kernel writes it over xGMI */
    int32_t reserved;
} cfx_halo_peer;

/* Mailboxes in device memory.  The receiver of a message owns its mailbox:
 *   cfx_halo_mailbox_alloc(e, message_bytes, &ptr, handle)  zeroed device memory on e's GPU for one incoming message
 *                              (CFX_HALO_MAILBOX_BYTES(message_bytes)), and a handle another PROCESS can open it with;
 *   cfx_halo_mailbox_open(e, handle, &ptr)                  in the sender's process: the peer's mailbox as a pointer e's GPU
 *                              can write through (hipIpcOpenMemHandle; peer HBM over xGMI between two GPUs of a node).
 * Tiles of ONE process pass `ptr` itself (the engine enables peer access between their GPUs in cfx_halo_attach).
 * CFX_ERR_STATE: this implementation / platform cannot do it — fall back to host-memory mailboxes. */
#define CFX_IPC_HANDLE_BYTES 64
int32_t cfx_halo_mailbox_alloc(cfx_engine *e, int32_t message_bytes, void **device_ptr, uint8_t handle[CFX_IPC_HANDLE_BYTES]);
int32_t cfx_halo_mailbox_open(cfx_engine *e, const uint8_t handle[CFX_IPC_HANDLE_BYTES], void **device_ptr);
/* 1: every mailbox cfx_halo_mailbox_alloc has handed out so far is fine-grained device memory (a kernel on ANOTHER GPU sees
 * the sender's stores while both kernels run); 0: some are plain allocations, because the platform would not export
 * fine-grained memory — good between processes that share one GPU, not guaranteed across two: the caller should then use the
 * host-memory mailboxes unless all tiles sit on one device. */
int32_t cfx_halo_mailbox_fine_grained(cfx_engine *e);
/* The PHYSICAL device the engine runs on, as a NUL-terminated string that is equal for two engines exactly if they share
 * a device whatever each process's visible-device numbering is (HIP: the PCI bus id, e.g. "0000:05:00.0"; a CPU
 * implementation: "cpu").  A caller that is offered plain (coarse-grained) device mailboxes compares it over all ranks. */
int32_t cfx_device_identity(cfx_engine *e, char *buf, int32_t capacity);
/* Free and total memory of that device in bytes (hipMemGetInfo; a CPU implementation: 0, 0) — what a long run is checked
 * against: a step of an ordinary run allocates nothing (tests/test_steady_state.py). */
int32_t cfx_device_memory(cfx_engine *e, int64_t *free_bytes, int64_t *total_bytes);
/* The staged exchange with the messages left in device memory (for a device-to-device transport such as RCCL send / recv
 * on these very buffers): cfx_halo_export(e, NULL) then writes the send buffer only on the device, cfx_halo_import(e, NULL)
 * reads the recv buffer from the device.  Both buffers are fixed for the life of the engine. */
int32_t cfx_halo_device_buffers(cfx_engine *e, void **send_dev, void **recv_dev);
int32_t cfx_halo_attach(cfx_engine *e, int32_t n_peers, const cfx_halo_peer *peers);
int32_t cfx_halo_post(cfx_engine *e);
int32_t cfx_halo_wait(cfx_engine *e);
/* spawn records whose lane is not part of this engine's sub-network carry lane = -1: only the per-vehicle
 * static table is filled (every tile knows every vehicle, so migrants need no static payload). */

/* Optional per-kernel timing with HIP events recorded on the engine's own stream (bench.py roofline).
 * Kernel ids are dense 0..cfx_profile_kernel_count()-1; cfx_profile_read() synchronises, adds the
 * elapsed time of every bracketed launch since the last read into total_ms[] / launches[] and clears. */
int32_t cfx_profile_kernel_count(void);
const char *cfx_profile_kernel_name(int32_t k);
int32_t cfx_profile_enable(cfx_engine *e, int32_t on);
int32_t cfx_profile_read(cfx_engine *e, double *total_ms, int64_t *launches);
/* The symbol of the kernel the engine launched LAST in timing slot k (the slots are roles — "k_action" is the car-following
 * launch whatever its organisation; this names the one that ran: "kr_action<256>", "kl_action", "kd_action", ...).  "" before
 * the first launch in that slot, and from CPU implementations. */
const char *cfx_profile_kernel_symbol(cfx_engine *e, int32_t k);

/* Where the HOST's time inside cfx_step went (never part of a result): every cfx_step call is timed with the host's steady
 * clock; a call that had to wait for the device or allocate says why.  `worst_*` describe the slowest call since the last
 * cfx_get_host_stats(e, out, 1) (reset = 1 clears everything but the *_total counters). */
#define CFX_STALL_TABLES 1       /* new templates / routes were uploaded: the stream was drained first (cfx_add_*) */
#define CFX_STALL_RING_REGROW 2  /* ring layout: the rings were rebuilt with doubled capacities (gather, drain, reallocate) */
#define CFX_STALL_SLOT_GROW 4    /* dense layout: the slot arrays grew (drains) */
#define CFX_STALL_VID_GROW 8     /* the per-vehicle tables grew (allocation only; the copy is ordered on the stream) */
#define CFX_STALL_SCALARS 16     /* the scalars were read back to decide something (overflow report, capacity bound) */
#define CFX_STALL_STAGE_GROW 32  /* the spawn staging buffers grew (drains) */
#define CFX_STALL_LIST_GROW 64   /* the step's vehicle list grew (allocation only) */
typedef struct cfx_host_stats {
    int64_t step_calls;           /* cfx_step calls since the last clearing read */
    double step_call_us_sum;      /* their host time */
    double worst_step_call_us;    /* the slowest of them ... */
    int64_t worst_step_call_at;   /* ... the engine step it submitted ... */
    int32_t worst_step_call_cause;/* ... and the CFX_STALL_* bits raised inside it (0: nothing of the above — the time went into
                                   * the HIP runtime's launch path, e.g. a full queue) */
    int32_t calls_over_1ms;       /* calls that took more than a millisecond */
    int64_t ring_regrows_total;   /* since the engine was created */
    int64_t table_grows_total;    /* vid-table + slot-array + list growths since the engine was created */
    /* the slowest cfx_get_vehicle_status call since the last clearing read (the spawner's priority-collision query: the one
     * call a free-running host makes that waits for the device), split into: launching a pending commit, enqueueing the copy,
     * waiting for the stream */
    double worst_status_query_us, worst_status_query_settle_us, worst_status_query_copy_us, worst_status_query_wait_us;
    int64_t status_queries;
} cfx_host_stats;
int32_t cfx_get_host_stats(cfx_engine *e, cfx_host_stats *out, int32_t reset);
/* Measurement aid: keeps the device busy with plain arithmetic for about `microseconds` on the engine's stream (returns at
 * once; the engine's next kernels queue behind it).  A benchmark calls it right before its warm-up steps so that a short
 * timed region is not measured on clocks that are still ramping up after host-only work.  Touches no engine state. */
int32_t cfx_device_spin(cfx_engine *e, int64_t microseconds);

#ifdef __cplusplus
}
#endif
#endif /* CITYFLOW_AMD_H */
