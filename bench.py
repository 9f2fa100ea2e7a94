#!/usr/bin/env python
"""Benchmark of the hot path (one `next_step()` of the MI355X-native CityFlow engine = one "step").

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): the reference generator's 30x30 grid (cityflow_amd/data/scenarios/grid_30x30,
produced by /root/reference/tools/generator) with ~100k concurrently running vehicles.  The stock generator's demand never
gets near 100k, so 3000 seeded interior-origin flows (one vehicle every 6 s each, for the first 240 s) are
added on top of the 120 stock flows.  The workload is the network at simulated time t = 300 s: building that state up
(BUILD_UP_STEPS untimed steps, ~97k vehicles running at the end) is part of constructing the input, like loading a dataset
— it does NOT depend on --warmup.  That state is written to an Archive file; every engine of the run (the GPU engine, the
reference CPU legs, the parity replay) starts from that very file.  Then W warm-up steps, then K timed steps.  All inputs
are resident in HBM when the timed region starts; per step the host only hands over the step's few spawn records.

N > 1 (BASELINE's "30x30 at 1/2/4/8 GPUs"): the SAME network and state cut into rows x cols tiles of intersections, one
tile per GPU / rank, one-lane ghost halo exchanged every step (cityflow_amd.tiled.DistributedEngine) —
"scaling": "strong".  With WORLD_SIZE unset, `--gpus N` re-launches this script under `torch.distributed.run` with N ranks
(one per GPU; ranks share devices when the box has fewer); with WORLD_SIZE set it must equal N.  `value` = vehicle-steps of
all tiles / the slowest rank's time.  `--weak` grows the grid with N instead (every rank a 30x30 block), `--replicas` runs
N independent copies (no exchange).  An N > 1 line also carries `scale_100x100`: the 100x100 / 1 M-vehicle network
(BASELINE configs[4]) tiled the same way — the N = 1 line times the same network on one GPU as `roofline_at_scale`.

Output: ONE JSON line on rank 0 with these extra objects:
  roofline           car-following kernel (k_action / kr_action) of the headline workload: algorithmic bytes (48 B per
                     running vehicle) / average launch duration measured with HIP events on the engine's
                     stream (the dispatch's own start / stop events, hipExtLaunchKernel: what a rocprofv3 kernel trace
                     reports); `traffic` from the committed PMC summary of the very kernel sources that run
  roofline_at_scale  the same figure where the bandwidth target means something: the 100x100 network,
                     ~1 M vehicles per launch (N = 1)
  cpu_baseline       the unmodified reference engine (oracle/_ref, prebuilt) timed on this box's host cores for a bounded
                     number of steps from the same state (falls back to the CPU twin, kind "port")
  parity             in-run parity against the reference (N = 1) / against a single GPU engine (N > 1) at several points
                     of the window: per-lane counts, vehicle count, signal phases, every vehicle's (speed, distance)
  rl_loop            RL-style loop on the headline network (set every signal, step, read per-lane counts): steps/s with the
                     array API, with the reference-style dict API, and the reference engine itself from the same state
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

# numpy (the array getters return numpy arrays) is imported HERE, with its BLAS worker pool cut to one thread: importing it
# starts one OpenBLAS worker per core (63 on the 256-core benchmark host), each of which spins for ~20 ms before it goes to
# sleep — 1.3 CPU-seconds inside 50 ms, against the container's CFS quota of 16 CPUs per 100 ms period (cgroup cpu.max
# "1600000 100000").  Rounds 4-6 imported it lazily, with the first array getter, i.e. right behind the timed region: the kernel
# then throttled the whole container — the stepping thread included — for the rest of the period, and ONE next_step() call of
# the first or second sustained window took 45-80 ms, whatever it happened to be doing (measured in round 6: inside cfx_step,
# inside the host spawner, inside a status query; tools/throttle_watch.py, profiles/r06_stall_*).  Nothing in this file uses BLAS.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy  # noqa: E402,F401  (see above: before anything is timed)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured achievable)
ACTION_BYTES_PER_VEHICLE = 48  # algorithmic bytes of the car-following (get-action) kernel

BUILD_UP_STEPS = 300           # simulated seconds of demand that define the workload state (see the module docstring)
N_EXTRA_FLOWS = 3000
EXTRA_INTERVAL = 6.0
EXTRA_END = 240
SCALE_GRID = 100               # the bandwidth-regime network: 100x100, BASELINE.json configs[4]
SCALE_FLOWS = 33000            # seeded interior flows on it (~1 M running vehicles at t = 300 s)


# ------------------------------------------------------------------------------------------------ workloads
def build_workload(workdir, seed, scenario="grid_30x30", n_extra=N_EXTRA_FLOWS):
    from cityflow_amd import scenarios
    generated = scenario.startswith("gen_")  # gen_RxC: a generator-format grid of any size
    if generated:
        r, c = (int(x) for x in scenario[4:].split("x"))
        base = scenarios.generate_grid(r, c, workdir, seed=seed)
    else:
        base = scenarios.materialize(scenario, workdir)
    d = os.path.dirname(base)
    flow = os.path.join(d, "flow_bench_%d.json" % n_extra)
    if not os.path.exists(flow):
        scenarios.dense_flows(os.path.join(d, "roadnet.json"), flow, n_extra, seed=12345, interval=EXTRA_INTERVAL,
                              base_flow=os.path.join(d, "flow.json"), end_time=EXTRA_END)
    if generated:
        cfg = dict(json.load(open(base)), flowFile=os.path.basename(flow))
        path = os.path.join(d, "config_bench.json")
        scenarios._write_json_atomic(path, cfg)
        return path
    return scenarios.materialize(scenario, workdir, flow_file=flow, seed=seed)


def tile_grid(n):
    rows = 1
    for r in range(1, int(n ** 0.5) + 1):
        if n % r == 0:
            rows = r
    return rows, n // rows


def build_tiled_workload(workdir, rows, cols, block, n_extra_per_tile):
    """(block*rows) x (block*cols) generated grid + the stock flows + n_extra_per_tile seeded interior flows per tile."""
    from cityflow_amd import scenarios
    base = scenarios.generate_grid(block * rows, block * cols, workdir)
    d = os.path.dirname(base)
    n_extra = n_extra_per_tile * rows * cols
    flow = os.path.join(d, "flow_bench_%d.json" % n_extra)
    if not os.path.exists(flow):
        scenarios.dense_flows(os.path.join(d, "roadnet.json"), flow, n_extra, seed=12345, interval=EXTRA_INTERVAL,
                              base_flow=os.path.join(d, "flow.json"), end_time=EXTRA_END)
    cfg = dict(json.load(open(base)), flowFile=os.path.basename(flow))
    path = os.path.join(d, "config_bench.json")
    scenarios._write_json_atomic(path, cfg)
    return path


def with_config(cfg_path, suffix, **overrides):
    """A sibling config file with some keys changed (e.g. rlTrafficLight for the RL loop)."""
    from cityflow_amd import scenarios
    c = dict(json.load(open(cfg_path)), **overrides)
    path = cfg_path.replace(".json", "_%s.json" % suffix)
    scenarios._write_json_atomic(path, c)
    return path


# ------------------------------------------------------------------------------------------------ parity records
def _lane_hash(counts):
    return hashlib.sha256(json.dumps(sorted(counts.items())).encode()).hexdigest()


def _state_hash(speed, distance):
    h = hashlib.sha256()
    for k in sorted(speed):
        h.update(("%s %s %s\n" % (k, float(speed[k]).hex(), float(distance[k]).hex())).encode())
    return h.hexdigest()


def _phase_hash(lights, real=None):
    """Signal state {intersection id: (phase index, remaining duration)} of the (non-virtual) intersections, exact bits."""
    h = hashlib.sha256()
    for k in sorted(lights):
        if real is None or k in real:
            h.update(("%s %d %s\n" % (k, int(lights[k][0]), float(lights[k][1]).hex())).encode())
    return h.hexdigest()


def engine_lights(eng):
    """{intersection id: (phase index, remaining duration)} of this repo's engine (cfx_get_tl_state)."""
    ph, rm = eng._tl_state()
    return {k: (int(p), float(r)) for k, p, r in zip(eng.intersection_ids(), ph, rm)}


def archive_lights(eng, scratch):
    """(phase, remain) of every intersection out of the engine's own Archive JSON — the only place the reference exposes
    them (reference src/engine/archive.cpp:326-343): {intersection id: (phase index, remaining duration)}."""
    eng.snapshot().dump(scratch)
    with open(scratch) as f:
        lights = json.load(f)["trafficLights"]
    os.unlink(scratch)
    return {k: (int(x["curPhaseIndex"]), float(x["remainDuration"])) for k, x in lights.items()}


def parity_record(eng, lights=None, real=None):
    """What the in-run parity check compares after the same number of steps from the same state: per-lane vehicle counts,
    vehicle count, signal phases (when `lights` = {id: (phase, remain)} is given) and every running vehicle's exact
    (speed, distance) bits."""
    speed, distance = eng.get_vehicle_speed(), eng.get_vehicle_distance()
    rec = {"vehicles": eng.get_vehicle_count(), "lane_hash": _lane_hash(eng.get_lane_vehicle_count()),
           "state_hash": _state_hash(speed, distance),
           "phase_hash": _phase_hash(lights, real) if lights is not None else None}
    return rec, {k: (speed[k], distance[k]) for k in speed}


def parity_detail_compare(gpu_sd, cpu_sd):
    """Per-vehicle comparison of two {id: (speed, distance)} maps: which vehicles differ at all and the largest relative
    deviation (BASELINE.json's tolerance is 1e-6)."""
    ids = set(gpu_sd) | set(cpu_sd)
    differing, worst = [], 0.0
    for k in ids:
        a, b = gpu_sd.get(k), cpu_sd.get(k)
        if a is None or b is None:
            differing.append(k)
            worst = float("inf")
            continue
        if tuple(a) != tuple(b):
            differing.append(k)
            for x, y in zip(a, b):
                worst = max(worst, abs(x - y) / max(abs(x), abs(y), 1e-12))
    return {"vehicles": len(ids), "vehicles_differing": len(differing), "max_relative_deviation": worst}, differing


def checkpoints_of(steps):
    """Where inside the timed window the parity records are taken (steps counted from the start of the window)."""
    return sorted({max(1, round(steps * i / 4)) for i in (1, 2, 3, 4)})


def tie_neighbourhood(flat, drivables):
    """The drivables around the intersections a tied drivable touches: where two correct engines may differ right after
    an exact-distance tie (the reference's std::sort leaves the pair's order to heap addresses, engine.cpp:480)."""
    L = int(flat["n_lanes"])
    ls, le, li = flat["ll_start_lane"], flat["ll_end_lane"], flat["ll_inter"]
    inters = set()
    for d in drivables:
        if d >= L:
            inters.add(int(li[d - L]))
        else:
            for k in range(len(ls)):
                if ls[k] == d or le[k] == d:
                    inters.add(int(li[k]))
    out = set(int(d) for d in drivables)
    for k in range(len(ls)):
        if int(li[k]) in inters:
            out.update((L + k, int(ls[k]), int(le[k])))
    return out


# ---- BASELINE configs[4] (100x100, ~1 M vehicles) against records the REFERENCE ITSELF produced -------------------------------
# tests/golden/reference_large.json (tests/golden/make_large_goldens.py: the unmodified reference, one thread, stepped from step 0
# on exactly the scale leg's workload).  Running the reference live at this size does not fit a bench run: constructing its
# engine on this network takes ~1 minute and loading a 1.1 GB Archive of 1 M vehicles ~2 more (measured), before its first step.
def scale_golden(scen, n_flows):
    path = os.path.join(ROOT, "tests", "golden", "reference_large.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        g = json.load(f)
    if not g.get("workload", "").startswith("%s " % scen) or ("%d seeded" % n_flows) not in g["workload"]:
        return None
    return g


def scale_record(eng, real=None):
    """What a checkpoint of the 1 M-vehicle run is compared by: vehicle count, every lane's count (array order), average travel
    time, the id-keyed hash of every vehicle's exact (speed, distance), the id-free hash of the multiset of those pairs, and
    (with `real` = ids of the non-virtual intersections) the signal phases."""
    import numpy as np
    arr = eng.get_lane_vehicle_count_array().astype(np.int32)
    speed, distance = eng.get_vehicle_speed(), eng.get_vehicle_distance()
    kin = hashlib.sha256()
    for pair in sorted((float(speed[k]).hex(), float(distance[k]).hex()) for k in speed):
        kin.update(("%s %s\n" % pair).encode())
    rec = {"vehicle_count": eng.get_vehicle_count(), "lane_sum": int(arr.sum()),
           "lane_array_sha256": hashlib.sha256(arr.tobytes()).hexdigest(),
           "state_hash": _state_hash(speed, distance), "kinematics_hash": kin.hexdigest(),
           "average_travel_time": float(eng.get_average_travel_time()).hex()}
    if real is not None:
        ph, rm = eng._tl_state()
        h = hashlib.sha256()
        for k, p, r in sorted(zip(eng.intersection_ids(), ph.tolist(), rm.tolist())):
            if k in real:
                h.update(("%s %d %s\n" % (k, int(p), float(r).hex())).encode())
        rec["phase_hash"] = h.hexdigest()
    return rec


def scale_compare(got, want, ties_here):
    """One checkpoint against the golden record.  Counts, every lane, the average travel time and the signal phases against
    the reference's at every checkpoint.  Every vehicle's exact (speed, distance) — id-keyed and as an id-free multiset —
    against the reference's while no exact-distance tie has happened; afterwards against the CPU twin's, recorded beside it:
    the reference's own order of a tied pair is a function of its unstable global sort, heap addresses and thread timing, and
    the order decides which of two vehicles at one position brakes for the other (tests/golden/make_large_goldens.py)."""
    out = {f: got.get(f) == want[f] for f in ("vehicle_count", "lane_sum", "lane_array_sha256", "average_travel_time", "phase_hash")
           if f in want and f in got}
    out["exact_distance_ties_so_far"] = ties_here
    out["ties_equal_twin"] = ties_here == want.get("twin_tie_events")
    out["every_vehicle_equal_twin"] = (got["state_hash"] == want.get("twin_state_hash") and
                                       got["kinematics_hash"] == want.get("twin_kinematics_hash"))
    out["every_vehicle_equal_reference"] = got["state_hash"] == want["state_hash"] and got["kinematics_hash"] == want["kinematics_hash"]
    need = [v for k, v in out.items() if k not in ("exact_distance_ties_so_far", "every_vehicle_equal_reference")]
    out["equal"] = bool(all(need) and (ties_here > 0 or out["every_vehicle_equal_reference"]))
    return out


def usable_cpus():
    """CPUs this process may actually use at once: the host's cores, the affinity mask, and the container's CFS quota (cgroup
    v2 cpu.max / v1 cpu.cfs_quota_us) — a leg that starts one thread per host core inside a 16-CPU quota is throttled by the
    kernel for most of its run, and measures the throttling."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return n


# ------------------------------------------------------------------------------------------------ CPU legs
def cpu_baseline(cfg, budget_s, threads, state_dump, warmup=0, checkpoints=(), detail_dir=None, gpu_hashes=None):
    """The unmodified reference engine (oracle/_ref) on the host cores, started from EXACTLY the state the GPU engine
    started from: that state is injected through the reference's own Archive JSON format (Engine.load_from_file,
    reference src/engine/archive.cpp:345-550), followed by the same `warmup` untimed steps.  Falls back to the CPU twin
    ("port").  At every step of `checkpoints` (counted from the end of the warm-up, untimed) the record of
    parity_record() is kept; where its (speed, distance) hash differs from the GPU's the per-vehicle values are written
    under `detail_dir`."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if ref_dir not in sys.path:
        sys.path.insert(0, ref_dir)
    kind = "reference"
    try:
        import cityflow_ref
        eng = cityflow_ref.Engine(cfg, threads)
    except Exception:  # not shipped / not built: time the CPU twin instead
        from cityflow_amd import _cityflow
        kind, threads = "port", 1
        eng = _cityflow.Engine._with_backend(cfg, 1, os.path.join(ref_dir, "libcfx_twin.so"))
    t_load = time.perf_counter()
    eng.load_from_file(state_dump)
    t_load = time.perf_counter() - t_load
    for _ in range(warmup):
        eng.next_step()
    start_running = eng.get_vehicle_count()
    veh_steps, steps, records = 0, 0, {}
    last_cp = max(checkpoints) if checkpoints else 0
    dt = 0.0
    while True:
        t0 = time.perf_counter()
        veh_steps += eng.get_vehicle_count()  # vehicles that take the coming step (admissions aside)
        eng.next_step()
        dt += time.perf_counter() - t0
        steps += 1
        if steps in checkpoints:  # untimed
            lights = archive_lights(eng, state_dump + ".lights_t%d.json" % threads)
            rec, per_vehicle = parity_record(eng, lights)
            rec["lights"] = {k: (p, float(r).hex()) for k, (p, r) in lights.items()}
            records[steps] = rec
            if detail_dir and (gpu_hashes is None or gpu_hashes.get(str(steps)) != rec["state_hash"]):
                with open(os.path.join(detail_dir, "cpu_t%d_step%d.json" % (threads, steps)), "w") as f:
                    json.dump(per_vehicle, f)
        if dt > budget_s and steps >= last_cp:
            break
    running = eng.get_vehicle_count()
    time.sleep(0.2)  # reference destructor race: settle before the engine is dropped
    del eng
    return {
        "value": veh_steps / dt, "unit": "vehicle-steps/s", "cores": threads, "kind": kind,
        "steps_per_sec": steps / dt,
        "sample": "%d steps from the state the GPU run's timed region starts from (the same Archive JSON, load %.1f s "
                  "untimed, then the same %d warm-up steps; %d -> %d running vehicles), %.1f s of wall time, %d thread(s) "
                  "of %d host cores (%d usable: affinity / container CPU quota)" % (
                      steps, t_load, warmup, start_running, running, dt, threads, os.cpu_count() or 1, usable_cpus()),
    }, records


def cpu_leg_subprocess(cfg, budget_s, threads, state_dump, warmup, checkpoints, detail_dir, gpu_hashes):
    """One reference leg in its own process, with Vehicle objects at ascending addresses (LD_PRELOAD of
    oracle/_ref/libmonotonic_new.so, oracle/monotonic_new.cpp).  The reference walks its vehicles in std::set<Vehicle*>
    order — by heap address — and its unstable sort of the vehicles that change drivable then leaves vehicles with EXACTLY
    equal distances in an order that depends on those addresses and, with several threads, on which thread finishes first
    (SURVEY.md App. C-6).  One thread + creation-ordered addresses is the reproducible reference; that is what the
    in-run parity check compares with."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    pre = os.path.join(ref_dir, "libmonotonic_new.so")
    size = os.path.join(ref_dir, "vehicle_size.txt")
    if not (os.path.exists(pre) and os.path.exists(size)):
        return None, None
    env = dict(os.environ, LD_PRELOAD=pre, CFX_VEHICLE_SIZE=open(size).read().strip())
    job = os.path.join(detail_dir, "cpu_leg_t%d.job.json" % threads)
    with open(job, "w") as f:
        json.dump({"cfg": cfg, "budget": budget_s, "threads": threads, "dump": state_dump, "warmup": warmup,
                   "checkpoints": list(checkpoints), "detail_dir": detail_dir, "gpu_hashes": gpu_hashes}, f)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", job], env=env, capture_output=True, text=True,
                         timeout=1200)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        sys.stderr.write("[bench] reference leg in its own process failed: %s\n" % out.stderr[-500:])
        return None, None
    d = json.loads(lines[-1])
    d["leg"]["sample"] += "; own process, Vehicle objects at creation-ordered addresses"
    return d["leg"], {int(k): v for k, v in d["records"].items()}


# ------------------------------------------------------------------------------------------------ PMC summaries
def kernel_source_sha():
    """sha256 over the HIP sources the device library is built from: a PMC summary counts only for the kernels it measured."""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cityflow_amd", "csrc", "hip", "*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_names, workload):
    """HBM bytes per launch of the action kernel from the committed rocprofv3 PMC summary (FETCH_SIZE and WRITE_SIZE in
    separate passes, tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on
    gfx950) — only if that summary was taken from THESE kernel sources (it carries their hash) on THIS workload (its
    `workload` tag: "bench" = the headline 30x30 workload, "100x100", ...); otherwise None."""
    import glob
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha") != sha or d.get("workload", "bench") != workload:
            continue
        for name in kernel_names:
            k = next((v for kn, v in sorted(d.get("kernels", {}).items()) if name in kn), None)
            if k:
                return k.get("hbm_bytes_fetch_doubled", k.get("hbm_bytes_raw")), "%s (%s; (2*FETCH_SIZE+WRITE_SIZE)*1024 per launch)" % (
                    os.path.relpath(path, ROOT), d.get("window", ""))
    return None, None


def history_note(roof, eng):
    """Lane::history (kept by default on networks up to 20 k lanes) is taken by trailing blocks of the action launch: say so where
    that launch is priced — their reads and writes are in `traffic` and in the launch's duration, not in the algorithmic bytes."""
    if roof is not None and eng is not None and hasattr(eng, "_keeps_lane_history") and eng._keeps_lane_history():
        roof["also_in_this_launch"] = ("Lane::history of the previous step (reference roadnet.cpp:900-915), one thread per lane in "
                                       "trailing blocks: part of `traffic` and of the duration, not of the 48 B per vehicle")


def roofline_from_profile(prof, vehicle_steps, workload_tag, note, with_traffic=True, symbols=None):
    """The car-following kernel's achieved algorithmic bandwidth from an instrumented run: `prof` = {kernel: (total ms,
    launches)} of cfx_profile_read, `vehicle_steps` = vehicles that took those steps."""
    act_ms, act_n = prof.get("k_action", (0.0, 0))
    if not act_n:
        return None
    vehicles_per_launch = vehicle_steps / float(act_n)
    avg_s = act_ms / act_n / 1e3
    achieved = ACTION_BYTES_PER_VEHICLE * vehicles_per_launch / avg_s / 1e9
    traffic, traffic_src = pmc_traffic(("kr_action", "kw_action", "kl_action", "kd_action", "k_action"), workload_tag) if with_traffic else (None, None)
    return {
        "bound": "hbm", "kernel": (symbols or {}).get("k_action", "k_action"), "profile_slot": "k_action",
        "kernels_by_slot": symbols or None, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_us": avg_s * 1e6, "vehicles_per_launch": vehicles_per_launch,
        "algorithmic_bytes_per_vehicle": ACTION_BYTES_PER_VEHICLE, "measured_over": note,
        "kernel_us_per_step": {k: ms / act_n * 1e3 for k, (ms, n) in prof.items() if n},  # (a kernel that runs in some steps only counts its share)
        "kernel_launches_per_step": {k: n / float(act_n) for k, (ms, n) in prof.items() if n},
        "sum_kernel_ms_per_step": sum(ms for ms, _n in prof.values()) / act_n,
    }


def chunk_medians(roofline, chunks):
    """`chunks` = the instrumented run's kernel times read in parts (cfx_profile_read after every part).  Adds the per-kernel
    median over the parts, and prices the roofline with the MEDIAN of the parts' average launch durations when the plain
    average over all launches is off it by more than a quarter: one launch that the box stretches to milliseconds (round 5:
    one of 100 launches took 76 ms) says nothing about the kernel, and the rocprofv3 summary of the same command
    (profiles/) agrees with the median.  Both figures stay in the line."""
    if not roofline or len(chunks) < 2:
        return
    per = {}
    for part in chunks:
        steps_here = max((n for _ms, n in part.values()), default=0)
        for k, (ms, n) in part.items():
            if n and steps_here:
                per.setdefault(k, []).append(ms / steps_here * 1e3)
    roofline["kernel_us_per_step_median_of_%d_chunks" % len(chunks)] = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
    act = [ms / n * 1e3 for part in chunks for k, (ms, n) in part.items() if k == "k_action" and n]
    if not act:
        return
    act_med = sorted(act)[len(act) // 2]
    # `avg_launch_us`, `achieved` and `frac` stay what they say — the plain average over every instrumented launch; the figures
    # priced with the median of the parts' averages stand beside them under their own names
    roofline["avg_launch_us_median_of_chunk_averages"] = act_med
    roofline["achieved_median_priced"] = ACTION_BYTES_PER_VEHICLE * roofline["vehicles_per_launch"] / (act_med * 1e-6) / 1e9
    roofline["frac_median_priced"] = roofline["achieved_median_priced"] / HBM_PEAK_GBS
    roofline["duration_estimator"] = "average over all instrumented launches"
    if abs(roofline["avg_launch_us"] - act_med) > 0.25 * act_med:
        roofline["outlier_note"] = "an outlier launch moved the plain average (%.1f us) off the median of %d chunk averages (%.1f us)" % (
            roofline["avg_launch_us"], len(act), act_med)


# ------------------------------------------------------------------------------------------------ launching
def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (the reference's counterpart
    is its thread pool, src/engine/engine.cpp:253-270 — here the workers are processes, one per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stderr.write("[bench] --gpus %d without WORLD_SIZE: launching %s\n" % (n, " ".join(cmd[1:9])))
    sys.exit(subprocess.call(cmd, env=env))


def warm_gpu_clocks(eng, tiled):
    """250 ms of plain device work on the engine's stream right before the warm-up steps (cfx_device_spin): after seconds of
    host-only work (JSON load) the GPU sits in its idle power state, and a timed region of a millisecond (20 steps) would
    otherwise be measured on ramping clocks (and whatever the power management does in the first tenths of a second of load
    would land in the windows behind it).  Not simulation work; touches no engine state.  The sustained figure is
    `ms_per_step_200`: a thousand steps behind the timed region."""
    (eng._eng if tiled else eng)._device_spin(250000)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--build-up-steps", type=int, default=BUILD_UP_STEPS, help=argparse.SUPPRESS)
    ap.add_argument("--profile-steps", type=int, default=100, help="instrumented steps for the roofline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the 8-thread cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="reference thread_num (default min(8, host cores))")
    ap.add_argument("--cpu-leg-seconds", type=float, default=6.0, help="budget of each extra leg (1 thread, all host cores)")
    ap.add_argument("--scale-steps", type=int, default=None,
                    help="steps of the 100x100 leg (roofline_at_scale / scale_100x100; default 50 on the GPU, 0 = skip)")
    ap.add_argument("--rl-seconds", type=float, default=5.0, help="budget of the reference leg of the RL loop (0 = skip the RL loop)")
    # test hooks (tests/test_distributed.py drives the N>1 code path on CPU with gloo and the CPU twin)
    ap.add_argument("--scenario", default="grid_30x30", help=argparse.SUPPRESS)
    ap.add_argument("--extra-flows", type=int, default=N_EXTRA_FLOWS, help=argparse.SUPPRESS)
    ap.add_argument("--scale-grid", type=int, default=SCALE_GRID, help=argparse.SUPPRESS)
    ap.add_argument("--scale-flows", type=int, default=SCALE_FLOWS, help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="auto", help=argparse.SUPPRESS)
    ap.add_argument("--backend-lib", default="", help=argparse.SUPPRESS)
    ap.add_argument("--no-halo", action="store_true", help=argparse.SUPPRESS)  # test: as if no halo transport came up
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of one tiled network")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: grow the grid with N (every GPU owns a 30x30 block) instead of tiling the N=1 workload itself")
    ap.add_argument("--tile-block", type=int, default=30, help=argparse.SUPPRESS)
    ap.add_argument("--cfx", default="", help=argparse.SUPPRESS)  # developer: implementation choices, "key=value,key=value"
    args = ap.parse_args()
    if args.scale_steps is None:
        args.scale_steps = 50 if args.backend_lib == "" else 0
    return args


class Job:
    """Rank / world / process group of this run."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.on_gpu = args.backend_lib == ""
        self.dist = None
        self.backend = None
        self.shared_devices = False
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torch.distributed.run "
                             "--nproc-per-node %d), or leave WORLD_SIZE unset and let --gpus start the ranks"
                             % (args.gpus, self.world, args.gpus))
        if self.world > 1:
            import torch
            import torch.distributed as dist
            backend = args.dist_backend
            if self.on_gpu:
                ndev = torch.cuda.device_count()
                if ndev == 0:
                    raise SystemExit("bench.py: no GPU visible (the product path has no CPU fallback)")
                self.shared_devices = ndev < int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
                local_dev = self.local_rank % ndev
                torch.cuda.set_device(local_dev)
                if backend == "auto":  # RCCL refuses two ranks on one device: such a run coordinates over gloo
                    backend = "gloo" if self.shared_devices else "nccl"
                if backend == "nccl":
                    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
                else:
                    dist.init_process_group(backend=backend)
            else:
                backend = "gloo" if backend == "auto" else backend
                dist.init_process_group(backend=backend)
            self.dist, self.backend = dist, backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, values, op):
        """all_reduce of a few doubles over the ranks."""
        if self.dist is None:
            return list(values)
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor(list(values), dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return [float(x) for x in t.tolist()]

    def all_true(self, ok):
        return self.reduce([1.0 if ok else 0.0], "MIN")[0] == 1.0


def make_tiled(job, cfg, rows, cols, backend_lib, notes):
    """One network over the ranks.  Probe the halo transports on this machine before committing to one, best first:
    mailboxes in the receiving GPU's HBM (hipIpc peer memory over xGMI), mailboxes in shared host memory, RCCL send / recv
    of device-resident messages, gloo through host buffers.  A transport counts only if EVERY rank ran a few steps on it
    without an error."""
    from cityflow_amd.tiled import DistributedEngine
    for transport in ("device", "host", "rccl", "gloo"):
        if transport == "rccl" and (not job.on_gpu or job.backend != "nccl"):
            continue
        cand, ok = None, True
        try:
            cand = DistributedEngine(cfg, rows, cols, backend_library=backend_lib, transport=transport)
            for _ in range(20):
                cand.next_step()
            cand.sync()
            cand.local_scalars()  # raises if a device-side halo wait timed out
        except Exception as exc:  # noqa: BLE001 - any failure disqualifies the transport
            ok = False
            notes.append("%s: %s" % (transport, str(exc)[:200]))
        if job.all_true(ok):
            cand.reset(True)  # back to step 0 with the configured seed: the probe leaves no trace
            return cand
        del cand
    return None


def total_scalars(job, eng, tiled):
    """Whole-job scalars: a tiled engine's are summed over the ranks."""
    s = eng.local_scalars() if tiled else eng._scalars()
    if tiled or job.world > 1:
        v = job.reduce([s["vehicle_steps"], s["active_vehicle_count"], s.get("tie_events", 0)], "SUM")
        return {"vehicle_steps": int(v[0]), "active_vehicle_count": int(v[1]), "tie_events": int(v[2]),
                "tie_drivables": []}
    return s


def timed_steps(job, eng, n, detail=None):
    """EXACTLY n steps bracketed by a barrier + device synchronisation on both sides; MAX over ranks.
    `detail` (a dict, filled on this rank): every next_step() call of the window is stamped with the host clock as well (two
    clock reads per step, ~0.1 us) — the slowest call and its index, how long the closing synchronisation waited, and what
    the device library says about its own cfx_step calls (cfx_get_host_stats: slowest call, the step it belonged to, and
    whether it drained the stream, regrew the rings or grew a table).  A window that comes out slower than its neighbours
    then says where the time went: inside one host call (which, and why), or on the device behind calls that all returned
    at once."""
    job.barrier()
    eng.sync()
    single = getattr(eng, "_host_stats", None) if detail is not None else None
    if single:
        single(True)
    calls = []
    t0 = time.perf_counter()
    if detail is None:
        for _ in range(n):
            eng.next_step()
    else:
        t1 = t0
        for _ in range(n):
            eng.next_step()
            t2 = time.perf_counter()
            calls.append(t2 - t1)
            t1 = t2
    t_sync = time.perf_counter()
    eng.sync()
    t_end = time.perf_counter()
    job.barrier()
    if detail is not None and calls:
        worst = max(range(len(calls)), key=lambda i: calls[i])
        detail.update({"ms_per_step": (t_end - t0) / n * 1e3, "worst_next_step_us": round(calls[worst] * 1e6, 1),
                       "worst_next_step_index": worst, "median_next_step_us": round(sorted(calls)[len(calls) // 2] * 1e6, 1),
                       "next_step_calls_over_1ms": sum(1 for c in calls if c > 1e-3),
                       "closing_sync_ms": round((t_end - t_sync) * 1e3, 3)})
        if single:
            hs = single(True)
            detail["device_library"] = {"worst_cfx_step_us": round(hs["worst_step_call_us"], 1), "at_step": hs["worst_step_call_at"],
                                        "cause_bits": hs["worst_step_call_cause"], "cfx_step_calls_over_1ms": hs["calls_over_1ms"],
                                        "ring_regrows_total": hs["ring_regrows_total"], "table_grows_total": hs["table_grows_total"],
                                        "status_queries": hs["status_queries"],
                                        "slowest_next_step_us": round(hs["slowest_next_step"][0], 1), "slowest_next_step_at": hs["slowest_next_step"][1],
                                        "slowest_next_step_parts_us": dict(zip(("phases", "take_records", "tables", "cfx_step", "replay", "spawner_ahead"),
                                                                               [round(x, 1) for x in hs["slowest_next_step"][2]])),
                                        "worst_status_query_us_total_settle_copy_wait": [round(x, 1) for x in hs["worst_status_query_us"]]}
    return job.reduce([time.perf_counter() - t0], "MAX")[0]


def lane_hash16(eng):
    return hashlib.sha256(eng.get_lane_vehicle_count_array().tobytes()).hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ the legs
def gpu_replay(eng, state_dump, warmup, steps, cps, real):
    """The parity side of the GPU engine: the same window again from the same file, with records at the checkpoints (and
    two steps after every exact-distance tie, so that what differs can be located while it is still local)."""
    eng.load_from_file(state_dump)
    for _ in range(warmup):
        eng.next_step()
    ties0 = eng._scalars()["tie_events"]
    records, details, want, tie_steps = {}, {}, set(cps), []
    for s in range(1, steps + 1):
        eng.next_step()
        sc = eng._scalars()
        if sc["tie_events"] - ties0 > len(tie_steps):
            tie_steps.append(s)
            if len(tie_steps) <= 3 and s + 2 <= steps:
                want.add(s + 2)
        if s in want:
            rec, per_vehicle = parity_record(eng, engine_lights(eng), real)
            rec["ties"] = sc["tie_events"] - ties0
            rec["tie_drivables"] = list(sc.get("tie_drivables", []))
            if rec["ties"] and s not in cps:  # a locality checkpoint: where every vehicle is
                vs = eng._vehicle_state()
                rec["_drivable_of"] = dict(zip(eng._vehicle_ids(vs["vid"]), vs["drivable"].tolist()))
            records[s], details[s] = rec, per_vehicle
    return records, details, sorted(want), tie_steps


def judge_parity(cps, gpu_recs, gpu_details, ref_recs, detail_dir, threads, flat):
    """In-run parity, checkpoint by checkpoint.  Exact everywhere while no two vehicles entered a drivable with EXACTLY
    equal distances (cfx_scalars::tie_events; the reference's unstable sort orders such a pair by heap address, so ANY two
    engines — two runs of the reference included — may differ in the vehicles around it from then on).  After a tie the
    counts and the signal phases must still be equal, the vehicles that differ must be few (<= 8 per tie) and, at the
    checkpoint two steps after the tie, lie around the intersections the tied drivable touches."""
    rows, ok, excused = [], True, False
    for s in cps:
        g, r = gpu_recs.get(s), (ref_recs or {}).get(s)
        if g is None or r is None:
            rows.append({"step": s, "compared": False})
            ok = False
            continue
        counts = g["vehicles"] == r["vehicles"] and g["lane_hash"] == r["lane_hash"]
        phases = (g["phase_hash"] == _phase_hash({k: (p, float.fromhex(x)) for k, (p, x) in r["lights"].items()}, flat["real"])
                  if "lights" in r else None)
        positions, differing = None, []
        if g["state_hash"] == r["state_hash"]:
            positions = {"vehicles": g["vehicles"], "vehicles_differing": 0, "max_relative_deviation": 0.0}
        else:
            pv = os.path.join(detail_dir, "cpu_t%d_step%d.json" % (threads, s))
            if os.path.exists(pv):
                with open(pv) as f:
                    positions, differing = parity_detail_compare(gpu_details[s], json.load(f))
        exact = bool(positions and positions["vehicles_differing"] == 0)
        row = {"step": s, "compared": True, "lane_counts_and_vehicle_count_equal": bool(counts), "signal_phases_equal": phases,
               "positions_bit_exact": exact, "positions": positions, "exact_distance_ties_so_far": g["ties"]}
        good = bool(counts) and phases is not False and positions is not None
        if good and not exact:
            if g["ties"] == 0:
                good = False
            else:
                excused = True
                good = positions["vehicles_differing"] <= 8 * g["ties"]
                if good and "_drivable_of" in g:  # the locality checkpoint
                    near = tie_neighbourhood(flat, g["tie_drivables"])
                    away = [k for k in differing if g["_drivable_of"].get(k, -1) not in near]
                    row["differing_vehicles_away_from_the_tie"] = len(away)
                    good = not away
        row["ok"] = good
        ok = ok and good
        rows.append(row)
    return ok, excused, rows


def _rl_loop(e, n, mode, offset, n_inter, real_ids):
    """n iterations of: a phase for every signal, one step, the per-lane vehicle counts; returns iterations per second."""
    import numpy as np
    t0 = time.perf_counter()
    for s in range(n):
        ph = ((offset + s) // 10) % 8
        if mode == "array":
            e.set_tl_phases(np.full(n_inter, ph, dtype=np.int32))
            e.next_step()
            obs = e.get_lane_vehicle_count_array()
        else:
            for iid in real_ids:
                e.set_tl_phase(iid, ph)
            e.next_step()
            obs = e.get_lane_vehicle_count()
    if hasattr(e, "sync"):
        e.sync()
    assert len(obs) > 0
    return n / (time.perf_counter() - t0)


def rl_loop_leg(job, args, cfg, workdir, state_dump):
    """RL-style use of the headline network (BASELINE configs[4]'s second half): every step set the phase of
    every signal, step, read the per-lane vehicle counts (reference calls: engine.cpp:628-634, 719-725).  Three ways: the
    array API of this engine, the reference-style dict API of this engine, and the reference engine itself (8 threads, dict
    API — the only one it has), all from the same state."""
    import numpy as np
    from cityflow_amd import _cityflow
    rl_cfg = with_config(cfg, "rl", rlTrafficLight=True)
    eng = _cityflow.Engine(rl_cfg, 1) if job.on_gpu else _cityflow.Engine._with_backend(rl_cfg, 1, args.backend_lib)
    eng.load_from_file(state_dump)
    ids = eng.intersection_ids()
    virt = eng._flat_net()["inter_virtual"]
    real_ids = [iid for i, iid in enumerate(ids) if not virt[i]]
    n_inter = len(ids)

    def loop(e, n, mode, offset):
        return _rl_loop(e, n, mode, offset, n_inter, real_ids)

    loop(eng, 20, "array", 0)  # warm-up
    out = {"network": "%s, rlTrafficLight, %d signals set and %d lane counts read every step" % (args.scenario, len(real_ids), len(eng.lane_ids())),
           "running_vehicles": eng.get_vehicle_count(),
           "array_api_steps_per_sec": loop(eng, 200, "array", 20),
           "dict_api_steps_per_sec": loop(eng, 20, "dict", 220)}
    del eng
    if args.rl_seconds > 0:
        ref_dir = os.path.join(ROOT, "oracle", "_ref")
        if ref_dir not in sys.path:
            sys.path.insert(0, ref_dir)
        try:
            import cityflow_ref
            threads = min(8, usable_cpus())
            ref = cityflow_ref.Engine(rl_cfg, threads)
            ref.load_from_file(state_dump)
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.rl_seconds:
                loop(ref, 1, "dict", n)
                n += 1
            out["reference_dict_api_steps_per_sec"] = n / (time.perf_counter() - t0)
            out["reference_threads"] = threads
            time.sleep(0.2)
            del ref
        except ImportError:
            out["reference_dict_api_steps_per_sec"] = None
    return out


def scale_leg(job, args, n_steps):
    """The bandwidth-regime network (BASELINE configs[4]: 100x100, ~1 M vehicles).  N = 1: one engine, instrumented —
    `roofline_at_scale`.  N > 1: the same network cut into the same rows x cols tiles — `scale_100x100`; the ratio of the
    two lines' steps/s is the strong-scaling figure at the size where tiling is expected to pay."""
    from cityflow_amd import _cityflow
    scen = "gen_%dx%d" % (args.scale_grid, args.scale_grid)
    workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_bench_scale")
    if job.rank == 0:
        build_workload(workdir, 0, scenario=scen, n_extra=args.scale_flows)
    job.barrier()
    cfg = build_workload(workdir, 0, scenario=scen, n_extra=args.scale_flows)
    tiled = job.world > 1
    notes = []
    if tiled:
        rows, cols = tile_grid(job.world)
        eng = make_tiled(job, cfg, rows, cols, args.backend_lib, notes)
        if eng is None:
            return {"error": "no halo transport: " + "; ".join(notes)}
    else:
        eng = _cityflow.Engine(cfg, 1) if job.on_gpu else _cityflow.Engine._with_backend(cfg, 1, args.backend_lib)
    # parity at this size: checkpoints of tests/golden/reference_large.json the run passes outside its timed region
    golden = None if tiled else scale_golden(scen, args.scale_flows)
    want = {int(k): v for k, v in golden["checkpoints"].items()} if golden else {}
    parity_cps, at = {}, 0
    real = None
    if golden:
        flat = eng._flat_net()
        real = {k for k, v in zip(eng.intersection_ids(), flat["inter_virtual"]) if not v}

    def advance(n):
        nonlocal at
        for _ in range(n):
            eng.next_step()
            at += 1
            if at in want and at > 300:  # (records of a million vehicles through the dict getters: ~10 s each)
                got = scale_record(eng, real if "phase_hash" in want[at] else None)
                parity_cps[at] = scale_compare(got, want[at], int(eng._scalars()["tie_events"]))

    advance(args.build_up_steps + 10)
    eng.sync()
    sc0 = total_scalars(job, eng, tiled)
    elapsed = timed_steps(job, eng, n_steps)
    at += n_steps
    if at in want:  # the end of the timed region itself
        parity_cps[at] = scale_compare(scale_record(eng, real if "phase_hash" in want[at] else None), want[at],
                                       int(eng._scalars()["tie_events"]))
    sc1 = total_scalars(job, eng, tiled)
    veh_steps = sc1["vehicle_steps"] - sc0["vehicle_steps"]
    out = {"workload": "%s (generator-format grid, --tlPlan, interval 1.0) + %d seeded interior flows; state after %d steps"
                       % (scen, args.scale_flows, args.build_up_steps + 10),
           "running_vehicles": sc1["active_vehicle_count"], "steps": n_steps, "ms_per_step": elapsed / n_steps * 1e3,
           "steps_per_sec": n_steps / elapsed, "vehicle_steps_per_sec": veh_steps / elapsed,
           "lane_count_hash_end": lane_hash16(eng)}
    if tiled:
        out["parallelism"] = "tiles %dx%d + halo" % tile_grid(job.world)
        out["halo"] = eng.halo_transport()
        return out
    out["layout"] = eng._layout()
    if job.on_gpu:
        eng._profile_enable(True)
        for _ in range(3):
            eng.next_step()
        at += 3 + n_steps  # (and the instrumented steps below)
        eng._profile_read()
        s0 = eng._scalars()
        prof, chunks = {}, []
        n_chunks = 5 if n_steps >= 25 else 1
        for ci in range(n_chunks):
            for _ in range(n_steps // n_chunks + (n_steps % n_chunks if ci == n_chunks - 1 else 0)):
                eng.next_step()
            part = eng._profile_read()
            chunks.append(part)
            prof = {k: (prof.get(k, (0.0, 0))[0] + ms, prof.get(k, (0.0, 0))[1] + n) for k, (ms, n) in part.items()}
        eng._profile_enable(False)
        s1 = eng._scalars()
        roof = roofline_from_profile(prof, s1["vehicle_steps"] - s0["vehicle_steps"], scen,
                                     "%d instrumented steps" % n_steps, symbols=eng._profile_symbols())
        chunk_medians(roof, chunks)
        history_note(roof, eng)
        if args.rl_seconds > 0:
            # BASELINE configs[4] as an RL agent drives it: the same state under rlTrafficLight, every signal set, one step
            # and the per-lane counts read, every iteration (array API)
            rl = _cityflow.Engine(with_config(cfg, "rl", rlTrafficLight=True), 1)
            rl.load(eng.snapshot())
            n_inter = len(rl.intersection_ids())
            _rl_loop(rl, 10, "array", 0, n_inter, None)
            out["rl_loop"] = {"signals_set_per_step": n_inter, "lane_counts_read_per_step": len(rl.lane_ids()),
                              "array_api_steps_per_sec": _rl_loop(rl, 60, "array", 10, n_inter, None)}
            del rl
        later = [s for s in sorted(want) if s > at]
        if later and later[0] - at <= 64:  # the next checkpoint behind the instrumented region
            advance(later[0] - at)
        # a longer window behind everything else (the 50-step region above starts on an idle device and is ~7 ms long): 200
        # more free-running steps, timed the same way — reported beside `ms_per_step`, never instead of it
        eng.sync()
        s200a = eng._scalars()
        t200 = timed_steps(job, eng, 200)
        s200b = eng._scalars()
        out["ms_per_step_200"] = t200 / 200 * 1e3
        out["vehicle_steps_per_sec_200"] = (s200b["vehicle_steps"] - s200a["vehicle_steps"]) / t200
        if golden:
            roof_parity = {
                "against": "records of the unmodified reference engine (tests/golden/reference_large.json; %d thread(s), Vehicle "
                           "objects at %s addresses), same workload from step 0" % (golden.get("reference_threads", 0),
                                                                                   golden.get("vehicle_addresses", "heap")),
                "kind": "reference (golden records)",
                "checkpoints": {str(s): r for s, r in sorted(parity_cps.items())},
                "timed_region": "steps %d..%d" % (args.build_up_steps + 11, args.build_up_steps + 10 + n_steps),
                "all_equal": bool(parity_cps) and all(r["equal"] for r in parity_cps.values()),
                "checked": "vehicle count, every lane's count, average travel time, signal phases where recorded: against the "
                           "reference at every checkpoint; every vehicle's exact (speed, distance), id-keyed and as a multiset: "
                           "against the reference until the first exact-distance tie, against the CPU twin's record afterwards"}
            out["parity"] = roof_parity
        if roof:
            roof["config"] = out
            roof["parity"] = out.get("parity")
            return roof
    return {"config": out}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--cpu-leg":  # cpu_leg_subprocess()
        j = json.load(open(sys.argv[2]))
        leg, recs = cpu_baseline(j["cfg"], j["budget"], j["threads"], j["dump"], j["warmup"], j["checkpoints"], j["detail_dir"],
                                 j["gpu_hashes"])
        print(json.dumps({"leg": leg, "records": recs}), flush=True)
        os._exit(0)  # (reference destructor race)
    args = parse_args()
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)  # `kill -USR1 <pid>` prints where a rank is (a hung collective)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    # stdout carries ONE line, the result.  Libraries that print from C++ (gloo announces every group it connects on fd 1)
    # would put theirs in front of it: from here on fd 1 is stderr, and the line goes to the descriptor saved here.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    job = Job(args)
    rank, world, on_gpu = job.rank, job.world, job.on_gpu
    strong = not args.weak

    from cityflow_amd import _cityflow

    tiled = world > 1 and not args.replicas
    tiling_failed = False
    eng, halo_notes, rows, cols = None, [], 1, 1
    if tiled:
        rows, cols = tile_grid(world)
        workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_bench_tiled")
        if strong:  # the very workload of the N=1 run, cut into rows x cols tiles
            if rank == 0:
                build_workload(workdir, seed=0, scenario=args.scenario, n_extra=args.extra_flows)
            job.barrier()
            cfg = build_workload(workdir, seed=0, scenario=args.scenario, n_extra=args.extra_flows)
        else:
            if rank == 0:  # one node: every rank reads the files rank 0 wrote
                build_tiled_workload(workdir, rows, cols, args.tile_block, args.extra_flows)
            job.barrier()
            cfg = os.path.join(workdir, "gen_%dx%d" % (args.tile_block * rows, args.tile_block * cols), "config_bench.json")
        if args.cfx:  # (developer: implementation choices for the tiles too, e.g. ringLanesPerWave=70000 = the dense tiles of rounds 2-5)
            if rank == 0:
                with_config(cfg, "cfx", cfx={k: (int(v) if v.lstrip("-").isdigit() else v)
                                              for k, v in (kv.split("=") for kv in args.cfx.split(","))})
            job.barrier()
            cfg = cfg.replace(".json", "_cfx.json")
        if args.no_halo:
            halo_notes.append("--no-halo")
        else:
            eng = make_tiled(job, cfg, rows, cols, args.backend_lib, halo_notes)
        if eng is None:
            # No halo transport works between these ranks (every rank agrees: make_tiled).  Rather than no line at all: N
            # independent replicas of the N = 1 workload — weak scaling, no exchange — and the line says so
            # (`config.parallelism`, `config.halo_probe_failures`).
            sys.stderr.write("bench.py: no halo transport works on this machine (%s): running independent replicas\n"
                             % "; ".join(halo_notes))
            tiled, tiling_failed = False, True
            args.replicas = True
            rows, cols = 1, 1
    if not tiled:
        workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_bench_rank%d" % rank)
        cfg = build_workload(workdir, seed=rank, scenario=args.scenario, n_extra=args.extra_flows)
        if args.cfx:
            cfg = with_config(cfg, "cfx", cfx={k: (int(v) if v.lstrip("-").isdigit() else v)
                                                for k, v in (kv.split("=") for kv in args.cfx.split(","))})
        if on_gpu:
            eng = _cityflow.Engine(cfg, 1)  # HIP engine on device LOCAL_RANK; raises if the extension/GPU is missing
            assert eng.backend_name() == "hip-gfx950"
        else:
            eng = _cityflow.Engine._with_backend(cfg, 1, args.backend_lib)
    backend_name = eng._eng.backend_name() if tiled else eng.backend_name()

    # ---- the workload state (not a warm-up, and not optional), written to the file every engine of this run starts from
    for _ in range(args.build_up_steps):
        eng.next_step()
    eng.sync()
    want_checks = args.cpu_seconds > 0 and not args.replicas and (world == 1 or strong)
    state_dump = os.path.join(workdir, "state_at_build_up.json")
    if want_checks:
        arch = eng.snapshot()  # (tiled: assembled from one part per rank, every rank takes part)
        if rank == 0:
            arch.dump(state_dump)
        del arch
        job.barrier()
        # The GPU engine re-reads the file too, so that vehicles are numbered in archive order on both sides.  (Where two
        # vehicles enter a lane with EXACTLY equal distances in one step the reference's std::sort leaves their order to
        # the order of its Vehicle objects; an engine that kept running and one that was restored from an archive may break
        # such a tie differently and would then drift apart although both are right.)
        eng.load_from_file(state_dump)
    if on_gpu:
        warm_gpu_clocks(eng, tiled)
    for _ in range(args.warmup):
        eng.next_step()
    eng.sync()
    sc0 = total_scalars(job, eng, tiled)
    host0 = eng._eng._host_seconds() if tiled else None

    # (every call of the timed region is stamped with the host clock too — two clock reads per step beside a 40 us device step
    # the host runs ahead of: a region that comes out slow says where, `timed_region_detail`)
    timed_detail = {}
    elapsed = timed_steps(job, eng, args.steps, timed_detail)

    host1 = eng._eng._host_seconds() if tiled else None  # this rank's host time inside the timed region
    sc1 = total_scalars(job, eng, tiled)
    veh_steps = sc1["vehicle_steps"] - sc0["vehicle_steps"]
    run0, run1 = sc0["active_vehicle_count"], sc1["active_vehicle_count"]
    ties_in_window = sc1.get("tie_events", 0) - sc0.get("tie_events", 0)
    lane_hash_end = lane_hash16(eng)  # (every rank takes part in a tiled run's getter)
    end_record = None
    if want_checks and tiled:  # what the single engine of rank 0 is compared with below (collective getters)
        sp, di = eng.get_vehicle_speed(), eng.get_vehicle_distance()
        end_arch = eng.snapshot()
        if rank == 0:
            end_arch.dump(state_dump + ".tiled_end.json")
            with open(state_dump + ".tiled_end.json") as f:
                lights = json.load(f)["trafficLights"]
            end_record = {"vehicles": run1, "lane_hash16": lane_hash_end, "state_hash": _state_hash(sp, di),
                          "lights": {k: (int(x["curPhaseIndex"]), float(x["remainDuration"])) for k, x in lights.items()}}
        del end_arch

    # ---- roofline leg: instrumented continuation of the same run (HIP events on the engine's stream), RIGHT BEHIND the timed
    #      region — on the vehicles the timed region stepped (until round 6's second half it followed the sustained windows:
    #      a thousand steps later the workload holds 15 % fewer vehicles, the launch takes as long, and the fraction read
    #      0.034 where the same kernel on the timed region's state reads 0.039).  Tiled runs: every
    #      rank keeps stepping (the tiles are coupled), rank 0 instruments its own tile, halo kernels included.
    # (not where several ranks share one device: their kernels take turns on it, so per-kernel times say nothing about a tile,
    # and the instrumented rank — every launch bracketed by events — falls behind its neighbour's bounded halo wait)
    roofline = None
    if on_gpu and args.profile_steps > 0 and (rank == 0 or tiled) and not (tiled and job.shared_devices):
        scp0 = eng.local_scalars() if tiled else eng._scalars()
        if rank == 0:
            (eng._eng._profile_enable(0, True) if tiled else eng._profile_enable(True))
            for _ in range(0 if tiled else 5):  # the first instrumented launches (event pool, lazy loads) are not the kernels
                eng.next_step()
            if not tiled:
                eng._profile_read()
                scp0 = eng._scalars()
        # (read in four chunks: the totals give the averages the roofline is priced with; the per-kernel MEDIAN over the chunks
        # is reported beside them — one launch that a box hiccup stretches to milliseconds moves a 100-launch average by tens of
        # microseconds, and says nothing about the kernel)
        prof, chunks = None, []
        n_chunks = 4 if args.profile_steps >= 40 else 1
        for ci in range(n_chunks):
            n_here = args.profile_steps // n_chunks + (args.profile_steps % n_chunks if ci == n_chunks - 1 else 0)
            for _ in range(n_here):
                eng.next_step()
            if rank == 0:
                part = eng._eng._profile_read(0) if tiled else eng._profile_read()
                chunks.append(part)
                if prof is None:
                    prof = {k: (ms, n) for k, (ms, n) in part.items()}
                else:
                    prof = {k: (prof.get(k, (0.0, 0))[0] + ms, prof.get(k, (0.0, 0))[1] + n) for k, (ms, n) in part.items()}
        if rank == 0:
            (eng._eng._profile_enable(0, False) if tiled else eng._profile_enable(False))
        eng.sync()
        scp1 = eng.local_scalars() if tiled else eng._scalars()
        if rank == 0:
            roofline = roofline_from_profile(
                prof, scp1["vehicle_steps"] - scp0["vehicle_steps"], "bench" if args.scenario == "grid_30x30" else args.scenario,
                "%d instrumented steps following the timed region%s" % (args.profile_steps, " (rank 0's tile)" if tiled else ""),
                with_traffic=not tiled, symbols=None if tiled else eng._profile_symbols())
            chunk_medians(roofline, chunks)
            history_note(roofline, None if tiled else eng)

    # ---- longer windows behind the driver-shaped one (a 20-step region is under a millisecond of device time): five more
    #      windows of 200 steps of the same run, timed the same way; reported beside the headline, never instead of it.
    # All windows are reported, their MEAN is the named figure — not the best one: a stall inside a window is part of what
    # a caller sees.  Every window carries where its slowest call was and what the device library did in it
    # (`sustained_windows`), so that a window that disagrees with its neighbours names its cause in the line itself.
    window_details = []
    if args.steps >= 200:
        ms_200_windows = [elapsed / args.steps * 1e3]
    else:
        ms_200_windows = []
        for _ in range(5):
            det = {}
            ms_200_windows.append(timed_steps(job, eng, 200, det) / 200 * 1e3)
            window_details.append(det)
    ms_per_step_200 = sum(ms_200_windows) / len(ms_200_windows)
    ms_per_step_200_median = sorted(ms_200_windows)[len(ms_200_windows) // 2]

    # ---- in-run parity and the CPU baseline (rank 0; a tiled run is compared with ONE engine on rank 0's device)
    cpu, legs, parity_in_run, parity_excused, parity_detail = None, None, None, None, None
    if want_checks and rank == 0:
        threads = args.cpu_threads or min(8, usable_cpus())
        cps = checkpoints_of(args.steps)
        detail_dir = tempfile.mkdtemp(prefix="parity_", dir=workdir)
        if tiled:
            single = _cityflow.Engine(cfg, 1) if on_gpu else _cityflow.Engine._with_backend(cfg, 1, args.backend_lib)
        else:
            single = eng
        flat = single._flat_net()
        flat["real"] = {k for k, v in zip(single.intersection_ids(), flat["inter_virtual"]) if not v}
        gpu_recs, gpu_details, all_cps, tie_steps = gpu_replay(single, state_dump, args.warmup, args.steps, cps, flat["real"])
        replay_end = {"vehicles": gpu_recs[args.steps]["vehicles"], "lane_hash16": lane_hash16(single)}
        timed_equals_replay = replay_end["lane_hash16"] == lane_hash_end and replay_end["vehicles"] == run1
        if tiled:  # tiled == single engine, bit for bit (both break ties by vehicle number)
            end_record["phase_hash"] = _phase_hash(end_record.pop("lights"), flat["real"])
            phases_same = end_record["phase_hash"] == _phase_hash(engine_lights(single), flat["real"])
            same = end_record["state_hash"] == gpu_recs[args.steps]["state_hash"] and timed_equals_replay and phases_same
            parity_detail = {"tiled_vs_single_engine": {
                "after_steps": args.warmup + args.steps, "same_lane_counts_and_vehicle_count": timed_equals_replay,
                "same_speed_and_distance_of_every_vehicle": end_record["state_hash"] == gpu_recs[args.steps]["state_hash"],
                "same_signal_phases": phases_same, "all_equal": same}}
        gpu_hashes = {str(s): r["state_hash"] for s, r in gpu_recs.items()}
        cpu, par8 = cpu_baseline(cfg, args.cpu_seconds, threads, state_dump, args.warmup, all_cps, detail_dir, gpu_hashes)
        legs, ref_recs, parity_threads = [], par8, threads
        for t in sorted({1, usable_cpus()} - {threads}):  # (one thread; every CPU the container may use at once)
            if args.cpu_leg_seconds > 0 and cpu["kind"] == "reference":
                if t == 1:  # the reproducible reference (see cpu_leg_subprocess): timing leg and parity check in one
                    leg, recs = cpu_leg_subprocess(cfg, args.cpu_leg_seconds, 1, state_dump, args.warmup, all_cps, detail_dir, gpu_hashes)
                    if leg is not None:
                        legs.append(leg)
                        if recs:
                            ref_recs, parity_threads = recs, 1
                        continue
                legs.append(cpu_baseline(cfg, args.cpu_leg_seconds, t, state_dump, args.warmup)[0])
        ok, parity_excused, cp_rows = judge_parity(all_cps, gpu_recs, gpu_details, ref_recs, detail_dir, parity_threads, flat)
        parity_in_run = bool(ok and timed_equals_replay and (not tiled or parity_detail["tiled_vs_single_engine"]["all_equal"]))
        parity_detail = dict(parity_detail or {}, **{
            "against": "%s, %d thread(s)" % (cpu["kind"], parity_threads),
            "window": "%d warm-up + %d steps from the Archive file every engine of the run starts from" % (args.warmup, args.steps),
            "checkpoints": cp_rows, "exact_distance_ties_in_window": gpu_recs[args.steps]["ties"],
            "tie_steps": tie_steps,
            "timed_region_equals_replay": timed_equals_replay,
            "cpu_%d_threads_state_hashes_equal_gpu" % threads: {str(s): (par8.get(s, {}).get("state_hash") == gpu_hashes.get(str(s)))
                                                                 for s in all_cps} if par8 else None,
            "checked": "per-lane vehicle counts, vehicle count, signal phases (index and remaining duration), every vehicle's "
                       "(speed, distance) bit for bit, at every checkpoint",
            "rule": "parity_in_run = every checkpoint equal in everything; after an exact-distance tie (the reference's own "
                    "result is then address-dependent) counts and phases must still be equal, <= 8 vehicles per tie may "
                    "differ and, two steps after the tie, only around the tied drivable's intersections — reported as "
                    "parity_excused_by_ties"})
        if tiled:
            del single

    # ---- the RL loop on the headline network and the bandwidth-regime network
    rl_loop = None
    if rank == 0 and world == 1 and args.rl_seconds > 0 and want_checks:
        rl_loop = rl_loop_leg(job, args, cfg, workdir, state_dump)
    halo_name = eng.halo_transport() if tiled else None
    # what every rank ended on, so that the first run across several physical GPUs describes itself: transport, the physical
    # device of its tile (PCI bus id), and whether the RCCL group came up with all ranks
    halo_by_rank, rccl_ranks = None, None
    if tiled and job.dist is not None:
        mine = {"rank": rank, "transport": eng.transport, "device": list(eng._eng.device_identities())}
        parts = [None] * world
        job.dist.all_gather_object(parts, mine)
        halo_by_rank = parts
        rccl_ranks = job.dist.get_world_size() if job.backend == "nccl" else 0
    layout = eng._eng._layout() if tiled else eng._layout()
    n_lanes = len(eng.lane_ids())
    del eng
    scale = None
    if args.scale_steps > 0 and not args.replicas and (world == 1 or strong):
        scale = scale_leg(job, args, args.scale_steps)

    if rank == 0:
        out = {
            "metric": "vehicle_steps_per_sec", "value": veh_steps / elapsed, "unit": "vehicle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_200": ms_per_step_200, "ms_per_step_200_windows": ms_200_windows,
            "ms_per_step_200_median": ms_per_step_200_median, "sustained_windows": window_details or None, "timed_region_detail": timed_detail or None,
            "higher_is_better": True, "scaling": None if world == 1 else ("strong" if (tiled and strong) else "weak"),
            # one tiled network was asked for and no halo transport came up: the line below is N independent replicas, NOT a
            # multi-GPU run of one network (also in config.parallelism / config.halo_probe_failures)
            "tiling_failed": bool(tiling_failed),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "steps_per_sec": args.steps * (1 if tiled else world) / elapsed,
            "config": {
                "workload": ("%s + %d seeded interior flows cut into %dx%d tiles, one tile per GPU, one-lane ghost halo "
                             "exchanged every step" % (args.scenario, args.extra_flows, rows, cols)) if (tiled and strong) else
                            ("grid_%dx%d (generator-format grid, --tlPlan, interval 1.0) + %d seeded interior flows "
                             "(1 veh / %.0f s each until t=%d s); one network, %dx%d tiles of %dx%d intersections, one "
                             "tile per GPU, one-lane ghost halo exchanged every step" % (
                                 args.tile_block * rows, args.tile_block * cols, args.extra_flows * world, EXTRA_INTERVAL,
                                 EXTRA_END, rows, cols, args.tile_block, args.tile_block)) if tiled else (
                            "%s (reference generator, --tlPlan, interval 1.0) + %d seeded interior flows "
                            "(1 veh / %.0f s each until t=%d s); %s" % (
                                args.scenario, args.extra_flows, EXTRA_INTERVAL, EXTRA_END,
                                "one replica per GPU" if world > 1 else "single engine")),
                "state": "network state after %d simulated seconds of demand build-up (untimed input construction, written to "
                         "an Archive file every engine of the run starts from), then %d warm-up steps" % (args.build_up_steps, args.warmup),
                "running_vehicles_start": run0, "running_vehicles_end": run1,
                "lanes": n_lanes,
                "lane_count_hash_end": lane_hash_end,
                "backend": backend_name,
                "process_group": job.backend,
                "ranks_share_devices": job.shared_devices if world > 1 else None,
                "halo": halo_name,
                "halo_by_rank": halo_by_rank,
                "rccl_ranks": rccl_ranks,
                "layout": layout,
                "cfx": args.cfx or None,
                "halo_probe_failures": halo_notes or None,
                "host_us_per_step": ({"spawner": round((host1[0] - host0[0]) / args.steps * 1e6, 1),
                                      "submit": round((host1[1] - host0[1]) / args.steps * 1e6, 1)} if tiled else None),
                "parallelism": ("tiles %dx%d + halo" % (rows, cols)) if tiled else (
                    ("replica x%d" % world + (" (one tiled network was asked for: no halo transport came up)" if tiling_failed else ""))
                    if world > 1 else "1 gpu"),
            },
            "roofline": roofline,
            "roofline_at_scale": scale if world == 1 else None,
            "scale_100x100": scale if world > 1 else None,
            "cpu_baseline": cpu,
            "cpu_baseline_legs": legs,
            "parity_in_run": parity_in_run,
            "parity_excused_by_ties": parity_excused,
            "parity": parity_detail,
            "rl_loop": rl_loop,
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    job.barrier()
    sys.stdout.flush()
    sys.stderr.flush()
    if job.dist is not None:
        # every rank is past the last barrier and the line is out: leave without the process group's teardown (gloo's
        # occasionally aborts a rank on the way out, which a launcher reports as a failed run)
        os._exit(0)


if __name__ == "__main__":
    main()
