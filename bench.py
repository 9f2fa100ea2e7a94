#!/usr/bin/env python
"""Benchmark of the hot path (one `next_step()` of the MI355X-native CityFlow engine = one "step").

Workload (N=1): BASELINE.json configs[2] — the reference generator's 30x30 grid (cityflow_amd/data/scenarios/
grid_30x30, produced by /root/reference/tools/generator) with ~100k concurrently running vehicles.  The stock
generator's demand never gets near 100k, so 3000 seeded interior-origin flows (one vehicle
every 6 s each, for the first 240 s) are added on top of the 120 stock flows.  The workload is the network at
simulated time t = 300 s: building that state up (BUILD_UP_STEPS untimed steps, ~97k vehicles running at the end) is
part of constructing the input, like loading a dataset — it does NOT depend on --warmup, so a short warm-up still
measures the named workload.  Then W warm-up steps, then K timed steps.  All inputs are resident in HBM when the timed
region starts; per step the host only uploads the handful of spawn records of that step (host-side mt19937 stream).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

N>1 (default): ONE road network tiled over the N GPUs: the grid grows with N — every rank owns a
30x30 block of intersections of a (30*rows)x(30*cols) grid (N=2: 1x2, 4: 2x2, 8: 2x4) with the same demand per
block as the N=1 workload, so per-GPU work is fixed ("scaling": "weak") — and the tiles exchange their one-lane
ghost halo every step (cityflow_amd.tiled.DistributedEngine).  `value` = vehicle-steps of all tiles / the slowest
rank's time.  `--replicas` runs N independent copies of the N=1 workload instead (no exchange).

Output: ONE JSON line on rank 0 (see README of the task contract) with two extra objects:
  roofline      car-following kernel (k_action): algorithmic bytes (48 B per running vehicle)
                / average launch duration measured with HIP events on the engine's stream (the dispatch's own
                start / stop events, hipExtLaunchKernel: the durations a rocprofv3 kernel trace reports)
  cpu_baseline  the unmodified reference engine (oracle/_ref, prebuilt) timed on this box's host cores for a bounded
                number of steps starting from the very state the GPU run had at the start of its timed region
                (injected through the reference's Archive JSON); falls back to the CPU twin (kind "port")
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured achievable)
ACTION_BYTES_PER_VEHICLE = 48  # algorithmic bytes of the car-following (get-action) kernel

BUILD_UP_STEPS = 300           # simulated seconds of demand that define the workload state (see the module docstring)
N_EXTRA_FLOWS = 3000
EXTRA_INTERVAL = 6.0
EXTRA_END = 240


def build_workload(workdir, seed, scenario="grid_30x30", n_extra=N_EXTRA_FLOWS):
    from cityflow_amd import scenarios
    generated = scenario.startswith("gen_")  # gen_RxC: a generator-format grid of any size (developer runs)
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
        with open(path, "w") as f:
            json.dump(cfg, f)
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


def _lane_hash(counts):
    import hashlib
    return hashlib.sha256(json.dumps(sorted(counts.items())).encode()).hexdigest()


def _state_hash(speed, distance):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(speed):
        h.update(("%s %s %s\n" % (k, float(speed[k]).hex(), float(distance[k]).hex())).encode())
    return h.hexdigest()


def parity_detail_compare(gpu_sd, cpu_sd):
    """Per-vehicle comparison of two {id: (speed, distance)} maps: how many vehicles differ at all and the largest relative
    deviation (BASELINE.json's tolerance is 1e-6)."""
    ids = set(gpu_sd) | set(cpu_sd)
    differing, worst = 0, 0.0
    for k in ids:
        a, b = gpu_sd.get(k), cpu_sd.get(k)
        if a is None or b is None:
            differing += 1
            worst = float("inf")
            continue
        if a != b:
            differing += 1
            for x, y in zip(a, b):
                worst = max(worst, abs(x - y) / max(abs(x), abs(y), 1e-12))
    return {"vehicles": len(ids), "vehicles_differing": differing, "max_relative_deviation": worst}


def parity_record(eng):
    """What the in-run parity check compares after the same number of steps from the same state: per-lane vehicle counts,
    vehicle count, and every running vehicle's exact (speed, distance) bits."""
    speed, distance = eng.get_vehicle_speed(), eng.get_vehicle_distance()
    rec = {"vehicles": eng.get_vehicle_count(), "lane_hash": _lane_hash(eng.get_lane_vehicle_count()),
           "state_hash": _state_hash(speed, distance)}
    return rec, {k: (speed[k], distance[k]) for k in speed}


def cpu_baseline(cfg, budget_s, threads, state_dump, parity_steps=0):
    """The unmodified reference engine (oracle/_ref) on the host cores, started from EXACTLY the state the GPU engine
    had when its timed region began: that state is injected through the reference's own Archive JSON format
    (Engine.load_from_file, reference src/engine/archive.cpp:345-550).  Falls back to the CPU twin ("port").
    With parity_steps > 0 the record of parity_record() after exactly that many steps is returned as well."""
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
    start_running = eng.get_vehicle_count()
    veh_steps, steps, parity = 0, 0, None
    dt = 0.0
    while True:
        t0 = time.perf_counter()
        veh_steps += eng.get_vehicle_count()  # vehicles that take the coming step (admissions aside)
        eng.next_step()
        dt += time.perf_counter() - t0
        steps += 1
        if steps == parity_steps:
            parity, per_vehicle = parity_record(eng)  # untimed
            with open(state_dump + ".parity_t%d.json" % threads, "w") as f:
                json.dump(per_vehicle, f)
        if dt > budget_s and steps >= parity_steps:
            break
    running = eng.get_vehicle_count()
    time.sleep(0.2)  # reference destructor race: settle before the engine is dropped
    del eng
    return {
        "value": veh_steps / dt, "unit": "vehicle-steps/s", "cores": threads, "kind": kind,
        "steps_per_sec": steps / dt,
        "sample": "%d steps from the GPU run's own state at the start of its timed region (%d -> %d running vehicles, "
                  "injected via Archive JSON, load %.1f s untimed), %.1f s of wall time, %d thread(s) of %d host cores"
                  % (steps, start_running, running, t_load, dt, threads, os.cpu_count() or 1),
    }, parity


def cpu_leg_subprocess(cfg, budget_s, threads, state_dump, parity_steps):
    """One reference leg in its own process, with Vehicle objects at ascending addresses (LD_PRELOAD of
    oracle/_ref/libmonotonic_new.so, oracle/monotonic_new.cpp).  The reference walks its vehicles in std::set<Vehicle*>
    order — by heap address — and its unstable sort of the vehicles that change drivable then leaves vehicles with EXACTLY
    equal distances in an order that depends on those addresses and, with several threads, on which thread finishes first
    (SURVEY.md App. C-6).  One thread + creation-ordered addresses is the reproducible reference; that is what the
    in-run parity check compares with."""
    import subprocess
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    pre = os.path.join(ref_dir, "libmonotonic_new.so")
    size = os.path.join(ref_dir, "vehicle_size.txt")
    if not (os.path.exists(pre) and os.path.exists(size)):
        return None, None
    env = dict(os.environ, LD_PRELOAD=pre, CFX_VEHICLE_SIZE=open(size).read().strip())
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", cfg, str(budget_s), str(threads), state_dump,
                          str(parity_steps)], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return None, None
    d = json.loads(lines[-1])
    d["leg"]["sample"] += "; own process, Vehicle objects at creation-ordered addresses"
    return d["leg"], d["parity"]


def kernel_source_sha():
    """sha256 over the HIP sources the device library is built from: a PMC summary counts only for the kernels it measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cityflow_amd", "csrc", "hip", "*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_names):
    """HBM bytes per launch of the action kernel from the committed rocprofv3 PMC summary (FETCH_SIZE and WRITE_SIZE in
    separate passes, tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on
    gfx950) — only if that summary was taken from THESE kernel sources (it carries their hash); otherwise None."""
    import glob
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha") != sha or d.get("workload", "bench") != "bench":
            continue
        for name in kernel_names:
            k = next((v for kn, v in sorted(d.get("kernels", {}).items()) if name in kn), None)
            if k:
                return k.get("hbm_bytes_fetch_doubled", k.get("hbm_bytes_raw")), "%s (%s; (2*FETCH_SIZE+WRITE_SIZE)*1024 per launch)" % (
                    os.path.relpath(path, ROOT), d.get("window", ""))
    return None, None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-leg":  # cpu_leg_subprocess()
        cfg, budget, threads, dump, psteps = sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
        leg, par = cpu_baseline(cfg, budget, threads, dump, parity_steps=psteps)
        print(json.dumps({"leg": leg, "parity": par}), flush=True)
        os._exit(0)  # (reference destructor race)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--build-up-steps", type=int, default=BUILD_UP_STEPS, help=argparse.SUPPRESS)
    ap.add_argument("--profile-steps", type=int, default=100, help="instrumented steps for the roofline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the 8-thread cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="reference thread_num (default min(8, host cores))")
    ap.add_argument("--cpu-leg-seconds", type=float, default=6.0, help="budget of each extra leg (1 thread, all host cores)")
    # test hooks (tests/test_distributed.py drives the N>1 code path on CPU with gloo and the CPU twin)
    ap.add_argument("--scenario", default="grid_30x30", help=argparse.SUPPRESS)
    ap.add_argument("--extra-flows", type=int, default=N_EXTRA_FLOWS, help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--backend-lib", default="", help=argparse.SUPPRESS)
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of one tiled network")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: grow the grid with N (every GPU owns a 30x30 block) instead of tiling the N=1 workload itself")
    ap.add_argument("--strong", action="store_true", help=argparse.SUPPRESS)  # the default since round 2
    ap.add_argument("--tile-block", type=int, default=30, help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.strong = not args.weak
    on_gpu = args.backend_lib == ""

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if on_gpu:
            local_dev = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_dev)
            if args.dist_backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
            else:
                dist.init_process_group(backend=args.dist_backend)
        else:
            dist.init_process_group(backend=args.dist_backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    from cityflow_amd import _cityflow

    tiled = world > 1 and not args.replicas
    eng = None
    halo_notes = []
    if tiled:
        import torch
        from cityflow_amd.tiled import DistributedEngine
        rows, cols = tile_grid(world)
        workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_bench_tiled")
        if args.strong:  # the very workload of the N=1 run, cut into rows x cols tiles
            if rank == 0:
                build_workload(workdir, seed=0, scenario=args.scenario, n_extra=args.extra_flows)
            barrier()
            cfg = build_workload(workdir, seed=0, scenario=args.scenario, n_extra=args.extra_flows)
        else:
            if rank == 0:  # one node: every rank reads the files rank 0 wrote
                build_tiled_workload(workdir, rows, cols, args.tile_block, args.extra_flows)
            barrier()
            cfg = os.path.join(workdir, "gen_%dx%d" % (args.tile_block * rows, args.tile_block * cols), "config_bench.json")
        # Probe the halo transports on this machine before committing to one, best first: mailboxes in the receiving
        # GPU's HBM (hipIpc peer memory over xGMI), mailboxes in shared host memory, RCCL send / recv of device-resident
        # messages, gloo through host buffers.  A transport counts only if EVERY rank ran a few steps on it without an error.
        for transport in ("device", "host", "rccl", "gloo"):
            if transport == "rccl" and (not on_gpu or dist.get_backend() != "nccl"):
                continue
            cand, ok = None, 1
            try:
                cand = DistributedEngine(cfg, rows, cols, backend_library=args.backend_lib, transport=transport)
                for _ in range(20):
                    cand.next_step()
                cand.sync()
                cand.local_scalars()  # raises if a device-side halo wait timed out
            except Exception as exc:  # noqa: BLE001 - any failure disqualifies the transport
                ok = 0
                halo_notes.append("%s: %s" % (transport, str(exc)[:200]))
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                cand.reset(True)  # back to step 0 with the configured seed: the probe leaves no trace
                eng = cand
                break
            del cand
        if eng is None:
            tiled = False  # neither transport works here: measure independent replicas instead (and say so)
            if rank == 0:
                print("[bench] tiled halo exchange unavailable (%s); falling back to replicas" % "; ".join(halo_notes),
                      file=sys.stderr)
        else:
            eng._scalars = eng.local_scalars  # this rank's tile; summed over ranks below
    if not tiled:
        workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_bench_rank%d" % rank)
        cfg = build_workload(workdir, seed=rank, scenario=args.scenario, n_extra=args.extra_flows)
        if on_gpu:
            eng = _cityflow.Engine(cfg, 1)  # HIP engine on device LOCAL_RANK; raises if the extension/GPU is missing
            assert eng.backend_name() == "hip-gfx950"
        else:
            eng = _cityflow.Engine._with_backend(cfg, 1, args.backend_lib)

    for _ in range(args.build_up_steps):  # the workload state: not a warm-up, and not optional
        eng.next_step()
    for _ in range(args.warmup):
        eng.next_step()
    eng.sync()
    sc0 = eng._scalars()
    state_dump = None
    if rank == 0 and args.cpu_seconds > 0 and not tiled:
        state_dump = os.path.join(workdir, "state_at_timed_region.json")
        eng.snapshot().dump(state_dump)
        # Both engines start the timed region from this very file: the GPU engine re-reads it too, so that vehicles are
        # numbered in archive order on both sides.  (Where two vehicles enter a lane with EXACTLY equal distances in one
        # step the reference's std::sort leaves their order to the order of its Vehicle objects; an engine that kept
        # running and one that was restored from an archive may break such a tie differently — seen once in ~500 steps
        # of this workload — and would then drift apart although both are right.)
        eng.load_from_file(state_dump)
        eng.sync()
        sc0 = eng._scalars()  # (the archive does not carry the vehicle-steps counter)

    barrier()
    eng.sync()
    host0 = eng._eng._host_seconds() if tiled else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.next_step()
    eng.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    host1 = eng._eng._host_seconds() if tiled else None  # rank 0's host time inside the timed region
    sc1 = eng._scalars()
    veh_steps = sc1["vehicle_steps"] - sc0["vehicle_steps"]
    gpu_parity, gpu_per_vehicle = parity_record(eng) if (rank == 0 and state_dump is not None) else (None, None)  # end of the timed region
    ties_in_window = sc1.get("tie_events", 0) - sc0.get("tie_events", 0)
    # per-lane vehicle counts at the end of the timed region (every rank takes part in a tiled run's getter)
    import hashlib
    lane_counts_end = eng.get_lane_vehicle_count_array()
    lane_hash_end = hashlib.sha256(lane_counts_end.tobytes()).hexdigest()[:16]

    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor([float(veh_steps), float(sc0["active_vehicle_count"]), float(sc1["active_vehicle_count"])],
                         dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        veh_steps, run0, run1 = int(v[0].item()), int(v[1].item()), int(v[2].item())
    else:
        run0, run1 = sc0["active_vehicle_count"], sc1["active_vehicle_count"]

    # ---- roofline leg: instrumented continuation of the same run (HIP events on the engine's stream).  Tiled runs: every
    #      rank keeps stepping (the tiles are coupled), rank 0 instruments its own tile, halo kernels included.
    roofline = None
    if on_gpu and args.profile_steps > 0 and (rank == 0 or tiled):
        scp0 = eng._scalars()
        if rank == 0:
            (eng._eng._profile_enable(0, True) if tiled else eng._profile_enable(True))
            for _ in range(0 if tiled else 5):  # the first instrumented launches (event pool, lazy loads) are not the kernels
                eng.next_step()
            if not tiled:
                eng._profile_read()
                scp0 = eng._scalars()
        for _ in range(args.profile_steps):
            eng.next_step()
        if rank == 0:
            prof = eng._eng._profile_read(0) if tiled else eng._profile_read()
            (eng._eng._profile_enable(0, False) if tiled else eng._profile_enable(False))
        eng.sync()
        scp1 = eng._scalars()
        act_ms, act_n = prof["k_action"] if rank == 0 else (0.0, 0)
        if rank == 0 and act_n:
            vehicles_per_launch = (scp1["vehicle_steps"] - scp0["vehicle_steps"]) / float(act_n)
            avg_s = act_ms / act_n / 1e3
            achieved = ACTION_BYTES_PER_VEHICLE * vehicles_per_launch / avg_s / 1e9
            step_ms = sum(ms for ms, _n in prof.values()) / act_n
            traffic, traffic_src = pmc_traffic(("kr_action", "k_action"))
            roofline = {
                "bound": "hbm", "kernel": "k_action", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None if tiled else traffic,
                "traffic_source": None if tiled else traffic_src,
                "avg_launch_us": avg_s * 1e6, "vehicles_per_launch": vehicles_per_launch,
                "algorithmic_bytes_per_vehicle": ACTION_BYTES_PER_VEHICLE,
                "measured_over": "%d instrumented steps following the timed region%s" % (
                    args.profile_steps, " (rank 0's tile)" if tiled else ""),
                "kernel_us_per_step": {k: ms / max(n, 1) * 1e3 for k, (ms, n) in prof.items() if n},
                "sum_kernel_ms_per_step": step_ms,
            }

    if rank == 0:
        cpu, legs, parity_in_run, parity_detail = None, None, None, None
        if args.cpu_seconds > 0 and not tiled:
            threads = args.cpu_threads or min(8, os.cpu_count() or 1)
            cpu, par8 = cpu_baseline(cfg, args.cpu_seconds, threads, state_dump, parity_steps=args.steps)
            legs, ref_parity, parity_threads = [], par8, threads
            for t in sorted({1, os.cpu_count() or 1} - {threads}):
                if args.cpu_leg_seconds > 0 and cpu["kind"] == "reference":
                    if t == 1:  # the reproducible reference (see cpu_leg_subprocess): timing leg and parity check in one
                        leg, par = cpu_leg_subprocess(cfg, args.cpu_leg_seconds, 1, state_dump, args.steps)
                        if leg is not None:
                            legs.append(leg)
                            if par is not None:
                                ref_parity, parity_threads = par, 1
                            continue
                    legs.append(cpu_baseline(cfg, args.cpu_leg_seconds, t, state_dump)[0])
            # In-run parity: the reference, from the same archive, after the same number of steps.
            #   counts    per-lane vehicle counts and the vehicle count — north_star's bit-exact items;
            #   positions every vehicle's (speed, distance), compared bit for bit and as relative deviation.
            # The second comparison is well defined only while no two vehicles entered a drivable with EXACTLY equal
            # distances (cfx_scalars::tie_events; the reference's unstable sort orders such a pair by heap address): a tie
            # in the window is reported, and the few vehicles behind it then differ between ANY two engines, two runs of
            # the reference included.
            counts_equal = bool(gpu_parity and ref_parity and gpu_parity["vehicles"] == ref_parity["vehicles"]
                                and gpu_parity["lane_hash"] == ref_parity["lane_hash"])
            positions = None
            pv = state_dump + ".parity_t%d.json" % parity_threads
            if gpu_per_vehicle is not None and os.path.exists(pv):
                with open(pv) as f:
                    positions = parity_detail_compare(gpu_per_vehicle, {k: tuple(v) for k, v in json.load(f).items()})
            parity_in_run = bool(counts_equal and positions is not None and (
                positions["vehicles_differing"] == 0 or ties_in_window > 0))
            parity_detail = {"after_steps": args.steps, "against": "%s, %d thread(s)" % (cpu["kind"], parity_threads),
                             "lane_counts_and_vehicle_count_equal": counts_equal,
                             "positions_bit_exact": bool(positions and positions["vehicles_differing"] == 0),
                             "positions": positions, "exact_distance_ties_in_window": ties_in_window,
                             "gpu": gpu_parity, "cpu": ref_parity, "cpu_%d_threads" % threads: par8,
                             "checked": "per-lane vehicle counts, vehicle count, every vehicle's (speed, distance) bit for bit"}
        out = {
            "metric": "vehicle_steps_per_sec", "value": veh_steps / elapsed, "unit": "vehicle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": None if world == 1 else ("strong" if (tiled and args.strong) else "weak"), "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "steps_per_sec": args.steps * (1 if tiled else world) / elapsed,
            "config": {
                "workload": ("%s + %d seeded interior flows cut into %dx%d tiles, one tile per GPU, one-lane ghost halo "
                             "exchanged every step" % (args.scenario, args.extra_flows, rows, cols)) if (tiled and args.strong) else
                            ("grid_%dx%d (generator-format grid, --tlPlan, interval 1.0) + %d seeded interior flows "
                             "(1 veh / %.0f s each until t=%d s); one network, %dx%d tiles of %dx%d intersections, one "
                             "tile per GPU, one-lane ghost halo exchanged every step" % (
                                 args.tile_block * rows, args.tile_block * cols, args.extra_flows * world, EXTRA_INTERVAL,
                                 EXTRA_END, rows, cols, args.tile_block, args.tile_block)) if tiled else (
                            "%s (reference generator, --tlPlan, interval 1.0) + %d seeded interior flows "
                            "(1 veh / %.0f s each until t=%d s); %s" % (
                                args.scenario, args.extra_flows, EXTRA_INTERVAL, EXTRA_END,
                                "one replica per GPU" if world > 1 else "single engine")),
                "state": "network state after %d simulated seconds of demand build-up (untimed input construction), then "
                         "%d warm-up steps" % (args.build_up_steps, args.warmup),
                "running_vehicles_start": run0, "running_vehicles_end": run1,
                "lanes": len(eng.lane_ids()),
                "lane_count_hash_end": lane_hash_end,
                "halo": eng.halo_transport() if tiled else None,
                "layout": eng._layout() if hasattr(eng, "_layout") else None,
                "halo_probe_failures": halo_notes or None,
                "host_us_per_step": ({"spawner": round((host1[0] - host0[0]) / args.steps * 1e6, 1),
                                      "submit": round((host1[1] - host0[1]) / args.steps * 1e6, 1)} if tiled else None),
                "parallelism": ("tiles %dx%d + halo" % (rows, cols)) if tiled else (
                    "replica x%d" % world if world > 1 else "1 gpu"),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_legs": legs,
            "parity_in_run": parity_in_run,
            "parity": parity_detail,
        }
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()


if __name__ == "__main__":
    main()
