"""Where does a next_step() stall?  Replays bench.py's own sequence (build-up -> Archive dump -> load_from_file ->
clock-warming spin -> warm-up -> steps) and timestamps EVERY next_step() call; prints the calls above a threshold with
what changed around them (ring capacity scale, device memory, running vehicles).

  python tools/stall_probe.py [--steps 600] [--threshold-us 500] [--no-load] [--cfx key=value,...]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--threshold-us", type=float, default=500.0)
    ap.add_argument("--no-load", action="store_true")
    ap.add_argument("--sync-every", type=int, default=0, help="eng.sync() every n steps (0: only at the windows' ends, as bench.py)")
    ap.add_argument("--cfx", default="")
    ap.add_argument("--scenario", default="grid_30x30")
    ap.add_argument("--extra-flows", type=int, default=bench.N_EXTRA_FLOWS)
    ap.add_argument("--build-up-steps", type=int, default=bench.BUILD_UP_STEPS)
    args = ap.parse_args()
    from cityflow_amd import _cityflow

    workdir = os.path.join(tempfile.gettempdir(), "cityflow_amd_stall_probe")
    cfg = bench.build_workload(workdir, seed=0, scenario=args.scenario, n_extra=args.extra_flows)
    if args.cfx:
        cfg = bench.with_config(cfg, "cfx", cfx={k: (int(v) if v.lstrip("-").isdigit() else v)
                                                  for k, v in (kv.split("=") for kv in args.cfx.split(","))})
    eng = _cityflow.Engine(cfg, 1)
    events = []

    def timed(label, n, sync_every=0):
        calls = []
        eng.sync()
        t0 = time.perf_counter()
        for i in range(n):
            ring0 = eng._ring_info()
            t1 = time.perf_counter()
            eng.next_step()
            dt = time.perf_counter() - t1
            calls.append(dt)
            if dt * 1e6 > args.threshold_us:
                events.append({"window": label, "index": i, "us": round(dt * 1e6), "ring_before": list(ring0),
                               "ring_after": list(eng._ring_info())})
            if sync_every and (i + 1) % sync_every == 0:
                eng.sync()
        t2 = time.perf_counter()
        eng.sync()
        t3 = time.perf_counter()
        return {"window": label, "steps": n, "ms_per_step": (t3 - t0) / max(n, 1) * 1e3, "final_sync_ms": (t3 - t2) * 1e3,
                "worst_call_us": round(max(calls) * 1e6) if calls else 0,
                "median_call_us": round(sorted(calls)[len(calls) // 2] * 1e6, 1) if calls else 0}

    out = [timed("build-up", args.build_up_steps)]
    print(json.dumps(out[-1]), flush=True)
    print("ring", eng._ring_info(), "vehicles", eng.get_vehicle_count(), "devmem", eng._device_memory(), flush=True)
    if not args.no_load:
        dump = os.path.join(workdir, "state.json")
        eng.snapshot().dump(dump)
        t = time.perf_counter()
        eng.load_from_file(dump)
        print("load_from_file: %.1f ms; ring %s" % ((time.perf_counter() - t) * 1e3, eng._ring_info()), flush=True)
    eng._device_spin(50000)
    out.append(timed("warm-up", args.warmup))
    done = 0
    for label, n in (("driver-20", 20), ("200-a", 200), ("200-b", 200)):
        out.append(timed(label, n, args.sync_every))
        done += n
    if args.steps > done:
        out.append(timed("rest", args.steps - done, args.sync_every))
    for o in out[1:]:
        print(json.dumps(o))
    print("ring", eng._ring_info(), "vehicles", eng.get_vehicle_count(), "devmem", eng._device_memory())
    print("events above %.0f us:" % args.threshold_us)
    for e in events:
        print("  ", json.dumps(e))


if __name__ == "__main__":
    main()
