"""Worker of tests/test_tiling.py::test_two_ranks_gloo: one tile per process over torch.distributed (gloo), compared
step by step on every rank with the same network on a single engine."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cityflow_amd import _cityflow as m  # noqa: E402
from cityflow_amd.tiled import DistributedEngine  # noqa: E402


def main():
    cfg, lib, rows, cols, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    backend = os.environ.get("CFX_TEST_DIST_BACKEND", "gloo")
    if backend == "nccl":  # one rank per physical GPU (RCCL refuses two ranks on one device)
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend=backend)
    rank = dist.get_rank()
    halo = None
    if os.environ.get("CFX_TEST_SUBGROUP") == "1":  # the layout of a GPU job: halo on its own host-side group
        import datetime
        halo = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
    transport = os.environ.get("CFX_TEST_TRANSPORT") or None
    compact = int(os.environ.get("CFX_TEST_COMPACT", "0"))
    single_cfg = cfg
    if compact:  # the tiles forget their finished vehicles every `compact` vehicle numbers (a collective), the single engine never
        import json
        base = json.load(open(cfg))
        paths = []
        for name, every in (("tiles", compact), ("one", 0)):
            path = cfg.replace(".json", "_compact_%s_r%d.json" % (name, rank))
            json.dump(dict(base, cfx=dict(base.get("cfx", {}), compactVehicles=every)), open(path, "w"))
            paths.append(path)
        cfg, single_cfg = paths
    eng = DistributedEngine(cfg, rows, cols, backend_library=lib, halo_group=halo, transport=transport,
                            mailboxes=None if (transport or os.environ.get("CFX_TEST_MAILBOXES") == "auto")
                            else os.environ.get("CFX_TEST_MAILBOXES", "1") == "1")
    if transport:
        assert eng.transport == transport, (eng.transport, transport)
    single = m.Engine._with_backend(single_cfg, 1, lib) if lib else m.Engine(single_cfg, 1)
    crossed = 0
    for s in range(steps):
        eng.next_step()
        single.next_step()
        a = single.get_lane_vehicle_count_array()
        b = eng.get_lane_vehicle_count_array()
        assert np.array_equal(a, b), "rank %d step %d: lane counts differ" % (rank, s)
        if s % 25 == 24:
            wa = single.get_lane_waiting_vehicle_count_array()
            wb = eng.get_lane_waiting_vehicle_count_array()
            assert np.array_equal(wa, wb), "rank %d step %d: waiting counts differ" % (rank, s)
            sa, sb = single._scalars(), eng.scalars()
            for k in ("active_vehicle_count", "finished_vehicle_count", "cumulative_travel_time", "vehicle_steps"):
                assert sa[k] == sb[k], (rank, s, k, sa[k], sb[k])
            if compact:  # (vehicle numbers differ after a compaction: ids instead)
                assert eng.get_vehicle_speed() == single.get_vehicle_speed(), (rank, s)
                assert eng.get_vehicle_distance() == single.get_vehicle_distance(), (rank, s)
                assert eng.get_lane_vehicles() == single.get_lane_vehicles(), (rank, s)
                assert eng.get_vehicles(True) == single.get_vehicles(True), (rank, s)
                assert eng.get_average_travel_time() == single.get_average_travel_time(), (rank, s)
                crossed = max(crossed, len(eng.local_vehicle_state()["vid"]))
                continue
            # this rank's vehicles: same per-vehicle state as on the single engine
            va, vb = single._vehicle_state(), eng.local_vehicle_state()
            pos = {int(v): i for i, v in enumerate(va["vid"])}
            idx = np.array([pos[int(v)] for v in vb["vid"]], dtype=np.int64)
            for k in ("drivable", "dis", "speed", "leader", "blocker", "route_pos", "enter_ll_time"):
                assert np.array_equal(va[k][idx], vb[k]), "rank %d step %d: %s differs" % (rank, s, k)
            crossed = max(crossed, len(vb["vid"]))
        if s == steps - 1:  # string getters: merged over ranks == single engine
            assert eng.get_vehicle_speed() == single.get_vehicle_speed()
            assert eng.get_vehicle_distance() == single.get_vehicle_distance()
            assert eng.get_lane_vehicles() == single.get_lane_vehicles()
    assert crossed > 0
    if compact:
        held, times = eng._eng._vehicle_table()
        assert times >= 2 and single._vehicle_table()[1] == 0 and held < single._vehicle_table()[0], (held, times, single._vehicle_table())
        eng.compact_vehicles()  # on request, too (a collective)
        for s in range(20):
            eng.next_step()
            single.next_step()
            assert np.array_equal(eng.get_lane_vehicle_count_array(), single.get_lane_vehicle_count_array()), (rank, s)
        assert eng.get_vehicle_speed() == single.get_vehicle_speed()
        if rank == 0:
            print("COMPACT_OK", times + 1, "compactions,", held, "of", single._vehicle_table()[0], "vehicle numbers held")
    if os.environ.get("CFX_TEST_ARCHIVE") == "1":
        # archive and routes over ranks: the snapshot assembled from every rank's part is the single engine's; after a load
        # (no communication: every rank keeps its tile's part) both go on identically; setRoute gives the same verdicts
        import json
        import tempfile
        arch, arch1 = eng.snapshot(), single.snapshot()
        with tempfile.TemporaryDirectory() as tmp:
            arch.dump(os.path.join(tmp, "a.json"))
            arch1.dump(os.path.join(tmp, "b.json"))
            one = json.load(open(os.path.join(tmp, "b.json")))
            assert eng._eng._keeps_lane_history()  # (Lane::history travels in the parts: the dumps are compared whole)
            assert json.load(open(os.path.join(tmp, "a.json"))) == one
        later = []
        for s in range(40):
            eng.next_step()
            single.next_step()
            later.append(single.get_lane_vehicle_count_array().copy())
        eng.load(arch1)
        single.load(arch)
        moved = 0
        for s in range(40):
            if s == 5:
                for v in sorted(single.get_vehicles(True))[:30]:
                    a, b = single.set_vehicle_route(v, ["road_3_2_1"]), eng.set_vehicle_route(v, ["road_3_2_1"])
                    assert a == b, (rank, v, a, b)
                    moved += int(a)
            eng.next_step()
            single.next_step()
            if s < 5:
                assert np.array_equal(eng.get_lane_vehicle_count_array(), later[s]), (rank, s)
            assert np.array_equal(eng.get_lane_vehicle_count_array(), single.get_lane_vehicle_count_array()), (rank, s)
        assert eng.get_vehicle_distance() == single.get_vehicle_distance()
        # the rest of the reference's query API over ranks
        assert eng.get_vehicles() == single.get_vehicles() and eng.get_vehicles(True) == single.get_vehicles(True)
        assert eng.get_average_travel_time() == single.get_average_travel_time()
        some = single.get_vehicles(True)
        for v in some[:25] + some[-25:]:
            assert eng.get_vehicle_info(v) == single.get_vehicle_info(v), (rank, v)
            assert eng.get_leader(v) == single.get_leader(v), (rank, v)
        try:
            eng.get_leader("flow_999_0")
            raise AssertionError("unknown vehicle accepted")
        except RuntimeError:
            pass
        # a vehicle pushed and asked about before its first step (listed once although every rank's spawner holds it)
        for e in (single, eng):
            e.push_vehicle({"length": 6.0, "maxSpeed": 12.0}, ["road_1_1_0", "road_2_1_0"])
        listed = single.get_vehicles(True)
        assert eng.get_vehicles(True) == listed and sum(v.startswith("manually_pushed") for v in listed) == 1
        pushed = [v for v in listed if v.startswith("manually_pushed")][0]
        assert eng.get_vehicle_info(pushed) == single.get_vehicle_info(pushed) == {"running": "0"}
        assert eng.get_leader(pushed) == single.get_leader(pushed) == ""
        assert eng.get_average_travel_time() == single.get_average_travel_time()
        for _ in range(3):
            eng.next_step()
            single.next_step()
        assert eng.get_vehicles(True) == single.get_vehicles(True)
        assert eng.get_vehicle_info(pushed) == single.get_vehicle_info(pushed)
        sa, sb = single._scalars(), eng.scalars()
        for k in ("active_vehicle_count", "finished_vehicle_count", "cumulative_travel_time"):
            assert sa[k] == sb[k], (rank, k, sa[k], sb[k])
        if rank == 0:
            print("ARCHIVE_OK routes changed", moved)
    final_transport, final_count = eng.transport, single.get_vehicle_count()
    if os.environ.get("CFX_TEST_REPLAY") == "1":
        # saveReplay with one tile per process: rank 0 writes the lines from every rank's part; same file as one engine's
        del eng, single
        import json
        base = json.load(open(cfg))
        d = os.path.dirname(cfg)
        outs = {}
        for kind in ("single", "tiled"):
            c = dict(base, saveReplay=True, roadnetLogFile="w_rn_%s.json" % kind, replayLogFile="w_rp_%s.txt" % kind)
            path = os.path.join(d, "w_cfg_%s.json" % kind)
            if rank == 0:
                json.dump(c, open(path, "w"))
            dist.barrier()
            if kind == "single":
                if rank == 0:
                    e = m.Engine._with_backend(path, 1, lib)
                    for s in range(60):
                        e.next_step()
                    del e
            else:
                e = DistributedEngine(path, rows, cols, backend_library=lib, halo_group=halo, transport=transport)
                for s in range(60):
                    if s == 20:
                        e.set_save_replay(False)
                    if s == 21:
                        e.set_save_replay(True)
                    e.next_step()
                del e
            dist.barrier()
        if rank == 0:
            a, b = open(os.path.join(d, "w_rp_single.txt")).read().split("\n"), open(os.path.join(d, "w_rp_tiled.txt")).read().split("\n")
            assert len(a) == 61 and len(b) == 60, (len(a), len(b))
            assert a[:20] + a[21:] == b, "replay lines differ"
            assert open(os.path.join(d, "w_rn_single.json")).read() == open(os.path.join(d, "w_rn_tiled.json")).read()
            print("REPLAY_OK")
        eng = single = None
    dist.barrier()
    if rank == 0:
        print("TILED_OK", steps, final_count, "transport", final_transport)
    dist.barrier()
    sys.stdout.flush()
    sys.stderr.flush()
    # no destroy_process_group(): gloo's teardown occasionally aborts a rank that has already printed its verdict
    # ("terminate called without an active exception", seen once in ~50 runs); every rank is past the last barrier here
    os._exit(0)


if __name__ == "__main__":
    main()
