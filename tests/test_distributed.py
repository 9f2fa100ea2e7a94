"""CPU: the N>1 path of bench.py (one process per device, barrier + max-over-ranks timing, whole-job value) with
world_size 2 on gloo and the CPU twin standing in for the device engine."""
import json
import os
import subprocess
import sys

from conftest import ROOT, TWIN_LIB, free_port


def test_bench_two_ranks_gloo(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "30", "--build-up-steps", "0",
           "--cpu-seconds", "0", "--scenario", "grid_6x6", "--extra-flows", "50", "--dist-backend", "gloo",
           "--backend-lib", TWIN_LIB, "--replicas"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 30 and d["scaling"] == "weak"
    assert d["metric"] == "vehicle_steps_per_sec" and d["unit"] == "vehicle-steps/s" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    # whole-job aggregate: two replicas => about twice the per-replica vehicle-steps of one step count
    assert d["config"]["parallelism"] == "replica x2"


def test_bench_two_ranks_strong_default_gloo(tmp_path):
    """The default N>1 mode (BASELINE.json: "30x30 at 1/2/4/8 GPUs"): the N=1 workload itself cut into tiles — strong
    scaling.  Here the 6x6 stand-in, cut 1x2, one tile per rank, halo exchanged every step."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20", "--build-up-steps", "100",
           "--cpu-seconds", "0", "--scenario", "grid_6x6", "--extra-flows", "40", "--dist-backend", "gloo", "--backend-lib", TWIN_LIB]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "tiles 1x2 + halo"
    assert "grid_6x6" in d["config"]["workload"] and d["config"]["halo"]
    assert d["value"] > 0 and d["config"]["running_vehicles_start"] > 100
    # the tiled run is the SAME simulation as the single engine's: same lane-count hash after the same steps
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "20", "--build-up-steps", "100",
                             "--cpu-seconds", "0", "--scenario", "grid_6x6", "--extra-flows", "40", "--backend-lib", TWIN_LIB],
                            env=dict(os.environ, TMPDIR=str(tmp_path)), capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    one = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][0])
    assert one["scaling"] is None
    assert d["config"]["lane_count_hash_end"] == one["config"]["lane_count_hash_end"]
    assert d["config"]["running_vehicles_end"] == one["config"]["running_vehicles_end"]


def test_bench_gpus_flag_launches_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself (what the reference's thread pool is to it,
    src/engine/engine.cpp:253-270) and prints an honest strong-scaling line: n_gpus 2, the CPU baseline, in-run parity
    against a single engine and against the reference, and the halo transport that won the probe."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = str(tmp_path)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10", "--build-up-steps", "150",
           "--cpu-seconds", "1", "--cpu-leg-seconds", "1", "--scenario", "grid_6x6", "--extra-flows", "60", "--backend-lib", TWIN_LIB]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "tiles 1x2 + halo"
    assert d["config"]["halo"] and d["config"]["backend"] == "cpu-twin" and d["config"]["process_group"] == "gloo"
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0
    if d["cpu_baseline"]["kind"] == "reference":  # (the twin as the CPU leg has no Archive-side light dump)
        assert d["parity_in_run"] is True and d["parity_excused_by_ties"] is False
        assert all(c["ok"] and c["signal_phases_equal"] for c in d["parity"]["checkpoints"]) and len(d["parity"]["checkpoints"]) >= 3
    assert d["parity"]["tiled_vs_single_engine"]["all_equal"] is True and d["parity"]["timed_region_equals_replay"] is True


def test_bench_refuses_a_world_that_is_not_gpus(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend-lib", TWIN_LIB], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_bench_single_engine_in_run_parity_twin(tmp_path):
    """N = 1 with the checks on: the parity object compares at >= 3 points of the window, signal phases included, and the
    line names the backend that ran."""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--build-up-steps", "120", "--cpu-seconds", "1",
           "--cpu-leg-seconds", "1", "--rl-seconds", "0.5", "--scenario", "grid_6x6", "--extra-flows", "40", "--backend-lib", TWIN_LIB]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["backend"] == "cpu-twin" and d["n_gpus"] == 1 and d["scaling"] is None
    assert d["roofline_at_scale"] is None  # (the 100x100 leg runs on the GPU only)
    if d["cpu_baseline"]["kind"] == "reference":
        cps = d["parity"]["checkpoints"]
        assert len(cps) >= 3 and all(c["ok"] and c["signal_phases_equal"] and c["positions_bit_exact"] for c in cps)
        assert d["parity_in_run"] is True
        assert d["rl_loop"]["array_api_steps_per_sec"] > 0 and d["rl_loop"]["reference_dict_api_steps_per_sec"] > 0


def test_parity_rule_after_a_tie():
    """The rule bench.py applies once an exact-distance tie has happened: few vehicles, and near the tied drivable."""
    sys.path.insert(0, ROOT)
    import bench
    flat = {"n_lanes": 4, "ll_start_lane": [0, 1], "ll_end_lane": [2, 3], "ll_inter": [0, 1], "real": None}
    assert bench.tie_neighbourhood(flat, [2]) == {2, 4, 0}
    g = {5: {"vehicles": 3, "lane_hash": "a", "phase_hash": None, "state_hash": "x", "ties": 1, "tie_drivables": [2],
             "_drivable_of": {"v1": 2, "v2": 3}}}
    r = {5: {"vehicles": 3, "lane_hash": "a", "state_hash": "y"}}
    import tempfile
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "cpu_t1_step5.json"), "w") as f:
        json.dump({"v1": [1.0, 2.0], "v2": [1.0, 2.0], "v3": [0.0, 0.0]}, f)
    ok, excused, rows = bench.judge_parity([5], g, {5: {"v1": (1.0, 2.5), "v2": (1.0, 2.0), "v3": (0.0, 0.0)}}, r, d, 1, flat)
    assert ok and excused and rows[0]["differing_vehicles_away_from_the_tie"] == 0
    ok, excused, rows = bench.judge_parity([5], g, {5: {"v1": (1.0, 2.0), "v2": (1.0, 2.5), "v3": (0.0, 0.0)}}, r, d, 1, flat)
    assert not ok and rows[0]["differing_vehicles_away_from_the_tie"] == 1  # v2 is on drivable 3: not around the tie
    g[5]["ties"] = 0
    ok, excused, rows = bench.judge_parity([5], g, {5: {"v1": (1.0, 2.5), "v2": (1.0, 2.0), "v3": (0.0, 0.0)}}, r, d, 1, flat)
    assert not ok  # no tie, no excuse


def test_bench_two_ranks_tiled_weak_gloo(tmp_path):
    """--weak: one network (3x6 here) that grows with N, tiled 1x2, one tile per rank, halo exchanged every step."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20", "--build-up-steps", "100",
           "--cpu-seconds", "0", "--tile-block", "3", "--extra-flows", "40", "--dist-backend", "gloo",
           "--backend-lib", TWIN_LIB, "--weak"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "tiles 1x2 + halo"
    assert d["value"] > 0 and d["config"]["running_vehicles_start"] > 100
    assert "grid_3x6" in d["config"]["workload"]
    # whole-job vehicle-steps = (running vehicles summed over both tiles) x steps, within the drift of the window
    per_step = d["value"] * d["ms_per_step"] / 1e3
    lo, hi = sorted((d["config"]["running_vehicles_start"], d["config"]["running_vehicles_end"]))
    assert lo * 0.9 <= per_step <= hi * 1.1


def test_bench_single_rank_twin(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "10", "--build-up-steps", "20", "--cpu-seconds", "0",
           "--scenario", "grid_6x6", "--extra-flows", "20", "--backend-lib", TWIN_LIB]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["roofline"] is None and d["cpu_baseline"] is None
    assert d["warmup"] == 10 and "20 simulated seconds" in d["config"]["state"]  # the workload state does not depend on --warmup


def test_bench_falls_back_to_replicas_when_no_halo_transport_comes_up(tmp_path):
    """An N > 1 run on a machine where no halo transport can be set up between the ranks still prints its line: N independent
    replicas (weak scaling), and says that this is not what was asked for."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--build-up-steps", "60",
           "--cpu-seconds", "1", "--scenario", "grid_6x6", "--extra-flows", "20", "--dist-backend", "gloo", "--backend-lib", TWIN_LIB,
           "--no-halo"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("replica x2") and "no halo transport" in d["config"]["parallelism"]
    assert d["config"]["halo_probe_failures"] == ["--no-halo"]


def test_bench_eight_ranks_on_config3_gloo(tmp_path):
    """BASELINE.json configs[3] as the driver will launch it on an 8-GPU node — `bench.py --gpus 8` under torch.distributed.run,
    the 30x30 network cut 2x4, one tile per rank, halo every step — with gloo and the CPU twin standing in for the devices: the
    plumbing of the first 8-rank run (tile cut, 8-way halo group, per-rank spawners, max-over-ranks timing, rank 0's single
    line) cannot be met for the first time on the hardware.  The tiles' lane counts equal the single engine's after the same
    steps (reference: one Engine over the whole network, src/engine/engine.cpp:566-594)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    common = ["--steps", "12", "--warmup", "6", "--build-up-steps", "60", "--cpu-seconds", "0", "--rl-seconds", "0", "--scale-steps", "0",
              "--profile-steps", "0", "--scenario", "grid_30x30", "--extra-flows", "400", "--backend-lib", TWIN_LIB]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo"] + common
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["tiling_failed"] is False
    assert d["config"]["parallelism"] == "tiles 2x4 + halo" and "grid_30x30" in d["config"]["workload"]
    assert d["value"] > 0 and d["config"]["running_vehicles_start"] > 1000
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=dict(os.environ, TMPDIR=str(tmp_path)),
                            capture_output=True, text=True, timeout=900)
    assert single.returncode == 0, single.stderr[-2000:]
    one = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["lane_count_hash_end"] == one["config"]["lane_count_hash_end"]
    assert d["config"]["running_vehicles_end"] == one["config"]["running_vehicles_end"]
