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
