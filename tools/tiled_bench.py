"""Developer timing of the tiled path.  Single process: python tools/tiled_bench.py ROWS COLS [steps] [warmup];
one tile per process: torchrun --nproc-per-node N tools/tiled_bench.py ROWS COLS ... (ROWS*COLS == N)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cityflow_amd import _cityflow as m  # noqa: E402
from cityflow_amd import scenarios  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    lib = os.environ.get("CFX_BACKEND_LIB", "")
    wd = tempfile.mkdtemp(prefix="tiled_bench_")
    base = scenarios.materialize("grid_30x30", wd)
    d = os.path.dirname(base)
    flow = scenarios.dense_flows(os.path.join(d, "roadnet.json"), os.path.join(d, "flow_dense.json"), 3000, seed=12345,
                                 interval=6.0, base_flow=os.path.join(d, "flow.json"), end_time=240)
    cfg = scenarios.materialize("grid_30x30", wd, flow_file=flow)
    distributed = "RANK" in os.environ
    if distributed:
        import torch.distributed as dist
        from cityflow_amd.tiled import DistributedEngine
        dist.init_process_group(backend=os.environ.get("CFX_DIST_BACKEND", "gloo"))
        eng = DistributedEngine(cfg, rows, cols, backend_library=lib)
        rank = dist.get_rank()
    else:
        eng = m.TiledEngine(cfg, rows, cols, [], lib)
        if os.environ.get("CFX_MAILBOXES", "1") == "1":
            eng.enable_mailboxes("tiled_bench_%d" % os.getpid())
        rank = 0
    for _ in range(warmup):
        eng.next_step()
    eng.sync()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.next_step()
    eng.sync()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    n = eng.get_vehicle_count()
    if rank == 0:
        print("tiles %dx%d %s: %.1f us/step, %d vehicles, %.3f G vehicle-steps/s" %
              (rows, cols, "distributed" if distributed else "one process", dt / steps * 1e6, n, n * steps / dt / 1e9))


if __name__ == "__main__":
    main()
