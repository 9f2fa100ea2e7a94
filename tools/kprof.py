"""Developer tool: section timers of k_action (needs a build with CFX_HIP_EXTRA_FLAGS=-DCFX_KPROF)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_kprof", 0)
eng = _cityflow.Engine(cfg, 1)
for _ in range(300):
    eng.next_step()
eng.sync()
lib = ctypes.CDLL(_cityflow._default_backend_path())
buf = (ctypes.c_ulonglong * 32)()
lib.cfx_debug_read_kprof(buf, 32)
N = 50
v0 = eng._scalars()["vehicle_steps"]
for _ in range(N):
    eng.next_step()
eng.sync()
nveh = eng._scalars()["vehicle_steps"] - v0
lib.cfx_debug_read_kprof(buf, 32)
names = ["leader", "carfollow", "inter-pre(light/canEnter)", "inter-cross-loop", "post(setDelta)", "store+classify", "load-own-state"]
print("vehicle-steps", nveh)
for i, n in enumerate(names):
    print("%-28s sum %12d cyc  avg/veh %8.1f  max %8d" % (n, buf[2 * i], buf[2 * i] / max(nveh, 1), buf[2 * i + 1]))
