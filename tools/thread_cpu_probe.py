"""Developer tool: which threads of the process burn CPU while the engine free-runs (the container's CFS quota throttles the
whole cgroup when the sum exceeds it).  Prints per-thread CPU seconds (utime + stime) after N steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cityflow_amd import _cityflow

def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out

def stat():
    d = {}
    for ln in open("/sys/fs/cgroup/cpu.stat"):
        k, v = ln.split()
        d[k] = int(v)
    return d

cfg = bench.build_workload("/tmp/cfa_thr", 0)
eng = _cityflow.Engine(cfg, 1)
for _ in range(300):
    eng.next_step()
eng.sync()
t0, s0 = threads(), stat()
w0 = time.time()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5000):
    eng.next_step()
eng.sync()
w1 = time.time()
t1, s1 = threads(), stat()
print("wall %.3f s; cgroup: usage %.3f s, throttled periods %d, throttled %.3f s" % (
    w1 - w0, (s1["usage_usec"] - s0["usage_usec"]) / 1e6, s1["nr_throttled"] - s0["nr_throttled"],
    (s1["throttled_usec"] - s0["throttled_usec"]) / 1e6))
for tid, (name, cpu) in sorted(t1.items(), key=lambda kv: -(kv[1][1] - t0.get(kv[0], ("", 0))[1])):
    d = cpu - t0.get(tid, ("", 0))[1]
    if d > 0.0:
        print("  tid %d %-20s %.3f s" % (tid, name, d))
print(len(t1), "threads")
