"""Developer tool: watches the container's CFS throttling counter (cgroup v2 cpu.stat) while a command runs and, whenever it
moves, lists the threads of the cgroup that were burning CPU in the last interval.  usage: throttle_watch.py <cmd...>"""
import os, subprocess, sys, time

def stat():
    d = {}
    for ln in open("/sys/fs/cgroup/cpu.stat"):
        k, v = ln.split()
        d[k] = int(v)
    return d

def threads():
    out = {}
    for pid in os.listdir("/proc"):
        if not pid.isdigit():
            continue
        try:
            for t in os.listdir("/proc/%s/task" % pid):
                f = open("/proc/%s/task/%s/stat" % (pid, t)).read()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[(int(pid), int(t))] = (name, int(rest[11]) + int(rest[12]), rest[0])
        except Exception:
            pass
    return out

p = subprocess.Popen(sys.argv[1:], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
hz = os.sysconf("SC_CLK_TCK")
s0, t0, w0 = stat(), threads(), time.time()
while p.poll() is None:
    time.sleep(0.05)
    s1, t1, w1 = stat(), threads(), time.time()
    if s1["nr_throttled"] != s0["nr_throttled"]:
        busy = sorted(((v[1] - t0.get(k, ("", 0, ""))[1]) / hz, k, v[0], v[2]) for k, v in t1.items())
        busy = [b for b in busy if b[0] > 0][-12:]
        print("t=%.2f s: throttled +%d periods (+%.3f s); cgroup usage %.3f s in %.3f s wall; %d threads alive, runnable now: %d"
              % (w1 - w0, s1["nr_throttled"] - s0["nr_throttled"], (s1["throttled_usec"] - s0["throttled_usec"]) / 1e6,
                 (s1["usage_usec"] - s0["usage_usec"]) / 1e6, 0.05, len(t1), sum(1 for v in t1.values() if v[2] == "R")))
        for b in reversed(busy):
            print("     %.3f s  pid %d tid %d %s (%s)" % (b[0], b[1][0], b[1][1], b[2], b[3]))
        sys.stdout.flush()
    s0, t0 = s1, t1
print("done; total throttled periods", stat()["nr_throttled"])
