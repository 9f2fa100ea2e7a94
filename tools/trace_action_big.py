"""Developer tool (needs tools/libexp_trace.so = the device library built with -DCFX_TRACE): per-block phase stamps of
kr_action on a large ring engine.  usage: python tools/trace_action_big.py [scenario] [n_extra] [ringLanesPerWave]"""
import ctypes, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
sys.argv = [sys.argv[0]]
scenario = args[0] if args else "gen_100x100"
n_extra = int(args[1]) if len(args) > 1 else 33000
rlw = int(args[2]) if len(args) > 2 else 2048
import bench
from cityflow_amd import _cityflow
lib = os.path.join(ROOT, "tools", "libexp_trace.so")
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=scenario, n_extra=n_extra)
c = json.load(open(cfg)); c["cfx"] = {"layout": "ring", "ringLanesPerWave": rlw}
cfg2 = cfg.replace(".json", "_trace.json"); json.dump(c, open(cfg2, "w"))
dll = ctypes.CDLL(lib)
eng = _cityflow.Engine._with_backend(cfg2, 1, lib)
dll.cfx_trace_dump(b"/tmp/x", 0)  # arm
for _ in range(320): eng.next_step()
eng.sync()
L = len(eng.get_lane_vehicle_count_array())
G = rlw % 1000
B = 1024 if rlw % 10000 >= 4000 else (512 if rlw % 10000 >= 2000 else 256)
nl = (L + G - 1) // G
dll.cfx_trace_dump(b"/tmp/trace.bin", 4096)
full = np.fromfile("/tmp/trace.bin", dtype=np.int64).reshape(-1, 8)
a = full[full[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
lane = full[:nl]
rest = full[nl:]
rest = rest[rest[:, 0] > 0]
ll = rest[rest[:, 1] > 0]
st = rest[rest[:, 1] == 0]
print("running", eng.get_vehicle_count(), "L", L, "B", B, "G", G, "blocks", len(a), "last start %.2f us, last end %.2f us" % (us(a[:, 0].max()), us(a[:, 4].max())))
for name, blk in (("lane blocks", lane), ("laneLink blocks", ll)):
    if not len(blk): continue
    d = blk[:, 4] - blk[:, 0]
    print(name, len(blk), "start avg %.2f max %.2f | end avg %.2f max %.2f | T avg %.0f max %d" % (
        us(blk[:, 0]).mean(), us(blk[:, 0]).max(), us(blk[:, 4]).mean(), us(blk[:, 4]).max(), blk[:, 5].mean(), blk[:, 5].max()))
    print("   block duration avg %.2f p50 %.2f p90 %.2f max %.2f us; phase avgs: preamble %.2f, loads %.2f, last pass compute %.2f" % (
        d.mean() / 100, np.percentile(d, 50) / 100, np.percentile(d, 90) / 100, d.max() / 100, (blk[:, 1] - blk[:, 0]).mean() / 100,
        (blk[:, 2] - blk[:, 1]).mean() / 100, (blk[:, 3] - blk[:, 2]).mean() / 100))
    # how many blocks are in flight over time
    ev = np.concatenate([np.stack([blk[:, 0], np.ones(len(blk))], 1), np.stack([blk[:, 4], -np.ones(len(blk))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    fl = np.cumsum(ev[:, 1])
    print("   blocks in flight: max %d, time-avg %.0f" % (fl.max(), (fl[:-1] * np.diff(ev[:, 0])).sum() / max(1, ev[-1, 0] - ev[0, 0])))
if len(st):
    d = st[:, 4] - st[:, 0]
    print("llstate blocks", len(st), "start avg %.2f end avg %.2f max %.2f; duration avg %.2f" % (us(st[:, 0]).mean(), us(st[:, 4]).mean(), us(st[:, 4]).max(), d.mean() / 100))
