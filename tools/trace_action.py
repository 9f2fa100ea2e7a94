import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cityflow_amd import _cityflow
lib = os.path.join(ROOT, "tools", "libexp_trace.so")
cfg = bench.with_config(bench.build_workload("/tmp/cfa_exp", 0), "trace", cfx={"ringLanesPerWave": 20014})  # block form, 14 lanes per block
dll = ctypes.CDLL(lib)
eng = _cityflow.Engine._with_backend(cfg, 1, lib)
dll.cfx_trace_dump(b"/tmp/x", 0)  # arm
for _ in range(320): eng.next_step()
eng.sync()
nb = (11160 + 13) // 14 + 2 * ((32400 + 255) // 256)
dll.cfx_trace_dump(b"/tmp/trace.bin", 4096 + 2048)
full = np.fromfile("/tmp/trace.bin", dtype=np.int64).reshape(-1, 8)
a = full[:nb]
x = full[4096:]
x = x[x[:, 0] > 0]
nJ = int(x[0, 5])
busy = x[: (nJ + 15) // 16]
xt0 = x[:, 0].min()
xus = lambda v: (v - xt0) / 100.0
print("kr_cross: %d blocks started, %d jobs in %d busy blocks; last start %.2f us" % (len(x), nJ, len(busy), xus(x[:, 0].max())))
print("  busy blocks: counts done avg %.2f | record loaded avg %.2f max %.2f | crosses done avg %.2f max %.2f | end avg %.2f max %.2f" % (
    xus(busy[:, 1]).mean(), xus(busy[:, 2]).mean(), xus(busy[:, 2]).max(), xus(busy[:, 3]).mean(), xus(busy[:, 3]).max(), xus(busy[:, 4]).mean(), xus(busy[:, 4]).max()))
print("  idle blocks end avg %.2f max %.2f" % (xus(x[len(busy):, 4]).mean(), xus(x[len(busy):, 4]).max()))
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0  # 100 MHz wall clock
nl = (11160 + 13) // 14
lane, ll, st = a[:nl], a[nl:nl + 127], a[nl + 127:]
print("blocks", len(a), "first start 0, last start %.2f us" % us(a[:, 0].max()))
for name, blk in (("lane blocks", lane), ("laneLink blocks", ll)):
    print(name, "start avg %.2f max %.2f | prefix done avg %.2f max %.2f | loads done avg %.2f max %.2f | end avg %.2f max %.2f | T avg %.0f max %d" % (
        us(blk[:, 0]).mean(), us(blk[:, 0]).max(), us(blk[:, 1]).mean(), us(blk[:, 1]).max(), us(blk[:, 2]).mean(), us(blk[:, 2]).max(),
        us(blk[:, 4]).mean(), us(blk[:, 4]).max(), blk[:, 5].mean(), blk[:, 5].max()))
    d = blk[:, 4] - blk[:, 0]
    print("   block duration avg %.2f p90 %.2f max %.2f us; phase avgs: preamble %.2f, loads %.2f, compute %.2f" % (
        d.mean() / 100, np.percentile(d, 90) / 100, d.max() / 100, (blk[:, 1] - blk[:, 0]).mean() / 100, (blk[:, 2] - blk[:, 1]).mean() / 100, (blk[:, 4] - blk[:, 2]).mean() / 100))
print("llstate blocks: start avg %.2f end avg %.2f max %.2f" % (us(st[:, 0]).mean(), us(st[:, 4]).mean(), us(st[:, 4]).max()))
