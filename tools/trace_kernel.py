"""Developer tool: per-block wall-clock stamps of ONE kernel of the dense step at 1 M vehicles (device library built with
-DCFX_TRACE -DCFX_TRACE_KERNEL=<id>, cfx_device.h lists the ids; python tools/build_variants.py tr8="-DCFX_TRACE -DCFX_TRACE_KERNEL=8").
usage: python tools/trace_kernel.py <id>=<denseForm>[,crossMode] ...   e.g. 8=262 1=270 2=270 3=270 4=263
Prints, for the last traced step: blocks that ran, when they started / ended relative to the first start, the spans between
consecutive stamps (us; 100 MHz clock), blocks in flight over time, and the notes the kernel left (slot 5 / 7)."""
import ctypes, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
specs = sys.argv[1:]
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload("/tmp/cfa_exp", 0, scenario=os.environ.get("CFX_TRACE_SCENARIO", "gen_100x100"), n_extra=int(os.environ.get("CFX_EXP_EXTRA", 33000)))
if os.environ.get("CFX_TRACE_LC"):  # the lane-change step: the state is built with lane change too
    c0 = json.load(open(cfg)); c0["laneChange"] = True
    cfg = cfg.replace(".json", "_lcbase.json"); json.dump(c0, open(cfg, "w"))
base = _cityflow.Engine(cfg, 1)
for _ in range(300):
    base.next_step()
arch = base.snapshot()
del base
for spec in specs:
    kid, form = spec.split("=")
    lib = os.path.join(ROOT, "gpurun_exp", "lib_tr%s.so" % kid)
    c = json.load(open(cfg)); c["cfx"] = {"layout": os.environ.get("CFX_TRACE_LAYOUT", "dense"), "denseForm": int(form)}
    if os.environ.get("CFX_TRACE_LC"):
        c["laneChange"] = True
    cfg2 = cfg.replace(".json", "_tr%s.json" % kid); json.dump(c, open(cfg2, "w"))
    dll = ctypes.CDLL(lib)
    eng = _cityflow.Engine._with_backend(cfg2, 1, lib)
    eng.load(arch)
    dll.cfx_trace_dump(b"/tmp/x", 0)  # arm
    for _ in range(19):
        eng.next_step()
    eng.sync()
    dll.cfx_trace_dump(b"/tmp/x", -1)  # clear: only the last step's stamps
    eng.next_step()
    eng.sync()
    path = "/tmp/trace_%s.bin" % kid
    dll.cfx_trace_dump(path.encode(), 65536)
    full = np.fromfile(path, dtype=np.int64).reshape(-1, 8)
    a = full[full[:, 0] > 0]
    sc = eng._scalars()
    print("kernel id", kid, "denseForm", form, "blocks that stamped", len(a), "diag", {k: v for k, v in sc.items() if k.startswith("diag_")})
    if not len(a):
        continue
    t0 = a[:, 0].min()
    endcol = 6 if (a[:, 6] > 0).any() and kid == "8" else (4 if (a[:, 4] > 0).any() else 3)
    a = a[a[:, endcol] > 0]
    us = lambda x: (x - t0) / 100.0
    print("   start: avg %.2f p90 %.2f max %.2f us | end: avg %.2f p50 %.2f p90 %.2f max %.2f us" % (
        us(a[:, 0]).mean(), np.percentile(us(a[:, 0]), 90), us(a[:, 0]).max(), us(a[:, endcol]).mean(),
        np.percentile(us(a[:, endcol]), 50), np.percentile(us(a[:, endcol]), 90), us(a[:, endcol]).max()))
    cols = [k for k in (0, 1, 2, 3, 4, 6) if (a[:, k] > 0).all()]
    for x, y in zip(cols[:-1], cols[1:]):
        d = (a[:, y] - a[:, x]) / 100.0
        print("   stamp %d -> %d: avg %.2f p50 %.2f p90 %.2f max %.2f us" % (x, y, d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
    if kid in ("5", "6"):  # vehicle blocks / laneLink (llstate) blocks of the action launch apart
        for name, sel in (("vehicle blocks", a[:, 5] == 0), ("laneLink blocks", a[:, 5] == 1)):
            b = a[sel]
            if len(b):
                d = (b[:, 4] - b[:, 0]) / 100.0
                print("   %s: %d, start avg %.2f max %.2f, end avg %.2f max %.2f, duration avg %.2f p90 %.2f max %.2f us" % (
                    name, len(b), us(b[:, 0]).mean(), us(b[:, 0]).max(), us(b[:, 4]).mean(), us(b[:, 4]).max(), d.mean(), np.percentile(d, 90), d.max()))
    if kid == "8":
        b = a[a[:, 7] > 0]
        if len(b):
            print("   stamp 7 (k_cross2: after pass A1) present in %d blocks: at avg %.2f us after the block's start" % (len(b), ((b[:, 7] - b[:, 0]) / 100.0).mean()))
    if os.environ.get("CFX_TRACE_RAW"):  # stamps in time order, whatever their slot numbers mean in this build
        for k in range(1, 8):
            if (a[:, k] > t0).all():
                d = (a[:, k] - a[:, 0]) / 100.0
                print("   raw stamp %d: avg %.2f p50 %.2f p90 %.2f max %.2f us after the block's start" % (k, d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
    part = [k for k in (2, 3) if not (a[:, k] > 0).all() and (a[:, k] > 0).any()]
    for k in part:
        b = a[a[:, k] > 0]
        print("   stamp %d present in %d blocks: at avg %.2f us after the block's start" % (k, len(b), ((b[:, k] - b[:, 0]) / 100.0).mean()))
    ev = np.concatenate([np.stack([a[:, 0], np.ones(len(a))], 1), np.stack([a[:, endcol], -np.ones(len(a))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    fl = np.cumsum(ev[:, 1])
    print("   blocks in flight: max %d, time-avg %.0f over %.2f us" % (fl.max(), (fl[:-1] * np.diff(ev[:, 0])).sum() / max(1, ev[-1, 0] - ev[0, 0]), (ev[-1, 0] - ev[0, 0]) / 100.0))
    print("   note[5]: min %d avg %.1f max %d   note[7]: avg %.1f max %d" % (a[:, 5].min(), a[:, 5].mean(), a[:, 5].max(), a[:, 7].mean(), a[:, 7].max()))
    del eng
