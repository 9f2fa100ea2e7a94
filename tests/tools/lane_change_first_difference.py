"""Developer aid: the HIP engine and the CPU twin side by side on the bench.py workload with laneChange true, every vehicle
field compared after EVERY step; prints the first step that differs, which fields, and the vehicles concerned."""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
sys.argv = [sys.argv[0]]
import bench
from cityflow_amd import _cityflow
TWIN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle', '_ref', 'libcfx_twin.so')
cfg = bench.build_workload("/tmp/cfa_lcdbg", 0, scenario="grid_30x30")
c = json.load(open(cfg)); c["laneChange"] = True
path = cfg.replace(".json", "_lc.json"); json.dump(c, open(path, "w"))
hip = _cityflow.Engine(path, 1)
tw = _cityflow.Engine._with_backend(path, 1, os.path.abspath(TWIN))
def st(e):
    s = e._vehicle_state()
    o = np.argsort(s["vid"], kind="stable")
    return {k: v[o] for k, v in s.items()}
for s in range(260):
    hip.next_step(); tw.next_step()
    a, b = st(hip), st(tw)
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    nsh = int((b["lc_flags"] & 1).sum())
    if bad:
        print("step", s, "differs in", bad, len(a["vid"]), len(b["vid"]), "twin shadows", nsh)
        print("scalars hip", hip._scalars()); print("scalars tw", tw._scalars())
        sa, sb = set(a["vid"].tolist()), set(b["vid"].tolist())
        miss = sorted(sb - sa); extra = sorted(sa - sb)
        print("missing in hip", miss[:20], "extra", extra[:20])
        idx = {v: i for i, v in enumerate(b["vid"].tolist())}
        for v in miss[:10]:
            i = idx[v]
            print({k: b[k][i].item() for k in b})
        if len(a["vid"]) == len(b["vid"]):
            for k in bad:
                w = np.nonzero(a[k] != b[k])[0][:8]
                print(k, [(int(a["vid"][i]), a[k][i].item(), b[k][i].item()) for i in w])
        break
    if s % 20 == 19: print("ok", s, len(a["vid"]), "shadows", nsh, flush=True)
