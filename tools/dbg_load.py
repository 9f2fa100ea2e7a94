import sys, os
ROOT='/root/repo'
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
import numpy as np
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload('/tmp/cfa_par', 0)
hip = _cityflow.Engine(cfg, 1)
for _ in range(320): hip.next_step()
hip.snapshot().dump('/tmp/cfa_par/s.json')
hip.load_from_file('/tmp/cfa_par/s.json')
tw = _cityflow.Engine._with_backend(cfg, 1, ROOT+'/oracle/_ref/libcfx_twin.so')
tw.load_from_file('/tmp/cfa_par/s.json')
def st(e):
    s=e._vehicle_state(); o=np.argsort(s['vid']); return {k:v[o] for k,v in s.items()}
for s in range(200):
    hip.next_step(); tw.next_step()
    if s % 10 == 9 or s > 170:
        a,b=st(hip),st(tw)
        for k in ('vid','drivable','dis','speed','blocker','leader','route_pos'):
            if a[k].shape!=b[k].shape or not np.array_equal(a[k],b[k]):
                i=np.nonzero(a[k]!=b[k])[0][:4] if a[k].shape==b[k].shape else []
                print('DIVERGE step',s+1,k,[(int(a['vid'][j]),int(a['drivable'][j]),float(a['dis'][j]),float(b['dis'][j]),float(a['speed'][j]),float(b['speed'][j])) for j in i]); sys.exit(1)
print('OK 200 steps hip(load)==twin(load)')
