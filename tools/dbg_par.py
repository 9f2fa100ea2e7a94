import sys, os, time
ROOT='/root/repo'
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/oracle/_ref')
import bench, cityflow_ref
from cityflow_amd import _cityflow
cfg = bench.build_workload('/tmp/cfa_par', 0)
hip = _cityflow.Engine(cfg, 1)
for _ in range(320): hip.next_step()
hip.snapshot().dump('/tmp/cfa_par/state.json')
ref = cityflow_ref.Engine(cfg, 8)
ref.load_from_file('/tmp/cfa_par/state.json')
tw = _cityflow.Engine._with_backend(cfg, 1, ROOT+'/oracle/_ref/libcfx_twin.so')
tw.load_from_file('/tmp/cfa_par/state.json')
for s in range(200):
    hip.next_step(); ref.next_step(); tw.next_step()
    if s in (0,1,2,5,20,50,100,199):
        a,b,c=hip.get_vehicle_speed(), ref.get_vehicle_speed(), tw.get_vehicle_speed()
        da,db,dc=hip.get_vehicle_distance(), ref.get_vehicle_distance(), tw.get_vehicle_distance()
        bad=[k for k in a if k not in b or a[k]!=b[k] or da[k]!=db[k]]
        bad2=[k for k in c if k not in b or c[k]!=b[k] or dc[k]!=db[k]]
        print(s+1, len(a), len(b), 'hip-vs-ref bad', len(bad), 'twin-vs-ref bad', len(bad2), [(k,a[k],b.get(k),da[k],db.get(k)) for k in bad[:3]], flush=True)
time.sleep(0.3)
