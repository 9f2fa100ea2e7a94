import sys, os, time
ROOT='/root/repo'
sys.path.insert(0,ROOT)
import numpy as np
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload('/tmp/cfa_par', 0)
mode = sys.argv[1]
hip = _cityflow.Engine(cfg, 1)
tw = _cityflow.Engine._with_backend(cfg, 1, ROOT+'/oracle/_ref/libcfx_twin.so')
for s in range(520):
    hip.next_step()
    if mode == 'snap' and s == 319: hip.snapshot()
    if mode == 'sync' : hip.sync()
    if mode == 'counts' and s % 7 == 0: hip.get_lane_vehicle_count_array()
    if mode == 'speed' and s in (320,321,322,325,340,370,420): hip.get_vehicle_speed(); hip.get_vehicle_distance()
    if mode == 'state' and s in (320,321,322,325,340,370,420): hip._vehicle_state()
    if mode == 'dump' and s == 319: hip.snapshot().dump('/tmp/cfa_par/x.json')
for s in range(520): tw.next_step()
a,b=hip._vehicle_state(), tw._vehicle_state()
oa,ob=np.argsort(a['vid']),np.argsort(b['vid'])
bad=0
for k in ('vid','drivable','dis','speed','blocker','leader'):
    x,y=a[k][oa],b[k][ob]
    n = int((x!=y).sum()) if x.shape==y.shape else -1
    print(mode, k, 'diff', n)
