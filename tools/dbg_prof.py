import sys, os, time
ROOT='/root/repo'
sys.path.insert(0,ROOT)
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload('/tmp/cfa_par', 0)
eng = _cityflow.Engine(cfg, 1)
for _ in range(320): eng.next_step()
mode = sys.argv[1]
if mode in ('load', 'loadget'):
    eng.snapshot().dump('/tmp/cfa_par/s.json'); eng.load_from_file('/tmp/cfa_par/s.json')
for _ in range(200): eng.next_step()
if mode in ('get', 'loadget'):
    eng.get_vehicle_speed(); eng.get_lane_vehicle_count_array()
eng._profile_enable(True)
for _ in range(100): eng.next_step()
prof = eng._profile_read()
print(mode, {k: round(ms / n * 1e3, 1) for k, (ms, n) in prof.items() if n})
