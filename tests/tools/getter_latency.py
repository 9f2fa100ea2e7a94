import sys, time
sys.path.insert(0, '.')
from cityflow_amd import _cityflow as m, scenarios
cfg = scenarios.materialize('grid_6x6', '/tmp/gl_wd')
e = m.Engine(cfg, 1)
for _ in range(600): e.next_step()
e.sync()
def t(f, n=200):
    f()
    t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print('vehicles', e.get_vehicle_count())
for name in ['get_vehicle_count','get_lane_vehicle_count','get_lane_waiting_vehicle_count','get_vehicle_speed','get_lane_vehicles','get_vehicles']:
    print('%-32s %8.1f us' % (name, t(getattr(e,name))))
def loop():
    e.next_step(); e.get_lane_vehicle_count(); e.get_lane_waiting_vehicle_count(); e.get_vehicle_speed(); e.get_lane_vehicles()
print('RL-style step with 4 getters: %.1f us' % t(loop, 300))
print('bare step: %.1f us' % t(e.next_step, 500))
