import sys, os, json
ROOT='/root/repo'
sys.path.insert(0,ROOT)
import bench
from cityflow_amd import _cityflow
cfg = bench.build_workload('/tmp/cfa_par', 0)
hip = _cityflow.Engine(cfg, 1)
tw = _cityflow.Engine._with_backend(cfg, 1, ROOT+'/oracle/_ref/libcfx_twin.so')
for s in range(320): hip.next_step(); tw.next_step()
hip.snapshot().dump('/tmp/cfa_par/a.json'); tw.snapshot().dump('/tmp/cfa_par/b.json')
a=json.load(open('/tmp/cfa_par/a.json')); b=json.load(open('/tmp/cfa_par/b.json'))
def diff(x,y,path,out):
    if len(out)>12: return
    if type(x)!=type(y): out.append((path,'type',str(x)[:60],str(y)[:60])); return
    if isinstance(x,dict):
        for k in sorted(set(x)|set(y)):
            if k not in x or k not in y: out.append((path+'/'+k,'missing', k in x, k in y)); continue
            diff(x[k],y[k],path+'/'+k,out)
    elif isinstance(x,list):
        if len(x)!=len(y): out.append((path,'len',len(x),len(y))); return
        for i,(p,q) in enumerate(zip(x,y)): diff(p,q,path+'[%d]'%i,out)
    elif x!=y: out.append((path,x,y))
out=[]; diff(a,b,'',out)
print(len(out)); 
for o in out[:12]: print(o)
