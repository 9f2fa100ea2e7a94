"""One-off fuzz (developer tool): the random API call sequences of tests/test_api_sequences.py — the unmodified reference
(oracle/_ref) against this host on the CPU twin — over a range of further seeds, one subprocess per run (the reference can
crash: e.g. `reset` after `push_vehicle` without a step in between leaves freed vehicles in a road's planRouteBuffer).
usage: python tests/tools/api_sequence_fuzz.py <first_seed> <end_seed>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, tempfile, faulthandler
faulthandler.enable()
ROOT = sys.argv[3]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
import cityflow_ref
from cityflow_amd import _cityflow as mod, scenarios as scen
import test_api_sequences as t
wd = tempfile.mkdtemp(prefix="fuzz_api_")
seed = int(sys.argv[1]); which = sys.argv[2]
if which == "calls": t.test_random_call_sequences_equal_reference(mod, cityflow_ref, scen, wd, seed % 2 == 0, seed)
elif which == "control": t.test_random_control_and_query_calls_equal_reference(mod, cityflow_ref, scen, wd, seed)
else: t.test_waiting_finished_and_reseeded_vehicles_equal_reference(mod, cityflow_ref, scen, wd, 0.5 if seed % 3 else 1.0, seed)
print("ok", flush=True); os._exit(0)
'''
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for which in ("calls", "control", "waiting"):
        r = subprocess.run([sys.executable, "-c", code, str(seed), which, ROOT], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr).strip().splitlines()[-12:]
        status = "ok" if r.returncode == 0 else ("CRASH(rc %d)" % r.returncode if "Segmentation" in r.stderr or r.returncode < 0 else "FAIL")
        print(status, which, seed, "" if status == "ok" else " | ".join(tail), flush=True)
