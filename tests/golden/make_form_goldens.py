"""Generates tests/golden/reference_irregular.json from the UNMODIFIED reference engine (oracle/_ref/cityflow_ref*.so, one
thread): checkpoint records (tests/conftest.py::checkpoint_record — vehicle count, per-lane counts, average travel time,
every vehicle's exact (speed, distance)) on the seeded IRREGULAR networks of tests/test_irregular.py, together with the
sha256 of the roadnet / flow files the seed produced — so that the GPU box, which has no /root/reference, (a) knows it
rebuilt the very same inputs and (b) compares the HIP engine, in every organisation of its step, with what the reference
itself computed on them (tests/test_reference_forms.py).

  python tests/golden/make_form_goldens.py
"""
import hashlib
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

SEEDS = [11, 14, 21]
STEPS = [100, 250, 400]


def file_sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    import cityflow_ref
    from cityflow_amd import scenarios
    from conftest import checkpoint_record
    from test_irregular import irregular

    work = tempfile.mkdtemp(prefix="goldens_forms_")
    out = {}
    for seed in SEEDS:
        cfg = irregular(scenarios, work, seed, n=6 if seed == 21 else 5)
        d = os.path.dirname(cfg)
        eng = cityflow_ref.Engine(cfg, 1)
        recs = {}
        for s in range(1, max(STEPS) + 1):
            eng.next_step()
            if s in STEPS:
                recs[str(s)] = checkpoint_record(eng)
        out[str(seed)] = {"n": 6 if seed == 21 else 5, "roadnet_sha256": file_sha(os.path.join(d, "roadnet.json")),
                          "flow_sha256": file_sha(os.path.join(d, "flow.json")), "checkpoints": recs}
        print("irregular", seed, {k: v["vehicle_count"] for k, v in recs.items()}, flush=True)
        time.sleep(0.2)  # reference destructor race
        del eng
    with open(os.path.join(HERE, "reference_irregular.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("written")
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
