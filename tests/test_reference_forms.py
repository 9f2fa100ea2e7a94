"""GPU (-m gpu): EVERY organisation of the HIP engine's step against records the unmodified REFERENCE produced — not only
against the CPU twin.  The forms `auto` would not pick on a small network are forced through the config's "cfx" object
(which never changes results): ring layout in block form, wave form, list form (kr_index + kl_action) and its ticketed
variant, with the cross phase in both organisations; dense layout in its round-2 form and with every `denseForm` bit.

  * the reference's golden checkpoints of the stock grids (tests/golden/reference_checkpoints.json, make_goldens.py):
    6x6 to step 1000, 30x30 to step 250;
  * the seeded irregular networks of tests/test_irregular.py (jittered geometry, removed roads, random signal plans,
    vehicle templates and routes) against tests/golden/reference_irregular.json (make_form_goldens.py), the inputs'
    sha256 first: the GPU box must have rebuilt the very files the reference ran on.

CPU: the twin against the same irregular goldens (the oracle is pinned there too)."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN, TWIN_LIB, checkpoint_record, assert_hip_backend
from test_irregular import irregular

FORMS = {
    "auto": {},
    "ring-block-latency": {"layout": "ring", "ringLanesPerWave": 20000, "crossMode": "latency"},
    "ring-wave-throughput": {"layout": "ring", "ringLanesPerWave": 10000, "crossMode": "throughput"},
    "ring-list-throughput": {"layout": "ring", "ringLanesPerWave": 30000, "crossMode": "throughput"},
    "ring-list-latency": {"layout": "ring", "ringLanesPerWave": 30000, "crossMode": "latency"},
    "ring-list-ticket": {"layout": "ring", "ringLanesPerWave": 60000, "crossMode": "throughput"},
    "ring-own-commit": {"layout": "ring", "ringLanesPerWave": 40000},
    "dense-round2": {"layout": "dense", "denseForm": 256, "crossMode": "latency"},
    "dense-lanes-bigbatch": {"layout": "dense", "denseForm": 256 + 2 + 4, "crossMode": "throughput"},
    "dense-lanes": {"layout": "dense", "denseForm": 256 + 2, "crossMode": "latency"},
}


def _with_cfx(path, tag, cfx):
    if not cfx:
        return path
    c = json.load(open(path))
    c["cfx"] = cfx
    out = path.replace(".json", "_form_%s.json" % tag)
    with open(out, "w") as f:
        json.dump(c, f)
    return out


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _irregular_golden():
    with open(os.path.join(GOLDEN, "reference_irregular.json")) as f:
        return json.load(f)


def _irregular_cfg(scen, workdir, seed, gold):
    cfg = irregular(scen, workdir, seed, n=gold["n"])
    d = os.path.dirname(cfg)
    assert _sha(os.path.join(d, "roadnet.json")) == gold["roadnet_sha256"], "seed %d: not the network the reference ran on" % seed
    assert _sha(os.path.join(d, "flow.json")) == gold["flow_sha256"], "seed %d: not the flows the reference ran on" % seed
    return cfg


def _run_against(eng, want, where):
    want = {int(k): v for k, v in want.items()}
    for s in range(1, max(want) + 1):
        eng.next_step()
        if s in want:
            assert checkpoint_record(eng) == want[s], "%s step %d" % (where, s)


@pytest.mark.parametrize("seed", [11, 14, 21])
def test_twin_matches_reference_goldens_on_irregular_networks(mod, scen, workdir, seed):
    gold = _irregular_golden()[str(seed)]
    eng = mod.Engine._with_backend(_irregular_cfg(scen, workdir, seed, gold), 1, TWIN_LIB)
    _run_against(eng, gold["checkpoints"], "twin, irregular %d" % seed)


@pytest.mark.gpu
@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("seed", [11, 14, 21])
def test_every_form_matches_reference_goldens_on_irregular_networks(mod, scen, workdir, seed, form):
    gold = _irregular_golden()[str(seed)]
    eng = mod.Engine(_with_cfx(_irregular_cfg(scen, workdir, seed, gold), form, FORMS[form]), 1)
    assert_hip_backend(eng)
    _run_against(eng, gold["checkpoints"], "%s, irregular %d" % (form, seed))


@pytest.mark.gpu
@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("name,last", [("grid_6x6", 1000), ("grid_30x30", 250)])
def test_every_form_matches_reference_goldens_on_the_stock_grids(mod, scen, workdir, golden, name, last, form):
    eng = mod.Engine(_with_cfx(scen.materialize(name, workdir), form, FORMS[form]), 1)
    assert_hip_backend(eng)
    want = {k: v for k, v in golden["reference_checkpoints"][name].items() if int(k) <= last}
    _run_against(eng, want, "%s, %s" % (form, name))
