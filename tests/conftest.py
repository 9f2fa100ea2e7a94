"""Shared fixtures.  Marker policy: tests that need an MI355X carry @pytest.mark.gpu; everything else runs on CPU.
oracle/ (CPU twin, unmodified reference build, golden vectors) is used here strictly as the checker."""
import hashlib
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
TWIN_LIB = os.path.join(ROOT, "oracle", "_ref", "libcfx_twin.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of the torch.distributed.run tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    spec = importlib.util.spec_from_file_location("cityflow_amd_build", os.path.join(ROOT, "cityflow_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build_all()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "all"])


_build_once()


@pytest.fixture(scope="session")
def mod():
    from cityflow_amd import _cityflow
    return _cityflow


@pytest.fixture(scope="session")
def scen():
    from cityflow_amd import scenarios
    return scenarios


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("scenarios"))


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("reference_checkpoints", "reference_spawns", "roadnet_probe"):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            out[name] = json.load(f)
    return out


@pytest.fixture(scope="session")
def ref_module():
    """The unmodified reference engine (oracle/_ref); skipped when it was not built / shipped."""
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    try:
        import cityflow_ref
    except ImportError:
        pytest.skip("oracle/_ref/cityflow_ref not built (needs /root/reference)")
    return cityflow_ref


def lane_hash(counts):
    return hashlib.sha256(json.dumps(sorted(counts.items())).encode()).hexdigest()


def state_hash(speed, distance):
    h = hashlib.sha256()
    for k in sorted(speed):
        h.update(("%s %s %s\n" % (k, float(speed[k]).hex(), float(distance[k]).hex())).encode())
    return h.hexdigest()


def checkpoint_record(eng):
    lc = eng.get_lane_vehicle_count()
    return {
        "vehicle_count": eng.get_vehicle_count(),
        "lane_sum": sum(lc.values()),
        "lane_hash": lane_hash(lc),
        "average_travel_time": float(eng.get_average_travel_time()).hex(),
        "state_hash": state_hash(eng.get_vehicle_speed(), eng.get_vehicle_distance()),
    }


def full_state(eng):
    """Per-vehicle state sorted by vid, for exact engine-vs-engine comparison."""
    import numpy as np
    s = eng._vehicle_state()
    order = np.argsort(s["vid"], kind="stable")
    return {k: v[order] for k, v in s.items()}


def assert_same_state(a, b, where="", skip=()):
    import numpy as np
    sa, sb = full_state(a), full_state(b)
    for k in ("vid", "drivable", "prev_drivable", "dis", "speed", "leader", "blocker", "enter_ll_time", "route_pos"):
        if k in skip:
            continue
        assert sa[k].shape == sb[k].shape, "%s: %s count differs (%s vs %s)" % (where, k, sa[k].shape, sb[k].shape)
        assert np.array_equal(sa[k], sb[k]), "%s: %s differs" % (where, k)
    has = sa["leader"] >= 0
    assert np.array_equal(sa["gap"][has], sb["gap"][has]), "%s: gap differs" % where
    assert np.array_equal(a.get_lane_vehicle_count_array(), b.get_lane_vehicle_count_array()), where + ": lane counts"
    ca, cb = a._scalars(), b._scalars()
    for k in ("active_vehicle_count", "finished_vehicle_count", "spawned_vehicle_count", "cumulative_travel_time"):
        assert ca[k] == cb[k], "%s: scalar %s differs (%r vs %r)" % (where, k, ca[k], cb[k])
    pa, pb = a._tl_state(), b._tl_state()
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1]), where + ": traffic-light state differs"


def dump_json_exact(obj, path):
    """json.dump for files the engines will read: every float as the literal Archive.dump itself would write — one that a
    correctly rounding reader AND the reference's reader (rapidjson's default number reader, which is an ulp or two off on
    many of the 16/17-digit literals `repr` produces; csrc/host/json_number.h) both turn back into exactly that float."""
    from cityflow_amd import _cityflow
    out = []

    def emit(x):
        if isinstance(x, dict):
            out.append("{")
            for i, (k, v) in enumerate(x.items()):
                if i:
                    out.append(",")
                out.append(json.dumps(str(k)))
                out.append(":")
                emit(v)
            out.append("}")
        elif isinstance(x, (list, tuple)):
            out.append("[")
            for i, v in enumerate(x):
                if i:
                    out.append(",")
                emit(v)
            out.append("]")
        elif isinstance(x, float):
            out.append(_cityflow._format_json_number(x))
        else:
            out.append(json.dumps(x))

    emit(obj)
    with open(path, "w") as f:
        f.write("".join(out))
