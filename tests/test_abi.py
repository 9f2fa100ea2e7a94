"""CPU: the product's C-ABI library loads without a GPU and exports every symbol include/cityflow_amd.h declares;
constructing an engine without a device fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, TWIN_LIB


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "cityflow_amd.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(cfx_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = _declared_symbols()
    for must in ("cfx_create", "cfx_destroy", "cfx_step", "cfx_reset", "cfx_set_tl_phase", "cfx_get_lane_counts",
                 "cfx_get_vehicles", "cfx_get_scalars", "cfx_last_error"):
        assert must in syms


@pytest.mark.parametrize("lib", ["hip", "twin"])
def test_library_exports_every_declared_symbol(mod, lib):
    path = mod._default_backend_path() if lib == "hip" else TWIN_LIB
    assert os.path.exists(path), path
    dll = ctypes.CDLL(path)
    for s in _declared_symbols():
        assert hasattr(dll, s), "%s does not export %s" % (path, s)
    dll.cfx_abi_version.restype = ctypes.c_int32
    assert dll.cfx_abi_version() == 9
    dll.cfx_backend_name.restype = ctypes.c_char_p
    assert dll.cfx_backend_name() == (b"hip-gfx950" if lib == "hip" else b"cpu-twin")


def test_no_silent_cpu_fallback(mod, scen, workdir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mod.Engine(scen.materialize("example_1x1", workdir), 1)


def test_drop_in_module_surface():
    import cityflow
    for name in ("next_step", "get_vehicle_count", "get_vehicles", "get_lane_vehicle_count",
                 "get_lane_waiting_vehicle_count", "get_lane_vehicles", "get_vehicle_speed", "get_vehicle_info",
                 "get_vehicle_distance", "get_leader", "get_current_time", "get_average_travel_time", "set_tl_phase",
                 "set_random_seed", "push_vehicle", "reset"):
        assert hasattr(cityflow.Engine, name), name
