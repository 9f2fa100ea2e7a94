"""In-tree native build for cityflow_amd (no pip, no cmake, no JIT cache).

Artefacts (git-ignored, shipped to the GPU box by gpurun):
  cityflow_amd/_cityflow<ext-suffix>.so   C++17 host: JSON loader, spawner, pybind11 `Engine` (g++)
  cityflow_amd/lib/libcfx_hip.so          C ABI of include/cityflow_amd.h + HIP kernels for gfx950 (hipcc)

`python cityflow_amd/build.py` builds both (run it by path: the package itself refuses to import unbuilt); `--host` / `--hip` select one.  Rebuilds are skipped when the
artefact is newer than every source it depends on.
"""
import glob
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")
HOST_DIR = os.path.join(HERE, "csrc", "host")
HIP_DIR = os.path.join(HERE, "csrc", "hip")
LIB_DIR = os.path.join(HERE, "lib")

HOST_SRCS = ["roadnet.cpp", "flow.cpp", "engine_host.cpp", "archive.cpp", "vector_engine.cpp", "tile_engine.cpp", "replay.cpp", "pymodule.cpp"]

# -ffp-contract=off: the reference is built by g++ for x86-64 without FMA contraction; every double
# expression on the parity path must round exactly like it.
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-sign-compare"]
HIP_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result"]


def _ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def host_target():
    return os.path.join(HERE, "_cityflow" + _ext_suffix())


def hip_target():
    return os.path.join(LIB_DIR, "libcfx_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("[cityflow_amd.build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _pybind_includes():
    import pybind11

    return ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]


def build_host(force=False):
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SRCS]
    deps = srcs + glob.glob(os.path.join(HOST_DIR, "*.h")) + [os.path.join(INCLUDE, "cityflow_amd.h")]
    tgt = host_target()
    if not force and not _stale(tgt, deps):
        return tgt
    cxx = os.environ.get("CXX", "g++")
    objs = []
    obj_dir = os.path.join(HERE, "build", "host")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            cmd = [cxx] + HOST_FLAGS + ["-I" + INCLUDE, "-I" + HOST_DIR] + _pybind_includes() + ["-c", s, "-o", o]
            print("[cityflow_amd.build] " + " ".join(cmd), flush=True)
            procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("host compile failed")
    _run([cxx, "-shared"] + objs + ["-ldl", "-o", tgt])
    return tgt


def build_hip(force=False):
    srcs = sorted(glob.glob(os.path.join(HIP_DIR, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources under " + HIP_DIR)
    deps = srcs + glob.glob(os.path.join(HIP_DIR, "*.h")) + glob.glob(os.path.join(HIP_DIR, "*.hpp")) + [
        os.path.join(INCLUDE, "cityflow_amd.h")]
    tgt = hip_target()
    if not force and not _stale(tgt, deps):
        return tgt
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIP_FLAGS if f] + os.environ.get("CFX_HIP_EXTRA_FLAGS", "").split()
    _run([hipcc] + flags + ["-shared", "-I" + INCLUDE, "-I" + HIP_DIR] + srcs + ["-o", tgt])
    return tgt


def build_all(force=False):
    return build_host(force), build_hip(force)


if __name__ == "__main__":
    args = sys.argv[1:]
    force = "--force" in args
    if "--host" in args:
        build_host(force)
    elif "--hip" in args:
        build_hip(force)
    else:
        build_all(force)
