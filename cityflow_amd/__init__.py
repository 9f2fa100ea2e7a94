"""cityflow_amd — MI355X-native CityFlow step engine.

`Engine` keeps the reference `cityflow.Engine` API (reference src/cityflow.cpp:10-47); the repo-root
`cityflow` module re-exports it so existing RL agents can `import cityflow` unchanged.  The native
extension is built in-tree by `cityflow_amd.build` (see `__graft_entry__.build()`); importing this
package never falls back to a Python/CPU implementation.
"""
# (the array API returns numpy arrays: imported with the package, not by the first getter in the middle of a caller's loop —
# importing numpy starts one BLAS worker per core, each spinning for ~20 ms, which inside a CPU-quota'd container throttles
# every thread of the process for the rest of the scheduler period: bench.py's docstring)
import numpy as _numpy  # noqa: F401

try:
    from . import _cityflow
except ImportError as exc:  # pragma: no cover - exercised only on unbuilt trees
    raise ImportError(
        "cityflow_amd: the native extension is not built; run `python cityflow_amd/build.py` "
        "(or `__graft_entry__.build()`) first"
    ) from exc

Engine = _cityflow.Engine
Archive = _cityflow.Archive
VectorEngine = _cityflow.VectorEngine
TiledEngine = _cityflow.TiledEngine  # one network over several engines; cityflow_amd.tiled.DistributedEngine = one per GPU
__version__ = _cityflow.__version__

__all__ = ["Engine", "Archive", "VectorEngine", "TiledEngine", "__version__"]
