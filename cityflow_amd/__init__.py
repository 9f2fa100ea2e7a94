"""cityflow_amd — MI355X-native CityFlow step engine.

`Engine` keeps the reference `cityflow.Engine` API (reference src/cityflow.cpp:10-47); the repo-root
`cityflow` module re-exports it so existing RL agents can `import cityflow` unchanged.  The native
extension is built in-tree by `cityflow_amd.build` (see `__graft_entry__.build()`); importing this
package never falls back to a Python/CPU implementation.
"""
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
