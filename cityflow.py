"""Drop-in `cityflow` module backed by the MI355X-native engine (see cityflow_amd/)."""
from cityflow_amd import Archive, Engine, __version__  # noqa: F401
