"""Drop-in `cityflow` module backed by the MI355X-native engine (see cityflow_amd/)."""
from cityflow_amd import Engine, __version__  # noqa: F401
