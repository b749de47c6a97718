"""nerfpp_amd -- MI355X-native (gfx950) NeRF / HashNeRF volume-rendering path.

Host-side mirror of the reference's renderer / embedder / MLP plugin surface over the C ABI of
libnerfpp_hip.so (include/nerfpp_hip.h).  Importing the package does not load the shared object; the first
compute call does, and raises if it has not been built (there is no CPU fallback).
"""
from . import synth  # noqa: F401

__all__ = ["synth", "modules", "renderer", "scene"]


def __getattr__(name):
    import importlib
    if name in ("modules", "renderer", "scene", "_lib", "dist"):
        return importlib.import_module(f"{__name__}.{name}")
    for mod in ("modules", "renderer"):
        m = importlib.import_module(f"{__name__}.{mod}")
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
