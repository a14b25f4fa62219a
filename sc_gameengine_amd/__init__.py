"""sc_gameengine_amd -- MI355X-native world-tick path for SandboxCityEngine (Skeletus/SC-GameEngine).

Only what the hot path needs: csrc/ (hand-written gfx950 kernels + the C ABI of include/sc_tick.h),
host/ (C++ adapter systems with the engine's `void(World&, float, void*)` signature), and a thin
Python host (capi / tick / synth_world) used by tests and bench.py.  No CPU fallback.
"""
from . import capi, synth_world  # noqa: F401
from .tick import WorldTick  # noqa: F401

__all__ = ["capi", "synth_world", "WorldTick"]
