"""spectrogram_amd -- MI355X-native STFT / analyser engine.

Only what the hot path needs: csrc/ (HIP kernels + C ABI), the ctypes binding,
and host-side mirrors of the reference's AnalyserNode interface.
"""
from ._lib import SpectroError, load, runtime_info, runtime_policy  # noqa: F401
from .engine import Engine, make_config  # noqa: F401
from .analyser import Analyser  # noqa: F401
from . import axis, wav  # noqa: F401

__all__ = ["Engine", "Analyser", "SpectroError", "load", "make_config", "wav", "axis", "runtime_info", "runtime_policy"]
