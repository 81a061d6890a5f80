"""sound_bubble_amd -- MI355X-native (gfx950) hot path of the Sound-Bubble separation model.

Python host code over hand-written HIP kernels behind a plain C ABI
(include/sound_bubble_hip.h -> sound_bubble_amd/lib/libsoundbubble_hip.so).
"""
from .net import NetDisEmbd3, NetOptim  # noqa: F401

__all__ = ["NetDisEmbd3", "NetOptim"]
