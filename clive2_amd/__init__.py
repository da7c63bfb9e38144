"""clive2_amd: MI355X-native bidirectional path tracer behind the Clive2 Python API.

Host plumbing (scene, camera, BVH, struct layouts) lives here as numpy code; the per-sample
hot path runs in hand-written HIP kernels behind the C-ABI declared in `include/clive2_amd.h`
(`clive2_amd/csrc/`), bound with ctypes in `_native.py` and driven by `renderer.Renderer`.
"""
from . import struct_types, constants  # noqa: F401
from .scene import (Scene, create_scene, create_scene_from_preset,  # noqa: F401
                    create_scene_from_preset_with_params, scene_presets)
from .camera import Camera, tone_map  # noqa: F401

__all__ = ["Scene", "create_scene", "create_scene_from_preset",
           "create_scene_from_preset_with_params", "scene_presets", "Camera", "tone_map",
           "struct_types", "constants"]
